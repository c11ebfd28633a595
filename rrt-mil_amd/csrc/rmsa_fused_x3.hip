// rmsa_fused_x3.hip -- the fused R-MSA core with the qkv projection EMULATED IN FP32 ON THE BF16 MATRIX CORES
// (RRT_COMPUTE_F32X3): every fp32 operand is carried as a (hi, lo) pair of bf16 values, x ~ hi + lo (16 significant
// bits), and a product a.b is three bf16 MFMAs with fp32 accumulation, ah.bh + ah.bl + al.bh (the dropped al.bl term
// is 2^-16 of the product).  SURVEY.md 7.3 H2 measured this split at 9.5e-7 max-abs on the encoder output against the
// reference -- the same distance true fp32 MFMA has -- while v_mfma_f32_16x16x32_bf16 runs 16x the rate of
// v_mfma_f32_16x16x4_f32: three of them per 32-wide K step against eight fp32 MFMAs is 5x less matrix-pipe time for
// the 84 % of the kernel's FLOPs that sit in the projection.
//
// Replaces InnerAttention.forward up to (not including) proj, modules/rmsa.py:100-122, exactly as rmsa_fused.hip
// does; only phase 1 differs:
//   phase 1  C[P x 192] = U_r . W_h^T from the split images of U (LayerNorm output, cast16.hip
//            ln_partition_split_kernel) and W (cast_split_kernel): a row of the image is [32 hi | 32 lo] per 32
//            elements = the fp32 kernels' 128-byte K tile, so the LDS ring, the XOR swizzle and the DMA addressing are
//            unchanged; slots 0..3 / 4..7 of a tile row are the hi / lo MFMA operands.  The loop is then bound by
//            LDS-DMA bandwidth like the 16-bit kernel's: four loader waves, 3-stage ring (rmsa_fused16.hip);
//   phases 2-4  fp32, as rmsa_fused.hip: Q / K / V tiles (fp32) -> EPEG stencil in place -> softmax(Q~ K^T) V on the
//            fp32 matrix cores (scores and probabilities keep full fp32 products: 16 % of the FLOPs);
//   output   O in the split layout, the A operand of the proj GEMM (linear_ws_kernel PREC_SPLIT).
// One wave owns one 16-column tile of each of Q, K and V (balanced phase 2).  MT <= 9 (regions of <= 144 tokens:
// 204 VGPRs at two waves per SIMD); larger regions keep the exact fp32 kernel.
#include <stdio.h>
#include <stdlib.h>

#include "internal.h"

namespace {

constexpr int HD = 64;
constexpr int BN = 3 * HD;
constexpr int ROWB = 128;           // bytes of one staged row = one K tile of 32 elements (hi | lo)
constexpr float NEG_BIG = -3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// (hi, lo) split of 4 floats (cast16.hip split4)
__device__ __forceinline__ void split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
  typedef __bf16 v4 __attribute__((ext_vector_type(4)));
  v4 h, l;
  h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
  l[0] = (__bf16)(a - (float)h[0]); l[1] = (__bf16)(b - (float)h[1]);
  l[2] = (__bf16)(c - (float)h[2]); l[3] = (__bf16)(d - (float)h[3]);
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
}

template <int MT>
__global__ __launch_bounds__(512, 2) void rmsa_fused_x3_kernel(const char* __restrict__ U, const char* __restrict__ W,
                                                               const float* __restrict__ bqkv,
                                                               const float* __restrict__ pe_w, char* __restrict__ O,
                                                               int n_rows, int P, int D, int heads_rt, int epeg_k,
                                                               float q_scale) {
  constexpr int BM = 16 * MT;
  constexpr int STAGE_B = (BM + BN) * ROWB;
  constexpr int NA = BM / 8, NB = BN / 8;
  constexpr int LA = (NA + 3) / 4, LB = NB / 4;
  constexpr int NT = 3;
  constexpr int TILE = BM * HD;                        // floats of one Q / K / V tile
  constexpr int RING_B = 3 * STAGE_B, TILES_B = 3 * TILE * 4;
  constexpr int LDS_MAIN = RING_B > TILES_B ? RING_B : TILES_B;
  constexpr int RUN = (BM + 31) / 32;
  constexpr int TAP_OFF = 12 + RUN - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = (float*)smem;                            // fp32 tiles alias the dead staging ring
  float* Ks = Qs + TILE;
  float* Vs = Qs + 2 * TILE;
  float* Qt = Qs;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const unsigned lds_b = lds_addr_of(smem);
  int head, reg;
  {
    const int b = blockIdx.x;
    const int n_regions = gridDim.x / heads_rt;
    const int full = (n_regions >> 3) * 8 * heads_rt;
    if (b < full) {
      const int xcd = b & 7, idx = b >> 3;
      const int grp = idx / heads_rt;
      reg = grp * 8 + xcd;
      head = idx - grp * heads_rt;
    } else {
      const int rem = b - full;
      reg = (n_regions >> 3) * 8 + rem / heads_rt;
      head = rem % heads_rt;
    }
  }
  const int row0 = reg * P;
  const int nk = D / 32;
  float* const taps = (float*)(smem + LDS_MAIN);
  if (tid < 128) {
    const int t = tid - TAP_OFF;
    float wt = (pe_w != nullptr && t >= 0 && t < epeg_k) ? pe_w[head * epeg_k + t] : 0.f;
    if (t == (epeg_k >> 1)) wt += 1.0f;
    taps[tid] = wt * LOG2E;
  }

  // ================================================================== phase 1: projection (split operands)
  if (wave >= 4) {
    const int lw = wave - 4;
    unsigned aoff[LA], boff[LB];
#pragma unroll
    for (int qi = 0; qi < LA; ++qi) {
      const int row = (qi * 4 + lw) * 8 + (lane >> 3), p = lane & 7;
      int gr = row0 + row;
      gr = gr < n_rows ? gr : n_rows - 1;
      aoff[qi] = (unsigned)gr * (unsigned)D * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int qi = 0; qi < LB; ++qi) {
      const int row = (qi * 4 + lw) * 8 + (lane >> 3), p = lane & 7;
      const int wr = (row >> 6) * D + head * HD + (row & 63);
      boff[qi] = (unsigned)wr * (unsigned)D * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
    auto stage = [&](int kt, unsigned buf) {
      const char* ub = U + kt * ROWB;
      const char* wb = W + kt * ROWB;
#pragma unroll
      for (int qi = 0; qi < LA; ++qi)
        if (qi * 4 + lw < NA) dma16s(ub, aoff[qi], buf + (qi * 4 + lw) * 1024);
#pragma unroll
      for (int qi = 0; qi < LB; ++qi) dma16s(wb, boff[qi], buf + BM * ROWB + (qi * 4 + lw) * 1024);
    };
    const bool full = (LA - 1) * 4 + lw < NA;
    stage(0, lds_b);
    if (nk > 1) stage(1, lds_b + STAGE_B);
    int slot = 2;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {
        if (full) wait_vmcnt<LA + LB>(); else wait_vmcnt<LA - 1 + LB>();
      } else {
        wait_vm0();
      }
      __syncthreads();
      if (kt + 2 < nk) stage(kt + 2, lds_b + slot * STAGE_B);
      slot = slot == 2 ? 0 : slot + 1;
    }
    __syncthreads();                                // "the staging ring is dead"
  } else {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int dq = 16 * wave + 4 * lg;              // first of the lane's 4 columns in each of its Q / K / V tiles
    float4 b4[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
      b4[j] = bqkv ? *(const float4*)(bqkv + j * D + head * HD + dq) : make_float4(0.f, 0.f, 0.f, 0.f);
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      const char* As = smem + slot * STAGE_B;
      const char* Bs = As + BM * ROWB;
      slot = slot == 2 ? 0 : slot + 1;
      bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = 64 * j + 16 * wave + lr, f = (row >> 1) & 7;
        bh[j] = *(const bf16x8*)(Bs + row * ROWB + ((lg ^ f) << 4));
        bl[j] = *(const bf16x8*)(Bs + row * ROWB + (((4 + lg) ^ f) << 4));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + lr, f = (row >> 1) & 7;
        ah[i] = *(const bf16x8*)(As + row * ROWB + ((lg ^ f) << 4));
        al[i] = *(const bf16x8*)(As + row * ROWB + (((4 + lg) ^ f) << 4));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    }
    // ================================================================ phase 2: Q / K / V tiles (fp32) -> LDS
    __syncthreads();                                // every wave is done with the staging ring
    // transposed accumulators: reg r of lane (lr, lg) is C[m = 16 i + lr][tile j, column dq + r]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float sc = j == 0 ? q_scale : 1.0f;
      float* dstm = j == 0 ? Qs : (j == 1 ? Ks : Vs);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = i * 16 + lr;
        *(float4*)(dstm + m * HD + (((dq >> 2) ^ (m & 15)) << 2)) =
            make_float4((acc[i][j][0] + b4[j].x) * sc, (acc[i][j][1] + b4[j].y) * sc,
                        (acc[i][j][2] + b4[j].z) * sc, (acc[i][j][3] + b4[j].w) * sc);
      }
    }
  }
  __syncthreads();                                  // Q / K / V tiles complete

  // ================================================================== phase 3: EPEG stencil -> Q~ (in place)
  {
    static_assert(RUN - 1 <= TAP_OFF && (TAP_OFF - (RUN - 1)) % 4 == 0 && 2 * RUN + 90 < 128, "tap table range / alignment");
    const int half = epeg_k >> 1;
    const int s = tid & 15, g = tid >> 4;
    const int r0 = g * RUN;
    float4 out[RUN];
#pragma unroll
    for (int o = 0; o < RUN; ++o) out[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < BM) {
      const int nsrc = RUN + 2 * half;
      constexpr int NTV = (RUN + 3 + 3) / 4;
      const float4* tp4 = (const float4*)(taps + TAP_OFF - (RUN - 1));
      for (int j0 = 0; j0 < nsrc; j0 += 4) {
        float4 v[4];
        float T[4 * NTV];
#pragma unroll
        for (int q = 0; q < NTV; ++q) {
          const float4 t4 = tp4[(j0 >> 2) + q];
          T[4 * q] = t4.x; T[4 * q + 1] = t4.y; T[4 * q + 2] = t4.z; T[4 * q + 3] = t4.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = r0 - half + j0 + u;
          const bool ok = rr >= 0 && rr < P;
          const int rc = ok ? rr : 0;
          v[u] = *(const float4*)(Qs + rc * HD + ((s ^ (rc & 15)) << 2));
          if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int o = 0; o < RUN; ++o) {
            const float wt = T[u + RUN - 1 - o];
            out[o].x += wt * v[u].x; out[o].y += wt * v[u].y; out[o].z += wt * v[u].z; out[o].w += wt * v[u].w;
          }
      }
    }
    __syncthreads();                                // all reads of Q done
    if (r0 < BM) {
#pragma unroll
      for (int o = 0; o < RUN; ++o) {
        const int m = r0 + o;
        if (m < BM) *(float4*)(Qt + m * HD + ((s ^ (m & 15)) << 2)) = out[o];
      }
    }
  }
  __syncthreads();

  // ================================================================== phase 4: attention from LDS (fp32 MFMA)
  for (int t = wave; t < MT; t += 8) {
    const int i0 = t * 16;
    if (i0 >= P) break;
    float4 bq[4];
    {
      const int m = i0 + lr;
#pragma unroll
      for (int c = 0; c < 4; ++c) bq[c] = *(const float4*)(Qt + m * HD + (((4 * c + lg) ^ (m & 15)) << 2));
    }
    f32x4 s[MT];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 a[MT];
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) {
        const int row = jt * 16 + lr;
        a[jt] = *(const float4*)(Ks + row * HD + (((4 * c + lg) ^ (row & 15)) << 2));
      }
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].x, bq[c].x, s[jt], 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].y, bq[c].y, s[jt], 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].z, bq[c].z, s[jt], 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].w, bq[c].w, s[jt], 0, 0, 0);
    }
    float cmax = NEG_BIG;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) {
      if ((jt + 1) * 16 > P) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (jt * 16 + 4 * lg + r >= P) s[jt][r] = NEG_BIG;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) cmax = fmaxf(cmax, s[jt][r]);
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
    float psum = 0.f;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[jt][r] - cmax);
        s[jt][r] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    const float inv = 1.0f / psum;
    f32x4 oacc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = jt * 16 + 4 * lg + r;
        const float4 v = *(const float4*)(Vs + row * HD + ((lr ^ (row & 15)) << 2));
        const float p = s[jt][r];
        oacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.x, oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.y, oacc[1], 0, 0, 0);
        oacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.z, oacc[2], 0, 0, 0);
        oacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.w, oacc[3], 0, 0, 0);
      }
    // O row (row0 + i), columns head * 64 + 4 lr .. + 3, as (hi, lo) quads of the split image
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ir = __shfl(inv, 4 * lg + r);
      const int i = i0 + 4 * lg + r;
      if (i < P) {
        uint2 hi, lo;
        split4(oacc[0][r] * ir, oacc[1][r] * ir, oacc[2][r] * ir, oacc[3][r] * ir, hi, lo);
        const int c = head * HD + (lr << 2);
        char* dst = O + (size_t)(row0 + i) * D * 4 + (c >> 5) * 128 + (c & 31) * 2;
        *(uint2*)dst = hi;
        *(uint2*)(dst + 64) = lo;
      }
    }
  }
}

template <int MT>
hipError_t launch_mt(const char* U, const char* W, const float* bqkv, const float* pe_w, char* O, int n_regions, int P,
                     int D, int heads, int epeg_k, hipStream_t st) {
  constexpr int BM = 16 * MT;
  constexpr size_t RING = (size_t)3 * (BM + BN) * ROWB, TILES = (size_t)3 * BM * HD * 4;
  constexpr size_t LDS = (RING > TILES ? RING : TILES) + 512;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = rmsa_fused_x3_kernel<MT>;
  static OncePerDevice once;
  if (once.first())
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
  const float q_scale = 1.0f / sqrtf((float)HD);
  kern<<<dim3(heads * n_regions), dim3(512), LDS, st>>>(U, W, bqkv, pe_w, O, n_regions * P, P, D, heads,
                                                       pe_w ? epeg_k : 0, q_scale);
  return hipGetLastError();
}

}  // namespace

bool rmsa_fused_x3_supported(int P, int D, int heads, int epeg_k) {
  static const bool off = getenv("RRT_NO_FUSED_X3") != nullptr;
  if (off) return false;
  return heads > 0 && D == heads * HD && D % 32 == 0 && P > 48 && P <= 144 && epeg_k >= 0 && epeg_k <= 63;
}

hipError_t launch_rmsa_fused_x3(const void* U, const void* W, const float* bqkv, const float* pe_w, void* O,
                                int n_regions, int P, int D, int heads, int epeg_k, hipStream_t st) {
#define RRT_X3(MT_) \
  return launch_mt<MT_>((const char*)U, (const char*)W, bqkv, pe_w, (char*)O, n_regions, P, D, heads, epeg_k, st);
  if (P > 128) { RRT_X3(9) }
  if (P > 112) { RRT_X3(8) }
  if (P > 96) { RRT_X3(7) }
  if (P > 64) { RRT_X3(6) }
  RRT_X3(4)
#undef RRT_X3
}
