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
//   phase 2  accumulators (+bias, q*scale) -> LDS: Q in fp32 (the EPEG stencil runs in fp32), K as two 16-bit row images
//            (hi, lo), V as two TRANSPOSED 16-bit images with the keys in MFMA operand order (rmsa_fused16.hip's layout);
//   phase 3  EPEG stencil in fp32, x log2(e), Q~ written as (hi, lo) row images over Q;
//   phase 4  S^T = K Q~^T and O^T = V^T P^T as three bf16 MFMAs per product (P split in registers), softmax statistics
//            in fp32 -- the attention's matrix-pipe time drops 5x like the projection's;
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
constexpr int VT_PITCH = 512;       // bytes per V^T row: 32 x 16-byte slots, XOR-swizzled (rmsa_fused16.hip)
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

__device__ __forceinline__ int vt_swz(int d) { return ((d >> 2) ^ ((d & 3) << 2)) & 15; }

// 8 floats -> (hi, lo) bf16x8 fragments
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (__bf16)a[e];
    lo[e] = (__bf16)(a[e] - (float)hi[e]);
    hi[4 + e] = (__bf16)b[e];
    lo[4 + e] = (__bf16)(b[e] - (float)hi[4 + e]);
  }
}
__device__ __forceinline__ f32x4 mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
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
  constexpr int MTP = (MT + 1) & ~1;                   // key tiles rounded up to whole 32-key MFMA blocks
  constexpr int QF_B = BM * 256, KP_B = BM * ROWB;     // fp32 Q tile; one 16-bit plane of K (or of Q~)
  constexpr int RING_B = 3 * STAGE_B, TILES_B = QF_B + 2 * KP_B + 2 * 64 * VT_PITCH;
  constexpr int LDS_MAIN = RING_B > TILES_B ? RING_B : TILES_B;
  constexpr int RUN = (BM + 31) / 32;
  constexpr int TAP_OFF = 12 + RUN - 1;
  static_assert(16 * MTP * 2 <= VT_PITCH, "V^T row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = (float*)smem;                            // fp32 Q [BM][64] (aliases the dead staging ring)
  char* const QTH = smem;                              // Q~ hi / lo [BM] x 128 B, written over Q after the stencil
  char* const QTL = smem + KP_B;
  char* const KH = smem + QF_B;                        // K hi / lo [BM] x 128 B, slot XOR ((row >> 1) & 7)
  char* const KL = KH + KP_B;
  char* const VTH = KL + KP_B;                         // V^T hi / lo [64] x 512 B, slot XOR vt_swz(d)
  char* const VTL = VTH + 64 * VT_PITCH;

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
    if (MT & 1) {                                   // the keyless half of the last 32-key block: finite (zero) V^T columns
      const int t2 = tid - 256, dd = t2 >> 2, g = t2 & 3;
      const int vslot = 4 * (MT >> 1) + g;
      const int off = dd * VT_PITCH + ((vslot ^ vt_swz(dd)) << 4) + 8;
      *(uint2*)(VTH + off) = make_uint2(0u, 0u);
      *(uint2*)(VTL + off) = make_uint2(0u, 0u);
    }
  } else {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int dq = 16 * wave + 4 * lg;              // first of the lane's 4 Q / K columns
    const int dv = 16 * wave + lr;                  // the lane's V column
    float4 bq4 = make_float4(0.f, 0.f, 0.f, 0.f), bk4 = bq4;
    float bv1 = 0.f;
    if (bqkv) {
      bq4 = *(const float4*)(bqkv + head * HD + dq);
      bk4 = *(const float4*)(bqkv + D + head * HD + dq);
      bv1 = bqkv[2 * D + head * HD + dv];
    }
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
      // Q and K tiles with the operand roles swapped (a lane ends up with 4 consecutive columns of a token row), the
      // V tile with the roles kept (4 consecutive TOKENS of one column: what V^T wants)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        acc[i][0] = mfma3(bh[0], bl[0], ah[i], al[i], acc[i][0]);
        acc[i][1] = mfma3(bh[1], bl[1], ah[i], al[i], acc[i][1]);
        acc[i][2] = mfma3(ah[i], al[i], bh[2], bl[2], acc[i][2]);
      }
    }
    // ================================================================ phase 2: Q (fp32), K and V^T as (hi, lo) images
    __syncthreads();                                // every wave is done with the staging ring
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = i * 16 + lr;
      *(float4*)((char*)Qs + m * 256 + (((dq >> 2) ^ (m & 15)) << 4)) =
          make_float4((acc[i][0][0] + bq4.x) * q_scale, (acc[i][0][1] + bq4.y) * q_scale,
                      (acc[i][0][2] + bq4.z) * q_scale, (acc[i][0][3] + bq4.w) * q_scale);
      uint2 hi, lo;
      split4(acc[i][1][0] + bk4.x, acc[i][1][1] + bk4.y, acc[i][1][2] + bk4.z, acc[i][1][3] + bk4.w, hi, lo);
      const int koff = m * ROWB + (((dq >> 3) ^ ((m >> 1) & 7)) << 4) + ((dq & 4) << 1);
      *(uint2*)(KH + koff) = hi;
      *(uint2*)(KL + koff) = lo;
      // tokens 16 i + 4 lg + r -> positions 32 (i / 2) + 8 lg + 4 (i % 2) + r of V^T row dv
      split4(acc[i][2][0] + bv1, acc[i][2][1] + bv1, acc[i][2][2] + bv1, acc[i][2][3] + bv1, hi, lo);
      const int voff = dv * VT_PITCH + (((4 * (i >> 1) + lg) ^ vt_swz(dv)) << 4) + ((i & 1) << 3);
      *(uint2*)(VTH + voff) = hi;
      *(uint2*)(VTL + voff) = lo;
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
        if (m < BM) {
          uint2 hi, lo;
          split4(out[o].x, out[o].y, out[o].z, out[o].w, hi, lo);
          const int qoff = m * ROWB + (((s >> 1) ^ ((m >> 1) & 7)) << 4) + ((s & 1) << 3);
          *(uint2*)(QTH + qoff) = hi;
          *(uint2*)(QTL + qoff) = lo;
        }
      }
    }
  }
  __syncthreads();

  // ================================================================== phase 4: attention from LDS (split operands)
  for (int t = wave; t < MT; t += 8) {
    const int i0 = t * 16;
    if (i0 >= P) break;
    bf16x8 bqh[2], bql[2];
    {
      const int m = i0 + lr;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int off = m * ROWB + (((4 * kk + lg) ^ ((m >> 1) & 7)) << 4);
        bqh[kk] = *(const bf16x8*)(QTH + off);
        bql[kk] = *(const bf16x8*)(QTL + off);
      }
    }
    f32x4 s[MTP];
#pragma unroll
    for (int jt = 0; jt < MTP; ++jt) s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) {
        const int row = jt * 16 + lr;
        const int off = row * ROWB + (((4 * kk + lg) ^ ((row >> 1) & 7)) << 4);
        s[jt] = mfma3(*(const bf16x8*)(KH + off), *(const bf16x8*)(KL + off), bqh[kk], bql[kk], s[jt]);
      }
    // s[jt][r] = log2e * score(query i0 + lr, key 16 jt + 4 lg + r)
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
    cmax = max_xor32(max_xor16(cmax));             // VALU lane swaps (common.h), not ds_bpermute
    float psum = 0.f;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[jt][r] - cmax);
        s[jt][r] = p;
        psum += p;
      }
    psum = sum_xor32(sum_xor16(psum));
    const float inv = 1.0f / psum;                  // of THIS lane's query (lr)
    // O^T = V^T P^T: A = V^T rows d = 4 a + c (a = lr), B = P^T straight from the score registers, both as (hi, lo)
    f32x4 oacc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < MTP / 2; ++b) {
      bf16x8 ph, pl;
      split8(s[2 * b], s[2 * b + 1], ph, pl);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd = 4 * lr + c;
        const int off = dd * VT_PITCH + (((4 * b + lg) ^ ((lr ^ (c << 2)) & 15)) << 4);
        oacc[c] = mfma3(*(const bf16x8*)(VTH + off), *(const bf16x8*)(VTL + off), ph, pl, oacc[c]);
      }
    }
    // oacc[c][r] = O[query i0 + lr][d = 16 lg + 4 r + c]: 16 consecutive columns -> 16 hi + 16 lo values of the split image
    const int i = i0 + lr;
    if (i < P) {
      uint2 h[4], l[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        split4(oacc[0][r] * inv, oacc[1][r] * inv, oacc[2][r] * inv, oacc[3][r] * inv, h[r], l[r]);
      const int c0 = head * HD + 16 * lg;
      char* dst = O + (size_t)(row0 + i) * D * 4 + (c0 >> 5) * 128 + (c0 & 31) * 2;
      *(uint4*)dst = make_uint4(h[0].x, h[0].y, h[1].x, h[1].y);
      *(uint4*)(dst + 16) = make_uint4(h[2].x, h[2].y, h[3].x, h[3].y);
      *(uint4*)(dst + 64) = make_uint4(l[0].x, l[0].y, l[1].x, l[1].y);
      *(uint4*)(dst + 80) = make_uint4(l[2].x, l[2].y, l[3].x, l[3].y);
    }
  }
}

template <int MT>
hipError_t launch_mt(const char* U, const char* W, const float* bqkv, const float* pe_w, char* O, int n_regions, int P,
                     int D, int heads, int epeg_k, hipStream_t st) {
  constexpr int BM = 16 * MT;
  constexpr size_t RING = (size_t)3 * (BM + BN) * ROWB, TILES = (size_t)BM * 512 + 2 * 64 * VT_PITCH;
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
  static const bool off = rrt_tune_env("RRT_NO_FUSED_X3") != nullptr;
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
