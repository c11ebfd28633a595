// crmsa.hip -- cross-region attention (CR-MSA) around the small inner MSA, plus the
// final LayerNorm epilogue.  All HBM/L2-streaming kernels (no matrix cores: k <= 8).
//
// Replaces CrossRegionAttntion.forward, modules/rmsa.py:303-335, and the tail of
// RRTEncoder.forward, modules/rrt.py:190-195:
//   logits  Lg[r,n,p] = <LN(x1)[token(r,p)], phi[:,n]>           (pad tokens: v = 0 -> Lg = 0)
//   combine C  = softmax_p(Lg)          dispatch Dk = softmax_n(Lg)
//   M = (Lg - min_p) / (max_p - min_p + 1e-8)
//   rep[n,r,:] = sum_p C[r,n,p] v[r,p,:]        -> inner MSA (linear / region_attn / linear) -> rep2
//   out[r,p,:] = sum_n M*Dk [r,n,p] rep2[n,r,:] ;  x2 = x1 + out ; y = LN(x2 (+x0))
// The reference materialises [R,k,P,D] three times (56 MB each at N=9000); here the
// combine is a k x P x D contraction per region and the dispatch a k-term axpy per token.
//
//   crmsa_logits_kernel    : 1 wave / token.  LN statistics + k dot products; writes
//                            mean/rstd [L,2] and Lg in REGION-MAJOR order [Np8, k].
//   crmsa_combine_kernel   : 1 block / (region, 64-column slab).  Region softmax/min/max
//                            statistics from Lg, the per-token dispatch weights M*Dk [Np8, k],
//                            then the weighted row sum over the region's P tokens.
//   crmsa_dispatch_ln_kernel: 2 tokens / wave.  k-term axpy + residual (+shortcut) + LayerNorm.
#include "internal.h"

namespace {

constexpr int KMAX = RRT_MAX_CRMSA_K;

// four fp32 values -> four 16-bit values (prec 1 bf16 / 2 fp16), the 16-bit copy of the representatives
template <int PREC>
__device__ __forceinline__ uint2 r4_pack4(float4 v) {
  if constexpr (PREC == 2) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 r; r[0] = (_Float16)v.x; r[1] = (_Float16)v.y; r[2] = (_Float16)v.z; r[3] = (_Float16)v.w;
    return __builtin_bit_cast(uint2, r);
  } else {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    b4 r; r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    return __builtin_bit_cast(uint2, r);
  }
}


// rows handled by one wave (independent loads in flight per wave = RW * NV float4)
constexpr int RW = 2;             // (4 until the column guards went: 9.1 -> 7.6 us at N = 9000)
#ifndef RRT_RW_DISPATCH
#define RRT_RW_DISPATCH 1
#endif
constexpr int RW_DISPATCH = RRT_RW_DISPATCH;    // (2 until the column guards went: 10.0 -> 9.0 us; round 5, with every request of both
                                                 // tokens hoisted and the representatives' rows shared: 8.8 against 7.9 -- more waves win)

// FULL: dim == NV * 256 exactly (every lane's columns exist): no column guards in the row loops
template <int NV, bool FULL>
__global__ __launch_bounds__(256) void crmsa_logits_kernel(const float* __restrict__ x1,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ phi,
                                                           float* __restrict__ mean_rstd,
                                                           float* __restrict__ logits, int dim, int k,
                                                           GridDev g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* phi_t = (float*)smem;                    // [k][dim]: phi transposed, one float4 read per (n, chunk)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int idx = threadIdx.x; idx < dim * k; idx += 256) {
    int d = idx / k, n = idx - d * k;
    phi_t[n * dim + d] = phi[idx];
  }
  __syncthreads();
  // persistent: phi is staged once per block, then the block grid-strides over groups of 4*RW rows
  const int ngroups = (g.Np + 4 * RW - 1) / (4 * RW);
  // the NEXT group's rows are requested before this group's are worked on (round 4: a trip was one exposed memory round
  // trip per RW rows -- 3.5 TB/s on bags whose regions take this kernel); rows past the bag re-read row 0 (unconditional
  // loads keep the request order straight-line) and are zeroed after they land
  float4 nxt[RW][NV];
  auto request = [&](const int grp_) {
    const int tt0 = (grp_ * 4 + wave) * RW;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int t = tt0 + i;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        nxt[i][v] = *(const float4*)(x1 + (size_t)(t < g.L ? t : 0) * dim + ((FULL || c < dim) ? c : 0));
      }
    }
  };
  if ((int)blockIdx.x < ngroups) request(blockIdx.x);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
  const int t0 = (grp * 4 + wave) * RW;
  float4 r[RW][NV];
  float sum[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int t = t0 + i;
    sum[i] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int c = (v * 64 + lane) * 4;
      r[i][v] = (t < g.L && (FULL || c < dim)) ? nxt[i][v] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (grp + (int)gridDim.x < ngroups) request(grp + gridDim.x);
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int v = 0; v < NV; ++v) sum[i] += (r[i][v].x + r[i][v].y) + (r[i][v].z + r[i][v].w);
  const float inv_d = 1.0f / (float)dim;
  float mean[RW], sq[RW], rstd[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) mean[i] = wave_sum(sum[i]) * inv_d;
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    sq[i] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int c = (v * 64 + lane) * 4;
      if (FULL || c < dim) {
        float a = r[i][v].x - mean[i], b = r[i][v].y - mean[i], cc = r[i][v].z - mean[i], d = r[i][v].w - mean[i];
        sq[i] += (a * a + b * b) + (cc * cc + d * d);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < RW; ++i) rstd[i] = 1.0f / sqrtf(wave_sum(sq[i]) * inv_d + LN_EPS);
  // normalise in place: r <- LN(x1) rows
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (FULL || c < dim) {
      const float4 gm = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        r[i][v].x = (r[i][v].x - mean[i]) * rstd[i] * gm.x + bt.x;
        r[i][v].y = (r[i][v].y - mean[i]) * rstd[i] * gm.y + bt.y;
        r[i][v].z = (r[i][v].z - mean[i]) * rstd[i] * gm.z + bt.z;
        r[i][v].w = (r[i][v].w - mean[i]) * rstd[i] * gm.w + bt.w;
      }
    }
  }
  for (int n = 0; n < k; ++n) {
    float acc[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) acc[i] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int c = (v * 64 + lane) * 4;
      if (FULL || c < dim) {
        const float4 ph = *(const float4*)(phi_t + n * dim + c);
#pragma unroll
        for (int i = 0; i < RW; ++i)
          acc[i] += (r[i][v].x * ph.x + r[i][v].y * ph.y) + (r[i][v].z * ph.z + r[i][v].w * ph.w);
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const float a = wave_sum(acc[i]);
      const int t = t0 + i;
      if (lane == 0 && t < g.Np)      // pad tokens (t >= L) carry zero rows -> zero logits
        logits[(size_t)token_to_slot(t, g) * k + n] = (t < g.L) ? a : 0.f;
    }
  }
  if (lane < RW) {
    const int t = t0 + lane;
    if (t < g.L) {
      float m = mean[0], rs = rstd[0];
#pragma unroll
      for (int i = 1; i < RW; ++i)
        if (lane == i) { m = mean[i]; rs = rstd[i]; }
      mean_rstd[2 * (size_t)t] = m;
      mean_rstd[2 * (size_t)t + 1] = rs;
    }
  }
  }
}

// One block per (region, 64-column slab): phase 1 builds the combine coefficients
//   W[n][p] = softmax_p(Lg)[n,p] * rstd_p   (0 for pad tokens), c0[n] = sum_p W*mean_p, c1[n] = sum_p C
// in LDS; phase 2 is the [k x P] . [P x 64] contraction over raw x1 rows with 16 row groups x 16
// float4 column lanes, every thread's row loads independent (deep memory-level parallelism);
// LN's affine is applied once at the end:  rep = gamma * (W.X1 - c0) + beta * c1.
template <bool VNORM>   // VNORM: x1 is already LN(x1) in region-major order [Np8, dim] (crmsa_mlp path)
__global__ __launch_bounds__(256) void crmsa_combine_kernel(const float* __restrict__ x1,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean_rstd,
                                                            const float* __restrict__ logits,
                                                            float* __restrict__ wdisp,
                                                            float* __restrict__ rep, uint16_t* __restrict__ rep16,
                                                            int prec16, int dim, int k, GridDev g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Wc = (float*)smem;                         // [P][KMAX] combine coefficient x rstd
  int* tok = (int*)(Wc + (size_t)g.P * KMAX);       // [P] token index or -1 (pad)
  float4* part = (float4*)(tok + ((g.P + 3) & ~3)); // [16 row groups][KMAX][16 col lanes]
  __shared__ float s_stat[KMAX][3];
  __shared__ float s_c0[KMAX][4], s_c1[KMAX][4];    // per-wave partials of c0, c1
  const int reg = blockIdx.x, slab = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = g.Rt;
  const float* lg = logits + (size_t)reg * g.P * k;

  // region statistics: wave n handles representative n (k <= 8, 4 waves -> 2 rounds)
  for (int n = wave; n < k; n += 4) {
    float mx = -3.0e38f, mn = 3.0e38f;
    for (int p = lane; p < g.P; p += 64) {
      float v = lg[(size_t)p * k + n];
      mx = fmaxf(mx, v);
      mn = fminf(mn, v);
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    float se = 0.f;
    for (int p = lane; p < g.P; p += 64) se += __expf(lg[(size_t)p * k + n] - mx);
    se = wave_sum(se);
    if (lane == 0) { s_stat[n][0] = mx; s_stat[n][1] = mn; s_stat[n][2] = 1.0f / se; }
  }
  __syncthreads();
  // dispatch weights of this region's tokens (one slab does it):
  //   wdisp[slot][n] = minmax_p(Lg)[n,p] * softmax_n(Lg)[n,p]      (rmsa.py:310-314, :324-325)
  if (slab == 0) {
    for (int p = tid; p < g.P; p += 256) {
      float v[KMAX], e[KMAX];
      float mx = -3.0e38f;
#pragma unroll
      for (int n = 0; n < KMAX; ++n)
        if (n < k) { v[n] = lg[(size_t)p * k + n]; mx = fmaxf(mx, v[n]); }
      float se = 0.f;
#pragma unroll
      for (int n = 0; n < KMAX; ++n)
        if (n < k) { e[n] = __expf(v[n] - mx); se += e[n]; }
      const float inv = 1.0f / se;
#pragma unroll
      for (int n = 0; n < KMAX; ++n)
        if (n < k)
          wdisp[((size_t)reg * g.P + p) * k + n] =
              (v[n] - s_stat[n][1]) / (s_stat[n][0] - s_stat[n][1] + 1e-8f) * (e[n] * inv);
    }
  }
  // phase 1: coefficients
  {
    const int ri = reg / g.rs, rj = reg - ri * g.rs;
    float c0[KMAX], c1[KMAX];
#pragma unroll
    for (int n = 0; n < KMAX; ++n) c0[n] = c1[n] = 0.f;
    for (int p = tid; p < g.P; p += 256) {
      int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
      int t = (ri * g.s + pi) * g.H + rj * g.s + pj;
      const bool real = t < g.L;
      tok[p] = real ? (VNORM ? reg * g.P + p : t) : -1;
      const float mean = (real && !VNORM) ? mean_rstd[2 * (size_t)t] : 0.f;
      const float rstd = real ? (VNORM ? 1.0f : mean_rstd[2 * (size_t)t + 1]) : 0.f;
#pragma unroll
      for (int n = 0; n < KMAX; ++n)
        if (n < k) {
          float c = real ? __expf(lg[(size_t)p * k + n] - s_stat[n][0]) * s_stat[n][2] : 0.f;
          Wc[p * KMAX + n] = c * rstd;
          c0[n] += c * rstd * mean;
          c1[n] += c;
        }
    }
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) {
        float a = wave_sum(c0[n]), b = wave_sum(c1[n]);
        if (lane == 0) { s_c0[n][wave] = a; s_c1[n][wave] = b; }
      }
  }
  __syncthreads();
  // phase 2: contraction over the region's rows
  const int cl = tid & 15, rg = tid >> 4;
  const int col = slab * 64 + cl * 4;
  float4 acc[KMAX];
#pragma unroll
  for (int n = 0; n < KMAX; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < dim) {
    // 8 independent rows per trip, requested unconditionally (rows that do not count re-read row 0 and meet weight 0):
    // four conditional loads per trip were four branches, a wait for everything in flight behind each (3.1 TB/s)
    constexpr int TR = 8;
    for (int p0 = rg; p0 < g.P; p0 += 16 * TR) {
      float4 xv[TR];
      int pp[TR];
#pragma unroll
      for (int u = 0; u < TR; ++u) {
        pp[u] = p0 + 16 * u;
        const int t = pp[u] < g.P ? tok[pp[u]] : -1;
        xv[u] = *(const float4*)(x1 + (size_t)(t >= 0 ? t : 0) * dim + col);
        if (t < 0) pp[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < TR; ++u) {
#pragma unroll
        for (int n = 0; n < KMAX; ++n)
          if (n < k) {
            const float w = pp[u] >= 0 ? Wc[pp[u] * KMAX + n] : 0.f;
            const float4 x = pp[u] >= 0 ? xv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[n].x += w * x.x; acc[n].y += w * x.y; acc[n].z += w * x.z; acc[n].w += w * x.w;
          }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < KMAX; ++n)
    if (n < k) part[(rg * KMAX + n) * 16 + cl] = acc[n];
  __syncthreads();
  // reduce the 16 row groups: thread (n, cl) for n < k
  for (int idx = tid; idx < k * 16; idx += 256) {
    int n = idx >> 4, c = idx & 15;
    int cc = slab * 64 + c * 4;
    if (cc >= dim) continue;
    float4 a = part[n * 16 + c];
#pragma unroll
    for (int q = 1; q < 16; ++q) {
      float4 b = part[(q * KMAX + n) * 16 + c];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const float c0 = (s_c0[n][0] + s_c0[n][1]) + (s_c0[n][2] + s_c0[n][3]);
    const float c1 = (s_c1[n][0] + s_c1[n][1]) + (s_c1[n][2] + s_c1[n][3]);
    float4 gm = make_float4(1.f, 1.f, 1.f, 1.f), bt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!VNORM) { gm = *(const float4*)(gamma + cc); bt = *(const float4*)(beta + cc); }
    float4 out;
    out.x = gm.x * (a.x - c0) + bt.x * c1;
    out.y = gm.y * (a.y - c0) + bt.y * c1;
    out.z = gm.z * (a.z - c0) + bt.z * c1;
    out.w = gm.w * (a.w - c0) + bt.w * c1;
    *(float4*)(rep + ((size_t)n * R + reg) * dim + cc) = out;   // rep [k, R, D]
    if (rep16)                                                  // reduced-precision modes: + the 16-bit A operand of the inner qkv GEMM
      *(uint2*)(rep16 + ((size_t)n * R + reg) * dim + cc) = prec16 == 2 ? r4_pack4<2>(out) : r4_pack4<1>(out);
  }
}

// ---- round 6: the two chip-wide kernels for regions of more than 144 tokens (N > ~10.4 k: BASELINE configs[3] and the large
// bags of configs[4]), dim = 512.  The round 1-5 pair above cost 8-9 us of FIXED time each on top of their bytes (N = 15000 /
// 30000: logits 13.1 / 17.8 us, combine 13.6 / 18.6 us for 30.7 / 61.4 MB -- 3.4 TB/s at best): the logits kernel was a
// persistent grid that staged phi through LDS behind a strided global read and then paid one exposed memory round trip per
// two rows; the combine kernel walked the region's logits three times from global memory before its first x1 row was
// requested, then fetched the rows in four dependent trips.  Here:
//   * crmsa_logits512_kernel: one wave per two rows, grid = rows / 8 (as ln_partition_kernel, which streams at 6.4 TB/s), the
//     lane's twelve gamma.phi products built in registers from direct (L2-hit) loads issued next to the row loads, the
//     constants B_n = sum_c beta_c phi_cn once per block; logits in the CENTERED form rstd * sum_c (x_c - mean) gamma_c phi_cn
//     + B_n (no cancellation at |mean| >> sigma);
//   * crmsa_combine512_kernel: the first eight x1 rows of every thread are requested at kernel ENTRY (their addresses are
//     geometry, not logits), the region's logits and (mean, rstd) are staged in LDS by one coalesced pass, every statistic is
//     taken from LDS, and the next trip's rows are in flight while the current one is contracted.
template <int KM>
__global__ __launch_bounds__(256) void crmsa_logits512_kernel(const float* __restrict__ x1, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ phi,
                                                              float* __restrict__ mean_rstd, float* __restrict__ logits,
                                                              GridDev g) {
  constexpr int DIM = 512, RW2 = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * 4 + wave) * RW2;
  if (t0 >= g.Np) return;
  float4 r[RW2][2];
#pragma unroll
  for (int i = 0; i < RW2; ++i) {
    const int t = t0 + i;
    const float* src = x1 + (size_t)(t < g.L ? t : 0) * DIM;      // rows past the bag: re-read row 0, never used
#pragma unroll
    for (int v = 0; v < 2; ++v) r[i][v] = *(const float4*)(src + (v * 64 + lane) * 4);
  }
  // gamma . phi of this lane's eight columns, B_n partial sums
  float gp[2][4][KM];
  float bsum[KM];
#pragma unroll
  for (int n = 0; n < KM; ++n) bsum[n] = 0.f;
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int c = (v * 64 + lane) * 4;
    const float4 gm4 = *(const float4*)(gamma + c), bt4 = *(const float4*)(beta + c);
    const float gm[4] = {gm4.x, gm4.y, gm4.z, gm4.w}, bt[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
    float pf[4 * KM];                              // phi[c .. c+3][0 .. KM): 4 KM contiguous floats = KM float4
#pragma unroll
    for (int j = 0; j < KM; ++j) {
      const float4 q = *(const float4*)(phi + (size_t)c * KM + 4 * j);
      pf[4 * j] = q.x; pf[4 * j + 1] = q.y; pf[4 * j + 2] = q.z; pf[4 * j + 3] = q.w;
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int n = 0; n < KM; ++n) {
        const float ph = pf[cc * KM + n];
        gp[v][cc][n] = gm[cc] * ph;
        bsum[n] += bt[cc] * ph;
      }
  }
  float Bn[KM];
#pragma unroll
  for (int n = 0; n < KM; ++n) Bn[n] = wave_sum(bsum[n]);
  const float inv_d = 1.0f / (float)DIM;
#pragma unroll
  for (int i = 0; i < RW2; ++i) {
    const int t = t0 + i;
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < 2; ++v) sum += (r[i][v].x + r[i][v].y) + (r[i][v].z + r[i][v].w);
    const float mean = wave_sum(sum) * inv_d;
    float sq = 0.f, d[KM];
#pragma unroll
    for (int n = 0; n < KM; ++n) d[n] = 0.f;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const float xc[4] = {r[i][v].x - mean, r[i][v].y - mean, r[i][v].z - mean, r[i][v].w - mean};
      sq += (xc[0] * xc[0] + xc[1] * xc[1]) + (xc[2] * xc[2] + xc[3] * xc[3]);
#pragma unroll
      for (int n = 0; n < KM; ++n)
        d[n] += (xc[0] * gp[v][0][n] + xc[1] * gp[v][1][n]) + (xc[2] * gp[v][2][n] + xc[3] * gp[v][3][n]);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
    float mine = 0.f;                              // lane n keeps logit n
#pragma unroll
    for (int n = 0; n < KM; ++n) {
      const float lg = rstd * wave_sum(d[n]) + Bn[n];
      mine = lane == n ? lg : mine;
    }
    if (t < g.Np) {
      if (lane < KM) logits[(size_t)token_to_slot(t, g) * KM + lane] = t < g.L ? mine : 0.f;   // pad tokens: zero rows -> zero logits
      if (lane == 0 && t < g.L) *(float2*)(mean_rstd + 2 * (size_t)t) = make_float2(mean, rstd);
    }
  }
}

template <int KM>
__global__ __launch_bounds__(256, 2) void crmsa_combine512_kernel(const float* __restrict__ x1, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ mean_rstd,
                                                               const float* __restrict__ logits, float* __restrict__ wdisp,
                                                               float* __restrict__ rep, uint16_t* __restrict__ rep16,
                                                               int prec16, GridDev g) {
  // TR rows per thread and trip, two trips in flight (a consumed while b lands): 16 float4 = 64 KB per block, two blocks per CU --
  // what the chip needs in flight per CU to stream at ~6 TB/s is ~47 KB.  (First version: TR = 16 -> 254 VGPRs, ONE block per CU,
  // 512 blocks in two rounds: 22.2 us at N = 30000 against 18.6 for the round-5 kernel.)
  constexpr int DIM = 512, TR = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lgs = (float*)smem;                                   // [P][KM] logits, then the combine coefficient x rstd in place
  float2* mr = (float2*)(lgs + (((size_t)g.P * KM + 3) & ~(size_t)3));   // [P] (mean, rstd) of the region's tokens, (0, 0) for pad
  float4* part = (float4*)(mr + ((g.P + 1) & ~1));             // [16 row groups][KM][16 column lanes]
  __shared__ float s_stat[KM][3];
  __shared__ float s_c0[KM][4], s_c1[KM][4];
  const int reg = blockIdx.x, slab = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = tid & 15, rg = tid >> 4;
  const int col = slab * 64 + cl * 4;
  const int ri = fdiv(reg, g.rs, g.inv_rs), rj = reg - ri * g.rs;
  auto tok_of = [&](int p) -> int {                            // token of the region's p-th slot, -1 for pad / past the region
    const int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
    const int t = (ri * g.s + pi) * g.H + rj * g.s + pj;
    return (p < g.P && t < g.L) ? t : -1;
  };
  // trip 0: this thread's first TR rows, requested before anything else
  float4 xa[TR], xb[TR];
  bool ra[TR], rb[TR];
#pragma unroll
  for (int u = 0; u < TR; ++u) {
    const int t = tok_of(rg + 16 * u);
    ra[u] = t >= 0;
    xa[u] = *(const float4*)(x1 + (size_t)(t >= 0 ? t : 0) * DIM + col);
  }
  // the region's logits and statistics -> LDS, one coalesced pass each
  const float* lg = logits + (size_t)reg * g.P * KM;
  for (int idx = tid; idx < g.P * KM; idx += 256) lgs[idx] = lg[idx];
  for (int p = tid; p < g.P; p += 256) {
    const int t = tok_of(p);
    mr[p] = t >= 0 ? *(const float2*)(mean_rstd + 2 * (size_t)t) : make_float2(0.f, 0.f);
  }
  __syncthreads();
  for (int n = wave; n < KM; n += 4) {                         // wave n: max, min, sum of exp of representative n
    float mx = -3.0e38f, mn = 3.0e38f;
    for (int p = lane; p < g.P; p += 64) {
      const float v = lgs[p * KM + n];
      mx = fmaxf(mx, v);
      mn = fminf(mn, v);
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    float se = 0.f;
    for (int p = lane; p < g.P; p += 64) se += __expf(lgs[p * KM + n] - mx);
    se = wave_sum(se);
    if (lane == 0) { s_stat[n][0] = mx; s_stat[n][1] = mn; s_stat[n][2] = 1.0f / se; }
  }
  __syncthreads();
  {
    float c0[KM], c1[KM];
#pragma unroll
    for (int n = 0; n < KM; ++n) c0[n] = c1[n] = 0.f;
    for (int p = tid; p < g.P; p += 256) {
      float v[KM];
#pragma unroll
      for (int n = 0; n < KM; ++n) v[n] = lgs[p * KM + n];
      if (slab == 0) {                                         // dispatch weights of the region's tokens (rmsa.py:310-314, :324-325)
        float mx = v[0];
#pragma unroll
        for (int n = 1; n < KM; ++n) mx = fmaxf(mx, v[n]);
        float e[KM], se = 0.f;
#pragma unroll
        for (int n = 0; n < KM; ++n) { e[n] = __expf(v[n] - mx); se += e[n]; }
        const float inv = 1.0f / se;
#pragma unroll
        for (int n = 0; n < KM; ++n)
          wdisp[((size_t)reg * g.P + p) * KM + n] =
              (v[n] - s_stat[n][1]) / (s_stat[n][0] - s_stat[n][1] + 1e-8f) * (e[n] * inv);
      }
      const float2 m = mr[p];                                  // pad: rstd = 0 -> coefficient 0
      const bool real = m.y != 0.f;
#pragma unroll
      for (int n = 0; n < KM; ++n) {
        const float c = real ? __expf(v[n] - s_stat[n][0]) * s_stat[n][2] : 0.f;
        lgs[p * KM + n] = c * m.y;
        c0[n] += c * m.y * m.x;
        c1[n] += c;
      }
    }
#pragma unroll
    for (int n = 0; n < KM; ++n) {
      const float a = wave_sum(c0[n]), b = wave_sum(c1[n]);
      if (lane == 0) { s_c0[n][wave] = a; s_c1[n][wave] = b; }
    }
  }
  __syncthreads();
  // contraction over the region's rows: trip q covers rows rg + 16 (TR q + u)
  float4 acc[KM];
#pragma unroll
  for (int n = 0; n < KM; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto request = [&](float4 (&xv)[TR], bool (&rv)[TR], int p0) {
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      const int t = tok_of(p0 + 16 * u);
      rv[u] = t >= 0;
      xv[u] = *(const float4*)(x1 + (size_t)(t >= 0 ? t : 0) * DIM + col);
    }
  };
  auto consume = [&](const float4 (&xv)[TR], const bool (&rv)[TR], int p0) {
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      const int p = p0 + 16 * u;
      const float4 x = rv[u] ? xv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < KM; ++n) {
        const float w = rv[u] ? lgs[p * KM + n] : 0.f;
        acc[n].x += w * x.x; acc[n].y += w * x.y; acc[n].z += w * x.z; acc[n].w += w * x.w;
      }
    }
  };
  for (int p0 = rg; p0 < g.P; p0 += 32 * TR) {                 // two trips per iteration: b requested while a is contracted
    const bool more = p0 + 16 * TR < g.P;
    if (more) request(xb, rb, p0 + 16 * TR);
    consume(xa, ra, p0);
    if (p0 + 32 * TR < g.P) request(xa, ra, p0 + 32 * TR);
    if (more) consume(xb, rb, p0 + 16 * TR);
  }
#pragma unroll
  for (int n = 0; n < KM; ++n) part[(rg * KM + n) * 16 + cl] = acc[n];
  __syncthreads();
  for (int idx = tid; idx < KM * 16; idx += 256) {             // reduce the 16 row groups: thread (n, column lane)
    const int n = idx >> 4, c = idx & 15;
    const int cc = slab * 64 + c * 4;
    float4 a = part[n * 16 + c];
#pragma unroll
    for (int q = 1; q < 16; ++q) {
      const float4 b = part[(q * KM + n) * 16 + c];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const float c0 = (s_c0[n][0] + s_c0[n][1]) + (s_c0[n][2] + s_c0[n][3]);
    const float c1 = (s_c1[n][0] + s_c1[n][1]) + (s_c1[n][2] + s_c1[n][3]);
    const float4 gm = *(const float4*)(gamma + cc), bt = *(const float4*)(beta + cc);
    float4 out;
    out.x = gm.x * (a.x - c0) + bt.x * c1;
    out.y = gm.y * (a.y - c0) + bt.y * c1;
    out.z = gm.z * (a.z - c0) + bt.z * c1;
    out.w = gm.w * (a.w - c0) + bt.w * c1;
    *(float4*)(rep + ((size_t)n * g.Rt + reg) * DIM + cc) = out;   // rep [k, R, D]
    if (rep16)
      *(uint2*)(rep16 + ((size_t)n * g.Rt + reg) * DIM + cc) = prec16 == 2 ? r4_pack4<2>(out) : r4_pack4<1>(out);
  }
}

// ---- combine from the projection slabs' row records (round 5) --------------------------------------------------------
// The last R-MSA layer's out-projection leaves, per (token, 64-column slab), the record (mean_64, M2_64, d_0 .. d_k-1),
// d_n = sum_c x1[c] gamma[c] phi[c, n] over the slab's columns (rmsa_fused.hip, proj_slab) -- LayerNorm 2's statistics and
// the logits' dot products taken from the x1 tiles while they were still in registers.  What is left of
// modules/rmsa.py:303-316 is ONE pass over x1 with no LayerNorm arithmetic in it and no hand-over between blocks:
//   block = (region, 64-column slab), as crmsa_combine_kernel.  Every block of a region re-derives the region's logits
//   from the records (P x D/64 x 8 floats, served by the L2: 37 KB at P = 144) -- Chan-merge of the slabs' (mean, M2),
//   logit_n = rstd (sum d_n - mean G_n) + B_n, G_n = sum_c gamma_c phi_cn, B_n = sum_c beta_c phi_cn (2 k numbers, by the
//   block's fourth wave while the other three merge their rows' records) -- then wave n does everything of representative
//   n (max, min, sum of exp, the combine coefficients c * rstd and the LayerNorm-fold sums c0, c1), and the block contracts
//   ITS 64 columns of the region's rows, whose loads were requested before any of that (they do not depend on it).
//   Slab 0 also writes the dispatch weights.
// 8 x the statistics work of one block per region, which is nothing (P x ~60 flops), instead of the write-through hand-over
// of crmsa_region4_kernel's quarters (14.6 us at N = 9000 for one 18 MB read: a chain of ten latencies).
// KM (4 / 8): representatives the instantiation carries (k <= KM; the table and phi-products of n >= k are zero) -- with a
// run-time k every per-n loop was a chain of scalar branches: 3.8 K instructions, 13 us (traced: 23 K cycles per wave).
template <int TR, int KM, bool COAL = false>   // x1 rows in flight per thread (16 row groups x 16 column lanes per block); representatives;
                                               // COAL: the records arrive coalesced and pass through LDS (see below)
__global__ __launch_bounds__(256) void crmsa_combine_parts_kernel(const float* __restrict__ x1, const float* __restrict__ part,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ phi, float* __restrict__ wdisp,
                                                                  float* __restrict__ rep, uint16_t* __restrict__ rep16,
                                                                  int prec16, int dim, int k, int n_slabs, int al16,
                                                                  GridDev g) {
  constexpr int MAXS = 8;                           // slabs whose records a thread keeps in flight at once (dim <= 512: all)
  constexpr int NL = KM / 4;                        // float4s of a row's logits / coefficients
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* s_lg = (float4*)smem;                     // [P][NL] logits, then the combine coefficients c * rstd in place
  float2* s_mr = (float2*)(s_lg + (size_t)g.P * NL);   // [P] mean, rstd (rstd = 0: pad token)
  float4* s_part = (float4*)(s_mr + ((g.P + 1) & ~1));  // [16 row groups][KM + 1][16 col lanes]
  __shared__ float s_stat[KM][5];                   // max, min, 1 / sum exp, c0, c1
  const int reg = blockIdx.x, slab = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = g.Rt, nf = (2 + k + 3) >> 2;        // float4s per record
  const int ri = fdiv(reg, g.rs, g.inv_rs), rj = reg - ri * g.rs;
  RRT_TRACE_INIT((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave);
  RRT_TRACE_MARK();                                 // [1] entry
  auto token_of = [&](const int p) {
    const int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
    const int t = (ri * g.s + pi) * g.H + rj * g.s + pj;
    return (p < g.P && t < g.L) ? t : -1;
  };
  // ---- every load of the kernel is requested up front, in the order it is consumed (the vector memory counter is in
  // order): this thread's row records, the G / B table, the contraction's rows, the affine of the last stage
  const int t_mine = token_of(tid);                 // row `tid` of the region (regions of > 256 rows: the loop below)
  constexpr int NFM = KM <= 2 ? 1 : KM <= 6 ? 2 : 3;
  float4 rc[MAXS][NFM];
  // The records of a row are n_slabs x nf float4s in a row of their own: a thread fetching ITS row's sixteen pieces makes every
  // wave-instruction touch 64 different 128-byte lines for 16 bytes each (traced: 6.1 K cycles until the last request of the
  // block was out -- the texture path, not the data).  Where a row's record is exactly 16 float4s (dim = 512, k = 3 / 4) and
  // the region's rows are all in the first batch (P <= 16 TR), the block fetches them the way it fetches x1 -- thread
  // (row group, column lane) takes piece `cl` of rows rg, rg + 16, ...: 256 contiguous bytes per row -- and passes them
  // through LDS (pitch 17 float4s); `coal` is decided (and the LDS provided) by the launcher.
  constexpr int RPITCH = 17;
  constexpr bool coal = COAL && KM == 4;            // (a run-time switch sent the staging registers to scratch: 17 us)
  float4* const s_rec = (float4*)(smem + (size_t)g.P * KM * 4 * 2 + (size_t)((g.P + 1) & ~1) * 8 + (size_t)16 * (KM + 1) * 16 * 16);
  float4 cq[TR];
  if constexpr (coal) {
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      const int t = token_of((tid >> 4) + 16 * u);
      cq[u] = ((const float4*)part)[(size_t)(t < 0 ? 0 : t) * 16 + (tid & 15)];
    }
  } else {
    const float4* r = (const float4*)(part + (size_t)(t_mine < 0 ? 0 : t_mine) * n_slabs * (4 * nf));
#pragma unroll
    for (int c = 0; c < MAXS; ++c)
#pragma unroll
      for (int f = 0; f < NFM; ++f) rc[c][f] = r[(c < n_slabs ? c : 0) * nf + (f < nf ? f : 0)];
  }
  // G_n = sum_c gamma_c phi_cn, B_n = sum_c beta_c phi_cn: wave 3's job (regions of <= 192 rows leave it idle in the row
  // pass; larger ones pay ~200 cycles).  Its loads go out with everybody's, the sums meet the rows' statistics at the barrier
  __shared__ float s_gb[2 * KM];
  float ag[KM], ab[KM];
  if (wave == 3) {
    // lane l: columns 8 l .. 8 l + 7 (dim <= 512), gamma / beta as two float4 each and the 8 k phi values as 2 k float4 --
    // every request out before the first use (a loop over columns with its loads inside was eight dependent round trips:
    // the block waited ~2.5 us for this wave, 8.2 -> 10.8 us per launch)
#pragma unroll
    for (int n = 0; n < KM; ++n) ag[n] = ab[n] = 0.f;
    const int c0 = 8 * lane;
    if (!(al16 & 1)) {                               // parameters not on 16-byte boundaries: the plain loop
      for (int c = lane; c < dim; c += 64) {
        const float gm = gamma[c], bt = beta[c];
#pragma unroll
        for (int n = 0; n < KM; ++n) {
          const float ph = n < k ? phi[(size_t)c * k + n] : 0.f;
          ag[n] += gm * ph;
          ab[n] += bt * ph;
        }
      }
    } else if (c0 < dim) {
      float gmv[8], btv[8], phv[8 * KM];
      {
        const float4 g0 = *(const float4*)(gamma + c0), g1 = *(const float4*)(gamma + c0 + 4);
        const float4 b0 = *(const float4*)(beta + c0), b1 = *(const float4*)(beta + c0 + 4);
        gmv[0] = g0.x; gmv[1] = g0.y; gmv[2] = g0.z; gmv[3] = g0.w; gmv[4] = g1.x; gmv[5] = g1.y; gmv[6] = g1.z; gmv[7] = g1.w;
        btv[0] = b0.x; btv[1] = b0.y; btv[2] = b0.z; btv[3] = b0.w; btv[4] = b1.x; btv[5] = b1.y; btv[6] = b1.z; btv[7] = b1.w;
        const float4* p4 = (const float4*)(phi + (size_t)c0 * k);      // 8 k floats, 16-byte aligned (32 k bytes per lane)
#pragma unroll
        for (int q = 0; q < 2 * KM; ++q)
          if (q < 2 * k) {
            const float4 v = p4[q];
            phv[4 * q] = v.x; phv[4 * q + 1] = v.y; phv[4 * q + 2] = v.z; phv[4 * q + 3] = v.w;
          }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int n = 0; n < KM; ++n)
          if (n < k) {
            // phi[(c0 + j) * k + n]: element j * k + n of the lane's 8 k values (k is a run-time 1 .. KM: a select over the
            // compile-time candidates instead of a dynamic register index)
            float ph = 0.f;
#pragma unroll
            for (int kk = 1; kk <= KM; ++kk)
              if (k == kk) ph = phv[j * kk + n < 8 * KM ? j * kk + n : 0];
            ag[n] += gmv[j] * ph;
            ab[n] += btv[j] * ph;
          }
    }
  }
  const int cl = tid & 15, rg = tid >> 4;
  const int col = slab * 64 + cl * 4;
  float4 xv[TR];
  int pp[TR];
#pragma unroll
  for (int u = 0; u < TR; ++u) {
    const int p = rg + 16 * u;
    const int t = token_of(p);
    pp[u] = t < 0 ? -1 : p;
    xv[u] = *(const float4*)(x1 + (size_t)(t < 0 ? 0 : t) * dim + col);
  }
  const float4 fgm = *(const float4*)(gamma + col), fbt = *(const float4*)(beta + col);   // (the last stage: threads < 16 KM)
  RRT_TRACE_MARK();                                 // [2] every load requested
  // ---- the row's statistics from its records (Chan et al., equal counts: slabs of 64 columns); its logits once G, B are there
  const float inv_d = 1.0f / (float)dim;
  auto row_stats = [&](const float4 (&q)[MAXS][NFM], const int t, float& mean, float& rstd, float (&dn)[KM]) {
    float m = q[0][0].x, m2 = q[0][0].y;
    auto dval = [&](const int c, const int n) {       // d_n of slab c: float 2 + n of the record
      const int e = 2 + n;
      const float4 v = q[c][(e >> 2) < NFM ? (e >> 2) : 0];
      return (e & 3) == 0 ? v.x : (e & 3) == 1 ? v.y : (e & 3) == 2 ? v.z : v.w;
    };
#pragma unroll
    for (int n = 0; n < KM; ++n) dn[n] = dval(0, n);
#pragma unroll
    for (int c = 1; c < MAXS; ++c)
      if (c < n_slabs) {
        const float mb = q[c][0].x, qb = q[c][0].y;
        const float dl = mb - m, w = 1.0f / (float)(c + 1);        // (compile-time reciprocal)
        m += dl * w;
        m2 += qb + dl * dl * (64.0f * (float)c * w);
#pragma unroll
        for (int n = 0; n < KM; ++n) dn[n] += dval(c, n);
      }
    mean = t >= 0 ? m : 0.f;
    rstd = t >= 0 ? 1.0f / sqrtf(m2 * inv_d + LN_EPS) : 0.f;
  };
  auto put_row = [&](const int p, const int t, const float mean, const float rstd, const float (&dn)[KM]) {
    float lg[KM];
#pragma unroll
    for (int n = 0; n < KM; ++n) lg[n] = t >= 0 ? rstd * (dn[n] - mean * s_gb[n]) + s_gb[KM + n] : 0.f;   // pad tokens: zero logits
#pragma unroll
    for (int f = 0; f < NL; ++f) s_lg[p * NL + f] = make_float4(lg[4 * f], lg[4 * f + 1], lg[4 * f + 2], lg[4 * f + 3]);
    s_mr[p] = make_float2(mean, rstd);
  };
  float mean_mine = 0.f, rstd_mine = 0.f, dn_mine[KM];
  if (wave == 3) {
#pragma unroll
    for (int n = 0; n < KM; ++n) {
      const float a_ = wave_sum(ag[n]), b_ = wave_sum(ab[n]);
      if (lane == 0) { s_gb[n] = a_; s_gb[KM + n] = b_; }
    }
  }
  if constexpr (coal) {
#pragma unroll
    for (int u = 0; u < TR; ++u) s_rec[((tid >> 4) + 16 * u) * RPITCH + (tid & 15)] = cq[u];
    lds_sync();
    if (tid < g.P) {
#pragma unroll
      for (int c = 0; c < MAXS; ++c)
#pragma unroll
        for (int f = 0; f < NFM; ++f) rc[c][f] = s_rec[tid * RPITCH + (c * 2 + f) % 16];
    }
  }
  if (tid < g.P) row_stats(rc, t_mine, mean_mine, rstd_mine, dn_mine);
  lds_sync();                                         // G, B published
  if (tid < g.P) put_row(tid, t_mine, mean_mine, rstd_mine, dn_mine);
  for (int p = tid + 256; p < g.P; p += 256) {        // regions of more than 256 rows (bags of > 16 k tokens)
    const int t = token_of(p);
    const float4* r = (const float4*)(part + (size_t)(t < 0 ? 0 : t) * n_slabs * (4 * nf));
    float4 q[MAXS][NFM];
#pragma unroll
    for (int c = 0; c < MAXS; ++c)
#pragma unroll
      for (int f = 0; f < NFM; ++f) q[c][f] = r[(c < n_slabs ? c : 0) * nf + (f < nf ? f : 0)];
    float mean, rstd, dn[KM];
    row_stats(q, t, mean, rstd, dn);
    put_row(p, t, mean, rstd, dn);
  }
  RRT_TRACE_MARK();                                 // [3] logits in LDS (records landed)
  lds_sync();
  RRT_TRACE_MARK();                                 // [4] barrier
  // ---- wave n owns representative n: max / min / sum of exp over the region, the combine coefficients c * rstd of every
  // row (in place of its logit: only this wave touches column n) and the two LayerNorm-fold sums (Identity 3)
  // (slab 0 keeps the logits for the dispatch weights: its coefficients go to a second table behind the partials)
  const float* lgs = (const float*)s_lg;
  float* const wtab = slab != 0 ? (float*)s_lg : (float*)(s_part + 16 * (KM + 1) * 16);
  for (int n = wave; n < KM; n += 4) {
    if (n >= k) {                                     // (wave-uniform) columns nobody owns: zero weights
      for (int p = lane; p < g.P; p += 64) wtab[p * KM + n] = 0.f;
      continue;
    }
    float mx = -3.0e38f, mn = 3.0e38f;
    for (int p = lane; p < g.P; p += 64) {
      const float v = lgs[p * KM + n];
      mx = fmaxf(mx, v);
      mn = fminf(mn, v);
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    float se = 0.f;
    for (int p = lane; p < g.P; p += 64) se += __expf(lgs[p * KM + n] - mx);
    se = wave_sum(se);
    const float inv = 1.0f / se;
    float c0 = 0.f, c1 = 0.f;
    for (int p = lane; p < g.P; p += 64) {
      const float2 mr = s_mr[p];
      const float c = mr.y != 0.f ? __expf(lgs[p * KM + n] - mx) * inv : 0.f;   // pad rows: in the softmax sum, not in the contraction
      c0 += c * mr.y * mr.x;
      c1 += c;
      wtab[p * KM + n] = c * mr.y;
    }
    c0 = wave_sum(c0);
    c1 = wave_sum(c1);
    if (lane == 0) { s_stat[n][0] = mx; s_stat[n][1] = mn; s_stat[n][2] = inv; s_stat[n][3] = c0; s_stat[n][4] = c1; }
  }
  RRT_TRACE_MARK();                                 // [5] region statistics + coefficients
  lds_sync();
  RRT_TRACE_MARK();                                 // [6] barrier
  // ---- contraction over the region's rows (this block's 64 columns)
  const float4* wt = (const float4*)wtab;
  float4 acc[KM];
#pragma unroll
  for (int n = 0; n < KM; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p0 = rg;; ) {
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      const int pc = pp[u] >= 0 ? pp[u] : 0;
      const float4 x = pp[u] >= 0 ? xv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int f = 0; f < NL; ++f) {
        const float4 w4 = wt[pc * NL + f];
        acc[4 * f].x += w4.x * x.x; acc[4 * f].y += w4.x * x.y; acc[4 * f].z += w4.x * x.z; acc[4 * f].w += w4.x * x.w;
        acc[4 * f + 1].x += w4.y * x.x; acc[4 * f + 1].y += w4.y * x.y; acc[4 * f + 1].z += w4.y * x.z; acc[4 * f + 1].w += w4.y * x.w;
        acc[4 * f + 2].x += w4.z * x.x; acc[4 * f + 2].y += w4.z * x.y; acc[4 * f + 2].z += w4.z * x.z; acc[4 * f + 2].w += w4.z * x.w;
        acc[4 * f + 3].x += w4.w * x.x; acc[4 * f + 3].y += w4.w * x.y; acc[4 * f + 3].z += w4.w * x.z; acc[4 * f + 3].w += w4.w * x.w;
      }
    }
    p0 += 16 * TR;
    if (p0 >= g.P) break;
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      const int p = p0 + 16 * u;
      const int t = token_of(p);
      pp[u] = t < 0 ? -1 : p;
      xv[u] = *(const float4*)(x1 + (size_t)(t < 0 ? 0 : t) * dim + col);
    }
  }
#pragma unroll
  for (int n = 0; n < KM; ++n) s_part[(rg * (KM + 1) + n) * 16 + cl] = acc[n];
  RRT_TRACE_MARK();                                 // [7] contraction (x1 rows landed)
  lds_sync();
  RRT_TRACE_MARK();                                 // [8] barrier
  if (tid < k * 16) {
    const int n = tid >> 4;                           // (cl = tid & 15: this thread's columns are `col`)
    float4 a = s_part[n * 16 + cl];
#pragma unroll
    for (int q = 1; q < 16; ++q) {
      const float4 b = s_part[(q * (KM + 1) + n) * 16 + cl];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const float c0 = s_stat[n][3], c1 = s_stat[n][4];
    float4 out;
    out.x = fgm.x * (a.x - c0) + fbt.x * c1;
    out.y = fgm.y * (a.y - c0) + fbt.y * c1;
    out.z = fgm.z * (a.z - c0) + fbt.z * c1;
    out.w = fgm.w * (a.w - c0) + fbt.w * c1;
    *(float4*)(rep + ((size_t)n * R + reg) * dim + col) = out;   // rep [k, R, D]
    if (rep16) *(uint2*)(rep16 + ((size_t)n * R + reg) * dim + col) = prec16 == 2 ? r4_pack4<2>(out) : r4_pack4<1>(out);
  }
  RRT_TRACE_MARK();                                 // [9] representatives stored
  // dispatch weights of the region's tokens (slab 0): minmax_p(Lg)[n, p] * softmax_n(Lg)[n, p]  (rmsa.py:310-314, :324-325)
  if (slab == 0) {
    for (int p = tid; p < g.P; p += 256) {
      float v[KM], e[KM];
      float mx = -3.0e38f;
#pragma unroll
      for (int n = 0; n < KM; ++n) {
        v[n] = lgs[p * KM + n];
        if (n < k) mx = fmaxf(mx, v[n]);
      }
      float se = 0.f;
#pragma unroll
      for (int n = 0; n < KM; ++n) {
        e[n] = n < k ? __expf(v[n] - mx) : 0.f;
        se += e[n];
      }
      const float inv = 1.0f / se;
#pragma unroll
      for (int n = 0; n < KM; ++n)
        if (n < k)
          wdisp[((size_t)reg * g.P + p) * k + n] = (v[n] - s_stat[n][1]) / (s_stat[n][0] - s_stat[n][1] + 1e-8f) * (e[n] * inv);
    }
    RRT_TRACE_MARK();                               // [10] dispatch weights written
  }
}

// ---- logits + combine in ONE pass over x1 (dim = 512, P <= 16 * NR_MAX, k <= 3) -------------------------------------
// One block of 16 waves per region.  Wave w owns the region's rows w, w + 16, ... (<= NR_MAX = 9 of them, two float4
// per lane each: 72 registers) and issues every load at once -- one memory round trip for the whole region, and the
// rows are read from HBM exactly once per bag (the two-kernel form read them twice and paid two launches, two ramps
// and two drains: 7.6 + 8.8 us).  Then, per row and exactly as crmsa_logits_kernel does it: two-pass LayerNorm
// statistics, normalised row, k dot products -> (mean, rstd, logits) in LDS.  After a barrier the region statistics
// and the combine coefficients are built as in crmsa_combine_kernel, and each wave contracts ITS rows, still in
// registers, into a partial rep [k x 512]; the 16 partials are summed through LDS in a fixed order (waves w + 8 into
// w, then eight per column) and LayerNorm's affine is applied once:  rep = gamma * (W.X1 - c0) + beta * c1.
constexpr int REGION_NR_MAX = 9;
constexpr int REGION_KMAX = 3;
// GK (round 6): 0 = the round-1 arithmetic (phi through LDS, normalised rows, one wave_sum per value, run-time k <= 3);
// 1, 2, 3 = k exactly, gamma . phi of the lane's eight columns in registers, centred logits, the rows' wave totals four at a
// time (wave_sum4) in chunks of three rows -- 45 wave reductions per wave become 12 groups -- and the representatives also as
// 16-bit values (rep16, may be null): the form the forward takes with SEVERAL bags in flight in exact fp32, where a front
// of 64 blocks leaves three quarters of the chip to the other bags' fused R-MSA launches (api.hip)
template <int GK>
__global__ __launch_bounds__(1024) void crmsa_region_kernel(const float* __restrict__ x1,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ phi,
                                                            float* __restrict__ mean_rstd,
                                                            float* __restrict__ logits,
                                                            float* __restrict__ wdisp,
                                                            float* __restrict__ rep, uint16_t* __restrict__ rep16, int prec16,
                                                            int k, GridDev g) {
  constexpr int DIM = 512, NR = REGION_NR_MAX, KM = REGION_KMAX;
  constexpr bool GPR = GK > 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* phi_t = (float*)smem;                      // [KM][DIM]
  float* s_lg = phi_t + KM * DIM;                   // [P][KM] logits
  float* s_mr = s_lg + NR * 16 * KM;                // [P][2] mean, rstd (rstd = 0: pad token)
  float* s_w = s_mr + NR * 16 * 2;                  // [P][KM] combine coefficient x rstd
  float4* s_part = (float4*)(s_w + NR * 16 * KM);   // [8][KM][128] float4: partial rep of waves w (+ w + 8)
  __shared__ float s_stat[KM][3];
  __shared__ float s_c0[KM][16], s_c1[KM][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int reg = blockIdx.x, R = g.Rt;
  const int ri = reg / g.rs, rj = reg - ri * g.rs;
  if constexpr (!GPR) {
    for (int idx = tid; idx < DIM * k; idx += 1024) {
      const int d = idx / k, n = idx - d * k;
      phi_t[n * DIM + d] = phi[idx];
    }
  }
  // ---- this wave's rows: every load in flight at once
  float4 r[NR][2];
  int tokv[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int p = wave + 16 * j;
    int t = -1;
    if (p < g.P) {
      const int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
      t = (ri * g.s + pi) * g.H + rj * g.s + pj;
      if (t >= g.L) t = -1;
    }
    tokv[j] = t;
    const float* src = x1 + (size_t)(t < 0 ? 0 : t) * DIM;
    r[j][0] = *(const float4*)(src + lane * 4);
    r[j][1] = *(const float4*)(src + 256 + lane * 4);
  }
  const float4 gm0 = *(const float4*)(gamma + lane * 4), gm1 = *(const float4*)(gamma + 256 + lane * 4);
  const float4 bt0 = *(const float4*)(beta + lane * 4), bt1 = *(const float4*)(beta + 256 + lane * 4);
  const float inv_d = 1.0f / (float)DIM;
  if constexpr (GPR) {
    constexpr int K = GK;
    float gp[2][4][K], Bn[K];
    {
      float bsum[K];
#pragma unroll
      for (int n = 0; n < K; ++n) bsum[n] = 0.f;
      float4 pq[2][K];
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int jj = 0; jj < K; ++jj) pq[v][jj] = *(const float4*)(phi + (size_t)((v * 64 + lane) * 4) * K + 4 * jj);
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const float4 g4 = v ? gm1 : gm0, b4 = v ? bt1 : bt0;
        const float gm[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
        float pf[4 * K];
#pragma unroll
        for (int jj = 0; jj < K; ++jj) { pf[4 * jj] = pq[v][jj].x; pf[4 * jj + 1] = pq[v][jj].y; pf[4 * jj + 2] = pq[v][jj].z; pf[4 * jj + 3] = pq[v][jj].w; }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
          for (int n = 0; n < K; ++n) {
            gp[v][cc][n] = gm[cc] * pf[cc * K + n];
            bsum[n] += bt[cc] * pf[cc * K + n];
          }
      }
#pragma unroll
      for (int n = 0; n < K; ++n) Bn[n] = wave_sum(bsum[n]);
    }
    static_assert(NR % 3 == 0, "rows in chunks of three");
#pragma unroll
    for (int c3 = 0; c3 < NR; c3 += 3) {
      if (wave + 16 * c3 >= g.P) continue;          // wave-uniform: the whole chunk lies past the region
      float sm[4] = {0.f, 0.f, 0.f, 0.f}, t4[4];
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) sm[u] += (r[c3 + u][v].x + r[c3 + u][v].y) + (r[c3 + u][v].z + r[c3 + u][v].w);
      wave_sum4(sm[0], sm[1], sm[2], sm[3], t4[0], t4[1], t4[2], t4[3]);
      constexpr int NV = 3 * (1 + K), NG = (NV + 3) / 4;
      float vals[4 * NG], tot[4 * NG], mean3[3];
#pragma unroll
      for (int i = 0; i < 4 * NG; ++i) vals[i] = 0.f;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const float mean = mean3[u] = t4[u] * inv_d;
        float sq = 0.f, d[K];
#pragma unroll
        for (int n = 0; n < K; ++n) d[n] = 0.f;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const float4 x = r[c3 + u][v];
          const float xc[4] = {x.x - mean, x.y - mean, x.z - mean, x.w - mean};
          sq += (xc[0] * xc[0] + xc[1] * xc[1]) + (xc[2] * xc[2] + xc[3] * xc[3]);
#pragma unroll
          for (int n = 0; n < K; ++n)
            d[n] += (xc[0] * gp[v][0][n] + xc[1] * gp[v][1][n]) + (xc[2] * gp[v][2][n] + xc[3] * gp[v][3][n]);
        }
        vals[u * (1 + K)] = sq;
#pragma unroll
        for (int n = 0; n < K; ++n) vals[u * (1 + K) + 1 + n] = d[n];
      }
#pragma unroll
      for (int gq = 0; gq < NG; ++gq)
        wave_sum4(vals[4 * gq], vals[4 * gq + 1], vals[4 * gq + 2], vals[4 * gq + 3], tot[4 * gq], tot[4 * gq + 1], tot[4 * gq + 2],
                  tot[4 * gq + 3]);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int p = wave + 16 * (c3 + u);
        if (p >= g.P) continue;
        const float rstd = 1.0f / sqrtf(tot[u * (1 + K)] * inv_d + LN_EPS);
        const bool real = tokv[c3 + u] >= 0;
        if (lane == 0) {
#pragma unroll
          for (int n = 0; n < K; ++n) s_lg[p * KM + n] = real ? rstd * tot[u * (1 + K) + 1 + n] + Bn[n] : 0.f;
          s_mr[2 * p] = real ? mean3[u] : 0.f;
          s_mr[2 * p + 1] = real ? rstd : 0.f;
          if (real && mean_rstd) { mean_rstd[2 * (size_t)tokv[c3 + u]] = mean3[u]; mean_rstd[2 * (size_t)tokv[c3 + u] + 1] = rstd; }
        }
      }
    }
  } else {
  __syncthreads();                                  // phi_t staged
  // ---- LayerNorm statistics + logits per row (the arithmetic of crmsa_logits_kernel)
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int p = wave + 16 * j;
    if (p >= g.P) continue;                         // wave-uniform
    const float4 a = r[j][0], b = r[j][1];
    const float mean = wave_sum(((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) * inv_d;
    float sq;
    {
      const float a0 = a.x - mean, a1 = a.y - mean, a2 = a.z - mean, a3 = a.w - mean;
      const float b0 = b.x - mean, b1 = b.y - mean, b2 = b.z - mean, b3 = b.w - mean;
      sq = ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3));
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
    float4 v0, v1;
    v0.x = (a.x - mean) * rstd * gm0.x + bt0.x; v0.y = (a.y - mean) * rstd * gm0.y + bt0.y;
    v0.z = (a.z - mean) * rstd * gm0.z + bt0.z; v0.w = (a.w - mean) * rstd * gm0.w + bt0.w;
    v1.x = (b.x - mean) * rstd * gm1.x + bt1.x; v1.y = (b.y - mean) * rstd * gm1.y + bt1.y;
    v1.z = (b.z - mean) * rstd * gm1.z + bt1.z; v1.w = (b.w - mean) * rstd * gm1.w + bt1.w;
    const bool real = tokv[j] >= 0;
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) {
        const float4 p0 = *(const float4*)(phi_t + n * DIM + lane * 4), p1 = *(const float4*)(phi_t + n * DIM + 256 + lane * 4);
        const float acc = wave_sum(((v0.x * p0.x + v0.y * p0.y) + (v0.z * p0.z + v0.w * p0.w)) +
                                   ((v1.x * p1.x + v1.y * p1.y) + (v1.z * p1.z + v1.w * p1.w)));
        if (lane == 0) s_lg[p * KM + n] = real ? acc : 0.f;      // pad tokens carry zero rows -> zero logits
      }
    if (lane == 0) {
      s_mr[2 * p] = real ? mean : 0.f;
      s_mr[2 * p + 1] = real ? rstd : 0.f;
      if (real && mean_rstd) { mean_rstd[2 * (size_t)tokv[j]] = mean; mean_rstd[2 * (size_t)tokv[j] + 1] = rstd; }
    }
  }
  }   // !GPR
  __syncthreads();
  // ---- region statistics: wave n handles representative n
  if (wave < k) {
    const int n = wave;
    float mx = -3.0e38f, mn = 3.0e38f;
    for (int p = lane; p < g.P; p += 64) {
      const float v = s_lg[p * KM + n];
      mx = fmaxf(mx, v);
      mn = fminf(mn, v);
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    float se = 0.f;
    for (int p = lane; p < g.P; p += 64) se += __expf(s_lg[p * KM + n] - mx);
    se = wave_sum(se);
    if (lane == 0) { s_stat[n][0] = mx; s_stat[n][1] = mn; s_stat[n][2] = 1.0f / se; }
  }
  __syncthreads();
  // ---- logits (region-major, for the training stash / debugging), dispatch weights, combine coefficients
  {
    float c0[KM], c1[KM];
#pragma unroll
    for (int n = 0; n < KM; ++n) c0[n] = c1[n] = 0.f;
    for (int p = tid; p < g.P; p += 1024) {
      float v[KM], e[KM];
      float mx = -3.0e38f;
#pragma unroll
      for (int n = 0; n < KM; ++n)
        if (n < k) { v[n] = s_lg[p * KM + n]; mx = fmaxf(mx, v[n]); }
      float se = 0.f;
#pragma unroll
      for (int n = 0; n < KM; ++n)
        if (n < k) { e[n] = __expf(v[n] - mx); se += e[n]; }
      const float inv = 1.0f / se;
      const float mean = s_mr[2 * p], rstd = s_mr[2 * p + 1];
      const bool real = rstd != 0.f;
#pragma unroll
      for (int n = 0; n < KM; ++n)
        if (n < k) {
          const size_t o = ((size_t)reg * g.P + p) * k + n;
          if (logits) logits[o] = v[n];
          wdisp[o] = (v[n] - s_stat[n][1]) / (s_stat[n][0] - s_stat[n][1] + 1e-8f) * (e[n] * inv);
          const float c = real ? __expf(v[n] - s_stat[n][0]) * s_stat[n][2] : 0.f;
          s_w[p * KM + n] = c * rstd;
          c0[n] += c * rstd * mean;
          c1[n] += c;
        }
    }
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) {
        const float a = wave_sum(c0[n]), b = wave_sum(c1[n]);
        if (lane == 0) { s_c0[n][wave] = a; s_c1[n][wave] = b; }
      }
  }
  __syncthreads();
  // ---- contraction over this wave's rows (registers), then the 16 partials through LDS
  float4 acc[KM][2];
#pragma unroll
  for (int n = 0; n < KM; ++n) acc[n][0] = acc[n][1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int p = wave + 16 * j;
    if (p >= g.P) continue;
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) {
        const float w = s_w[p * KM + n];            // 0 for pad tokens
        acc[n][0].x += w * r[j][0].x; acc[n][0].y += w * r[j][0].y; acc[n][0].z += w * r[j][0].z; acc[n][0].w += w * r[j][0].w;
        acc[n][1].x += w * r[j][1].x; acc[n][1].y += w * r[j][1].y; acc[n][1].z += w * r[j][1].z; acc[n][1].w += w * r[j][1].w;
      }
  }
  if (wave >= 8) {
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) {
        s_part[((wave - 8) * KM + n) * 128 + lane] = acc[n][0];
        s_part[((wave - 8) * KM + n) * 128 + 64 + lane] = acc[n][1];
      }
  }
  __syncthreads();
  if (wave < 8) {
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) {
        float4 a = s_part[(wave * KM + n) * 128 + lane], b = s_part[(wave * KM + n) * 128 + 64 + lane];
        a.x += acc[n][0].x; a.y += acc[n][0].y; a.z += acc[n][0].z; a.w += acc[n][0].w;
        b.x += acc[n][1].x; b.y += acc[n][1].y; b.z += acc[n][1].z; b.w += acc[n][1].w;
        acc[n][0] = a; acc[n][1] = b;
      }
  }
  __syncthreads();
  if (wave < 8) {
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) {
        s_part[(wave * KM + n) * 128 + lane] = acc[n][0];
        s_part[(wave * KM + n) * 128 + 64 + lane] = acc[n][1];
      }
  }
  __syncthreads();
  for (int idx = tid; idx < k * 128; idx += 1024) {
    const int n = idx >> 7, c = idx & 127;
    float4 a = s_part[n * 128 + c];
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      const float4 b = s_part[(q * KM + n) * 128 + c];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { c0 += s_c0[n][w]; c1 += s_c1[n][w]; }
    const float4 gm = *(const float4*)(gamma + c * 4), bt = *(const float4*)(beta + c * 4);
    float4 out;
    out.x = gm.x * (a.x - c0) + bt.x * c1;
    out.y = gm.y * (a.y - c0) + bt.y * c1;
    out.z = gm.z * (a.z - c0) + bt.z * c1;
    out.w = gm.w * (a.w - c0) + bt.w * c1;
    *(float4*)(rep + ((size_t)n * R + reg) * DIM + c * 4) = out;   // rep [k, R, D]
    if (rep16) *(uint2*)(rep16 + ((size_t)n * R + reg) * DIM + c * 4) = prec16 == 2 ? r4_pack4<2>(out) : r4_pack4<1>(out);
  }
}

// ---- the same at full chip width: FOUR blocks per region -------------------------------------------------------------
// Block (region, q) owns a quarter of the region's rows (<= 36: 12 waves x 3 rows, every load in flight at once) and does
// for them what crmsa_region_kernel does for all of them -- LayerNorm statistics and logits (global: the region's last
// block needs every row's), LOCAL softmax statistics, the contraction of its rows with exp(Lg - local max) * rstd -- and
// leaves a partial record (rep_b [k][512], local max, sum, c0, c1, min) in the workspace.  The quarter that arrives
// last (an atomic counter per region; nobody waits for anybody) merges the four records like an online softmax,
// applies LayerNorm's affine, and writes the region's dispatch weights.  256 blocks: the VALU work that made the
// one-block form slow sits on every CU again.  The counters must be zero at the start: an earlier GEMM of the same
// forward zeroes them (LinearEpilogue.zero64), the merging block leaves them zero.
constexpr int R4_KMAX = RRT_MAX_CRMSA_K;            // k <= 8: the kernel is instantiated for KM = 3 and KM = 8 representatives
constexpr int R4_NB_MAX = 16;                       // blocks per region, at most
// Device-coherent accesses for the hand-over between the quarters of a region.  The eight XCDs have private L2s, so an
// ordinary store may sit dirty in the writer's L2 and an ordinary load may hit a stale line in the reader's; a
// __threadfence() repairs that by writing the whole L2 back (measured: 100 us for the 3072 waves of this kernel).  Relaxed
// agent-scope atomics are single write-through stores / L2-bypassing loads (sc1) instead, and a wave that waited for its
// stores (vmcnt 0) before the block barrier has them in memory before the arrival counter moves.
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The same scope for whole float4s (round 5): one `buffer_store_dwordx4 ... sc1` / `buffer_load_dwordx4 ... sc1` where the dword
// forms were four instructions, each touching a quarter of every 16-byte piece -- four partial writes of every 128-byte line of
// a record (the PMC's 7.8 MB written for ~2 MB of records) and sixteen loads per thread in the merge.  aux 0x10 = sc1 = what the
// relaxed agent-scope atomics above compile to; the compiler tracks these as ordinary vector memory operations.
typedef unsigned r4_u32x4 __attribute__((ext_vector_type(4)));
#ifdef RRT_NO_REGION4_VEC
constexpr bool R4_VEC = false;
#else
constexpr bool R4_VEC = true;
#endif
__device__ __forceinline__ void st_agent4(__amdgpu_buffer_rsrc_t rs, float* base, float* p, const float4 v) {
  if constexpr (R4_VEC) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(r4_u32x4, v), rs, (unsigned)((p - base) * 4), 0, 0x10);
  } else {
    st_agent(p, v.x); st_agent(p + 1, v.y); st_agent(p + 2, v.z); st_agent(p + 3, v.w);
  }
}
__device__ __forceinline__ float4 ld_agent4(__amdgpu_buffer_rsrc_t rs, const float* base, const float* p) {
  if constexpr (R4_VEC) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((p - base) * 4), 0, 0x10));
  } else {
    return make_float4(ld_agent(p), ld_agent(p + 1), ld_agent(p + 2), ld_agent(p + 3));
  }
}

// KM: representatives the instantiation has registers for (k <= KM); KC: how many of them go through the block's
// LDS reduction at a time (the [waves / 2][KC][512] buffer is 48 KiB at KC = 4)
// GPR (round 6; k == KM_ exactly): gamma . phi of the lane's eight columns in registers and the logits in crmsa_logits512's
// centred form -- no phi table staged through LDS behind a block barrier in front of the first row, no per-element
// normalisation -- and the rows' wave totals four at a time (wave_sum4), as crmsa_stream4_kernel
template <int NB, int R4_WAVES, int R4_ROWS, int KM_, bool GPR = false>     // blocks per region, waves per block, rows per wave
__global__ __launch_bounds__(64 * R4_WAVES) void crmsa_region4_kernel(const float* __restrict__ x1,
                                                                      const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta,
                                                                      const float* __restrict__ phi,
                                                                      float* __restrict__ mean_rstd,
                                                                      float* __restrict__ logits,
                                                                      float* __restrict__ wdisp,
                                                                      float* __restrict__ rep,
                                                                      uint16_t* __restrict__ rep16, int prec16,
                                                                      float* __restrict__ part_g, int* __restrict__ counters,
                                                                      int k, GridDev g) {
  constexpr int DIM = 512, NR = R4_ROWS, NW = R4_WAVES, KM = KM_, KC = KM_ < 4 ? KM_ : 4, PQM = NR * NW;
  constexpr int R4_REC = KM * (DIM + 8);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* phi_t = (float*)smem;                      // [KM][DIM]
  float* s_lg = phi_t + KM * DIM;                   // [PQM][KM]
  float* s_mr = s_lg + PQM * KM;                    // [PQM][2]
  float* s_w = s_mr + PQM * 2;                      // [PQM][KM]
  float4* s_part = (float4*)(s_w + PQM * KM);       // [NW / 2][KC][128] float4
  __shared__ float s_stat[KM][3];                   // local max, min, sum of exp (all rows of the quarter)
  __shared__ float s_c0[KM][NW], s_c1[KM][NW];
  __shared__ float s_mrg[KM][4];                    // merge: M, 1 / L, c0 / L, c1 / L ... of the region
  __shared__ float s_scb[KM][NB];                   // merge: exp(max_b - M) / L of block b
  __shared__ float s_mm[KM][2];                     // region min, max
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int reg = blockIdx.x / NB, q = blockIdx.x - reg * NB, R = g.Rt;
  RRT_TRACE_INIT(blockIdx.x * NW + wave);
  RRT_TRACE_MARK();                                 // [1] entry
  const int ri = reg / g.rs, rj = reg - ri * g.rs;
  const int PQ = (g.P + NB - 1) / NB;               // rows per part ("quarter": NB = 4)
  float4 r[NR][2];
  int tokv[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int rq = wave + NW * j, p = q * PQ + rq;
    int t = -1;
    if (rq < PQ && p < g.P) {
      const int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
      t = (ri * g.s + pi) * g.H + rj * g.s + pj;
      if (t >= g.L) t = -1;
    }
    tokv[j] = t;
    const float* src = x1 + (size_t)(t < 0 ? 0 : t) * DIM;
    r[j][0] = *(const float4*)(src + lane * 4);
    r[j][1] = *(const float4*)(src + 256 + lane * 4);
  }
  const float4 gm0 = *(const float4*)(gamma + lane * 4), gm1 = *(const float4*)(gamma + 256 + lane * 4);
  const float4 bt0 = *(const float4*)(beta + lane * 4), bt1 = *(const float4*)(beta + 256 + lane * 4);
  float gp[GPR ? 2 : 1][4][KM], Bn[KM];
  if constexpr (GPR) {
    float bsum[KM];
#pragma unroll
    for (int n = 0; n < KM; ++n) bsum[n] = 0.f;
    float4 pq[2][KM];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int jj = 0; jj < KM; ++jj) pq[v][jj] = *(const float4*)(phi + (size_t)((v * 64 + lane) * 4) * KM + 4 * jj);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const float4 g4 = v ? gm1 : gm0, b4 = v ? bt1 : bt0;
      const float gm[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
      float pf[4 * KM];
#pragma unroll
      for (int jj = 0; jj < KM; ++jj) { pf[4 * jj] = pq[v][jj].x; pf[4 * jj + 1] = pq[v][jj].y; pf[4 * jj + 2] = pq[v][jj].z; pf[4 * jj + 3] = pq[v][jj].w; }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int n = 0; n < KM; ++n) {
          gp[v][cc][n] = gm[cc] * pf[cc * KM + n];
          bsum[n] += bt[cc] * pf[cc * KM + n];
        }
    }
#pragma unroll
    for (int n = 0; n < KM; ++n) Bn[n] = wave_sum(bsum[n]);
  } else {
    for (int idx = tid; idx < DIM * k; idx += 64 * NW) {      // after the row loads: one memory round trip, not two
      const int d = idx / k, n = idx - d * k;
      phi_t[n * DIM + d] = phi[idx];
    }
    lds_sync();
  }
  RRT_TRACE_MARK();                                 // [2] rows requested, phi in LDS
  const float inv_d = 1.0f / (float)DIM;
  // The wave's NR rows in three straight-line stages -- means, variances, logits -- and the stores behind them: 2 + k
  // wave reductions per row are dependent DPP chains, and with a row's stores (a lane-0 branch) between one row and the
  // next the compiler ran the 15 chains of three rows one after the other (round 5, traced: 6.0 K cycles for this phase).
  // Rows past the quarter's end compute on row 0's data and are not stored.
  float mean_[NR], rstd_[NR], lgs[NR][KM];
  if constexpr (GPR) {
    static_assert(NR <= 4, "one group of sums");
    float sm[4] = {0.f, 0.f, 0.f, 0.f}, t4[4];
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int v = 0; v < 2; ++v) sm[j] += (r[j][v].x + r[j][v].y) + (r[j][v].z + r[j][v].w);
    wave_sum4(sm[0], sm[1], sm[2], sm[3], t4[0], t4[1], t4[2], t4[3]);
    constexpr int NV = NR * (1 + KM), NG = (NV + 3) / 4;
    float vals[4 * NG], tot[4 * NG];
#pragma unroll
    for (int i = 0; i < 4 * NG; ++i) vals[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const float mean = mean_[j] = t4[j] * inv_d;
      float sq = 0.f, d[KM];
#pragma unroll
      for (int n = 0; n < KM; ++n) d[n] = 0.f;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const float xc[4] = {r[j][v].x - mean, r[j][v].y - mean, r[j][v].z - mean, r[j][v].w - mean};
        sq += (xc[0] * xc[0] + xc[1] * xc[1]) + (xc[2] * xc[2] + xc[3] * xc[3]);
#pragma unroll
        for (int n = 0; n < KM; ++n)
          d[n] += (xc[0] * gp[v][0][n] + xc[1] * gp[v][1][n]) + (xc[2] * gp[v][2][n] + xc[3] * gp[v][3][n]);
      }
      vals[j * (1 + KM)] = sq;
#pragma unroll
      for (int n = 0; n < KM; ++n) vals[j * (1 + KM) + 1 + n] = d[n];
    }
#pragma unroll
    for (int gq = 0; gq < NG; ++gq)
      wave_sum4(vals[4 * gq], vals[4 * gq + 1], vals[4 * gq + 2], vals[4 * gq + 3], tot[4 * gq], tot[4 * gq + 1], tot[4 * gq + 2],
                tot[4 * gq + 3]);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      rstd_[j] = 1.0f / sqrtf(tot[j * (1 + KM)] * inv_d + LN_EPS);
#pragma unroll
      for (int n = 0; n < KM; ++n) lgs[j][n] = rstd_[j] * tot[j * (1 + KM) + 1 + n] + Bn[n];
    }
  } else {
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const float4 a = r[j][0], b = r[j][1];
    mean_[j] = wave_sum(((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) * inv_d;
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const float4 a = r[j][0], b = r[j][1];
    const float mean = mean_[j];
    const float a0 = a.x - mean, a1 = a.y - mean, a2 = a.z - mean, a3 = a.w - mean;
    const float b0 = b.x - mean, b1 = b.y - mean, b2 = b.z - mean, b3 = b.w - mean;
    const float sq = ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3));
    rstd_[j] = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const float4 a = r[j][0], b = r[j][1];
    const float mean = mean_[j], rstd = rstd_[j];
    float4 v0, v1;
    v0.x = (a.x - mean) * rstd * gm0.x + bt0.x; v0.y = (a.y - mean) * rstd * gm0.y + bt0.y;
    v0.z = (a.z - mean) * rstd * gm0.z + bt0.z; v0.w = (a.w - mean) * rstd * gm0.w + bt0.w;
    v1.x = (b.x - mean) * rstd * gm1.x + bt1.x; v1.y = (b.y - mean) * rstd * gm1.y + bt1.y;
    v1.z = (b.z - mean) * rstd * gm1.z + bt1.z; v1.w = (b.w - mean) * rstd * gm1.w + bt1.w;
#pragma unroll
    for (int n = 0; n < KM; ++n) {
      lgs[j][n] = 0.f;
      if (n < k) {
        const float4 p0 = *(const float4*)(phi_t + n * DIM + lane * 4), p1 = *(const float4*)(phi_t + n * DIM + 256 + lane * 4);
        lgs[j][n] = wave_sum(((v0.x * p0.x + v0.y * p0.y) + (v0.z * p0.z + v0.w * p0.w)) +
                             ((v1.x * p1.x + v1.y * p1.y) + (v1.z * p1.z + v1.w * p1.w)));
      }
    }
  }
  }   // !GPR
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int rq = wave + NW * j, p = q * PQ + rq;
    if (rq >= PQ || p >= g.P) continue;             // wave-uniform
    const bool real = tokv[j] >= 0;
    if (lane == 0) {
#pragma unroll
      for (int n = 0; n < KM; ++n)
        if (n < k) {
          const float lgv = real ? lgs[j][n] : 0.f;   // pad tokens carry zero rows -> zero logits
          s_lg[rq * KM + n] = lgv;
          st_agent(logits + ((size_t)reg * g.P + p) * k + n, lgv);
        }
      s_mr[2 * rq] = real ? mean_[j] : 0.f;
      s_mr[2 * rq + 1] = real ? rstd_[j] : 0.f;
      if (real && mean_rstd) { mean_rstd[2 * (size_t)tokv[j]] = mean_[j]; mean_rstd[2 * (size_t)tokv[j] + 1] = rstd_[j]; }
    }
  }
  RRT_TRACE_MARK();                                 // [3] LayerNorm statistics + logits of this wave's rows
  lds_sync();
  const int nrow = min(PQ, g.P - q * PQ);           // rows of this quarter (>= 1: P >= 4 is checked by the launcher)
  // wave n < k owns representative n: local max / min / sum of exp over the quarter's rows, the combine coefficients
  // c * rstd of every row, and the two LayerNorm-fold sums (Identity 3) -- five wave reductions on k waves, one barrier
  // (round 2 did the sums on all NW waves behind a second barrier: 4.3 K cycles of mostly idle reductions)
  if (wave < k) {
    const int n = wave;
    float mx = -3.0e38f, mn = 3.0e38f;
    for (int p = lane; p < nrow; p += 64) {
      const float v = s_lg[p * KM + n];
      mx = fmaxf(mx, v);
      mn = fminf(mn, v);
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    float se = 0.f, c0 = 0.f, c1 = 0.f;
    for (int p = lane; p < nrow; p += 64) {
      const float e = __expf(s_lg[p * KM + n] - mx);
      se += e;
      const float mean = s_mr[2 * p], rstd = s_mr[2 * p + 1];
      const float c = rstd != 0.f ? e : 0.f;        // pad rows (rstd = 0): in the softmax sum, not in the contraction
      s_w[p * KM + n] = c * rstd;
      c0 += c * rstd * mean;
      c1 += c;
    }
    se = wave_sum(se);
    c0 = wave_sum(c0);
    c1 = wave_sum(c1);
    if (lane == 0) {
      s_stat[n][0] = mx; s_stat[n][1] = mn; s_stat[n][2] = se;
      s_c0[n][0] = c0; s_c1[n][0] = c1;
    }
  }
  lds_sync();
  RRT_TRACE_MARK();                                 // [4] local softmax statistics + coefficients
  float4 acc[KM][2];
#pragma unroll
  for (int n = 0; n < KM; ++n) acc[n][0] = acc[n][1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int rq = wave + NW * j;
    if (rq >= nrow) continue;
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) {
        const float w = s_w[rq * KM + n];
        acc[n][0].x += w * r[j][0].x; acc[n][0].y += w * r[j][0].y; acc[n][0].z += w * r[j][0].z; acc[n][0].w += w * r[j][0].w;
        acc[n][1].x += w * r[j][1].x; acc[n][1].y += w * r[j][1].y; acc[n][1].z += w * r[j][1].z; acc[n][1].w += w * r[j][1].w;
      }
  }
  RRT_TRACE_MARK();                                 // [5] contraction of this wave's rows
  constexpr int HW = NW / 2;
  // ---- this quarter's record -> workspace: the NW per-wave partials summed through LDS in a fixed order, KC
  // representatives at a time
  float* rec = part_g + (size_t)(reg * NB + q) * R4_REC;
  const __amdgpu_buffer_rsrc_t rs_part =
      __builtin_amdgcn_make_buffer_rsrc((void*)part_g, 0, (int)((size_t)gridDim.x * R4_REC * 4), 0x00020000);
#pragma unroll
  for (int n0 = 0; n0 < KM; n0 += KC) {
    if (n0 >= k) break;
    if (n0 > 0) lds_sync();                    // the previous chunk's reads of s_part are done
    if (wave >= HW) {
#pragma unroll
      for (int n = n0; n < n0 + KC && n < KM; ++n)
        if (n < k) {
          s_part[((wave - HW) * KC + n - n0) * 128 + lane] = acc[n][0];
          s_part[((wave - HW) * KC + n - n0) * 128 + 64 + lane] = acc[n][1];
        }
    }
    lds_sync();
    if (wave < HW) {
#pragma unroll
      for (int n = n0; n < n0 + KC && n < KM; ++n)
        if (n < k) {
          float4 a = s_part[(wave * KC + n - n0) * 128 + lane], b = s_part[(wave * KC + n - n0) * 128 + 64 + lane];
          a.x += acc[n][0].x; a.y += acc[n][0].y; a.z += acc[n][0].z; a.w += acc[n][0].w;
          b.x += acc[n][1].x; b.y += acc[n][1].y; b.z += acc[n][1].z; b.w += acc[n][1].w;
          s_part[(wave * KC + n - n0) * 128 + lane] = a;
          s_part[(wave * KC + n - n0) * 128 + 64 + lane] = b;
        }
    }
    lds_sync();
    const int kc = min(KC, k - n0);
    for (int idx = tid; idx < kc * 128; idx += 64 * NW) {
      const int n = idx >> 7, c = idx & 127;
      float4 a = s_part[n * 128 + c];
#pragma unroll
      for (int w = 1; w < HW; ++w) {
        const float4 b = s_part[(w * KC + n) * 128 + c];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      st_agent4(rs_part, part_g, rec + (n0 + n) * (DIM + 8) + c * 4, a);
    }
  }
  if (tid < k) {
    const int n = tid;
    const float c0 = s_c0[n][0], c1 = s_c1[n][0];
    float* st = rec + n * (DIM + 8) + DIM;            // the record's eight trailing floats: max, min, sum, c0 | c1, -, -, -
    st_agent4(rs_part, part_g, st, make_float4(s_stat[n][0], s_stat[n][1], s_stat[n][2], c0));
    st_agent4(rs_part, part_g, st + 4, make_float4(c1, 0.f, 0.f, 0.f));
  }
  RRT_TRACE_MARK();                                 // [6] record stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record (and the logits) are in memory ...
  __syncthreads();
  RRT_TRACE_MARK();                                 // [7] record in memory (write-through stores acknowledged)
  if (tid == 0)                                     // ... before this quarter counts as arrived
    s_last = __hip_atomic_fetch_add(counters + reg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NB - 1;
  __syncthreads();
  RRT_TRACE_MARK();                                 // [8] arrival counted
  if (!s_last) return;
  if (tid == 0) __hip_atomic_store(counters + reg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next forward
  const float* rec0 = part_g + (size_t)(reg * NB) * R4_REC;
  // the logits of this thread's row (the dispatch weights at the very end) are requested here, in front of the merge: their
  // round trip hides behind it (regions of more than 64 NW rows: the loop at the end fetches the rest)
  float v_first[KM];
  {
    const int p = tid < g.P ? tid : 0;
#pragma unroll
    for (int n = 0; n < KM; ++n) v_first[n] = n < k ? ld_agent(logits + ((size_t)reg * g.P + p) * k + n) : 0.f;
  }
  if (tid < k) {
    const int n = tid;
    // every block's statistics requested at once (two float4 per block), then the arithmetic; the blocks' weights
    // exp(max_b - M) / L go to LDS for the merge below (it fetched max_b again, per thread and block, in front of its sums)
    float4 sa[NB], sb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* st = rec0 + b * R4_REC + n * (DIM + 8) + DIM;
      sa[b] = ld_agent4(rs_part, part_g, st);
      sb[b] = ld_agent4(rs_part, part_g, st + 4);
    }
    float M = -3.0e38f, mn = 3.0e38f;
#pragma unroll
    for (int b = 0; b < NB; ++b) { M = fmaxf(M, sa[b].x); mn = fminf(mn, sa[b].y); }
    float L = 0.f, c0 = 0.f, c1 = 0.f, scb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      scb[b] = __expf(sa[b].x - M);
      L += scb[b] * sa[b].z; c0 += scb[b] * sa[b].w; c1 += scb[b] * sb[b].x;
    }
    const float invL = 1.0f / L;
#pragma unroll
    for (int b = 0; b < NB; ++b) s_scb[n][b] = scb[b] * invL;
    s_mrg[n][0] = M; s_mrg[n][1] = invL; s_mrg[n][2] = c0 / L; s_mrg[n][3] = c1 / L;
    s_mm[n][0] = mn; s_mm[n][1] = M;
  }
  __syncthreads();
  for (int idx = tid; idx < k * 128; idx += 64 * NW) {
    const int n = idx >> 7, c = idx & 127;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < NB; ++b) {
      const float* rb = rec0 + b * R4_REC + n * (DIM + 8);
      const float sc = s_scb[n][b];
      const float4 v4 = ld_agent4(rs_part, part_g, rb + c * 4);
      a.x += sc * v4.x; a.y += sc * v4.y; a.z += sc * v4.z; a.w += sc * v4.w;
    }
    const float c0 = s_mrg[n][2], c1 = s_mrg[n][3];
    const float4 gm = *(const float4*)(gamma + c * 4), bt = *(const float4*)(beta + c * 4);
    float4 out;
    out.x = gm.x * (a.x - c0) + bt.x * c1;
    out.y = gm.y * (a.y - c0) + bt.y * c1;
    out.z = gm.z * (a.z - c0) + bt.z * c1;
    out.w = gm.w * (a.w - c0) + bt.w * c1;
    *(float4*)(rep + ((size_t)n * R + reg) * DIM + c * 4) = out;
    // reduced-precision modes: the representatives also leave as the 16-bit A operand of their qkv projection
    if (rep16) {
      uint16_t* d16 = rep16 + ((size_t)n * R + reg) * DIM + c * 4;
      *(uint2*)d16 = prec16 == 2 ? r4_pack4<2>(out) : r4_pack4<1>(out);
    }
  }
  RRT_TRACE_MARK();                                 // [9] (last arrival) records merged, rep written
  // dispatch weights of the whole region (rmsa.py:310-314, :324-325) from the four quarters' logits
  for (int p = tid; p < g.P; p += 64 * NW) {
    float v[KM], e[KM];
    float mx = -3.0e38f;
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) { v[n] = p == tid ? v_first[n] : ld_agent(logits + ((size_t)reg * g.P + p) * k + n); mx = fmaxf(mx, v[n]); }
    float se = 0.f;
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) { e[n] = __expf(v[n] - mx); se += e[n]; }
    const float inv = 1.0f / se;
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k)
        wdisp[((size_t)reg * g.P + p) * k + n] = (v[n] - s_mm[n][0]) / (s_mm[n][1] - s_mm[n][0] + 1e-8f) * (e[n] * inv);
  }
  RRT_TRACE_MARK();                                 // [10] dispatch weights written
}

// ---- round 6: the one-pass form for regions of MORE than 144 tokens (BASELINE configs[3]: 484 per region; the large bags of
// configs[4]) -- FOUR blocks per region that STREAM their quarter of the region's rows.
// crmsa_region4_kernel keeps every row of a block in flight at once (36 per block), so larger regions meant 8 / 16 blocks per
// region and a merge of 8 / 16 partial records by one block: slower than two chip-wide passes over x1 (history section 3,
// "measured without gain"; 46-74 against 39 us at N = 30000), which is what the forward used above 144 tokens -- x1 read TWICE
// (2 x 61.4 MB at N = 30000: 15.1 + 15.3 us).  Here a block is still (region, quarter) and the merge still takes four records,
// but a wave walks its rows in trips of NR (the next trip's rows requested before the current one is reduced) and keeps an
// ONLINE softmax of its own: running max m_n, sum Z_n, the LayerNorm-fold sums c0_n / c1_n and the contraction
// acc_n = sum_p exp(Lg_np - m_n) rstd_p x1_p, rescaled once per trip when the max moves.  The waves' partials meet in LDS
// (scaled to the block's max, summed in a fixed order), the block's record has crmsa_region4's layout, and the tail -- arrival
// counter, merge by the last quarter, LayerNorm's affine, the region's dispatch weights -- is crmsa_region4's.  x1 is read ONCE.
constexpr int S4_PQMAX = 144;                       // rows of a quarter, at most (P <= 576)
template <int NW, int NR, int KM_>                  // waves per block, rows per wave and trip, representatives (k == KM_ exactly)
__global__ __launch_bounds__(64 * NW) void crmsa_stream4_kernel(const float* __restrict__ x1, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ phi,
                                                                float* __restrict__ mean_rstd, float* __restrict__ logits,
                                                                float* __restrict__ wdisp, float* __restrict__ rep,
                                                                uint16_t* __restrict__ rep16, int prec16,
                                                                float* __restrict__ part_g, int* __restrict__ counters,
                                                                GridDev g) {
  constexpr int NB = 4, DIM = 512, KM = KM_, KC = KM_ < 4 ? KM_ : 4, k = KM_;
  constexpr int R4_REC = KM * (DIM + 8);            // (the scratch is sized for crmsa_region4's records: at least as large)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_wst = (float*)smem;                      // [KM][NW][8]: the waves' m, min, Z, c0, c1
  float* s_lg = s_wst + KM * NW * 8;                // [S4_PQMAX][KM]: the quarter's logits, written out after the loop
  float* s_mr = s_lg + S4_PQMAX * KM;               // [S4_PQMAX][2]
  float4* s_part = (float4*)(s_mr + S4_PQMAX * 2);  // [NW / 2][KC][128] float4
  __shared__ float s_stat[KM][5];                   // the block's max, min, Z, c0, c1
  __shared__ float s_mrg[KM][4];
  __shared__ float s_scb[KM][NB];
  __shared__ float s_mm[KM][2];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: the rows' validity tests become scalar branches)
  const int reg = blockIdx.x / NB, q = blockIdx.x - reg * NB, R = g.Rt;
  const int ri = reg / g.rs, rj = reg - ri * g.rs;
  const int PQ = (g.P + NB - 1) / NB;
  const int nrow = min(PQ, g.P - q * PQ);           // rows of this quarter (>= 1: the launcher checks P >= 4 NB)
  // token of the block's row rq: >= 0 real, -1 pad slot of the region grid (zero row: zero logits, in the softmax, not in the
  // contraction), -2 past the quarter (nothing)
  auto tok_of = [&](int rq) -> int {
    if (rq >= nrow) return -2;
    const int p = q * PQ + rq;
    const int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
    const int t = (ri * g.s + pi) * g.H + rj * g.s + pj;
    return t < g.L ? t : -1;
  };
  float4 ra[NR][2], rb[NR][2];
  int ta[NR], tb[NR];
  auto request = [&](float4 (&rv)[NR][2], int (&tk)[NR], int j0) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int t = tok_of(wave + NW * (j0 + j));
      tk[j] = t;
      const float* src = x1 + (size_t)(t < 0 ? 0 : t) * DIM;
      rv[j][0] = *(const float4*)(src + lane * 4);
      rv[j][1] = *(const float4*)(src + 256 + lane * 4);
    }
  };
  // gamma . phi of this lane's eight columns and B_n = sum_c beta_c phi_cn (crmsa_logits512_kernel's arithmetic: the logits
  // are bit-identical to the two-pass form's).  These loads come FIRST: they are read inside the loop, and a load that is
  // younger than the trips' requests at loop entry makes the loop's static wait for it (vmcnt(0), executed every iteration)
  // drain the prefetched trip as well
  float gp[2][4][KM], Bn[KM];
  {
    float bsum[KM];
#pragma unroll
    for (int n = 0; n < KM; ++n) bsum[n] = 0.f;
    float4 gm4[2], bt4[2], pq[2][KM];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int c = (v * 64 + lane) * 4;
      gm4[v] = *(const float4*)(gamma + c);
      bt4[v] = *(const float4*)(beta + c);
#pragma unroll
      for (int j = 0; j < KM; ++j) pq[v][j] = *(const float4*)(phi + (size_t)c * KM + 4 * j);
    }
    request(ra, ta, 0);
    request(rb, tb, NR);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const float gm[4] = {gm4[v].x, gm4[v].y, gm4[v].z, gm4[v].w}, bt[4] = {bt4[v].x, bt4[v].y, bt4[v].z, bt4[v].w};
      float pf[4 * KM];                              // phi[c .. c+3][0 .. KM): 4 KM contiguous floats
#pragma unroll
      for (int j = 0; j < KM; ++j) { pf[4 * j] = pq[v][j].x; pf[4 * j + 1] = pq[v][j].y; pf[4 * j + 2] = pq[v][j].z; pf[4 * j + 3] = pq[v][j].w; }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int n = 0; n < KM; ++n) {
          const float ph = pf[cc * KM + n];
          gp[v][cc][n] = gm[cc] * ph;
          bsum[n] += bt[cc] * ph;
        }
    }
#pragma unroll
    for (int n = 0; n < KM; ++n) Bn[n] = wave_sum(bsum[n]);
  }
  const float inv_d = 1.0f / (float)DIM;
  float m_[KM], mn_[KM], Z_[KM], c0_[KM], c1_[KM];
  float4 acc[KM][2];
#pragma unroll
  for (int n = 0; n < KM; ++n) {
    m_[n] = -3.0e38f; mn_[n] = 3.0e38f; Z_[n] = c0_[n] = c1_[n] = 0.f;
    acc[n][0] = acc[n][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto consume = [&](const float4 (&r)[NR][2], const int (&tk)[NR], int j0) {
    float mean_[NR], rstd_[NR], lgs[NR][KM];
    // the wave totals of a trip -- NR sums, then NR x (1 + KM) centred sums -- in groups of four (wave_sum4): the reductions
    // were a quarter of the kernel's instructions, and the kernel is bound by its instruction count (24 us at N = 30000 with
    // one wave_sum per value, for 61 MB)
    {
      float sm[4] = {0.f, 0.f, 0.f, 0.f};
      static_assert(NR <= 4, "one group of sums");
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int v = 0; v < 2; ++v) sm[j] += (r[j][v].x + r[j][v].y) + (r[j][v].z + r[j][v].w);
      float t4[4];
      wave_sum4(sm[0], sm[1], sm[2], sm[3], t4[0], t4[1], t4[2], t4[3]);
#pragma unroll
      for (int j = 0; j < NR; ++j) mean_[j] = t4[j] * inv_d;
    }
    float sq_[NR], d_[NR][KM];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const float mean = mean_[j];
      float sq = 0.f;
#pragma unroll
      for (int n = 0; n < KM; ++n) d_[j][n] = 0.f;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const float xc[4] = {r[j][v].x - mean, r[j][v].y - mean, r[j][v].z - mean, r[j][v].w - mean};
        sq += (xc[0] * xc[0] + xc[1] * xc[1]) + (xc[2] * xc[2] + xc[3] * xc[3]);
#pragma unroll
        for (int n = 0; n < KM; ++n)
          d_[j][n] += (xc[0] * gp[v][0][n] + xc[1] * gp[v][1][n]) + (xc[2] * gp[v][2][n] + xc[3] * gp[v][3][n]);
      }
      sq_[j] = sq;
    }
    {
      constexpr int NV = NR * (1 + KM), NG = (NV + 3) / 4;
      float vals[4 * NG], tot[4 * NG];
#pragma unroll
      for (int i = 0; i < 4 * NG; ++i) vals[i] = 0.f;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        vals[j * (1 + KM)] = sq_[j];
#pragma unroll
        for (int n = 0; n < KM; ++n) vals[j * (1 + KM) + 1 + n] = d_[j][n];
      }
#pragma unroll
      for (int gq = 0; gq < NG; ++gq)
        wave_sum4(vals[4 * gq], vals[4 * gq + 1], vals[4 * gq + 2], vals[4 * gq + 3], tot[4 * gq], tot[4 * gq + 1], tot[4 * gq + 2],
                  tot[4 * gq + 3]);
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        rstd_[j] = 1.0f / sqrtf(tot[j * (1 + KM)] * inv_d + LN_EPS);
#pragma unroll
        for (int n = 0; n < KM; ++n) lgs[j][n] = rstd_[j] * tot[j * (1 + KM) + 1 + n] + Bn[n];
      }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      if (tk[j] == -2) continue;                    // wave-uniform
      const bool real = tk[j] >= 0;
      const int rq = wave + NW * (j0 + j);
#pragma unroll
      for (int n = 0; n < KM; ++n) lgs[j][n] = real ? lgs[j][n] : 0.f;     // pad tokens carry zero rows -> zero logits
      // (to LDS, not to memory: a global store between the trips' loads leaves the wait-count pass with loads AND stores
      //  outstanding -- one counter, no order between the two kinds -- and every wait of the loop became vmcnt(0): the
      //  prefetched trip was waited for on the spot; the logits leave in one pass behind the loop)
      if (lane == 0) {
#pragma unroll
        for (int n = 0; n < KM; ++n) s_lg[rq * KM + n] = lgs[j][n];
        s_mr[2 * rq] = real ? mean_[j] : 0.f;
        s_mr[2 * rq + 1] = real ? rstd_[j] : 0.f;
      }
    }
    // the trip's rows into the wave's online softmax: one rescale per trip
#pragma unroll
    for (int n = 0; n < KM; ++n) {
      float mx = m_[n];
#pragma unroll
      for (int j = 0; j < NR; ++j)
        if (tk[j] != -2) { mx = fmaxf(mx, lgs[j][n]); mn_[n] = fminf(mn_[n], lgs[j][n]); }
      const float sc = __expf(m_[n] - mx);
      m_[n] = mx;
      Z_[n] *= sc; c0_[n] *= sc; c1_[n] *= sc;
      acc[n][0].x *= sc; acc[n][0].y *= sc; acc[n][0].z *= sc; acc[n][0].w *= sc;
      acc[n][1].x *= sc; acc[n][1].y *= sc; acc[n][1].z *= sc; acc[n][1].w *= sc;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        if (tk[j] == -2) continue;
        const float e = __expf(lgs[j][n] - mx);
        Z_[n] += e;
        if (tk[j] >= 0) {
          const float w = e * rstd_[j];
          c0_[n] += w * mean_[j];
          c1_[n] += e;
          acc[n][0].x += w * r[j][0].x; acc[n][0].y += w * r[j][0].y; acc[n][0].z += w * r[j][0].z; acc[n][0].w += w * r[j][0].w;
          acc[n][1].x += w * r[j][1].x; acc[n][1].y += w * r[j][1].y; acc[n][1].z += w * r[j][1].z; acc[n][1].w += w * r[j][1].w;
        }
      }
    }
  };
  const int nj = (nrow + NW - 1) / NW;              // rows of the longest wave
  // two trips per iteration: one lands while the other is reduced.  The requests are UNCONDITIONAL (rows past the quarter
  // re-read row 0 and are skipped by `consume`): a request under `if (more rows)` leaves the wait-count pass not knowing
  // whether the younger loads exist, and it then waits for the older trip with vmcnt(0) -- i.e. for the prefetch as well.
  for (int j0 = 0; j0 < nj; j0 += 2 * NR) {
    consume(ra, ta, j0);
    request(ra, ta, j0 + 2 * NR);
    consume(rb, tb, j0 + NR);
    request(rb, tb, j0 + 3 * NR);
  }
  // ---- the waves' partials -> the block's record: statistics first, then the contraction scaled to the block's max
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < KM; ++n) {
      float* w = s_wst + (n * NW + wave) * 8;
      w[0] = m_[n]; w[1] = mn_[n]; w[2] = Z_[n]; w[3] = c0_[n]; w[4] = c1_[n];
    }
  }
  lds_sync();
  // the quarter's logits (the merging block reads the whole region's) and, for the stage entry point, mean / rstd
  for (int idx = tid; idx < nrow * k; idx += 64 * NW) {
    const int rq = idx / k, n = idx - rq * k;
    st_agent(logits + ((size_t)reg * g.P + q * PQ + rq) * k + n, s_lg[rq * KM + n]);
  }
  if (mean_rstd)
    for (int rq = tid; rq < nrow; rq += 64 * NW) {
      const int t = tok_of(rq);
      if (t >= 0) *(float2*)(mean_rstd + 2 * (size_t)t) = make_float2(s_mr[2 * rq], s_mr[2 * rq + 1]);
    }
#pragma unroll
  for (int n = 0; n < KM; ++n) {
    if (n >= k) continue;
    float M = -3.0e38f;
    for (int w = 0; w < NW; ++w) M = fmaxf(M, s_wst[(n * NW + w) * 8]);
    const float sc = __expf(m_[n] - M);
    acc[n][0].x *= sc; acc[n][0].y *= sc; acc[n][0].z *= sc; acc[n][0].w *= sc;
    acc[n][1].x *= sc; acc[n][1].y *= sc; acc[n][1].z *= sc; acc[n][1].w *= sc;
  }
  if (tid < k) {
    const int n = tid;
    float M = -3.0e38f, mn = 3.0e38f;
    for (int w = 0; w < NW; ++w) { M = fmaxf(M, s_wst[(n * NW + w) * 8]); mn = fminf(mn, s_wst[(n * NW + w) * 8 + 1]); }
    float Z = 0.f, c0 = 0.f, c1 = 0.f;
    for (int w = 0; w < NW; ++w) {
      const float* ws = s_wst + (n * NW + w) * 8;
      const float sc = __expf(ws[0] - M);
      Z += sc * ws[2]; c0 += sc * ws[3]; c1 += sc * ws[4];
    }
    s_stat[n][0] = M; s_stat[n][1] = mn; s_stat[n][2] = Z; s_stat[n][3] = c0; s_stat[n][4] = c1;
  }
  constexpr int HW = NW / 2;
  float* rec = part_g + (size_t)(reg * NB + q) * R4_REC;
  const __amdgpu_buffer_rsrc_t rs_part =
      __builtin_amdgcn_make_buffer_rsrc((void*)part_g, 0, (int)((size_t)gridDim.x * R4_REC * 4), 0x00020000);
#pragma unroll
  for (int n0 = 0; n0 < KM; n0 += KC) {
    if (n0 > 0) lds_sync();
    if (wave >= HW) {
#pragma unroll
      for (int n = n0; n < n0 + KC && n < KM; ++n) {
        s_part[((wave - HW) * KC + n - n0) * 128 + lane] = acc[n][0];
        s_part[((wave - HW) * KC + n - n0) * 128 + 64 + lane] = acc[n][1];
      }
    }
    lds_sync();
    if (wave < HW) {
#pragma unroll
      for (int n = n0; n < n0 + KC && n < KM; ++n) {
        float4 a = s_part[(wave * KC + n - n0) * 128 + lane], b = s_part[(wave * KC + n - n0) * 128 + 64 + lane];
        a.x += acc[n][0].x; a.y += acc[n][0].y; a.z += acc[n][0].z; a.w += acc[n][0].w;
        b.x += acc[n][1].x; b.y += acc[n][1].y; b.z += acc[n][1].z; b.w += acc[n][1].w;
        s_part[(wave * KC + n - n0) * 128 + lane] = a;
        s_part[(wave * KC + n - n0) * 128 + 64 + lane] = b;
      }
    }
    lds_sync();
    const int kc = min(KC, k - n0);
    for (int idx = tid; idx < kc * 128; idx += 64 * NW) {
      const int n = idx >> 7, c = idx & 127;
      float4 a = s_part[n * 128 + c];
#pragma unroll
      for (int w = 1; w < HW; ++w) {
        const float4 b = s_part[(w * KC + n) * 128 + c];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      st_agent4(rs_part, part_g, rec + (n0 + n) * (DIM + 8) + c * 4, a);
    }
  }
  if (tid < k) {                                    // (s_stat was written by this very thread)
    const int n = tid;
    float* st = rec + n * (DIM + 8) + DIM;          // max, min, sum, c0 | c1, -, -, -  (crmsa_region4's record)
    st_agent4(rs_part, part_g, st, make_float4(s_stat[n][0], s_stat[n][1], s_stat[n][2], s_stat[n][3]));
    st_agent4(rs_part, part_g, st + 4, make_float4(s_stat[n][4], 0.f, 0.f, 0.f));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record (and the logits) are in memory ...
  __syncthreads();
  if (tid == 0)                                     // ... before this quarter counts as arrived
    s_last = __hip_atomic_fetch_add(counters + reg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NB - 1;
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) __hip_atomic_store(counters + reg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next forward
  // ---- the last quarter merges the region (as crmsa_region4_kernel)
  const float* rec0 = part_g + (size_t)(reg * NB) * R4_REC;
  if (tid < k) {
    const int n = tid;
    float4 sa[NB], sb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* st = rec0 + b * R4_REC + n * (DIM + 8) + DIM;
      sa[b] = ld_agent4(rs_part, part_g, st);
      sb[b] = ld_agent4(rs_part, part_g, st + 4);
    }
    float M = -3.0e38f, mn = 3.0e38f;
#pragma unroll
    for (int b = 0; b < NB; ++b) { M = fmaxf(M, sa[b].x); mn = fminf(mn, sa[b].y); }
    float L = 0.f, c0 = 0.f, c1 = 0.f, scb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      scb[b] = __expf(sa[b].x - M);
      L += scb[b] * sa[b].z; c0 += scb[b] * sa[b].w; c1 += scb[b] * sb[b].x;
    }
    const float invL = 1.0f / L;
#pragma unroll
    for (int b = 0; b < NB; ++b) s_scb[n][b] = scb[b] * invL;
    s_mrg[n][0] = M; s_mrg[n][1] = invL; s_mrg[n][2] = c0 / L; s_mrg[n][3] = c1 / L;
    s_mm[n][0] = mn; s_mm[n][1] = M;
  }
  __syncthreads();
  for (int idx = tid; idx < k * 128; idx += 64 * NW) {
    const int n = idx >> 7, c = idx & 127;
    float4 v4[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) v4[b] = ld_agent4(rs_part, part_g, rec0 + b * R4_REC + n * (DIM + 8) + c * 4);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float sc = s_scb[n][b];
      a.x += sc * v4[b].x; a.y += sc * v4[b].y; a.z += sc * v4[b].z; a.w += sc * v4[b].w;
    }
    const float c0 = s_mrg[n][2], c1 = s_mrg[n][3];
    const float4 gm = *(const float4*)(gamma + c * 4), bt = *(const float4*)(beta + c * 4);
    float4 out;
    out.x = gm.x * (a.x - c0) + bt.x * c1;
    out.y = gm.y * (a.y - c0) + bt.y * c1;
    out.z = gm.z * (a.z - c0) + bt.z * c1;
    out.w = gm.w * (a.w - c0) + bt.w * c1;
    *(float4*)(rep + ((size_t)n * R + reg) * DIM + c * 4) = out;
    if (rep16) {
      uint16_t* d16 = rep16 + ((size_t)n * R + reg) * DIM + c * 4;
      *(uint2*)d16 = prec16 == 2 ? r4_pack4<2>(out) : r4_pack4<1>(out);
    }
  }
  // dispatch weights of the whole region (rmsa.py:310-314, :324-325) from the four quarters' logits
  for (int p = tid; p < g.P; p += 64 * NW) {
    float v[KM], e[KM];
    float mx = -3.0e38f;
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) { v[n] = ld_agent(logits + ((size_t)reg * g.P + p) * k + n); mx = fmaxf(mx, v[n]); }
    float se = 0.f;
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k) { e[n] = __expf(v[n] - mx); se += e[n]; }
    const float inv = 1.0f / se;
#pragma unroll
    for (int n = 0; n < KM; ++n)
      if (n < k)
        wdisp[((size_t)reg * g.P + p) * k + n] = (v[n] - s_mm[n][0]) / (s_mm[n][1] - s_mm[n][0] + 1e-8f) * (e[n] * inv);
  }
}

// KB > 0 (round 4): the dispatch weights and the representatives' rows of the first KB representatives are requested UP FRONT,
// together with the token row, and gamma / beta with them -- the loop over n issued one dependent L2 round trip per
// representative behind the row's HBM round trip (and a fourth one for gamma / beta behind the statistics): 8.7 us for the
// bytes LayerNorm + partition moves in 5.9.  Representatives n >= k are fetched clamped and weighted 0.  KB = 0: the loop.
template <int NV, bool CRMSA, bool FULL, int KB = 0>   // FULL: dim == NV * 256, no lane predication (see ln_partition.hip)
__global__ __launch_bounds__(256) void crmsa_dispatch_ln_kernel(
    const float* __restrict__ x1, const float* __restrict__ x0, const float* __restrict__ wdisp,
    const float* __restrict__ rep2, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, int L, int dim, int k, GridDev g,
    uint16_t* __restrict__ y16, int prec16) {
  // y16 (round 6, may be null): the output rows once more as 16-bit values (prec16 = 1 bf16 / 2 fp16) -- the A operand of the
  // slide classifier's pooling-score Linear under autocast (modules/datten.py:28-38), which then needs no fp32-operand GEMM
  constexpr int RW = RW_DISPATCH;
  const int lane = threadIdx.x & 63;
  const int t0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * RW);   // wave-uniform: scalar index math
  if (t0 >= L) return;
  const int R = g.Rt;
  float4 r[RW][NV];
  // issue every row load first (x1, shortcut), then the dispatch weights
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int t = t0 + i < L ? t0 + i : L - 1;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int c = (v * 64 + lane) * 4;
      r[i][v] = (FULL || c < dim) ? ld_row<NT_DISPATCH>(x1 + (size_t)t * dim + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 && (FULL || c < dim)) {
        float4 s = *(const float4*)(x0 + (size_t)t * dim + c);
        r[i][v].x += s.x; r[i][v].y += s.y; r[i][v].z += s.z; r[i][v].w += s.w;
      }
    }
  }
  float4 gm_pre[NV], bt_pre[NV];
  if constexpr (KB > 0) {
    if (gamma != nullptr) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        gm_pre[v] = (FULL || c < dim) ? *(const float4*)(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        bt_pre[v] = (FULL || c < dim) ? *(const float4*)(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  if (CRMSA && KB > 0) {
    // every request of every token of the wave before the first use (round 5: the per-token form put token 1's L2 round
    // trip behind token 0's arithmetic); tokens of one region -- neighbours in the bag almost always are -- share the
    // representatives' rows: one fetch (the k rows are 3 x the token row's bytes through the texture path)
    float w[RW][KB > 0 ? KB : 1];
    float4 q[RW][KB > 0 ? KB : 1][NV];
    int reg[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int t = t0 + i < L ? t0 + i : L - 1;
      const int slot = token_to_slot(t, g);
      reg[i] = fdiv(slot, g.P, g.inv_P);
      const float* wd = wdisp + (size_t)slot * k;
#pragma unroll
      for (int n = 0; n < KB; ++n) w[i][n] = wd[n < k ? n : k - 1];      // in bounds; its weight is zeroed below
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const bool fetch = i == 0 || reg[i] != reg[0];       // wave-uniform (t0 is)
#pragma unroll
      for (int n = 0; n < KB; ++n) {
        const int nn = n < k ? n : k - 1;
        const float* rp = rep2 + ((size_t)nn * R + reg[i]) * dim;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int c = (v * 64 + lane) * 4;
          if (fetch) q[i][n][v] = (FULL || c < dim) ? *(const float4*)(rp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          else q[i][n][v] = q[0][n][v];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);                   // every request is out before the first use waits
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int n = 0; n < KB; ++n) {
        const float wn = n < k ? w[i][n] : 0.f;            // branch-free; same summation order as the loop form
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          r[i][v].x += wn * q[i][n][v].x; r[i][v].y += wn * q[i][n][v].y; r[i][v].z += wn * q[i][n][v].z; r[i][v].w += wn * q[i][n][v].w;
        }
      }
  } else if (CRMSA) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int t = t0 + i < L ? t0 + i : L - 1;
      const int slot = token_to_slot(t, g);
      const int reg = fdiv(slot, g.P, g.inv_P);
      const float* wd = wdisp + (size_t)slot * k;
#pragma unroll
      for (int n = 0; n < KMAX; ++n)
        if (n < k) {
          const float w = wd[n];
          const float* rp = rep2 + ((size_t)n * R + reg) * dim;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            int c = (v * 64 + lane) * 4;
            if (FULL || c < dim) {
              const float4 q = *(const float4*)(rp + c);
              r[i][v].x += w * q.x; r[i][v].y += w * q.y; r[i][v].z += w * q.z; r[i][v].w += w * q.w;
            }
          }
        }
    }
  }
  if (gamma == nullptr) {              // no LayerNorm: an FFN follows (ffn = 1), rows go out as they are
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      if (t0 + i >= L) break;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        int c = (v * 64 + lane) * 4;
        if (FULL || c < dim) *(float4*)(y + (size_t)(t0 + i) * dim + c) = r[i][v];
      }
    }
    return;
  }
  const float inv_d = 1.0f / (float)dim;
  float mean[RW], rstd[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) sum += (r[i][v].x + r[i][v].y) + (r[i][v].z + r[i][v].w);
    mean[i] = wave_sum(sum) * inv_d;
  }
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int c = (v * 64 + lane) * 4;
      if (FULL || c < dim) {
        float a = r[i][v].x - mean[i], b = r[i][v].y - mean[i], cc = r[i][v].z - mean[i], d = r[i][v].w - mean[i];
        sq += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    rstd[i] = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (FULL || c < dim) {
      float4 gm, bt;
      if constexpr (KB > 0) { gm = gm_pre[v]; bt = bt_pre[v]; }
      else { gm = *(const float4*)(gamma + c); bt = *(const float4*)(beta + c); }
#pragma unroll
      for (int i = 0; i < RW; ++i)
        if (t0 + i < L) {
          float4 o;
          o.x = (r[i][v].x - mean[i]) * rstd[i] * gm.x + bt.x;
          o.y = (r[i][v].y - mean[i]) * rstd[i] * gm.y + bt.y;
          o.z = (r[i][v].z - mean[i]) * rstd[i] * gm.z + bt.z;
          o.w = (r[i][v].w - mean[i]) * rstd[i] * gm.w + bt.w;
          st_row<NT_DISPATCH>(y + (size_t)(t0 + i) * dim + c, o);
          if (y16 != nullptr)
            *(uint2*)(y16 + (size_t)(t0 + i) * dim + c) = prec16 == 2 ? r4_pack4<2>(o) : r4_pack4<1>(o);
        }
    }
  }
}

// crmsa_mlp logits (rmsa.py:248-252, :305): logits[row][n] = sum_j tanh(hid[row][j]) * W2[n][j].
// One wave per row of hid [rows, hdim]; zero pad rows give zero logits by themselves.
__global__ __launch_bounds__(256) void crmsa_mlp_logits_kernel(const float* __restrict__ hid,
                                                               const float* __restrict__ w2,
                                                               float* __restrict__ logits, int rows,
                                                               int hdim, int k) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float acc[KMAX];
#pragma unroll
  for (int n = 0; n < KMAX; ++n) acc[n] = 0.f;
  for (int j = lane; j < hdim; j += 64) {
    const float t = tanhf(hid[(size_t)row * hdim + j]);
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) acc[n] += t * w2[(size_t)n * hdim + j];
  }
#pragma unroll
  for (int n = 0; n < KMAX; ++n)
    if (n < k) {
      const float a = wave_sum(acc[n]);
      if (lane == 0) logits[(size_t)row * k + n] = a;
    }
}

#ifdef RRT_NO_DISPATCH_KB5
constexpr bool DISPATCH_KB5 = false;
#else
constexpr bool DISPATCH_KB5 = true;
#endif
template <bool CRMSA>
hipError_t launch_dispatch(const float* x1, const float* x0, const float* wdisp,
                           const float* rep2, const float* gamma, const float* beta, float* y, int L,
                           int dim, int k, const GridDev& g, hipStream_t st, uint16_t* y16 = nullptr, int prec16 = 0) {
  dim3 grid((L + 4 * RW_DISPATCH - 1) / (4 * RW_DISPATCH)), block(256);
#define RRT_DISPATCH(NV)                                                                                         \
  do {                                                                                                           \
    if (RRT_ALLOW_FULL && dim == NV * 256) {                                                                     \
      if (NV <= 2 && CRMSA && k <= 3)                                                                            \
        crmsa_dispatch_ln_kernel<NV, CRMSA, true, (NV <= 2 ? 3 : 0)><<<grid, block, 0, st>>>(x1, x0, wdisp, rep2, gamma, beta, y, L, dim, k, g, y16, prec16); \
      else if (NV <= 2 && CRMSA && DISPATCH_KB5 && k <= 5) /* (configs[4]: crmsa_k = 5 -- five rows fetched, not eight) */  \
        crmsa_dispatch_ln_kernel<NV, CRMSA, true, (NV <= 2 ? 5 : 0)><<<grid, block, 0, st>>>(x1, x0, wdisp, rep2, gamma, beta, y, L, dim, k, g, y16, prec16); \
      else if (NV <= 2 && CRMSA)                                                                                 \
        crmsa_dispatch_ln_kernel<NV, CRMSA, true, (NV <= 2 ? 8 : 0)><<<grid, block, 0, st>>>(x1, x0, wdisp, rep2, gamma, beta, y, L, dim, k, g, y16, prec16); \
      else                                                                                                       \
        crmsa_dispatch_ln_kernel<NV, CRMSA, true><<<grid, block, 0, st>>>(x1, x0, wdisp, rep2, gamma, beta, y, L, dim, k, g, y16, prec16);  \
    } else                                                                                                       \
      crmsa_dispatch_ln_kernel<NV, CRMSA, false><<<grid, block, 0, st>>>(x1, x0, wdisp, rep2, gamma, beta, y, L, dim, k, g, y16, prec16); \
  } while (0)
  if (dim <= 256) RRT_DISPATCH(1);
  else if (dim <= 512) RRT_DISPATCH(2);
  else if (dim <= 1024) RRT_DISPATCH(4);
  else RRT_DISPATCH(8);
#undef RRT_DISPATCH
  return hipGetLastError();
}

}  // namespace

#ifdef RRT_TRACE
RRT_TRACE_DEFINE_READER(rrt_debug_trace_crmsa)
#endif

hipError_t launch_crmsa_logits(const float* x1, const float* gamma, const float* beta,
                               const float* phi, float* mean_rstd, float* logits, int dim, int k,
                               const GridDev& g8, hipStream_t st) {
  static const bool old_pair = rrt_tune_env("RRT_CRMSA_OLD_PAIR") != nullptr;      // (A/B in a tuning build)
  if (dim == 512 && k >= 1 && k <= KMAX && !old_pair) {
    const dim3 grid2((g8.Np + 7) / 8);
    switch (k) {
#define RRT_LG512(K_) case K_: crmsa_logits512_kernel<K_><<<grid2, 256, 0, st>>>(x1, gamma, beta, phi, mean_rstd, logits, g8); break;
      RRT_LG512(1) RRT_LG512(2) RRT_LG512(3) RRT_LG512(4) RRT_LG512(5) RRT_LG512(6) RRT_LG512(7) RRT_LG512(8)
#undef RRT_LG512
    }
    return hipGetLastError();
  }
  const int ngroups = (g8.Np + 4 * RW - 1) / (4 * RW);
  dim3 grid(ngroups < 1024 ? ngroups : 1024), block(256);     // <= 4 resident blocks per CU, grid-stride
  const size_t lds = (size_t)dim * k * sizeof(float);
#define RRT_LOGITS(NV) \
  do {                                                                                                      \
    if (RRT_ALLOW_FULL && dim == NV * 256) crmsa_logits_kernel<NV, true><<<grid, block, lds, st>>>(x1, gamma, beta, phi, mean_rstd, logits, dim, k, g8);  \
    else crmsa_logits_kernel<NV, false><<<grid, block, lds, st>>>(x1, gamma, beta, phi, mean_rstd, logits, dim, k, g8);               \
  } while (0)
  if (dim <= 256) RRT_LOGITS(1);
  else if (dim <= 512) RRT_LOGITS(2);
  else if (dim <= 1024) RRT_LOGITS(4);
  else RRT_LOGITS(8);
#undef RRT_LOGITS
  return hipGetLastError();
}

// logits + combine as one kernel where it applies (dim 512, regions of <= 144 tokens, k <= 3); false -> the two kernels
bool crmsa_region_supported(int dim, int k, const GridDev& g8) {
  return dim == 512 && k >= 1 && k <= REGION_KMAX && g8.P <= 16 * REGION_NR_MAX;
}
// Whether the encoder forward uses it: opt-in (RRT_CRMSA_REGION=1).  Measured at N = 9000:
// the 64-block kernel takes 20.2 us against 7.6 + 8.8 for the two chip-wide ones (its LayerNorm / dot-product VALU work
// sits on a quarter of the CUs), so one bag in flight loses 1.7 %; two bags in flight gain 4 % (it leaves 192 CUs to
// the other bag's R-MSA kernel), at the price of that kernel's in-flight duration (0.74 -> 0.65-0.70 of peak).
bool crmsa_region_enabled() {
  static const bool on = rrt_tune_env("RRT_CRMSA_REGION") != nullptr;
  return on;
}
template <int GK>
static hipError_t launch_region_gk(const float* x1, const float* gamma, const float* beta, const float* phi, float* mean_rstd,
                                   float* logits, float* wdisp, float* rep, uint16_t* rep16, int prec16, int k, const GridDev& g8,
                                   hipStream_t st) {
  const size_t lds = (size_t)(REGION_KMAX * 512 + REGION_NR_MAX * 16 * (2 * REGION_KMAX + 2)) * 4 +
                     (size_t)8 * REGION_KMAX * 128 * 16;
  auto kern = crmsa_region_kernel<GK>;
  static OncePerDevice once;
  if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<dim3(g8.rs * g8.rs), dim3(1024), lds, st>>>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, k, g8);
  return hipGetLastError();
}
hipError_t launch_crmsa_region(const float* x1, const float* gamma, const float* beta, const float* phi,
                               float* mean_rstd, float* logits, float* wdisp, float* rep, int k, const GridDev& g8,
                               hipStream_t st, uint16_t* rep16, int prec16) {
  static const bool no_gpr = rrt_tune_env("RRT_NO_REGION_GPR") != nullptr;
  const bool al16 = ((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)phi) & 15) == 0);
  if (!no_gpr && al16) {
    if (k == 1) return launch_region_gk<1>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, k, g8, st);
    if (k == 2) return launch_region_gk<2>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, k, g8, st);
    if (k == 3) return launch_region_gk<3>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, k, g8, st);
  }
  return launch_region_gk<0>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, k, g8, st);
}

// Shapes.  Regions of <= 144 tokens (bags up to ~9.2 k patches): four blocks of 12 waves x 3 rows per region (fewest
// hand-overs: best with one bag in flight; the tuning build also has eight blocks of 4 waves x 5 rows, RRT_REGION4_CFG=8:
// one wave per SIMD at ~100 VGPRs, which fits next to two waves of the other bag's fused R-MSA kernel).  Round 3: regions
// of <= 288 tokens take 8 blocks, <= 576 tokens 16 blocks of the same 12 x 3 shape (N = 30000: P8 = 484), and k <= 8
// representatives (BASELINE configs[4]: crmsa_k = 5) the KM = 8 instantiation.
bool crmsa_region4_supported(int dim, int k, const GridDev& g8) {
  static const bool off = rrt_tune_env("RRT_NO_CRMSA_REGION4") != nullptr;
  return !off && dim == 512 && k >= 1 && k <= R4_KMAX && g8.P >= 4 && g8.P <= 36 * R4_NB_MAX;
}
size_t crmsa_region4_scratch_floats(const GridDev& g8, int k) {
  const int nb = g8.P > 288 ? 16 : 8;               // (8 also covers the tuning build's eight-block shape of small regions)
  return (size_t)g8.rs * g8.rs * nb * (k <= 3 ? 3 : R4_KMAX) * (512 + 8);
}
template <int NB, int NW, int NR, int KM, bool GPR = false>
static hipError_t launch_region4_cfg(const float* x1, const float* gamma, const float* beta, const float* phi,
                                     float* mean_rstd, float* logits, float* wdisp, float* rep, uint16_t* rep16, int prec16,
                                     float* part_g, int* counters, int k, const GridDev& g8, hipStream_t st) {
  static_assert(NB <= R4_NB_MAX && NW >= KM && NW % 2 == 0 && KM <= R4_KMAX, "region4 shape");
  constexpr int PQM = NR * NW, KC = KM < 4 ? KM : 4;
  const size_t lds = (size_t)(KM * 512 + PQM * (2 * KM + 2)) * 4 + (size_t)(NW / 2) * KC * 128 * 16;
  auto kern = crmsa_region4_kernel<NB, NW, NR, KM, GPR>;
  static OncePerDevice once;
  if (lds > 64 * 1024 && once.first())
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<dim3(g8.rs * g8.rs * NB), dim3(64 * NW), lds, st>>>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16,
                                                             part_g, counters, k, g8);
  return hipGetLastError();
}
hipError_t launch_crmsa_region4(const float* x1, const float* gamma, const float* beta, const float* phi,
                                float* mean_rstd, float* logits, float* wdisp, float* rep, uint16_t* rep16, int prec16,
                                float* part_g, int* counters, int k, const GridDev& g8, hipStream_t st) {
#define RRT_R4(NB_, NW_, NR_)                                                                                          \
  return k <= 3 ? launch_region4_cfg<NB_, NW_, NR_, 3>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, \
                                                       part_g, counters, k, g8, st)                                   \
                : launch_region4_cfg<NB_, NW_, NR_, 8>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, \
                                                       part_g, counters, k, g8, st)
  static const int cfg = rrt_tune_env("RRT_REGION4_CFG") ? atoi(rrt_tune_env("RRT_REGION4_CFG")) : 4;
  static const bool r4_gpr = rrt_tune_env("RRT_NO_REGION4_GPR") == nullptr;
#ifdef RRT_TUNING
  if (cfg == 86 && g8.P <= 576) { RRT_R4(8, 12, 6); }        // 8 blocks x 12 waves x 6 rows
  if (cfg == 164 && g8.P <= 512) { RRT_R4(16, 8, 4); }       // 16 blocks x 8 waves x 4 rows
  if (cfg == 166 && g8.P <= 576) { RRT_R4(16, 12, 3); }
  if (cfg == 84 && g8.P <= 384) { RRT_R4(8, 12, 4); }
#endif
  if (g8.P > 288) { RRT_R4(16, 12, 3); }
  if (g8.P > 144) { RRT_R4(8, 12, 3); }
  if (cfg == 8 && g8.P > 96 && k <= 3)
    return launch_region4_cfg<8, 4, 5, 3>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, part_g, counters, k, g8, st);
  // (round 6) k = 4, 5 (BASELINE configs[4]: crmsa_k = 5) at the four-block shape: registers and loops for five representatives,
  // not eight (the records are smaller than the scratch's eight-representative ones: the layout is the kernel's own)
  if (g8.P <= 144 && cfg == 4 && k > 3 && k <= 5 && !(k == 5 && r4_gpr))
    return launch_region4_cfg<4, 12, 3, 5>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, part_g, counters, k, g8, st);
  // (round 6) the representative counts of the BASELINE configs with gamma . phi in registers and wave totals four at a time
  // (16-byte-aligned gamma / beta / phi: float4 loads)
  if (g8.P <= 144 && cfg == 4 && r4_gpr && ((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)phi) & 15) == 0)) {
#define RRT_R4G(K_) if (k == K_) return launch_region4_cfg<4, 12, 3, K_, true>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16, part_g, counters, k, g8, st)
    RRT_R4G(1); RRT_R4G(3); RRT_R4G(5);
#undef RRT_R4G
  }
  RRT_R4(4, 12, 3);
#undef RRT_R4
}

// one pass over x1 for regions of more than 144 tokens (crmsa_stream4_kernel): dim 512, k <= 8, 16 <= P
bool crmsa_stream4_supported(int dim, int k, const GridDev& g8) {
  static const bool off = rrt_tune_env("RRT_NO_CRMSA_STREAM4") != nullptr;
  return !off && dim == 512 && k >= 1 && k <= R4_KMAX && g8.P >= 16 && g8.P <= 4 * S4_PQMAX;
}
template <int NW, int NR, int KM>
static hipError_t launch_stream4_cfg(const float* x1, const float* gamma, const float* beta, const float* phi,
                                     float* mean_rstd, float* logits, float* wdisp, float* rep, uint16_t* rep16, int prec16,
                                     float* part_g, int* counters, const GridDev& g8, hipStream_t st) {
  static_assert(NW >= KM && NW % 2 == 0 && KM <= R4_KMAX, "stream4 shape");
  constexpr int KC = KM < 4 ? KM : 4;
  const size_t lds = (size_t)(KM * NW * 8 + S4_PQMAX * (KM + 2)) * 4 + (size_t)(NW / 2) * KC * 128 * 16;
  auto kern = crmsa_stream4_kernel<NW, NR, KM>;
  static OncePerDevice once;
  if (lds > 64 * 1024 && once.first())
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<dim3(g8.rs * g8.rs * 4), dim3(64 * NW), lds, st>>>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, rep16, prec16,
                                                            part_g, counters, g8);
  return hipGetLastError();
}
// part_g: crmsa_region4_scratch_floats(g8, k) floats (its records are at least as large and at least as many); counters zero;
// gamma, beta, phi on 16-byte boundaries (float4 loads, as launch_crmsa_logits' dim-512 kernel)
hipError_t launch_crmsa_stream4(const float* x1, const float* gamma, const float* beta, const float* phi,
                                float* mean_rstd, float* logits, float* wdisp, float* rep, uint16_t* rep16, int prec16,
                                float* part_g, int* counters, int k, const GridDev& g8, hipStream_t st) {
  if (!crmsa_stream4_supported(512, k, g8)) return hipErrorInvalidValue;
  // twelve waves (three per SIMD) while the accumulators and gamma . phi of <= 3 representatives fit 168 registers; more
  // representatives: eight waves with the 256-register budget (at twelve the k = 5 / 8 forms spilled)
#define RRT_S4(NW_, NR_, KM_) case KM_: return launch_stream4_cfg<NW_, NR_, KM_>(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, \
                                                                                rep16, prec16, part_g, counters, g8, st)
  switch (k) {      // (k = 7, 8: two rows per trip -- with three the 256 registers of an eight-wave block spilled)
    RRT_S4(12, 3, 1); RRT_S4(12, 3, 2); RRT_S4(12, 3, 3); RRT_S4(8, 3, 4); RRT_S4(8, 3, 5); RRT_S4(8, 3, 6); RRT_S4(8, 2, 7); RRT_S4(8, 2, 8);
  }
#undef RRT_S4
  return hipErrorInvalidValue;
}

hipError_t launch_crmsa_combine(const float* x1, const float* gamma, const float* beta,
                                const float* mean_rstd, const float* logits, float* wdisp,
                                float* rep, uint16_t* rep16, int prec16, int dim, int k, const GridDev& g8, hipStream_t st) {
  dim3 grid(g8.rs * g8.rs, (dim + 63) / 64), block(256);
  const size_t lds = ((size_t)g8.P * KMAX + ((g8.P + 3) & ~3)) * 4 + (size_t)16 * KMAX * 16 * sizeof(float4);
  if (lds > 150 * 1024) return hipErrorInvalidValue;   // P8 > ~3500 tokens per region (N > 220k)
  const bool vnorm = mean_rstd == nullptr;   // crmsa_mlp path: x1 holds LN(x1) rows in region-major order
  static const bool old_pair = rrt_tune_env("RRT_CRMSA_OLD_PAIR") != nullptr;      // (A/B in a tuning build)
  if (!vnorm && dim == 512 && k >= 1 && k <= KMAX && !old_pair) {
    const size_t lds2 = (((size_t)g8.P * k + 3) & ~(size_t)3) * 4 + (size_t)((g8.P + 1) & ~1) * 8 + (size_t)16 * k * 16 * sizeof(float4);
    if (lds2 <= 64 * 1024) {
      switch (k) {
#define RRT_CB512(K_) case K_: crmsa_combine512_kernel<K_><<<grid, block, lds2, st>>>(x1, gamma, beta, mean_rstd, logits, wdisp, rep, rep16, prec16, g8); break;
        RRT_CB512(1) RRT_CB512(2) RRT_CB512(3) RRT_CB512(4) RRT_CB512(5) RRT_CB512(6) RRT_CB512(7) RRT_CB512(8)
#undef RRT_CB512
      }
      return hipGetLastError();
    }
  }
  auto kern = vnorm ? crmsa_combine_kernel<true> : crmsa_combine_kernel<false>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<grid, block, lds, st>>>(x1, gamma, beta, mean_rstd, logits, wdisp, rep, rep16, prec16, dim, k, g8);
  return hipGetLastError();
}

// dim % 64 == 0 (the slabs are 64 columns wide), dim <= 512; any k <= 8, any region size whose tables fit the LDS
static size_t combine_parts_lds(int P, int KM) {
  // logits / coefficients [P][KM], (mean, rstd) [P], the row groups' partials [16][KM + 1][16] float4 and (slab 0) a
  // second coefficient table [P][KM]
  return (size_t)P * KM * 4 * 2 + (size_t)((P + 1) & ~1) * 8 + (size_t)16 * (KM + 1) * 16 * 16;
}
bool crmsa_combine_parts_supported(int dim, int k, const GridDev& g8) {
  static const bool off = rrt_tune_env("RRT_NO_CRMSA_PARTS") != nullptr;
  return !off && dim % 64 == 0 && dim >= 64 && dim <= 512 && k >= 1 && k <= KMAX && combine_parts_lds(g8.P, k <= 4 ? 4 : 8) <= 150 * 1024;
}
size_t crmsa_parts_floats(long n_tokens, int dim, int k) { return (size_t)n_tokens * (dim / 64) * ((2 + k + 3) & ~3); }
hipError_t launch_crmsa_combine_parts(const float* x1, const float* part, const float* gamma, const float* beta,
                                      const float* phi, float* wdisp, float* rep, uint16_t* rep16, int prec16, int dim,
                                      int k, const GridDev& g8, hipStream_t st) {
  if (!crmsa_combine_parts_supported(dim, k, g8)) return hipErrorInvalidValue;
  dim3 grid(g8.rs * g8.rs, dim / 64), block(256);
  constexpr int TR = 9;                               // P = 144: every row of the region in flight at once
  static const bool no_coal = rrt_tune_env("RRT_NO_CPARTS_COAL") != nullptr;
  // bit 0: gamma / beta / phi on 16-byte boundaries; bit 1: the records come in coalesced and pass through LDS (see the kernel)
  const bool coal = !no_coal && k >= 3 && k <= 4 && dim == 512 && g8.P <= 16 * TR && (((uintptr_t)part) & 15) == 0;
  const int al16 = ((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)phi) & 15) == 0 ? 1 : 0) | (coal ? 2 : 0);
#define RRT_CPARTS(KM_)                                                                                              \
  do {                                                                                                               \
    const size_t lds = combine_parts_lds(g8.P, KM_) + (coal ? (size_t)16 * TR * 17 * 16 : 0);                        \
    const bool ck = coal && KM_ == 4;                                                                                \
    auto kern = ck ? crmsa_combine_parts_kernel<TR, KM_, true> : crmsa_combine_parts_kernel<TR, KM_, false>;         \
    static OncePerDevice once_coal, once_plain;      /* one per concrete kernel: the attribute is the kernel's own */ \
    if (lds > 64 * 1024 && (ck ? once_coal : once_plain).first())                                                    \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);          \
    kern<<<grid, block, lds, st>>>(x1, part, gamma, beta, phi, wdisp, rep, rep16, prec16, dim, k, dim / 64, al16, g8); \
  } while (0)
  if (k <= 4) RRT_CPARTS(4);
  else RRT_CPARTS(8);
#undef RRT_CPARTS
  return hipGetLastError();
}

hipError_t launch_crmsa_mlp_logits(const float* hid, const float* w2, float* logits, int rows, int hdim,
                                   int k, hipStream_t st) {
  crmsa_mlp_logits_kernel<<<dim3((rows + 3) / 4), 256, 0, st>>>(hid, w2, logits, rows, hdim, k);
  return hipGetLastError();
}

hipError_t launch_crmsa_dispatch_ln(const float* x1, const float* x0, const float* wdisp,
                                    const float* rep2, const float* gamma,
                                    const float* beta, float* y, int dim, int k, const GridDev& g8,
                                    hipStream_t st, uint16_t* y16, int prec16) {
  if (y16 != nullptr && (gamma == nullptr || (prec16 != 1 && prec16 != 2) || dim % 4 != 0)) return hipErrorInvalidValue;
  return launch_dispatch<true>(x1, x0, wdisp, rep2, gamma, beta, y, g8.L, dim, k, g8, st, y16, prec16);
}

hipError_t launch_layernorm(const float* x1, const float* x0, const float* gamma,
                            const float* beta, float* y, int L, int dim, hipStream_t st) {
  GridDev g{};
  g.L = L;
  g.H = g.s = g.rs = g.P = 1;
  return launch_dispatch<false>(x1, x0, nullptr, nullptr, gamma, beta, y, L, dim, 0, g, st);
}
