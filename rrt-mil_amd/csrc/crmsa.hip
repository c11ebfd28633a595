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

// rows handled by one wave (independent loads in flight per wave = RW * NV float4)
constexpr int RW = 4;
constexpr int RW_DISPATCH = 2;

template <int NV>
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
      r[i][v] = (t < g.L && c < dim) ? *(const float4*)(x1 + (size_t)t * dim + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      sum[i] += (r[i][v].x + r[i][v].y) + (r[i][v].z + r[i][v].w);
    }
  }
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
      if (c < dim) {
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
    if (c < dim) {
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
      if (c < dim) {
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
                                                            float* __restrict__ rep, int dim, int k,
                                                            GridDev g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Wc = (float*)smem;                         // [P][KMAX] combine coefficient x rstd
  int* tok = (int*)(Wc + (size_t)g.P * KMAX);       // [P] token index or -1 (pad)
  float4* part = (float4*)(tok + ((g.P + 3) & ~3)); // [16 row groups][KMAX][16 col lanes]
  __shared__ float s_stat[KMAX][3];
  __shared__ float s_c0[KMAX][4], s_c1[KMAX][4];    // per-wave partials of c0, c1
  const int reg = blockIdx.x, slab = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = g.rs * g.rs;
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
    for (int p0 = rg; p0 < g.P; p0 += 64) {        // 4 independent rows per trip
      float4 xv[4];
      int pp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pp[u] = p0 + 16 * u;
        const int t = pp[u] < g.P ? tok[pp[u]] : -1;
        xv[u] = t >= 0 ? *(const float4*)(x1 + (size_t)t * dim + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < 0) pp[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (pp[u] >= 0) {
#pragma unroll
          for (int n = 0; n < KMAX; ++n)
            if (n < k) {
              const float w = Wc[pp[u] * KMAX + n];
              acc[n].x += w * xv[u].x; acc[n].y += w * xv[u].y; acc[n].z += w * xv[u].z; acc[n].w += w * xv[u].w;
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
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Inference form of logits + combine: ONE pass over x1 (crmsa_scan_kernel) and a small merge.
// A block owns a chunk of one region's tokens; a wave takes four rows at a time, normalises them in registers
// (LN2), forms the k logits, and feeds the rows straight into the combine as an ONLINE softmax over the region's
// tokens -- running max m[n], normaliser l[n] and the weighted row sum acc[n][:] = sum_p exp(Lg[n,p] - m[n]) v_p
// (pad tokens: v = 0 and Lg = 0, so they count in l and add no row, exactly like the reference's zero rows after
// LayerNorm).  x1 is read once; the [R, k, P, D] temporaries of the reference and the second pass of
// crmsa_combine_kernel do not exist.  The block's four waves merge through 8 KiB of LDS; chunk partials
// (m, l, min, row) go to `part`, crmsa_merge_kernel folds the chunks of a region into rep [k, R, D] and the
// region statistics (min, max) the dispatch weights need.
constexpr int SCAN_RW = 4;                 // rows in flight per wave
constexpr int SCAN_HDR = 4;                // floats in front of a partial row: m, l, min, (unused)

template <int NV, int KK>   // KK >= k: representatives held in registers
__global__ __launch_bounds__(256) void crmsa_scan_kernel(const float* __restrict__ x1,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ phi,
                                                         float* __restrict__ logits, float* __restrict__ part,
                                                         int dim, int k, int nch, int CH, GridDev g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* phi_t = (float*)smem;                    // [k][dim]
  float* mrg = phi_t + (size_t)k * dim;           // [4 waves][SCAN_HDR + dim]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int n = 0; n < k; ++n)
    for (int d = threadIdx.x; d < dim; d += 256) phi_t[n * dim + d] = phi[(size_t)d * k + n];
  __syncthreads();
  const int reg = blockIdx.x / nch, ch = blockIdx.x - reg * nch;
  const int p_lo = ch * CH, p_hi = min(p_lo + CH, g.P);
  const int ri = fdiv(reg, g.rs, g.inv_rs), rj = reg - ri * g.rs;
  float4 gm[NV], bt[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * 64 + lane) * 4;
    gm[v] = c < dim ? *(const float4*)(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    bt[v] = c < dim ? *(const float4*)(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float m[KK], l[KK], mn[KK];
  float4 acc[KK][NV];
#pragma unroll
  for (int n = 0; n < KK; ++n) {
    m[n] = -3.0e38f; l[n] = 0.f; mn[n] = 3.0e38f;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[n][v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_d = 1.0f / (float)dim;
  for (int p0 = p_lo + wave * SCAN_RW; p0 < p_hi; p0 += 4 * SCAN_RW) {
    float4 r[SCAN_RW][NV];
    bool real[SCAN_RW];
    float sum[SCAN_RW];
#pragma unroll
    for (int i = 0; i < SCAN_RW; ++i) {
      const int p = p0 + i;
      const int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
      const int t = (ri * g.s + pi) * g.H + rj * g.s + pj;
      real[i] = p < p_hi && t < g.L;
      sum[i] = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        r[i][v] = (real[i] && c < dim) ? *(const float4*)(x1 + (size_t)t * dim + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        sum[i] += (r[i][v].x + r[i][v].y) + (r[i][v].z + r[i][v].w);
      }
    }
    float mean[SCAN_RW], rstd[SCAN_RW];
#pragma unroll
    for (int i = 0; i < SCAN_RW; ++i) mean[i] = wave_sum(sum[i]) * inv_d;
#pragma unroll
    for (int i = 0; i < SCAN_RW; ++i) {
      float sq = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < dim) {
          const float a = r[i][v].x - mean[i], b = r[i][v].y - mean[i], cc = r[i][v].z - mean[i], d = r[i][v].w - mean[i];
          sq += (a * a + b * b) + (cc * cc + d * d);
        }
      }
      rstd[i] = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
    }
    // rows <- LN2(x1) (pad rows stay exact zeros: they are not layer-normed in the reference)
#pragma unroll
    for (int i = 0; i < SCAN_RW; ++i)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (real[i]) {
          r[i][v].x = (r[i][v].x - mean[i]) * rstd[i] * gm[v].x + bt[v].x;
          r[i][v].y = (r[i][v].y - mean[i]) * rstd[i] * gm[v].y + bt[v].y;
          r[i][v].z = (r[i][v].z - mean[i]) * rstd[i] * gm[v].z + bt[v].z;
          r[i][v].w = (r[i][v].w - mean[i]) * rstd[i] * gm[v].w + bt[v].w;
        }
      }
#pragma unroll
    for (int n = 0; n < KK; ++n) {
      if (n >= k) break;
      float a[SCAN_RW];
#pragma unroll
      for (int i = 0; i < SCAN_RW; ++i) a[i] = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < dim) {
          const float4 ph = *(const float4*)(phi_t + n * dim + c);
#pragma unroll
          for (int i = 0; i < SCAN_RW; ++i)
            a[i] += (r[i][v].x * ph.x + r[i][v].y * ph.y) + (r[i][v].z * ph.z + r[i][v].w * ph.w);
        }
      }
      float lgv[SCAN_RW], top = m[n];
#pragma unroll
      for (int i = 0; i < SCAN_RW; ++i) {
        lgv[i] = real[i] ? wave_sum(a[i]) : 0.f;             // wave-uniform; pad tokens: exactly 0
        if (p0 + i < p_hi) {
          if (lane == 0) logits[((size_t)reg * g.P + p0 + i) * k + n] = lgv[i];
          top = fmaxf(top, lgv[i]);
          mn[n] = fminf(mn[n], lgv[i]);
        }
      }
      if (top > m[n]) {                                        // (wave-uniform) new running max: rescale
        const float sc = __expf(m[n] - top);
        l[n] *= sc;
#pragma unroll
        for (int v = 0; v < NV; ++v) { acc[n][v].x *= sc; acc[n][v].y *= sc; acc[n][v].z *= sc; acc[n][v].w *= sc; }
        m[n] = top;
      }
#pragma unroll
      for (int i = 0; i < SCAN_RW; ++i)
        if (p0 + i < p_hi) {
          const float e = __expf(lgv[i] - m[n]);
          l[n] += e;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            acc[n][v].x += e * r[i][v].x; acc[n][v].y += e * r[i][v].y;
            acc[n][v].z += e * r[i][v].z; acc[n][v].w += e * r[i][v].w;
          }
        }
    }
  }
  // merge the four waves, one representative at a time (8 KiB of LDS): partial = (m, l, min | row)
  const int stride = SCAN_HDR + dim;
  for (int n = 0; n < k; ++n) {
    float* mine = mrg + wave * stride;
    float mm = 0.f, ll = 0.f, mi = 0.f;
    float4 row[NV];
#pragma unroll
    for (int q = 0; q < KK; ++q)
      if (q == n) {
        mm = m[q]; ll = l[q]; mi = mn[q];
#pragma unroll
        for (int v = 0; v < NV; ++v) row[v] = acc[q][v];
      }
    if (lane == 0) { mine[0] = mm; mine[1] = ll; mine[2] = mi; }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < dim) *(float4*)(mine + SCAN_HDR + c) = row[v];
    }
    __syncthreads();
    float* out = part + ((size_t)(reg * nch + ch) * k + n) * stride;
    const float M = fmaxf(fmaxf(mrg[0], mrg[stride]), fmaxf(mrg[2 * stride], mrg[3 * stride]));
    float sc[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) sc[w] = mrg[w * stride + 1] > 0.f ? __expf(mrg[w * stride] - M) : 0.f;   // idle wave: l = 0
    if (threadIdx.x == 0) {
      out[0] = M;
      out[1] = (sc[0] * mrg[1] + sc[1] * mrg[stride + 1]) + (sc[2] * mrg[2 * stride + 1] + sc[3] * mrg[3 * stride + 1]);
      out[2] = fminf(fminf(mrg[2], mrg[stride + 2]), fminf(mrg[2 * stride + 2], mrg[3 * stride + 2]));
      out[3] = 0.f;
    }
    for (int c = threadIdx.x * 4; c < dim; c += 1024) {
      const float4 a0 = *(const float4*)(mrg + SCAN_HDR + c), a1 = *(const float4*)(mrg + stride + SCAN_HDR + c);
      const float4 a2 = *(const float4*)(mrg + 2 * stride + SCAN_HDR + c), a3 = *(const float4*)(mrg + 3 * stride + SCAN_HDR + c);
      float4 o;
      o.x = (sc[0] * a0.x + sc[1] * a1.x) + (sc[2] * a2.x + sc[3] * a3.x);
      o.y = (sc[0] * a0.y + sc[1] * a1.y) + (sc[2] * a2.y + sc[3] * a3.y);
      o.z = (sc[0] * a0.z + sc[1] * a1.z) + (sc[2] * a2.z + sc[3] * a3.z);
      o.w = (sc[0] * a0.w + sc[1] * a1.w) + (sc[2] * a2.w + sc[3] * a3.w);
      *(float4*)(out + SCAN_HDR + c) = o;
    }
    __syncthreads();
  }
}

// rep[n, reg, :] = sum_c exp(m_c - M) row_c / sum_c exp(m_c - M) l_c ; stats[reg][n] = (region min, region max) of the logits
__global__ __launch_bounds__(128) void crmsa_merge_kernel(const float* __restrict__ part, float* __restrict__ rep,
                                                          float* __restrict__ stats, int dim, int k, int nch, int R) {
  const int reg = blockIdx.x / k, n = blockIdx.x - reg * k;
  const int stride = SCAN_HDR + dim;
  const float* base = part + ((size_t)reg * nch * k + n) * stride;
  float M = -3.0e38f, mi = 3.0e38f;
  for (int c = 0; c < nch; ++c) {
    const float* pc = base + (size_t)c * k * stride;
    if (pc[1] > 0.f) M = fmaxf(M, pc[0]);
    mi = fminf(mi, pc[2]);
  }
  float L = 0.f;
  for (int c = 0; c < nch; ++c) {
    const float* pc = base + (size_t)c * k * stride;
    if (pc[1] > 0.f) L += __expf(pc[0] - M) * pc[1];
  }
  const float inv = 1.0f / L;
  for (int col = threadIdx.x * 4; col < dim; col += 512) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < nch; ++c) {
      const float* pc = base + (size_t)c * k * stride;
      if (pc[1] > 0.f) {
        const float sc = __expf(pc[0] - M);
        const float4 a = *(const float4*)(pc + SCAN_HDR + col);
        o.x += sc * a.x; o.y += sc * a.y; o.z += sc * a.z; o.w += sc * a.w;
      }
    }
    *(float4*)(rep + ((size_t)n * R + reg) * dim + col) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
  }
  if (threadIdx.x == 0) {
    stats[((size_t)reg * k + n) * 2] = mi;
    stats[((size_t)reg * k + n) * 2 + 1] = M;
  }
}

// WFLY: `wdisp` holds the raw logits [Np8, k] and `stats` the regions' (min, max) per representative; the dispatch
// weight minmax_p(Lg) * softmax_k(Lg) (rmsa.py:310-314,324-325) is formed here instead of being stored and re-read
template <int NV, bool CRMSA, bool WFLY = false>
__global__ __launch_bounds__(256) void crmsa_dispatch_ln_kernel(
    const float* __restrict__ x1, const float* __restrict__ x0, const float* __restrict__ wdisp,
    const float* __restrict__ rep2, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, int L, int dim, int k, GridDev g,
    const float* __restrict__ stats = nullptr) {
  constexpr int RW = RW_DISPATCH;
  const int lane = threadIdx.x & 63;
  const int t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (t0 >= L) return;
  const int R = g.rs * g.rs;
  float4 r[RW][NV];
  // issue every row load first (x1, shortcut), then the dispatch weights
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int t = t0 + i < L ? t0 + i : L - 1;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int c = (v * 64 + lane) * 4;
      r[i][v] = c < dim ? *(const float4*)(x1 + (size_t)t * dim + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 && c < dim) {
        float4 s = *(const float4*)(x0 + (size_t)t * dim + c);
        r[i][v].x += s.x; r[i][v].y += s.y; r[i][v].z += s.z; r[i][v].w += s.w;
      }
    }
  }
  if (CRMSA) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int t = t0 + i < L ? t0 + i : L - 1;
      const int slot = token_to_slot(t, g);
      const int reg = fdiv(slot, g.P, g.inv_P);
      const float* wd = wdisp + (size_t)slot * k;
      float wfly[KMAX];
      if (WFLY) {
        float lgv[KMAX], mxk = -3.0e38f, se = 0.f;
#pragma unroll
        for (int n = 0; n < KMAX; ++n)
          if (n < k) { lgv[n] = wd[n]; mxk = fmaxf(mxk, lgv[n]); }
#pragma unroll
        for (int n = 0; n < KMAX; ++n)
          if (n < k) { wfly[n] = __expf(lgv[n] - mxk); se += wfly[n]; }
        const float inv = 1.0f / se;
#pragma unroll
        for (int n = 0; n < KMAX; ++n)
          if (n < k) {
            const float lo = stats[((size_t)reg * k + n) * 2], hi = stats[((size_t)reg * k + n) * 2 + 1];
            wfly[n] = (lgv[n] - lo) / (hi - lo + 1e-8f) * (wfly[n] * inv);
          }
      }
#pragma unroll
      for (int n = 0; n < KMAX; ++n)
        if (n < k) {
          const float w = WFLY ? wfly[n] : wd[n];
          const float* rp = rep2 + ((size_t)n * R + reg) * dim;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            int c = (v * 64 + lane) * 4;
            if (c < dim) {
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
        if (c < dim) *(float4*)(y + (size_t)(t0 + i) * dim + c) = r[i][v];
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
      if (c < dim) {
        float a = r[i][v].x - mean[i], b = r[i][v].y - mean[i], cc = r[i][v].z - mean[i], d = r[i][v].w - mean[i];
        sq += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    rstd[i] = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (c < dim) {
      const float4 gm = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
#pragma unroll
      for (int i = 0; i < RW; ++i)
        if (t0 + i < L) {
          float4 o;
          o.x = (r[i][v].x - mean[i]) * rstd[i] * gm.x + bt.x;
          o.y = (r[i][v].y - mean[i]) * rstd[i] * gm.y + bt.y;
          o.z = (r[i][v].z - mean[i]) * rstd[i] * gm.z + bt.z;
          o.w = (r[i][v].w - mean[i]) * rstd[i] * gm.w + bt.w;
          *(float4*)(y + (size_t)(t0 + i) * dim + c) = o;
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

template <bool CRMSA, bool WFLY = false>
hipError_t launch_dispatch(const float* x1, const float* x0, const float* wdisp,
                           const float* rep2, const float* gamma, const float* beta, float* y, int L,
                           int dim, int k, const GridDev& g, hipStream_t st, const float* stats = nullptr) {
  dim3 grid((L + 4 * RW_DISPATCH - 1) / (4 * RW_DISPATCH)), block(256);
#define RRT_DISPATCH(NV)                                                                          \
  crmsa_dispatch_ln_kernel<NV, CRMSA, WFLY><<<grid, block, 0, st>>>(x1, x0, wdisp, rep2, gamma, \
                                                                    beta, y, L, dim, k, g, stats)
  if (dim <= 256) RRT_DISPATCH(1);
  else if (dim <= 512) RRT_DISPATCH(2);
  else if (dim <= 1024) RRT_DISPATCH(4);
  else RRT_DISPATCH(8);
#undef RRT_DISPATCH
  return hipGetLastError();
}

}  // namespace

hipError_t launch_crmsa_logits(const float* x1, const float* gamma, const float* beta,
                               const float* phi, float* mean_rstd, float* logits, int dim, int k,
                               const GridDev& g8, hipStream_t st) {
  const int ngroups = (g8.Np + 4 * RW - 1) / (4 * RW);
  dim3 grid(ngroups < 1024 ? ngroups : 1024), block(256);     // <= 4 resident blocks per CU, grid-stride
  const size_t lds = (size_t)dim * k * sizeof(float);
#define RRT_LOGITS(NV) \
  crmsa_logits_kernel<NV><<<grid, block, lds, st>>>(x1, gamma, beta, phi, mean_rstd, logits, dim, k, g8)
  if (dim <= 256) RRT_LOGITS(1);
  else if (dim <= 512) RRT_LOGITS(2);
  else if (dim <= 1024) RRT_LOGITS(4);
  else RRT_LOGITS(8);
#undef RRT_LOGITS
  return hipGetLastError();
}

hipError_t launch_crmsa_combine(const float* x1, const float* gamma, const float* beta,
                                const float* mean_rstd, const float* logits, float* wdisp,
                                float* rep, int dim, int k, const GridDev& g8, hipStream_t st) {
  dim3 grid(g8.rs * g8.rs, (dim + 63) / 64), block(256);
  const size_t lds = ((size_t)g8.P * KMAX + ((g8.P + 3) & ~3)) * 4 + (size_t)16 * KMAX * 16 * sizeof(float4);
  if (lds > 150 * 1024) return hipErrorInvalidValue;   // P8 > ~3500 tokens per region (N > 220k)
  const bool vnorm = mean_rstd == nullptr;   // crmsa_mlp path: x1 holds LN(x1) rows in region-major order
  auto kern = vnorm ? crmsa_combine_kernel<true> : crmsa_combine_kernel<false>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<grid, block, lds, st>>>(x1, gamma, beta, mean_rstd, logits, wdisp, rep, dim, k, g8);
  return hipGetLastError();
}

// one-pass logits + combine (inference): `part` = crmsa_scan_workspace bytes; logits [Np8, k] region-major,
// rep [k, R, dim], stats [R, k, 2]
int crmsa_scan_chunks(const GridDev& g8) {
  const int R = g8.rs * g8.rs;
  int nch = (256 + R - 1) / R;                                   // fill the chip: R * nch >= 256 blocks ...
  const int cap = (g8.P + 4 * SCAN_RW - 1) / (4 * SCAN_RW);     // ... but at least one four-row trip per wave
  if (nch > cap) nch = cap;
  return nch < 1 ? 1 : nch;
}
size_t crmsa_scan_workspace(int dim, int k, const GridDev& g8) {
  return (size_t)g8.rs * g8.rs * crmsa_scan_chunks(g8) * k * (SCAN_HDR + dim) * sizeof(float);
}
hipError_t launch_crmsa_scan(const float* x1, const float* gamma, const float* beta, const float* phi,
                             float* logits, float* rep, float* stats, float* part, int dim, int k,
                             const GridDev& g8, hipStream_t st) {
  const int R = g8.rs * g8.rs, nch = crmsa_scan_chunks(g8), CH = (g8.P + nch - 1) / nch;
  const size_t lds = ((size_t)dim * k + 4 * (SCAN_HDR + dim)) * sizeof(float);
  if (lds > 64 * 1024) return hipErrorInvalidValue;              // dim <= 2048 at k = 8 is 80 KiB: not reached (dim <= 1024 here)
  dim3 grid(R * nch), block(256);
#define RRT_SCAN(NV)                                                                                                   \
  do {                                                                                                                 \
    if (k <= 2) crmsa_scan_kernel<NV, 2><<<grid, block, lds, st>>>(x1, gamma, beta, phi, logits, part, dim, k, nch, CH, g8);      \
    else if (k <= 4) crmsa_scan_kernel<NV, 4><<<grid, block, lds, st>>>(x1, gamma, beta, phi, logits, part, dim, k, nch, CH, g8); \
    else crmsa_scan_kernel<NV, 8><<<grid, block, lds, st>>>(x1, gamma, beta, phi, logits, part, dim, k, nch, CH, g8);             \
  } while (0)
  if (dim <= 256) RRT_SCAN(1);
  else if (dim <= 512) RRT_SCAN(2);
  else if (dim <= 1024) RRT_SCAN(4);
  else return hipErrorInvalidValue;
#undef RRT_SCAN
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  crmsa_merge_kernel<<<dim3(R * k), 128, 0, st>>>(part, rep, stats, dim, k, nch, R);
  return hipGetLastError();
}

hipError_t launch_crmsa_dispatch_fly_ln(const float* x1, const float* x0, const float* logits, const float* stats,
                                        const float* rep2, const float* gamma, const float* beta, float* y, int dim,
                                        int k, const GridDev& g8, hipStream_t st) {
  return launch_dispatch<true, true>(x1, x0, logits, rep2, gamma, beta, y, g8.L, dim, k, g8, st, stats);
}

hipError_t launch_crmsa_mlp_logits(const float* hid, const float* w2, float* logits, int rows, int hdim,
                                   int k, hipStream_t st) {
  crmsa_mlp_logits_kernel<<<dim3((rows + 3) / 4), 256, 0, st>>>(hid, w2, logits, rows, hdim, k);
  return hipGetLastError();
}

hipError_t launch_crmsa_dispatch_ln(const float* x1, const float* x0, const float* wdisp,
                                    const float* rep2, const float* gamma,
                                    const float* beta, float* y, int dim, int k, const GridDev& g8,
                                    hipStream_t st) {
  return launch_dispatch<true>(x1, x0, wdisp, rep2, gamma, beta, y, g8.L, dim, k, g8, st);
}

hipError_t launch_layernorm(const float* x1, const float* x0, const float* gamma,
                            const float* beta, float* y, int L, int dim, hipStream_t st) {
  GridDev g{};
  g.L = L;
  g.H = g.s = g.rs = g.P = 1;
  return launch_dispatch<false>(x1, x0, nullptr, nullptr, gamma, beta, y, L, dim, 0, g, st);
}
