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
//   crmsa_combine_kernel   : 1 block / (region, 128-column slab).  Region softmax/min/max
//                            statistics from Lg, then the weighted row sum over P tokens.
//   crmsa_dispatch_ln_kernel: 1 wave / token.  k-term axpy + residual (+shortcut) + LayerNorm.
#include "internal.h"

namespace {

constexpr int KMAX = RRT_MAX_CRMSA_K;

template <int NV>
__global__ __launch_bounds__(256) void crmsa_logits_kernel(const float* __restrict__ x1,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ phi,
                                                           float* __restrict__ mean_rstd,
                                                           float* __restrict__ logits, int dim, int k,
                                                           GridDev g) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= g.Np) return;
  float* lg = logits + (size_t)token_to_slot(t, g) * k;
  if (t >= g.L) {   // zero pad token -> zero logits
    if (lane < k) lg[lane] = 0.f;
    return;
  }
  const float* src = x1 + (size_t)t * dim;
  float4 r[NV];
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    r[v] = (c < dim) ? *(const float4*)(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    sum += (r[v].x + r[v].y) + (r[v].z + r[v].w);
  }
  const float inv_d = 1.0f / (float)dim;
  const float mean = wave_sum(sum) * inv_d;
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (c < dim) {
      float a = r[v].x - mean, b = r[v].y - mean, cc = r[v].z - mean, d = r[v].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
  float acc[KMAX];
#pragma unroll
  for (int n = 0; n < KMAX; ++n) acc[n] = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (c < dim) {
      float4 gm = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
      float u[4] = {(r[v].x - mean) * rstd * gm.x + bt.x, (r[v].y - mean) * rstd * gm.y + bt.y,
                    (r[v].z - mean) * rstd * gm.z + bt.z, (r[v].w - mean) * rstd * gm.w + bt.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* ph = phi + (size_t)(c + e) * k;
#pragma unroll
        for (int n = 0; n < KMAX; ++n)
          if (n < k) acc[n] += u[e] * ph[n];
      }
    }
  }
#pragma unroll
  for (int n = 0; n < KMAX; ++n)
    if (n < k) acc[n] = wave_sum(acc[n]);
  if (lane == 0) {
    mean_rstd[2 * (size_t)t] = mean;
    mean_rstd[2 * (size_t)t + 1] = rstd;
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) lg[n] = acc[n];
  }
}

// stats layout per (region, n): {max, min, 1/sum_p exp(Lg - max)}
__global__ __launch_bounds__(256) void crmsa_combine_kernel(const float* __restrict__ x1,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean_rstd,
                                                            const float* __restrict__ logits,
                                                            float* __restrict__ stats,
                                                            float* __restrict__ rep, int dim, int k,
                                                            GridDev g) {
  __shared__ float s_stat[KMAX][3];
  __shared__ float4 s_part[8][KMAX][32];   // [row-group][n][column lane]
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
    if (lane == 0) {
      s_stat[n][0] = mx; s_stat[n][1] = mn; s_stat[n][2] = 1.0f / se;
      if (slab == 0) {
        float* st = stats + ((size_t)reg * k + n) * 3;
        st[0] = mx; st[1] = mn; st[2] = 1.0f / se;
      }
    }
  }
  __syncthreads();

  // weighted row sum: thread = (row group rg of 8, column lane cl of 32) on a 128-column slab
  const int cl = tid & 31, rg = tid >> 5;
  const int col = slab * 128 + cl * 4;
  float4 acc[KMAX];
#pragma unroll
  for (int n = 0; n < KMAX; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < dim) {
    const float4 gm = *(const float4*)(gamma + col), bt = *(const float4*)(beta + col);
    const int ri = reg / g.rs, rj = reg - ri * g.rs;
    for (int p = rg; p < g.P; p += 8) {
      int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
      int t = (ri * g.s + pi) * g.H + rj * g.s + pj;
      if (t >= g.L) continue;                       // pad token: v = 0 contributes nothing
      const float mean = mean_rstd[2 * (size_t)t], rstd = mean_rstd[2 * (size_t)t + 1];
      float4 xv = *(const float4*)(x1 + (size_t)t * dim + col);
      float4 v;
      v.x = (xv.x - mean) * rstd * gm.x + bt.x;
      v.y = (xv.y - mean) * rstd * gm.y + bt.y;
      v.z = (xv.z - mean) * rstd * gm.z + bt.z;
      v.w = (xv.w - mean) * rstd * gm.w + bt.w;
#pragma unroll
      for (int n = 0; n < KMAX; ++n)
        if (n < k) {
          float c = __expf(lg[(size_t)p * k + n] - s_stat[n][0]) * s_stat[n][2];
          acc[n].x += c * v.x; acc[n].y += c * v.y; acc[n].z += c * v.z; acc[n].w += c * v.w;
        }
    }
  }
#pragma unroll
  for (int n = 0; n < KMAX; ++n)
    if (n < k) s_part[rg][n][cl] = acc[n];
  __syncthreads();
  // reduce the 8 row groups: thread (n, cl) for n < k
  for (int idx = tid; idx < k * 32; idx += 256) {
    int n = idx >> 5, c = idx & 31;
    int cc = slab * 128 + c * 4;
    if (cc >= dim) continue;
    float4 a = s_part[0][n][c];
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      float4 b = s_part[q][n][c];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *(float4*)(rep + ((size_t)n * R + reg) * dim + cc) = a;   // rep [k, R, D]
  }
}

template <int NV, bool CRMSA>
__global__ __launch_bounds__(256) void crmsa_dispatch_ln_kernel(
    const float* __restrict__ x1, const float* __restrict__ x0, const float* __restrict__ logits,
    const float* __restrict__ stats, const float* __restrict__ rep2, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, int L, int dim, int k, GridDev g) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= L) return;
  float wgt[KMAX];
  int reg = 0;
  if (CRMSA) {
    const int slot = token_to_slot(t, g);
    reg = fdiv(slot, g.P, g.inv_P);
    const float* lg = logits + (size_t)slot * k;
    float mx = -3.0e38f;
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) { wgt[n] = lg[n]; mx = fmaxf(mx, wgt[n]); }
    float se = 0.f;
    float e[KMAX];
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) { e[n] = __expf(wgt[n] - mx); se += e[n]; }
    const float inv = 1.0f / se;
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) {
        const float* st = stats + ((size_t)reg * k + n) * 3;
        float mm = (wgt[n] - st[1]) / (st[0] - st[1] + 1e-8f);
        wgt[n] = mm * (e[n] * inv);
      }
  }
  const int R = g.rs * g.rs;
  float4 r[NV];
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (c < dim) {
      float4 a = *(const float4*)(x1 + (size_t)t * dim + c);
      if (CRMSA) {
#pragma unroll
        for (int n = 0; n < KMAX; ++n)
          if (n < k) {
            float4 rp = *(const float4*)(rep2 + ((size_t)n * R + reg) * dim + c);
            a.x += wgt[n] * rp.x; a.y += wgt[n] * rp.y; a.z += wgt[n] * rp.z; a.w += wgt[n] * rp.w;
          }
      }
      if (x0) {
        float4 s = *(const float4*)(x0 + (size_t)t * dim + c);
        a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w;
      }
      r[v] = a;
    } else {
      r[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    sum += (r[v].x + r[v].y) + (r[v].z + r[v].w);
  }
  const float inv_d = 1.0f / (float)dim;
  const float mean = wave_sum(sum) * inv_d;
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (c < dim) {
      float a = r[v].x - mean, b = r[v].y - mean, cc = r[v].z - mean, d = r[v].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (c < dim) {
      float4 gm = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
      float4 o;
      o.x = (r[v].x - mean) * rstd * gm.x + bt.x;
      o.y = (r[v].y - mean) * rstd * gm.y + bt.y;
      o.z = (r[v].z - mean) * rstd * gm.z + bt.z;
      o.w = (r[v].w - mean) * rstd * gm.w + bt.w;
      *(float4*)(y + (size_t)t * dim + c) = o;
    }
  }
}

template <bool CRMSA>
hipError_t launch_dispatch(const float* x1, const float* x0, const float* logits, const float* stats,
                           const float* rep2, const float* gamma, const float* beta, float* y, int L,
                           int dim, int k, const GridDev& g, hipStream_t st) {
  dim3 grid((L + 3) / 4), block(256);
#define RRT_DISPATCH(NV)                                                                          \
  crmsa_dispatch_ln_kernel<NV, CRMSA><<<grid, block, 0, st>>>(x1, x0, logits, stats, rep2, gamma, \
                                                              beta, y, L, dim, k, g)
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
  dim3 grid((g8.Np + 3) / 4), block(256);
#define RRT_LOGITS(NV) \
  crmsa_logits_kernel<NV><<<grid, block, 0, st>>>(x1, gamma, beta, phi, mean_rstd, logits, dim, k, g8)
  if (dim <= 256) RRT_LOGITS(1);
  else if (dim <= 512) RRT_LOGITS(2);
  else if (dim <= 1024) RRT_LOGITS(4);
  else RRT_LOGITS(8);
#undef RRT_LOGITS
  return hipGetLastError();
}

hipError_t launch_crmsa_combine(const float* x1, const float* gamma, const float* beta,
                                const float* mean_rstd, const float* logits, float* stats,
                                float* rep, int dim, int k, const GridDev& g8, hipStream_t st) {
  dim3 grid(g8.rs * g8.rs, (dim + 127) / 128), block(256);
  crmsa_combine_kernel<<<grid, block, 0, st>>>(x1, gamma, beta, mean_rstd, logits, stats, rep, dim,
                                               k, g8);
  return hipGetLastError();
}

hipError_t launch_crmsa_dispatch_ln(const float* x1, const float* x0, const float* logits,
                                    const float* stats, const float* rep2, const float* gamma,
                                    const float* beta, float* y, int dim, int k, const GridDev& g8,
                                    hipStream_t st) {
  return launch_dispatch<true>(x1, x0, logits, stats, rep2, gamma, beta, y, g8.L, dim, k, g8, st);
}

hipError_t launch_layernorm(const float* x1, const float* x0, const float* gamma,
                            const float* beta, float* y, int L, int dim, hipStream_t st) {
  GridDev g{};
  g.L = L;
  g.H = g.s = g.rs = g.P = 1;
  return launch_dispatch<false>(x1, x0, nullptr, nullptr, nullptr, gamma, beta, y, L, dim, 0, g, st);
}
