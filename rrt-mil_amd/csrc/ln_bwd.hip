// ln_bwd.hip -- LayerNorm backward (row f2 building block).
//     y = (x - mean) * rstd * gamma + beta        (nn.LayerNorm, eps 1e-5, biased variance)
//     g = dy * gamma ; dx = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat)) ; dgamma = sum_t dy * xhat ; dbeta = sum_t dy
// One wave per token, grid-stride; mean / rstd are recomputed from x (nothing but x is stashed by the forward).
// The upstream gradient may live in region-major padded order (the output of the qkv-linear backward, [Np, D]):
// then token t reads row token_to_slot(t) -- the adjoint of "zero-pad + region_partition" is this gather, the pad
// rows' gradients are simply never read.  `add` (optional) is the residual branch's gradient, summed into dx.
// dgamma / dbeta: per-block column partials [blocks][2][D], reduced in a fixed order by reduce_partials_kernel.
#include "internal.h"

namespace {

constexpr int LNB_BLOCKS = 512;

template <int NV>   // float4 per lane per row: dim <= NV * 256
__global__ __launch_bounds__(256) void ln_backward_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ add, float* __restrict__ dx,
                                                          float* __restrict__ part, int L, int dim, int mapped,
                                                          GridDev g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* red = (float4*)smem;                       // [3 waves][2][NV * 64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const float inv_d = 1.0f / (float)dim;
  float4 gm[NV], dg[NV], db[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * 64 + lane) * 4;
    gm[v] = c < dim ? *(const float4*)(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[v] = db[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int t = wid; t < L; t += nw) {
    const size_t xrow = (size_t)t * dim;
    const size_t grow = (size_t)(mapped ? token_to_slot(t, g) : t) * dim;
    float4 xv[NV], gv[NV];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      const bool ok = c < dim;
      xv[v] = ok ? *(const float4*)(x + xrow + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      gv[v] = ok ? *(const float4*)(dy + grow + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      sum += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
    }
    const float mean = wave_sum(sum) * inv_d;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < dim) {
        xv[v].x -= mean; xv[v].y -= mean; xv[v].z -= mean; xv[v].w -= mean;
        sq += (xv[v].x * xv[v].x + xv[v].y * xv[v].y) + (xv[v].z * xv[v].z + xv[v].w * xv[v].w);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      // xv <- xhat ; column partials on the raw dy ; gv <- dy * gamma
      xv[v].x *= rstd; xv[v].y *= rstd; xv[v].z *= rstd; xv[v].w *= rstd;
      dg[v].x += gv[v].x * xv[v].x; dg[v].y += gv[v].y * xv[v].y; dg[v].z += gv[v].z * xv[v].z; dg[v].w += gv[v].w * xv[v].w;
      db[v].x += gv[v].x; db[v].y += gv[v].y; db[v].z += gv[v].z; db[v].w += gv[v].w;
      gv[v].x *= gm[v].x; gv[v].y *= gm[v].y; gv[v].z *= gm[v].z; gv[v].w *= gm[v].w;
      s1 += (gv[v].x + gv[v].y) + (gv[v].z + gv[v].w);
      s2 += (gv[v].x * xv[v].x + gv[v].y * xv[v].y) + (gv[v].z * xv[v].z + gv[v].w * xv[v].w);
    }
    const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < dim) {
        float4 o;
        o.x = rstd * (gv[v].x - c1 - xv[v].x * c2);
        o.y = rstd * (gv[v].y - c1 - xv[v].y * c2);
        o.z = rstd * (gv[v].z - c1 - xv[v].z * c2);
        o.w = rstd * (gv[v].w - c1 - xv[v].w * c2);
        if (add) {
          const float4 a = *(const float4*)(add + xrow + c);
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        *(float4*)(dx + xrow + c) = o;
      }
    }
  }
  // block partial of dgamma / dbeta: waves 1..3 -> LDS, wave 0 adds and writes
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      red[((wave - 1) * 2 + 0) * (NV * 64) + v * 64 + lane] = dg[v];
      red[((wave - 1) * 2 + 1) * (NV * 64) + v * 64 + lane] = db[v];
    }
  }
  __syncthreads();
  if (wave == 0) {
    float* out = part + (size_t)blockIdx.x * 2 * dim;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < dim) {
        float4 a = dg[v], b = db[v];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 p = red[(w * 2 + 0) * (NV * 64) + v * 64 + lane];
          const float4 q = red[(w * 2 + 1) * (NV * 64) + v * 64 + lane];
          a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
          b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
        }
        *(float4*)(out + c) = a;
        *(float4*)(out + dim + c) = b;
      }
    }
  }
}

// dst [Np, dim] region-major <- src [L, dim] token order, pad slots = 0 (region_partition of a gradient)
__global__ __launch_bounds__(256) void partition_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int dim, GridDev g, unsigned thresh, unsigned seed,
                                                             float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int slot = blockIdx.x * 4 + wave; slot < g.Np; slot += gridDim.x * 4) {
    const int t = slot_to_token(slot, g);
    for (int c = lane * 4; c < dim; c += 256) {
      float4 v = t < g.L ? *(const float4*)(src + (size_t)t * dim + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (thresh || scale != 1.0f) {       // adjoint of the forward's dropout / branch multiplier on this layer's proj output (same mask)
        const unsigned long long i = (unsigned long long)slot * dim + c;
        v.x = rrt_drop_keep(seed, i, thresh) ? v.x * scale : 0.f;
        v.y = rrt_drop_keep(seed, i + 1, thresh) ? v.y * scale : 0.f;
        v.z = rrt_drop_keep(seed, i + 2, thresh) ? v.z * scale : 0.f;
        v.w = rrt_drop_keep(seed, i + 3, thresh) ? v.w * scale : 0.f;
      }
      *(float4*)(dst + (size_t)slot * dim + c) = v;
    }
  }
}

// FFN activation (TransLayer's Mlp, rrt.py:25-41) as its own pass in training: the pre-activation is stashed
// (GELU's derivative needs it), h = act(hpre) is rebuilt where it is consumed.
// (thresh != 0: the Mlp's first Dropout, rrt.py:38, on the activation's output; same stateless mask as proj_drop)
__global__ __launch_bounds__(256) void act_forward_kernel(const float* __restrict__ hpre, float* __restrict__ h,
                                                          size_t n, int act, unsigned thresh, unsigned seed,
                                                          float scale) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 v = *(const float4*)(hpre + i);
  if (act == RRT_ACT_GELU) {
    v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f));
    v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
    v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f));
    v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
  } else {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  if (thresh) {
    v.x = rrt_drop_keep(seed, i, thresh) ? v.x * scale : 0.f;
    v.y = rrt_drop_keep(seed, i + 1, thresh) ? v.y * scale : 0.f;
    v.z = rrt_drop_keep(seed, i + 2, thresh) ? v.z * scale : 0.f;
    v.w = rrt_drop_keep(seed, i + 3, thresh) ? v.w * scale : 0.f;
  }
  *(float4*)(h + i) = v;
}

__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == RRT_ACT_GELU)      // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
  return x > 0.f ? 1.0f : 0.f;
}

__global__ __launch_bounds__(256) void act_backward_kernel(float* __restrict__ dh, const float* __restrict__ hpre,
                                                           size_t n, int act, unsigned thresh, unsigned seed,
                                                           float scale) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 g = *(const float4*)(dh + i);
  const float4 x = *(const float4*)(hpre + i);
  if (thresh) {
    g.x = rrt_drop_keep(seed, i, thresh) ? g.x * scale : 0.f;
    g.y = rrt_drop_keep(seed, i + 1, thresh) ? g.y * scale : 0.f;
    g.z = rrt_drop_keep(seed, i + 2, thresh) ? g.z * scale : 0.f;
    g.w = rrt_drop_keep(seed, i + 3, thresh) ? g.w * scale : 0.f;
  }
  g.x *= act_grad(x.x, act); g.y *= act_grad(x.y, act); g.z *= act_grad(x.z, act); g.w *= act_grad(x.w, act);
  *(float4*)(dh + i) = g;
}

__global__ __launch_bounds__(256) void apply_drop_mask_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              size_t n, unsigned thresh, unsigned seed, float scale) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = rrt_drop_keep(seed, i, thresh) ? src[i] * scale : 0.f;
}

}  // namespace

hipError_t launch_partition_rows(const float* src, float* dst, int dim, const GridDev& g, unsigned drop_thresh,
                                 unsigned drop_seed, float drop_scale, hipStream_t st) {
  const int need = (g.Np + 3) / 4;
  partition_rows_kernel<<<dim3(need < 4096 ? need : 4096), 256, 0, st>>>(src, dst, dim, g, drop_thresh, drop_seed,
                                                                      drop_scale);
  return hipGetLastError();
}

hipError_t launch_act_forward(const float* hpre, float* h, size_t n, int act, unsigned thresh, unsigned seed,
                              float scale, hipStream_t st) {
  act_forward_kernel<<<dim3((unsigned)((n / 4 + 255) / 256)), 256, 0, st>>>(hpre, h, n, act, thresh, seed, scale);
  return hipGetLastError();
}

hipError_t launch_act_backward(float* dh, const float* hpre, size_t n, int act, unsigned thresh, unsigned seed,
                               float scale, hipStream_t st) {
  act_backward_kernel<<<dim3((unsigned)((n / 4 + 255) / 256)), 256, 0, st>>>(dh, hpre, n, act, thresh, seed, scale);
  return hipGetLastError();
}

hipError_t launch_apply_drop_mask(float* buf, int rows, int cols, unsigned drop_thresh, unsigned drop_seed,
                                  float drop_scale, hipStream_t st) {
  if (!drop_thresh && drop_scale == 1.0f) return hipSuccess;
  const size_t n = (size_t)rows * cols;
  apply_drop_mask_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(buf, buf, n, drop_thresh, drop_seed,
                                                                            drop_scale);
  return hipGetLastError();
}

hipError_t launch_copy_drop_mask(const float* src, float* dst, size_t n, unsigned drop_thresh, unsigned drop_seed,
                                 float drop_scale, hipStream_t st) {
  apply_drop_mask_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(src, dst, n, drop_thresh, drop_seed,
                                                                            drop_scale);
  return hipGetLastError();
}

size_t ln_bwd_workspace(int dim) { return (size_t)LNB_BLOCKS * 2 * dim * sizeof(float); }

// dgb: [2, dim] = dgamma then dbeta.  g == nullptr: dy is in token order.
hipError_t launch_ln_backward(const float* dy, const float* x, const float* gamma, const float* add, float* dx,
                              float* dgb, float* part, int L, int dim, const GridDev* g, hipStream_t st,
                              ReduceJobs* defer) {
  const int need = (L + 3) / 4;
  const int blocks = need < LNB_BLOCKS ? need : LNB_BLOCKS;
  GridDev gd{};
  if (g) gd = *g;
#define RRT_LNB(NV)                                                                                      \
  ln_backward_kernel<NV><<<dim3(blocks), 256, (size_t)3 * 2 * NV * 64 * sizeof(float4), st>>>(          \
      dy, x, gamma, add, dx, part, L, dim, g != nullptr, gd)
  if (dim <= 256) RRT_LNB(1);
  else if (dim <= 512) RRT_LNB(2);
  else if (dim <= 1024) RRT_LNB(4);
  else RRT_LNB(8);
#undef RRT_LNB
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return reduce_or_defer(defer, part, dgb, blocks, (size_t)2 * dim, st);
}
