// mil_pool.hip -- ABMIL attention pooling + predictor behind the encoder.
//
// Replaces modules/datten.py:28-38 (Attention.forward) / :69-83 (AttentionGated.forward) after their
// first Linear(+activation), and modules/rrt.py:241 (predictor):
//     a_n    = wc . h_n (+ bc)            h_n = hid_a[n]  or  hid_a[n] * hid_b[n] (gated)
//     A      = softmax_n(a)               over the N tokens of the bag
//     pooled = sum_n A_n y_n              [dim]
//     logits = pred_w pooled + pred_b     [n_classes]
// The softmax over N is an online softmax over POOL_CHUNK-token chunks: every block emits
// (m_b, l_b, v_b = sum exp(a_n - m_b) y_n), one merge block rescales and adds them.  HBM-bound:
// reads y (N*dim) and the hidden rows (N*hid) once; the 18 MB of y never has to be re-read by a
// separate softmax / matmul pair, and nothing of size N x dim is written.
#include "internal.h"

namespace {

// part layout per chunk: [dim] weighted sum, then m, l (padded to dim + 4 floats)
__device__ __forceinline__ size_t part_stride(int dim) { return (size_t)dim + 4; }

__global__ __launch_bounds__(256) void pool_partial_kernel(const float* __restrict__ y,
                                                           const float* __restrict__ hid_a,
                                                           const float* __restrict__ hid_b,
                                                           const float* __restrict__ wc,
                                                           const float* __restrict__ bc,
                                                           float* __restrict__ a_raw,
                                                           float* __restrict__ part, int N, int dim, int hid) {
  __shared__ float s_a[POOL_CHUNK];
  __shared__ float s_e[POOL_CHUNK];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* red = (float4*)smem;                       // [dim/4] second row group's partial
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * POOL_CHUNK;
  const int cnt = min(POOL_CHUNK, N - n0);

  // ---- scores: one wave per token, POOL_CHUNK / 4 tokens per wave
  const float b = bc ? bc[0] : 0.f;
  for (int t = wave; t < cnt; t += 4) {
    const size_t row = (size_t)(n0 + t) * hid;
    float acc = 0.f;
    for (int c = lane * 4; c < hid; c += 256) {
      float4 h = *(const float4*)(hid_a + row + c);
      if (hid_b) {
        const float4 g = *(const float4*)(hid_b + row + c);
        h.x *= g.x; h.y *= g.y; h.z *= g.z; h.w *= g.w;
      }
      const float4 w = *(const float4*)(wc + c);
      acc += (h.x * w.x + h.y * w.y) + (h.z * w.z + h.w * w.w);
    }
    acc = wave_sum(acc) + b;
    if (lane == 0) {
      s_a[t] = acc;
      a_raw[n0 + t] = acc;
    }
  }
  __syncthreads();
  // ---- chunk max / exp / sum (every thread redundantly: cnt <= 32 LDS reads)
  float m = -3.0e38f;
  for (int t = 0; t < cnt; ++t) m = fmaxf(m, s_a[t]);
  if (tid < cnt) s_e[tid] = __expf(s_a[tid] - m);
  __syncthreads();
  float l = 0.f;
  for (int t = 0; t < cnt; ++t) l += s_e[t];

  // ---- weighted sum of the chunk's rows: thread = (row group rg of 2, float4 column lane of 128)
  const int cl = tid & 127, rg = tid >> 7;
  float* out = part + (size_t)blockIdx.x * part_stride(dim);
  for (int cb = 0; cb < dim; cb += 512) {
    const int c = cb + cl * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < dim) {
      for (int t0 = rg; t0 < cnt; t0 += 8) {          // 4 independent rows in flight
        float4 v[4];
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = t0 + 2 * u;
          const bool ok = t < cnt;
          e[u] = ok ? s_e[t] : 0.f;
          v[u] = ok ? *(const float4*)(y + (size_t)(n0 + t) * dim + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc.x += e[u] * v[u].x; acc.y += e[u] * v[u].y; acc.z += e[u] * v[u].z; acc.w += e[u] * v[u].w;
        }
      }
    }
    if (rg == 1 && c < dim) red[cl] = acc;
    __syncthreads();
    if (rg == 0 && c < dim) {
      const float4 o = red[cl];
      *(float4*)(out + c) = make_float4(acc.x + o.x, acc.y + o.y, acc.z + o.z, acc.w + o.w);
    }
    __syncthreads();
  }
  if (tid == 0) {
    out[dim] = m;
    out[dim + 1] = l;
  }
}

// One block: global max / normaliser over the chunk partials, pooled vector, predictor, and the
// attention row the reference returns (normalised, or raw scores for no_norm=True).
__global__ __launch_bounds__(1024) void pool_merge_kernel(const float* __restrict__ part,
                                                          const float* __restrict__ a_raw,
                                                          const float* __restrict__ pred_w,
                                                          const float* __restrict__ pred_b,
                                                          float* __restrict__ pooled,
                                                          float* __restrict__ logits, float* __restrict__ attn,
                                                          int no_norm, int N, int dim, int n_classes, int nb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_scale = (float*)smem;                    // [nb] exp(m_b - M)
  float* s_pool = s_scale + ((nb + 3) & ~3);        // [dim]
  float4* s_red = (float4*)(s_pool + dim);          // [8 groups][128 column lanes]
  __shared__ float s_w[16];
  __shared__ float s_M, s_invL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t ps = part_stride(dim);

  float m = -3.0e38f;
  for (int b = tid; b < nb; b += 1024) m = fmaxf(m, part[b * ps + dim]);
  m = wave_max(m);
  if (lane == 0) s_w[wave] = m;
  __syncthreads();
  if (tid == 0) {
    float M = s_w[0];
    for (int w = 1; w < 16; ++w) M = fmaxf(M, s_w[w]);
    s_M = M;
  }
  __syncthreads();
  const float M = s_M;
  float l = 0.f;
  for (int b = tid; b < nb; b += 1024) {
    const float sc = __expf(part[b * ps + dim] - M);
    s_scale[b] = sc;
    l += part[b * ps + dim + 1] * sc;
  }
  l = wave_sum(l);
  __syncthreads();                                  // s_w reuse
  if (lane == 0) s_w[wave] = l;
  __syncthreads();
  if (tid == 0) {
    float L = 0.f;
    for (int w = 0; w < 16; ++w) L += s_w[w];
    s_invL = 1.0f / L;
  }
  __syncthreads();
  const float invL = s_invL;

  // pooled[c] = invL * sum_b scale_b part[b][c] : thread = (chunk group of 8, float4 column lane of 128)
  const int cl = tid & 127, grp = tid >> 7;
  for (int cb = 0; cb < dim; cb += 512) {
    const int c = cb + cl * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < dim) {
      for (int b0 = grp; b0 < nb; b0 += 32) {
        float4 v[4];
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int b = b0 + 8 * u;
          const bool ok = b < nb;
          sc[u] = ok ? s_scale[b] : 0.f;
          v[u] = ok ? *(const float4*)(part + b * ps + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc.x += sc[u] * v[u].x; acc.y += sc[u] * v[u].y; acc.z += sc[u] * v[u].z; acc.w += sc[u] * v[u].w;
        }
      }
    }
    s_red[grp * 128 + cl] = acc;
    __syncthreads();
    if (grp == 0 && c < dim) {
      float4 a = s_red[cl];
#pragma unroll
      for (int q = 1; q < 8; ++q) {
        const float4 o = s_red[q * 128 + cl];
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
      a.x *= invL; a.y *= invL; a.z *= invL; a.w *= invL;
      *(float4*)(s_pool + c) = a;
      if (pooled) *(float4*)(pooled + c) = a;
    }
    __syncthreads();
  }

  // predictor: one wave per class
  for (int j = wave; j < n_classes; j += 16) {
    float acc = 0.f;
    for (int c = lane * 4; c < dim; c += 256) {
      const float4 w = *(const float4*)(pred_w + (size_t)j * dim + c);
      const float4 p = *(const float4*)(s_pool + c);
      acc += (w.x * p.x + w.y * p.y) + (w.z * p.z + w.w * p.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) logits[j] = acc + (pred_b ? pred_b[j] : 0.f);
  }
  if (attn) {
    for (int n = tid; n < N; n += 1024) {
      const float a = a_raw[n];
      attn[n] = no_norm ? a : __expf(a - M) * invL;
    }
  }
}

// ---- backward of the pooling (training of the slide classifier; adjoint of datten.py:28-38 / :69-83 after the first Linear):
//     s_n = wc . h_n + bc ,  A = softmax_n(s) ,  pooled = sum_n A_n y_n         (h = hid_a, or hid_a * hid_b when gated)
// given d pooled [dim] (and optionally d A [N], d s [N] for callers that use the returned attention / raw scores):
//     dA_n = y_n . dpooled (+ dA_ext_n)      ds_n = A_n (dA_n - c) (+ ds_ext_n) ,  c = sum_m A_m dA_m = pooled . dpooled (+ c_ext)
//     dy_n = A_n dpooled      dh_n = ds_n wc      d wc = sum_n ds_n h_n      d bc = sum_n ds_n
// One wave per token, POOL_BWD_ROWS tokens per block; per-block partials of (d wc | d bc) -> launch_reduce_partials.
constexpr int POOL_BWD_ROWS = 32;
__global__ __launch_bounds__(256) void pool_backward_kernel(const float* __restrict__ y, const float* __restrict__ hid_a,
                                                            const float* __restrict__ hid_b, const float* __restrict__ wc,
                                                            const float* __restrict__ attn, const float* __restrict__ pooled,
                                                            const float* __restrict__ d_pooled, const float* __restrict__ d_attn,
                                                            const float* __restrict__ d_raw, const float* __restrict__ c_ext,
                                                            float* __restrict__ dy, float* __restrict__ dhid_a,
                                                            float* __restrict__ dhid_b, float* __restrict__ part, int N, int dim,
                                                            int hid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_dp = (float*)smem;                        // d pooled [dim]
  float* s_acc = s_dp + dim;                         // [4 waves][hid + 4]: this block's d wc | d bc partial
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float cpart = 0.f;
  for (int c = tid; c < dim; c += 256) {
    const float v = d_pooled[c];
    s_dp[c] = v;
    cpart += v * pooled[c];
  }
  for (int c = tid; c < 4 * (hid + 4); c += 256) s_acc[c] = 0.f;
  __shared__ float s_c[4];
  cpart = wave_sum(cpart);
  if (lane == 0) s_c[wave] = cpart;
  __syncthreads();
  const float cc = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]) + (c_ext ? c_ext[0] : 0.f);
  float* myacc = s_acc + wave * (hid + 4);
  const int n0 = blockIdx.x * POOL_BWD_ROWS;
  float dbc = 0.f;
  for (int t = wave; t < POOL_BWD_ROWS; t += 4) {
    const int n = n0 + t;
    if (n >= N) break;
    const float a = attn[n];
    const float* yr = y + (size_t)n * dim;
    float da = 0.f;
    for (int c = lane * 4; c < dim; c += 256) {
      const float4 v = *(const float4*)(yr + c);
      const float4 d = *(const float4*)(s_dp + c);
      da += (v.x * d.x + v.y * d.y) + (v.z * d.z + v.w * d.w);
      *(float4*)(dy + (size_t)n * dim + c) = make_float4(a * d.x, a * d.y, a * d.z, a * d.w);
    }
    da = wave_sum(da) + (d_attn ? d_attn[n] : 0.f);
    const float ds = a * (da - cc) + (d_raw ? d_raw[n] : 0.f);
    dbc += ds;
    for (int c = lane * 4; c < hid; c += 256) {
      const float4 w = *(const float4*)(wc + c);
      float4 ha = *(const float4*)(hid_a + (size_t)n * hid + c);
      float4 g = make_float4(ds * w.x, ds * w.y, ds * w.z, ds * w.w);          // d (h_a h_b) (or d h_a)
      float4 h = ha;
      if (hid_b) {
        const float4 hb = *(const float4*)(hid_b + (size_t)n * hid + c);
        *(float4*)(dhid_b + (size_t)n * hid + c) = make_float4(g.x * ha.x, g.y * ha.y, g.z * ha.z, g.w * ha.w);
        h = make_float4(ha.x * hb.x, ha.y * hb.y, ha.z * hb.z, ha.w * hb.w);
        g = make_float4(g.x * hb.x, g.y * hb.y, g.z * hb.z, g.w * hb.w);
      }
      *(float4*)(dhid_a + (size_t)n * hid + c) = g;
      float4 acc = *(float4*)(myacc + c);
      acc.x += ds * h.x; acc.y += ds * h.y; acc.z += ds * h.z; acc.w += ds * h.w;
      *(float4*)(myacc + c) = acc;
    }
  }
  if (lane == 0) myacc[hid] = dbc;
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * (hid + 4);
  for (int c = tid; c < hid + 1; c += 256)
    out[c] = (s_acc[c] + s_acc[(hid + 4) + c]) + (s_acc[2 * (hid + 4) + c] + s_acc[3 * (hid + 4) + c]);
}

}  // namespace

// d wc [hid] and d bc [1] come out as dwcb [hid + 4] (d bc at [hid]); part: ceil(N / 32) * (hid + 4) floats
hipError_t launch_pool_backward(const float* y, const float* hid_a, const float* hid_b, const float* wc, const float* attn,
                                const float* pooled, const float* d_pooled, const float* d_attn, const float* d_raw,
                                const float* c_ext, float* dy, float* dhid_a, float* dhid_b, float* dwcb, float* part, int N,
                                int dim, int hid, hipStream_t st) {
  const int nb = (N + POOL_BWD_ROWS - 1) / POOL_BWD_ROWS;
  const size_t lds = ((size_t)dim + 4 * (hid + 4)) * sizeof(float);
  pool_backward_kernel<<<dim3(nb), dim3(256), lds, st>>>(y, hid_a, hid_b, wc, attn, pooled, d_pooled, d_attn, d_raw, c_ext, dy,
                                                         dhid_a, dhid_b, part, N, dim, hid);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_reduce_partials(part, dwcb, nb, (size_t)hid + 4, st);
}
size_t pool_backward_part_floats(int N, int hid) { return (size_t)((N + POOL_BWD_ROWS - 1) / POOL_BWD_ROWS) * (hid + 4); }

hipError_t launch_pool_partial(const float* y, const float* hid_a, const float* hid_b, const float* wc,
                               const float* bc, float* a_raw, float* part, int N, int dim, int hid,
                               hipStream_t st) {
  const int nb = (N + POOL_CHUNK - 1) / POOL_CHUNK;
  const size_t lds = (size_t)128 * sizeof(float4);
  pool_partial_kernel<<<dim3(nb), dim3(256), lds, st>>>(y, hid_a, hid_b, wc, bc, a_raw, part, N, dim, hid);
  return hipGetLastError();
}

hipError_t launch_pool_merge(const float* part, const float* a_raw, const float* pred_w, const float* pred_b,
                             float* pooled, float* logits, float* attn, int no_norm, int N, int dim,
                             int n_classes, hipStream_t st) {
  const int nb = (N + POOL_CHUNK - 1) / POOL_CHUNK;
  const size_t lds = ((size_t)((nb + 3) & ~3) + dim) * sizeof(float) + (size_t)8 * 128 * sizeof(float4);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  auto kern = pool_merge_kernel;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<dim3(1), dim3(1024), lds, st>>>(part, a_raw, pred_w, pred_b, pooled, logits, attn, no_norm, N, dim,
                                         n_classes, nb);
  return hipGetLastError();
}
