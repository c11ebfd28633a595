// epeg_variants.hip -- the reference's EPEG ablations (row f4): epeg_2d and epeg_type = 'value_bf' / 'value_af'
// (modules/rmsa.py:76-85 constructors, :106-129 forward).  Correctness-first VALU kernels around the unfused
// R-MSA path (qkv linear -> ... -> proj linear); the default 1-D 'attn' EPEG stays on the fused MFMA kernels.
//
//  * 2-D 'attn' EPEG (rmsa.py:78-79,106-108): pe = Conv2d(h, h, k, padding = k/2, groups = h) over the SCORE MAP,
//    i.e. a k x k stencil over (query, key), zero padded at the region edge: S~ = S + conv2d(S) + b.  It does not
//    factor through Q like the (k, 1) kernel does, so the block of a (region, head) holds S [P, P] in LDS
//    (P <= ~180 tokens), forms one row of S~ at a time, soft-maxes it and multiplies with V.  The conv bias is a
//    per-head constant on every score: it cancels in the softmax (DESIGN.md identity 2) and is not read.
//  * value EPEG (rmsa.py:80-85,114-129): pe = depth-wise Conv2d over the C = h * hd channels of v laid out as a
//    sqrt(P) x sqrt(P) image, kernel (k, 1) or k x k.  The reference builds the image with
//    v.permute(0, 3, 1, 2).reshape(B_, C, s, s): channel c reads v[head = c % h, :, d = c / h]; its output goes to
//    v[head = c / hd, :, d = c % hd] ('value_bf': v += pe before attn @ v) or to x[:, :, c] ('value_af': after it).
//    value_pe_kernel writes pe [rows, C] (bias included); add_cols_kernel adds it into the v columns of the qkv
//    buffer or into the attention output.
#include "internal.h"

namespace {

__global__ __launch_bounds__(256) void value_pe_kernel(const float* __restrict__ qkv, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ pe,
                                                       int P, int s, int dim, int heads, int k, int two_d) {
  const int hd = dim / heads, half = k >> 1;
  const int kw = two_d ? k : 1;
  const int row = blockIdx.x;                       // region-major slot: region = row / P, token n = row % P
  const int reg = row / P, n = row - reg * P;
  const int ni = n / s, nj = n - ni * s;
  const float* vbase = qkv + (size_t)reg * P * 3 * dim + 2 * dim;
  for (int c = threadIdx.x; c < dim; c += 256) {
    const int src_col = (c % heads) * hd + c / heads;          // image channel c = v[head = c % h][d = c / h]
    const float* wc = w + (size_t)c * k * kw;
    float acc = bias ? bias[c] : 0.f;
    for (int a = 0; a < k; ++a) {
      const int ii = ni + a - half;
      if (ii < 0 || ii >= s) continue;
      for (int b = 0; b < kw; ++b) {
        const int jj = two_d ? nj + b - half : nj;
        if (jj < 0 || jj >= s) continue;
        acc += wc[a * kw + b] * vbase[(size_t)(ii * s + jj) * 3 * dim + src_col];
      }
    }
    pe[(size_t)row * dim + c] = acc;
  }
}

// dst[row * ld + c] += src[row * dim + c]
__global__ __launch_bounds__(256) void add_cols_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t rows,
                                                       int dim, int ld) {
  const size_t n4 = rows * (size_t)(dim / 4);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t row = i / (dim / 4);
    const int c = (int)(i - row * (dim / 4)) * 4;
    float4 a = *(float4*)(dst + row * ld + c);
    const float4 b = *(const float4*)(src + row * dim + c);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    *(float4*)(dst + row * ld + c) = a;
  }
}

// One block per (region, head).  LDS: S [P][P] | conv taps [k][k] | 4 probability rows [P]; regions whose score map
// does not fit the CU's LDS (P > ~180) keep S in a caller-provided global scratch (smap: [regions * heads][P][P], L2-
// resident while the block works on it) -- same code, the block only reads what it wrote itself.
__global__ __launch_bounds__(256) void attn_scoremap_kernel(const float* __restrict__ qkv, const float* __restrict__ pe_w,
                                                            float* __restrict__ o, float* __restrict__ smap, int P, int dim,
                                                            int heads, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int head = blockIdx.x, reg = blockIdx.y;
  float* S = smap ? smap + ((size_t)reg * heads + head) * P * P : (float*)smem;
  float* W = smap ? (float*)smem : S + (size_t)P * P;
  float* prow = W + k * k;
  const int hd = dim / heads, half = k >> 1, ld = 3 * dim;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* qb = qkv + (size_t)reg * P * ld + head * hd;      // q already carries head_dim^-0.5
  const float* kb = qb + dim;
  const float* vb = qb + 2 * dim;
  for (int i = tid; i < k * k; i += 256) W[i] = pe_w[(size_t)head * k * k + i];
  // phase A: S[i][j] = <q_i, k_j>; thread = key j, its k row walks past every query (q_i is block-uniform)
  for (int j = tid; j < P; j += 256) {
    const float* kr = kb + (size_t)j * ld;
    for (int i = 0; i < P; ++i) {
      const float* qr = qb + (size_t)i * ld;
      float a = 0.f;
      for (int d = 0; d < hd; ++d) a += qr[d] * kr[d];
      S[(size_t)i * P + j] = a;
    }
  }
  __syncthreads();
  // phase B: wave = query row: S~ row (k x k stencil, zero padded), softmax, P.V
  float* pr = prow + wave * P;
  for (int i = wave; i < P; i += 4) {
    float mx = -3.0e38f;
    for (int j = lane; j < P; j += 64) {
      float a = S[(size_t)i * P + j];
      for (int aa = 0; aa < k; ++aa) {
        const int ii = i + aa - half;
        if (ii < 0 || ii >= P) continue;
        for (int bb = 0; bb < k; ++bb) {
          const int jj = j + bb - half;
          if (jj < 0 || jj >= P) continue;
          a += W[aa * k + bb] * S[(size_t)ii * P + jj];
        }
      }
      pr[j] = a;
      mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < P; j += 64) {
      const float p = __expf(pr[j] - mx);
      pr[j] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < hd; d += 64) {
      float acc = 0.f;
      for (int j = 0; j < P; ++j) acc += pr[j] * vb[(size_t)j * ld + d];
      o[((size_t)reg * P + i) * dim + head * hd + d] = acc * inv;
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Backward of the ablations (round 3: row f4 trains).  Correctness-first VALU kernels, deterministic (fixed summation
// orders, no atomics).
//
// value EPEG.  pe[row, c] = b[c] + sum_{a,b} w[c,a,b] v[row + (a - k/2, b - k/2), src(c)],  src(c) = (c % h) hd + c / h.
// Given dpe [rows, C]:
//   dv[m, src(c)] += sum_{a,b} w[c,a,b] dpe[m - (a - k/2, b - k/2), c]      (value_pe_bwd_kernel; c = (sc % hd) h + sc / hd)
//   dw[c,a,b] = sum_rows dpe[row, c] v[row + shift, src(c)],  db[c] = sum_rows dpe[row, c]     (value_pe_wgrad_kernel)
__global__ __launch_bounds__(256) void value_pe_bwd_kernel(const float* __restrict__ dpe, const float* __restrict__ w,
                                                           float* __restrict__ dqkv, int P, int s, int dim, int heads,
                                                           int k, int two_d) {
  const int hd = dim / heads, half = k >> 1;
  const int kw = two_d ? k : 1;
  const int row = blockIdx.x;
  const int reg = row / P, n = row - reg * P;
  const int ni = n / s, nj = n - ni * s;
  const float* dbase = dpe + (size_t)reg * P * dim;
  for (int sc = threadIdx.x; sc < dim; sc += 256) {
    const int c = (sc % hd) * heads + sc / hd;                 // the image channel that reads v column sc
    const float* wc = w + (size_t)c * k * kw;
    float acc = 0.f;
    for (int a = 0; a < k; ++a) {
      const int ii = ni - (a - half);
      if (ii < 0 || ii >= s) continue;
      for (int b = 0; b < kw; ++b) {
        const int jj = two_d ? nj - (b - half) : nj;
        if (jj < 0 || jj >= s) continue;
        acc += wc[a * kw + b] * dbase[(size_t)(ii * s + jj) * dim + c];
      }
    }
    dqkv[(size_t)row * 3 * dim + 2 * dim + sc] += acc;
  }
}

// one block per image channel c; per region the channel's dpe column and v column go through LDS, thread t owns taps
// t, t + 256, ... (and the bias as one more "tap" against a column of ones) across ALL regions: no cross-thread sum.
// vsub (may be null): subtracted from v -- 'value_bf' stashes v' = v + pe, the conv saw v = v' - pe.
__global__ __launch_bounds__(256) void value_pe_wgrad_kernel(const float* __restrict__ dpe, const float* __restrict__ qkv,
                                                             const float* __restrict__ vsub, float* __restrict__ dw,
                                                             float* __restrict__ db, int n_regions, int P, int s, int dim,
                                                             int heads, int k, int two_d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dcol = (float*)smem;                     // [P]
  float* vcol = dcol + P;                         // [P]
  const int c = blockIdx.x, hd = dim / heads, half = k >> 1, kw = two_d ? k : 1, ntap = k * kw;
  const int sc = (c % heads) * hd + c / heads;
  constexpr int QMAX = 16;                        // taps per thread: 63 x 63 / 256 < 16
  float acc[QMAX];
#pragma unroll
  for (int q = 0; q < QMAX; ++q) acc[q] = 0.f;
  for (int reg = 0; reg < n_regions; ++reg) {
    for (int n = threadIdx.x; n < P; n += 256) {
      const size_t row = (size_t)reg * P + n;
      dcol[n] = dpe[row * dim + c];
      float v = qkv[row * 3 * dim + 2 * dim + sc];
      if (vsub) v -= vsub[row * dim + sc];
      vcol[n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
      const int t = q * 256 + threadIdx.x;
      if (t > ntap) break;
      float a_ = 0.f;
      if (t == ntap) {                            // the bias
        for (int n = 0; n < P; ++n) a_ += dcol[n];
      } else {
        const int a = t / kw, b = t - a * kw;
        const int di = a - half, dj = two_d ? b - half : 0;
        for (int ni = 0; ni < s; ++ni) {
          const int ii = ni + di;
          if (ii < 0 || ii >= s) continue;
          for (int nj = 0; nj < s; ++nj) {
            const int jj = nj + dj;
            if (jj < 0 || jj >= s) continue;
            a_ += dcol[ni * s + nj] * vcol[ii * s + jj];
          }
        }
      }
      acc[q] += a_;
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < QMAX; ++q) {
    const int t = q * 256 + threadIdx.x;
    if (t < ntap) dw[(size_t)c * ntap + t] = acc[q];
    else if (t == ntap && db) db[c] = acc[q];
  }
}

// dst[row, c] = src[row * ld + off + c]   (pull dv' out of the interleaved dqkv rows before they are updated in place)
__global__ __launch_bounds__(256) void copy_cols_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t rows,
                                                        int dim, int ld, int off) {
  const size_t n4 = rows * (size_t)(dim / 4);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t row = i / (dim / 4);
    const int c = (int)(i - row * (dim / 4)) * 4;
    *(float4*)(dst + row * dim + c) = *(const float4*)(src + row * ld + off + c);
  }
}
// dst = a - b (elementwise, n4 float4 groups)
__global__ __launch_bounds__(256) void sub_kernel(float* __restrict__ dst, const float* __restrict__ a, const float* __restrict__ b,
                                                  size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i];
    ((float4*)dst)[i] = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
  }
}

// 2-D 'attn' EPEG backward, one block per (region, head).  Maps (LDS when 3 P^2 floats fit, else global scratch):
//   S = Q K^T (q pre-scaled), A = softmax(S + conv2d(S)), G = dS~ = A o (dA - <dA, A>_row), then dS = G + conv2d^T(G) in A's
//   place.  Outputs: dq (x scale: the gradient w.r.t. the unscaled projection), dk, dv into dqkv; this block's partial of
//   the tap gradients dW[a,b] = sum_{i,j} G[i,j] S[i + a - k/2, j + b - k/2] into dw_part[(region, head)][k*k] (summed over
//   regions, in order, by scoremap_wsum_kernel).  The conv bias has an exactly zero gradient (it cancels in the softmax).
__global__ __launch_bounds__(256) void attn_scoremap_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ pe_w,
                                                                const float* __restrict__ dO, float* __restrict__ dqkv,
                                                                float* __restrict__ dw_part, float* __restrict__ maps_g, int P,
                                                                int dim, int heads, int k, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int head = blockIdx.x, reg = blockIdx.y;
  const size_t PP = (size_t)P * P;
  float* base = maps_g ? maps_g + ((size_t)reg * heads + head) * 3 * PP : (float*)smem;
  float* S = base;
  float* A = base + PP;
  float* G = base + 2 * PP;
  float* W = maps_g ? (float*)smem : base + 3 * PP;
  float* prow = W + k * k;                          // [4][P] scratch rows
  const int hd = dim / heads, half = k >> 1, ld = 3 * dim;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* qb = qkv + (size_t)reg * P * ld + head * hd;
  const float* kb = qb + dim;
  const float* vb = qb + 2 * dim;
  const float* dob = dO + (size_t)reg * P * dim + head * hd;
  float* dqb = dqkv + (size_t)reg * P * ld + head * hd;
  for (int i = tid; i < k * k; i += 256) W[i] = pe_w[(size_t)head * k * k + i];
  for (int j = tid; j < P; j += 256) {
    const float* kr = kb + (size_t)j * ld;
    for (int i = 0; i < P; ++i) {
      const float* qr = qb + (size_t)i * ld;
      float a = 0.f;
      for (int d = 0; d < hd; ++d) a += qr[d] * kr[d];
      S[(size_t)i * P + j] = a;
    }
  }
  __syncthreads();
  // A rows: softmax(S + conv2d(S)); G rows: A o (dA - <dA, A>)
  float* pr = prow + wave * P;
  for (int i = wave; i < P; i += 4) {
    float mx = -3.0e38f;
    for (int j = lane; j < P; j += 64) {
      float a = S[(size_t)i * P + j];
      for (int aa = 0; aa < k; ++aa) {
        const int ii = i + aa - half;
        if (ii < 0 || ii >= P) continue;
        for (int bb = 0; bb < k; ++bb) {
          const int jj = j + bb - half;
          if (jj < 0 || jj >= P) continue;
          a += W[aa * k + bb] * S[(size_t)ii * P + jj];
        }
      }
      pr[j] = a;
      mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < P; j += 64) {
      const float p = __expf(pr[j] - mx);
      pr[j] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float dot = 0.f;
    const float* dor = dob + (size_t)i * dim;
    for (int j = lane; j < P; j += 64) {
      const float a = pr[j] * inv;
      const float* vr = vb + (size_t)j * ld;
      float da = 0.f;
      for (int d = 0; d < hd; ++d) da += dor[d] * vr[d];
      A[(size_t)i * P + j] = a;
      pr[j] = da;
      dot += da * a;
    }
    dot = wave_sum(dot);
    for (int j = lane; j < P; j += 64) G[(size_t)i * P + j] = A[(size_t)i * P + j] * (pr[j] - dot);
  }
  __syncthreads();
  // dV[j, d] = sum_i A[i, j] dO[i, d]
  for (int idx = tid; idx < P * hd; idx += 256) {
    const int j = idx / hd, d = idx - j * hd;
    float a = 0.f;
    for (int i = 0; i < P; ++i) a += A[(size_t)i * P + j] * dob[(size_t)i * dim + d];
    dqb[(size_t)j * ld + 2 * dim + d] = a;
  }
  // tap gradients of this (region, head)
  float* dwp = dw_part + ((size_t)reg * heads + head) * k * k;
  for (int t = tid; t < k * k; t += 256) {
    const int di = t / k - half, dj = t % k - half;
    float a = 0.f;
    const int i0 = di < 0 ? -di : 0, i1 = di > 0 ? P - di : P;
    const int j0 = dj < 0 ? -dj : 0, j1 = dj > 0 ? P - dj : P;
    for (int i = i0; i < i1; ++i)
      for (int j = j0; j < j1; ++j) a += G[(size_t)i * P + j] * S[(size_t)(i + di) * P + j + dj];
    dwp[t] = a;
  }
  __syncthreads();                                  // every read of A is done: it becomes dS
  for (int idx = tid; idx < P * P; idx += 256) {
    const int i = idx / P, j = idx - i * P;
    float a = G[idx];
    for (int aa = 0; aa < k; ++aa) {
      const int ii = i - (aa - half);
      if (ii < 0 || ii >= P) continue;
      for (int bb = 0; bb < k; ++bb) {
        const int jj = j - (bb - half);
        if (jj < 0 || jj >= P) continue;
        a += W[aa * k + bb] * G[(size_t)ii * P + jj];
      }
    }
    A[idx] = a;
  }
  __syncthreads();
  // dq[i, d] = scale sum_j dS[i, j] k[j, d];  dk[j, d] = sum_i dS[i, j] q[i, d]  (q is the pre-scaled one)
  for (int idx = tid; idx < P * hd; idx += 256) {
    const int r = idx / hd, d = idx - r * hd;
    float aq = 0.f, ak = 0.f;
    for (int t = 0; t < P; ++t) {
      aq += A[(size_t)r * P + t] * kb[(size_t)t * ld + d];
      ak += A[(size_t)t * P + r] * qb[(size_t)t * ld + d];
    }
    dqb[(size_t)r * ld + d] = aq * scale;
    dqb[(size_t)r * ld + dim + d] = ak;
  }
}

// C [M, N] = A [M, Kd] . W [Kd, N] for a small inner dimension that is not a GEMM K tile (crmsa_mlp's hidden width
// dim / 4 when dim % 128 != 0): thread = output column, the row's Kd values are block-uniform
__global__ __launch_bounds__(256) void small_k_matmul_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                             float* __restrict__ Cm, int M, int N, int Kd) {
  const int row = blockIdx.x;
  const float* a = A + (size_t)row * Kd;
  for (int c = threadIdx.x; c < N; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < Kd; ++j) acc += a[j] * W[(size_t)j * N + c];
    Cm[(size_t)row * N + c] = acc;
  }
}

// dw[head, t] = sum over regions (in order) of the per-(region, head) partials
__global__ __launch_bounds__(256) void scoremap_wsum_kernel(const float* __restrict__ part, float* __restrict__ dw, int n_regions,
                                                            int heads, int kk) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= heads * kk) return;
  const int head = idx / kk, t = idx - head * kk;
  float a = 0.f;
  for (int r = 0; r < n_regions; ++r) a += part[((size_t)r * heads + head) * kk + t];
  dw[idx] = a;
}

}  // namespace

size_t attn_scoremap_lds(int P, int k) { return ((size_t)P * P + (size_t)k * k + 4 * (size_t)P) * sizeof(float); }
// floats of global scratch the 2-D score-map kernels need when a region's map does not fit the LDS (0: it fits)
size_t attn_scoremap_scratch_floats(int n_regions, int P, int heads, int k, int maps) {
  const size_t lds = ((size_t)maps * P * P + (size_t)k * k + 4 * (size_t)P) * sizeof(float);
  return lds <= 160 * 1024 ? 0 : (size_t)maps * n_regions * heads * P * P;
}

hipError_t launch_attn_scoremap(const float* qkv, const float* pe_w, float* o, float* smap, int n_regions, int P, int dim,
                                int heads, int k, hipStream_t st) {
  size_t lds = attn_scoremap_lds(P, k);
  if (lds > 160 * 1024) {
    if (!smap) return hipErrorInvalidValue;
    lds = ((size_t)k * k + 4 * (size_t)P) * sizeof(float);
  } else {
    smap = nullptr;
  }
  if (lds > 64 * 1024) {
    static OncePerDevice once;
    if (once.first())
      (void)hipFuncSetAttribute((const void*)attn_scoremap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  attn_scoremap_kernel<<<dim3(heads, n_regions), 256, lds, st>>>(qkv, pe_w, o, smap, P, dim, heads, k);
  return hipGetLastError();
}

hipError_t launch_value_pe(const float* qkv, const float* w, const float* bias, float* pe, int n_regions, int P, int s,
                           int dim, int heads, int k, int two_d, hipStream_t st) {
  value_pe_kernel<<<dim3(n_regions * P), 256, 0, st>>>(qkv, w, bias, pe, P, s, dim, heads, k, two_d);
  return hipGetLastError();
}

hipError_t launch_add_cols(float* dst, const float* src, size_t rows, int dim, int ld, hipStream_t st) {
  const size_t n4 = rows * (size_t)(dim / 4);
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  add_cols_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(dst, src, rows, dim, ld);
  return hipGetLastError();
}

hipError_t launch_value_pe_backward(const float* dpe, const float* qkv, const float* vsub, const float* w, float* dqkv, float* dw,
                                    float* db, int n_regions, int P, int s, int dim, int heads, int k, int two_d,
                                    hipStream_t st) {
  if ((size_t)2 * P * sizeof(float) > 160 * 1024 || (size_t)k * (two_d ? k : 1) + 1 > 16 * 256) return hipErrorInvalidValue;
  const size_t lds = (size_t)2 * P * sizeof(float);
  if (lds > 64 * 1024) {
    static OncePerDevice once;
    if (once.first())
      (void)hipFuncSetAttribute((const void*)value_pe_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  value_pe_wgrad_kernel<<<dim3(dim), 256, lds, st>>>(dpe, qkv, vsub, dw, db, n_regions, P, s, dim, heads, k, two_d);
  value_pe_bwd_kernel<<<dim3(n_regions * P), 256, 0, st>>>(dpe, w, dqkv, P, s, dim, heads, k, two_d);
  return hipGetLastError();
}

hipError_t launch_copy_cols(float* dst, const float* src, size_t rows, int dim, int ld, int off, hipStream_t st) {
  const size_t n4 = rows * (size_t)(dim / 4);
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  copy_cols_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(dst, src, rows, dim, ld, off);
  return hipGetLastError();
}

hipError_t launch_sub(float* dst, const float* a, const float* b, size_t n, hipStream_t st) {
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  sub_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(dst, a, b, n4);
  return hipGetLastError();
}

// scratch: attn_scoremap_scratch_floats(R, P, heads, k, 3) floats of maps (0 when they fit the LDS) + R * heads * k * k of
// tap partials behind them
hipError_t launch_attn_scoremap_backward(const float* qkv, const float* pe_w, const float* dO, float* dqkv, float* dpe_w,
                                         float* scratch, int n_regions, int P, int dim, int heads, int k, hipStream_t st) {
  const size_t maps = attn_scoremap_scratch_floats(n_regions, P, heads, k, 3);
  float* part = scratch + maps;
  size_t lds = ((size_t)k * k + 4 * (size_t)P) * sizeof(float);
  if (!maps) lds += (size_t)3 * P * P * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    static OncePerDevice once;
    if (once.first())
      (void)hipFuncSetAttribute((const void*)attn_scoremap_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const float scale = 1.0f / sqrtf((float)(dim / heads));
  attn_scoremap_bwd_kernel<<<dim3(heads, n_regions), 256, lds, st>>>(qkv, pe_w, dO, dqkv, part, maps ? scratch : nullptr, P, dim,
                                                                     heads, k, scale);
  scoremap_wsum_kernel<<<dim3((heads * k * k + 255) / 256), 256, 0, st>>>(part, dpe_w, n_regions, heads, k * k);
  return hipGetLastError();
}

hipError_t launch_small_k_matmul(const float* A, const float* W, float* Cm, int M, int N, int Kd, hipStream_t st) {
  small_k_matmul_kernel<<<dim3(M), 256, 0, st>>>(A, W, Cm, M, N, Kd);
  return hipGetLastError();
}
