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

// One block per (region, head).  LDS: S [P][P] | conv taps [k][k] | 4 probability rows [P].
__global__ __launch_bounds__(256) void attn_scoremap_kernel(const float* __restrict__ qkv, const float* __restrict__ pe_w,
                                                            float* __restrict__ o, int P, int dim, int heads, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* S = (float*)smem;
  float* W = S + (size_t)P * P;
  float* prow = W + k * k;
  const int head = blockIdx.x, reg = blockIdx.y;
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

}  // namespace

size_t attn_scoremap_lds(int P, int k) { return ((size_t)P * P + (size_t)k * k + 4 * (size_t)P) * sizeof(float); }

hipError_t launch_attn_scoremap(const float* qkv, const float* pe_w, float* o, int n_regions, int P, int dim, int heads,
                                int k, hipStream_t st) {
  const size_t lds = attn_scoremap_lds(P, k);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    static OncePerDevice once;
    if (once.first())
      (void)hipFuncSetAttribute((const void*)attn_scoremap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  attn_scoremap_kernel<<<dim3(heads, n_regions), 256, lds, st>>>(qkv, pe_w, o, P, dim, heads, k);
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
