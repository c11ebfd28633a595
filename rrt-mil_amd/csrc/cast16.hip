// cast16.hip -- producers of the 16-bit GEMM operands of the reduced-precision modes (bf16 / fp16).
//
// In RRT_COMPUTE_BF16 / F16 every tensor that is consumed ONLY as a matrix-core operand lives in HBM in
// 16 bits: half the bytes over HBM / L2 / LDS-DMA and no conversion inside the GEMM loops (the round-1
// kernels moved fp32 and rounded after LDS -- the same values, twice the traffic and twice the DMA issue).
//   * cast16_kernel       : the nn.Linear weights of the R-MSA layers (qkv.weight, proj.weight), once per call;
//   * ln_partition16_kernel: LayerNorm + zero-pad + region_partition (modules/rrt.py:121-123,
//                            modules/rmsa.py:199-200,28-39) with the normalised rows rounded to 16 bits -- exactly
//                            what torch.autocast feeds nn.Linear (LayerNorm runs in fp32, Linear casts its input).
// Rounding is round-to-nearest-even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).
#include "internal.h"

namespace {

template <int PREC>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  if constexpr (PREC == 1) {
    typedef __bf16 v4 __attribute__((ext_vector_type(4)));
    v4 r;
    r[0] = (__bf16)a; r[1] = (__bf16)b; r[2] = (__bf16)c; r[3] = (__bf16)d;
    return __builtin_bit_cast(uint2, r);
  } else {
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    v4 r;
    r[0] = (_Float16)a; r[1] = (_Float16)b; r[2] = (_Float16)c; r[3] = (_Float16)d;
    return __builtin_bit_cast(uint2, r);
  }
}

template <int PREC>
__global__ __launch_bounds__(256) void cast16_kernel(Cast16Jobs jobs) {
  const int j = blockIdx.y;
  const float4* src = (const float4*)jobs.src[j];
  uint2* dst = (uint2*)jobs.dst[j];
  const size_t n4 = jobs.n4[j];
  // four independent loads per trip (a bag-sized job -- the classifier's feature matrix, round 5 -- is ~4 float4 per thread
  // at the launcher's grid: one memory round trip; the one-load loop was a chain of them)
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < n4) {
#ifdef RRT_NO_NT_CAST
        v[u] = src[i + u * stride];
#else
        v[u] = ld_nt((const float*)(src + i + u * stride));      // the fp32 source is read once (configs[2]: +0.7 %)
#endif
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < n4) dst[i + u * stride] = pack4<PREC>(v[u].x, v[u].y, v[u].z, v[u].w);
  }
}

template <int NV, int PREC, bool FULL>   // float4 per lane: dim <= NV*256; FULL: dim == NV*256, no lane predication (ln_partition.hip)
__global__ __launch_bounds__(256) void ln_partition16_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             uint16_t* __restrict__ u, int dim, GridDev g,
                                                             int* __restrict__ zero, int n_zero) {
  if (zero != nullptr && blockIdx.x == 0)              // side job (as ln_partition_kernel): arrival counters of a later kernel
    for (int i = threadIdx.x; i < n_zero; i += 256) zero[i] = 0;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);   // padded-grid token index
  if (t >= g.Np) return;
  uint16_t* dst = u + (size_t)token_to_slot(t, g) * dim;
  if (t >= g.L) {   // pad row: exact zeros (they are NOT layer-normed in the reference)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int c = (v * 64 + lane) * 4;
      if (FULL || c < dim) *(uint2*)(dst + c) = make_uint2(0u, 0u);
    }
    return;
  }
  const float* src = x + (size_t)t * dim;
  float4 r[NV];
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    r[v] = (FULL || c < dim) ? ld_row<NT_LN1>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    sum += (r[v].x + r[v].y) + (r[v].z + r[v].w);
  }
  const float inv_d = 1.0f / (float)dim;
  const float mean = wave_sum(sum) * inv_d;
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (FULL || c < dim) {
      float a = r[v].x - mean, b = r[v].y - mean, cc = r[v].z - mean, d = r[v].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (FULL || c < dim) {
      float4 gm = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
      *(uint2*)(dst + c) = pack4<PREC>((r[v].x - mean) * rstd * gm.x + bt.x, (r[v].y - mean) * rstd * gm.y + bt.y,
                                       (r[v].z - mean) * rstd * gm.z + bt.z, (r[v].w - mean) * rstd * gm.w + bt.w);
    }
  }
}

// ---- RRT_COMPUTE_F32X3: fp32 values as (hi, lo) bf16 pairs -----------------------------------------------------
// x ~ hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significant bits; both round-to-nearest-even).  A row of K
// floats becomes K / 32 groups of 128 bytes: [32 x hi | 32 x lo] -- the SAME 4 bytes per element and the same
// 128-byte K tiles as fp32, so the GEMM kernels stage it with their fp32 code and read slots 0..3 (hi) / 4..7 (lo)
// of a tile row as bf16 MFMA operands.
__device__ __forceinline__ void split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
  typedef __bf16 v4 __attribute__((ext_vector_type(4)));
  v4 h, l;
  h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
  l[0] = (__bf16)(a - (float)h[0]); l[1] = (__bf16)(b - (float)h[1]);
  l[2] = (__bf16)(c - (float)h[2]); l[3] = (__bf16)(d - (float)h[3]);
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
}
// element index e (multiple of 4) of a flat fp32 array -> byte offset of its hi quad in the split image (+64: lo quad)
__device__ __forceinline__ size_t split_off(size_t e) { return (e >> 5) * 128 + (e & 31) * 2; }

__global__ __launch_bounds__(256) void cast_split_kernel(Cast16Jobs jobs) {
  const int j = blockIdx.y;
  const float4* src = (const float4*)jobs.src[j];
  char* dst = (char*)jobs.dst[j];
  const size_t n4 = jobs.n4[j];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = src[i];
    uint2 hi, lo;
    split4(v.x, v.y, v.z, v.w, hi, lo);
    const size_t off = split_off(i * 4);
    *(uint2*)(dst + off) = hi;
    *(uint2*)(dst + off + 64) = lo;
  }
}

template <int NV, bool FULL>
__global__ __launch_bounds__(256) void ln_partition_split_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 char* __restrict__ u, int dim, GridDev g) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= g.Np) return;
  char* dst = u + (size_t)token_to_slot(t, g) * dim * 4;
  if (t >= g.L) {   // pad row: exact zeros (hi = lo = 0)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int c = (v * 64 + lane) * 4;
      if (FULL || c < dim) *(float4*)(dst + (size_t)c * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const float* src = x + (size_t)t * dim;
  float4 r[NV];
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    r[v] = (FULL || c < dim) ? ld_row<NT_LN1>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    sum += (r[v].x + r[v].y) + (r[v].z + r[v].w);
  }
  const float inv_d = 1.0f / (float)dim;
  const float mean = wave_sum(sum) * inv_d;
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (FULL || c < dim) {
      float a = r[v].x - mean, b = r[v].y - mean, cc = r[v].z - mean, d = r[v].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + LN_EPS);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = (v * 64 + lane) * 4;
    if (FULL || c < dim) {
      float4 gm = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
      uint2 hi, lo;
      split4((r[v].x - mean) * rstd * gm.x + bt.x, (r[v].y - mean) * rstd * gm.y + bt.y,
             (r[v].z - mean) * rstd * gm.z + bt.z, (r[v].w - mean) * rstd * gm.w + bt.w, hi, lo);
      const size_t off = split_off((size_t)c);
      *(uint2*)(dst + off) = hi;
      *(uint2*)(dst + off + 64) = lo;
    }
  }
}

}  // namespace

hipError_t launch_cast_split(const Cast16Jobs& jobs, hipStream_t st) {
  if (jobs.count <= 0) return hipSuccess;
  size_t mx = 0;
  for (int j = 0; j < jobs.count; ++j) mx = jobs.n4[j] > mx ? jobs.n4[j] : mx;
  size_t blocks = (mx + 256 * 4 - 1) / (256 * 4);
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  cast_split_kernel<<<dim3((unsigned)blocks, jobs.count), 256, 0, st>>>(jobs);
  return hipGetLastError();
}

hipError_t launch_ln_partition_split(const float* x, const float* gamma, const float* beta, void* u, int dim,
                                     const GridDev& g, hipStream_t st) {
  if (dim % 32) return hipErrorInvalidValue;
  dim3 grid((g.Np + 3) / 4), block(256);
#define RRT_LNPS(NV)                                                                                          \
  do {                                                                                                      \
    if (RRT_ALLOW_FULL && dim == NV * 256) ln_partition_split_kernel<NV, true><<<grid, block, 0, st>>>(x, gamma, beta, (char*)u, dim, g);  \
    else ln_partition_split_kernel<NV, false><<<grid, block, 0, st>>>(x, gamma, beta, (char*)u, dim, g);                 \
  } while (0)
  if (dim <= 256) RRT_LNPS(1);
  else if (dim <= 512) RRT_LNPS(2);
  else if (dim <= 1024) RRT_LNPS(4);
  else RRT_LNPS(8);
#undef RRT_LNPS
  return hipGetLastError();
}

hipError_t launch_cast16(const Cast16Jobs& jobs, int prec, hipStream_t st) {
  if (jobs.count <= 0) return hipSuccess;
  if (prec != 1 && prec != 2) return hipErrorInvalidValue;
  size_t mx = 0;
  for (int j = 0; j < jobs.count; ++j) mx = jobs.n4[j] > mx ? jobs.n4[j] : mx;
  size_t blocks = (mx + 256 * 4 - 1) / (256 * 4);           // ~4 float4 per thread
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;                         // (512 until round 5: weight-sized jobs only)
  dim3 grid((unsigned)blocks, jobs.count);
  if (prec == 1) cast16_kernel<1><<<grid, 256, 0, st>>>(jobs);
  else cast16_kernel<2><<<grid, 256, 0, st>>>(jobs);
  return hipGetLastError();
}

hipError_t launch_ln_partition16(const float* x, const float* gamma, const float* beta, uint16_t* u, int dim,
                                 const GridDev& g, int prec, hipStream_t st, int* zero, int n_zero) {
  if (prec != 1 && prec != 2) return hipErrorInvalidValue;
  dim3 grid((g.Np + 3) / 4), block(256);
#define RRT_LNP16(NV)                                                                               \
  do {                                                                                              \
    if (RRT_ALLOW_FULL && dim == NV * 256) {                                                                          \
      if (prec == 1) ln_partition16_kernel<NV, 1, true><<<grid, block, 0, st>>>(x, gamma, beta, u, dim, g, zero, n_zero);  \
      else ln_partition16_kernel<NV, 2, true><<<grid, block, 0, st>>>(x, gamma, beta, u, dim, g, zero, n_zero);           \
    } else {                                                                                        \
      if (prec == 1) ln_partition16_kernel<NV, 1, false><<<grid, block, 0, st>>>(x, gamma, beta, u, dim, g, zero, n_zero); \
      else ln_partition16_kernel<NV, 2, false><<<grid, block, 0, st>>>(x, gamma, beta, u, dim, g, zero, n_zero);          \
    }                                                                                               \
  } while (0)
  if (dim <= 256) RRT_LNP16(1);
  else if (dim <= 512) RRT_LNP16(2);
  else if (dim <= 1024) RRT_LNP16(4);
  else RRT_LNP16(8);
#undef RRT_LNP16
  return hipGetLastError();
}
