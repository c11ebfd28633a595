// ln_partition.hip -- LayerNorm + zero-pad + region partition in one pass over the bag.
//
// Replaces: nn.LayerNorm (modules/rrt.py:121-123), the fp32 zero-pad torch.cat
// (modules/rmsa.py:199-200) and region_partition's permute+contiguous copy
// (modules/rmsa.py:28-39).  HBM-bound: reads L*D, writes Np*D floats, one wave per
// token row, float4 (16 B/lane) accesses, two-pass mean/variance held in registers.
#include "internal.h"

// FULL: dim == NV * 256 (every lane's columns exist) -> no lane predication in the row loops.  Not only a few
// instructions: the predicated form (v_cmp -> SGPR mask -> v_cndmask / s_and_saveexec around packed fp32 ops) gave
// wrong values in lanes 48..63 of a row now and then when the wave shared its SIMD with bf16-MFMA waves of another
// kernel (found as run-to-run differences of crmsa_logits_kernel next to rmsa_fused_x3_kernel)
template <int NV, bool FULL>   // float4 per lane: supports dim <= NV*256
__global__ __launch_bounds__(256) void ln_partition_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ u, int dim, GridDev g,
                                                           int* __restrict__ zero, int n_zero) {
  if (zero != nullptr && blockIdx.x == 0)
    for (int i = threadIdx.x; i < n_zero; i += 256) zero[i] = 0;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);   // padded-grid token index
  if (t >= g.Np) return;
  float* dst = u + (size_t)token_to_slot(t, g) * dim;
  if (t >= g.L) {   // pad row: exact zeros (they are NOT layer-normed in the reference)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int c = (v * 64 + lane) * 4;
      if (FULL || c < dim) *(float4*)(dst + c) = make_float4(0.f, 0.f, 0.f, 0.f);
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
      float4 o;
      o.x = (r[v].x - mean) * rstd * gm.x + bt.x;
      o.y = (r[v].y - mean) * rstd * gm.y + bt.y;
      o.z = (r[v].z - mean) * rstd * gm.z + bt.z;
      o.w = (r[v].w - mean) * rstd * gm.w + bt.w;
      *(float4*)(dst + c) = o;
    }
  }
}

hipError_t launch_ln_partition(const float* x, const float* gamma, const float* beta, float* u,
                               int dim, const GridDev& g, hipStream_t st, int* zero, int n_zero) {
  dim3 grid((g.Np + 3) / 4), block(256);
#define RRT_LNP(NV)                                                                              \
  do {                                                                                         \
    if (RRT_ALLOW_FULL && dim == NV * 256) ln_partition_kernel<NV, true><<<grid, block, 0, st>>>(x, gamma, beta, u, dim, g, zero, n_zero);  \
    else ln_partition_kernel<NV, false><<<grid, block, 0, st>>>(x, gamma, beta, u, dim, g, zero, n_zero);                 \
  } while (0)
  if (dim <= 256) RRT_LNP(1);
  else if (dim <= 512) RRT_LNP(2);
  else if (dim <= 1024) RRT_LNP(4);
  else RRT_LNP(8);
#undef RRT_LNP
  return hipGetLastError();
}
