// common.h -- shared device helpers for the gfx950 RRTEncoder kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RRT_WAVE 64

// Device copy of the region grid (modules/rmsa.py:175-202 geometry, 32-bit).
struct GridDev {
  int L;    // real tokens
  int H;    // padded grid side
  int s;    // region side
  int rs;   // regions per side
  int P;    // tokens per region (s*s)
  int Np;   // H*H
};

// padded-grid token index -> region-major slot (region_partition, rmsa.py:28-39)
__device__ __forceinline__ int token_to_slot(int t, const GridDev& g) {
  int i = t / g.H, j = t - i * g.H;
  int ri = i / g.s, pi = i - ri * g.s;
  int rj = j / g.s, pj = j - rj * g.s;
  return (ri * g.rs + rj) * g.P + pi * g.s + pj;
}
// region-major slot -> padded-grid token index (region_reverse, rmsa.py:41-54)
__device__ __forceinline__ int slot_to_token(int slot, const GridDev& g) {
  int reg = slot / g.P, p = slot - reg * g.P;
  int ri = reg / g.rs, rj = reg - ri * g.rs;
  int pi = p / g.s, pj = p - pi * g.s;
  return (ri * g.s + pi) * g.H + rj * g.s + pj;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}

// 16-byte global -> LDS DMA (global_load_lds_dwordx4): the LDS destination is the
// wave-uniform `lds_wave_base` + lane*16; the global source is per lane.
__device__ __forceinline__ void dma16(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base,
                                   16, 0, 0);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#define LN_EPS 1e-5f
