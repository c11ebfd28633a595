// fused16.h -- helpers shared by the 16-bit fused R-MSA kernels (rmsa_fused16.hip: one (region, head) per block;
// rmsa_pair16.hip: two regions x one head per block).
#pragma once
#include "internal.h"

namespace f16k {

constexpr int HD = 64;
constexpr int BN = 3 * HD;          // q | k | v columns of one head
constexpr int ROWB = 128;           // bytes of one staged row = 64 16-bit elements = one K tile
constexpr int VT_PITCH = 512;       // bytes per V^T row: 32 x 16-byte slots (keys <= 256), XOR-swizzled over 16
constexpr float NEG_BIG = -3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;

template <int PREC>
struct H16;
template <>
struct H16<1> {
  typedef __bf16 frag __attribute__((ext_vector_type(8)));
  typedef __bf16 v4 __attribute__((ext_vector_type(4)));
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  typedef __bf16 elem;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct H16<2> {
  typedef _Float16 frag __attribute__((ext_vector_type(8)));
  typedef _Float16 v4 __attribute__((ext_vector_type(4)));
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  typedef _Float16 elem;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <int PREC>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  typename H16<PREC>::v4 r;
  using E = typename H16<PREC>::elem;
  r[0] = (E)a; r[1] = (E)b; r[2] = (E)c; r[3] = (E)d;
  return __builtin_bit_cast(uint2, r);
}
template <int PREC>
__device__ __forceinline__ unsigned pack2(float a, float b) {
  typename H16<PREC>::v2 r;
  using E = typename H16<PREC>::elem;
  r[0] = (E)a; r[1] = (E)b;
  return __builtin_bit_cast(unsigned, r);
}
template <int PREC>
__device__ __forceinline__ typename H16<PREC>::frag pack8(const f32x4& a, const f32x4& b) {
  typename H16<PREC>::frag r;
  using E = typename H16<PREC>::elem;
  r[0] = (E)a[0]; r[1] = (E)a[1]; r[2] = (E)a[2]; r[3] = (E)a[3];
  r[4] = (E)b[0]; r[5] = (E)b[1]; r[6] = (E)b[2]; r[7] = (E)b[3];
  return r;
}

// XOR key of the 16-byte slots of V^T row d: injective over the 16 rows a wave WRITES together (d = 16 w + lr: bits
// 1..0 = lr >> 2, bits 3..2 = (lr & 3) ^ (w & 3)) and over the 16 rows it READS together (d = 4 lr + c: lr ^ 4 c)
__device__ __forceinline__ int vt_swz(int d) { return ((d >> 2) ^ ((d & 3) << 2)) & 15; }

// wait until at most N of this wave's vector-memory operations (the LDS-DMA pieces) are still in flight
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace f16k
