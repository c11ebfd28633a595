// Does VALU work overlap with fp32 MFMA on one SIMD?  (a) from another wave, (b) from the same wave.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/_abl/mfma_valu_overlap && ./tools/_abl/mfma_valu_overlap
// One block of 512 threads per CU: waves w and w + 4 share SIMD w.  Cycles from s_memtime, max over the block's waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool BF = false, bool PRIO = false>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[8];
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b + i); }
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // MODE 0: every wave MFMA.  1: waves 0-3 MFMA, 4-7 idle.  2: waves 0-3 MFMA, 4-7 VALU fma.  3: waves 0-3 idle, 4-7 VALU.
  // 4: every wave MFMA + 6 independent VALU fma per MFMA (same wave).  5: like 2 with v_exp instead of fma.
  // 6: like 4 with 2 v_exp per MFMA.   7: waves 4-7 VALU exp alone
  const bool do_mfma = MODE == 0 || MODE == 4 || MODE == 6 || ((MODE == 1 || MODE == 2 || MODE == 5) && wave < 4);
  const bool do_valu = ((MODE == 2 || MODE == 3 || MODE == 5 || MODE == 7) && wave >= 4);
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (BF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if (MODE == 4) {
#pragma unroll
          for (int q = 0; q < 6; ++q) v[q] = __builtin_fmaf(v[q], b, a);
        }
        if (MODE == 6) {
          v[0] = __builtin_amdgcn_exp2f(v[0]);
          v[1] = __builtin_amdgcn_exp2f(v[1]);
        }
      }
    }
  } else if (do_valu) {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    if (MODE == 5 || MODE == 7) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[0] = __builtin_amdgcn_exp2f(v[0]);
          v[1] = __builtin_amdgcn_exp2f(v[1]);
        }
      }
    } else {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(v[q], b, a);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, bool BF = false, bool PRIO = false>
void run(const char* what, float* out, unsigned long long* cyc, int iters) {
  const int nb = 256;
  std::vector<unsigned long long> h(nb * 8);
  k<MODE, BF, PRIO><<<nb, 512>>>(out, cyc, iters);
  k<MODE, BF, PRIO><<<nb, 512>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> lo, hi;
  for (int b = 0; b < nb; ++b) {
    unsigned long long m03 = 0, m47 = 0;
    for (int w = 0; w < 4; ++w) m03 = std::max(m03, h[b * 8 + w]);
    for (int w = 4; w < 8; ++w) m47 = std::max(m47, h[b * 8 + w]);
    lo.push_back((double)m03);
    hi.push_back((double)m47);
  }
  std::sort(lo.begin(), lo.end());
  std::sort(hi.begin(), hi.end());
  printf("%-70s waves0-3 %9.0f  waves4-7 %9.0f  (s_memtime ticks, median block)\n", what, lo[nb / 2], hi[nb / 2]);
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;   // 16000 MFMAs per wave = 512 K cycles at 32 cycles each
  run<1>("1: waves 0-3 MFMA (16 K each), waves 4-7 idle", out, cyc, iters);
  run<0>("0: all 8 waves MFMA (2 per SIMD)", out, cyc, iters);
  run<3>("3: waves 4-7 VALU fma (128 K each), waves 0-3 idle", out, cyc, iters);
  run<2>("2: waves 0-3 MFMA, waves 4-7 VALU fma", out, cyc, iters);
  run<7>("7: waves 4-7 v_exp (32 K each), waves 0-3 idle", out, cyc, iters);
  run<5>("5: waves 0-3 MFMA, waves 4-7 v_exp", out, cyc, iters);
  run<4>("4: all waves MFMA + 6 independent fma per MFMA in the same wave", out, cyc, iters);
  run<6>("6: all waves MFMA + 2 v_exp per MFMA in the same wave", out, cyc, iters);
  run<2, false, true>("2p: waves 0-3 MFMA, waves 4-7 VALU fma at s_setprio 3", out, cyc, iters);
  run<5, false, true>("5p: waves 0-3 MFMA, waves 4-7 v_exp at s_setprio 3", out, cyc, iters);
  printf("---- the same with v_mfma_f32_16x16x32_bf16\n");
  run<1, true>("1: waves 0-3 MFMA bf16 (16 K each), waves 4-7 idle", out, cyc, iters);
  run<0, true>("0: all 8 waves MFMA bf16 (2 per SIMD)", out, cyc, iters);
  run<2, true>("2: waves 0-3 MFMA bf16, waves 4-7 VALU fma", out, cyc, iters);
  run<5, true>("5: waves 0-3 MFMA bf16, waves 4-7 v_exp", out, cyc, iters);
  run<4, true>("4: all waves MFMA bf16 + 6 independent fma per MFMA in the same wave", out, cyc, iters);
  run<6, true>("6: all waves MFMA bf16 + 2 v_exp per MFMA in the same wave", out, cyc, iters);
  run<2, true, true>("2p: waves 0-3 MFMA bf16, waves 4-7 VALU fma at s_setprio 3", out, cyc, iters);
  return 0;
}
