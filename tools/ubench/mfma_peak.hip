// What does the chip SUSTAIN on the matrix cores?  All CUs, pure MFMA loops, wall clock (HIP events) + cycles (s_memtime at 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/_abl/mfma_peak && ./tools/_abl/mfma_peak
// The roofline denominators of bench.py are the data-sheet peaks (fp32 157.3 TF, bf16 2.5 PF = 256 CUs x 4 SIMDs x
// 2.4 GHz x 64 / 1024 flops per cycle).  This program measures the rate a kernel that does nothing else reaches, for
// launches of ~0.1 / 1 / 20 ms, with 1 or 2 waves per SIMD -- i.e. the clock the chip holds under matrix load.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (BF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}

template <bool BF>
void run(const char* name, int threads, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256;
  for (int w = 0; w < 3; ++w) k<BF><<<blocks, threads>>>(out, iters);
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<BF><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flops = (double)blocks * (threads / 64) * iters * 8.0 * (BF ? 2.0 * 16 * 16 * 32 : 2.0 * 16 * 16 * 4);
  const double tf = flops / (ms * 1e-3) * 1e-12;
  const double peak = BF ? 2516.6 : 157.3;
  printf("%-5s %d waves/SIMD  %8.3f ms/launch  %8.1f TFLOP/s  = %.3f of %.1f  (implied clock %.2f GHz)\n", name, threads / 256,
         ms, tf, tf / peak, peak, 2.4 * tf / peak);
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  const int f32_its[3] = {1200, 12000, 240000}, bf_its[3] = {2400, 24000, 480000};
  for (int t = 0; t < 3; ++t) {
    run<false>("f32", 256, f32_its[t] * 2, out);
    run<false>("f32", 512, f32_its[t], out);
  }
  // the headline kernel's FLOPs as nothing but MFMAs: rmsa_fused_kernel at N = 9000 does 17.21 GFLOP per launch (512 blocks of
  // 8 waves; here 256 blocks x 8 waves x 513 x 8 MFMAs = 17.2 GFLOP) -- what a launch of that size can reach at all
  printf("-- a launch with the fused fp32 R-MSA kernel's FLOPs (17.2 G), and the 16-bit pair kernel's as bf16 MFMAs:\n");
  run<false>("f32", 512, 513, out);
  run<true>("bf16", 512, 64, out);
  for (int t = 0; t < 3; ++t) {
    run<true>("bf16", 256, bf_its[t] * 2, out);
    run<true>("bf16", 512, bf_its[t], out);
  }
  return 0;
}
