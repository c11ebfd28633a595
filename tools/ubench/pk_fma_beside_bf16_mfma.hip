// pk_fma_beside_bf16_mfma.hip -- minimal reproducer ATTEMPT for the round-5 hazard: compiler-formed packed
// fp32 instructions (v_pk_fma_f32 with a cross-half op_sel) in a streaming kernel returned run-to-run different values when
// the wave shared its SIMD with bf16-MFMA waves of ANOTHER kernel.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_fma_beside_bf16_mfma.hip -o tools/_abl/pk_fma && tools/_abl/pk_fma
//
// Two kernels on two streams:
//   * contract_{pk,ref}: the contraction of crmsa_combine_parts_kernel, acc[n] += w[n] * x over a region's rows with the weights
//     read from LDS as float4 (what compiled to v_pk_fma_f32 ... op_sel:[1,0,0] in the product).  ONE body, compiled twice in
//     this translation unit through the function-level target attribute: with packed fp32 (`pk`) and without (`ref`).
//   * hog<BF>: a pure MFMA loop (bf16 16x16x32 or fp32 16x16x4), one or two waves per SIMD, small enough (no LDS, < 64 VGPRs) for the
//     contraction's waves to be co-resident on the same SIMDs.
// Protocol: reference = `ref` alone on an idle chip.  Then `pk` and `ref`, each alone / beside the bf16 hog / beside the fp32
// hog, LAUNCHES times; every launch's output is compared bit for bit with the reference on the device; the program prints the
// number of launches (and elements) that differ.  Exit code 0 always: it is a measurement.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int P = 144, KM = 3, COLS = 64;      // rows of a region, representatives, columns of a slab

#define CONTRACT_BODY                                                                                          \
  __shared__ float4 s_w[P];                                                                                    \
  const int tid = threadIdx.x, cl = tid & 15, rg = tid >> 4;                                                   \
  const int reg = blockIdx.x, slab = blockIdx.y;                                                               \
  for (int p = tid; p < P; p += 256) s_w[p] = W[(size_t)reg * P + p];                                          \
  __syncthreads();                                                                                             \
  float4 acc[KM];                                                                                              \
  for (int n = 0; n < KM; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);                                       \
  float4 xv[9];                                                                                                \
  for (int u = 0; u < 9; ++u) xv[u] = *(const float4*)(X + ((size_t)reg * P + rg + 16 * u) * dim + slab * COLS + cl * 4); \
  for (int u = 0; u < 9; ++u) {                                                                                \
    const float4 w = s_w[rg + 16 * u];                                                                         \
    const float4 x = xv[u];                                                                                    \
    acc[0].x += w.x * x.x; acc[0].y += w.x * x.y; acc[0].z += w.x * x.z; acc[0].w += w.x * x.w;                \
    acc[1].x += w.y * x.x; acc[1].y += w.y * x.y; acc[1].z += w.y * x.z; acc[1].w += w.y * x.w;                \
    acc[2].x += w.z * x.x; acc[2].y += w.z * x.y; acc[2].z += w.z * x.z; acc[2].w += w.z * x.w;                \
  }                                                                                                            \
  __shared__ float4 part[16][KM][16];                                                                          \
  for (int n = 0; n < KM; ++n) part[rg][n][cl] = acc[n];                                                       \
  __syncthreads();                                                                                             \
  if (tid < KM * 16) {                                                                                         \
    const int n = tid >> 4, c = tid & 15;                                                                      \
    float4 a = part[0][n][c];                                                                                  \
    for (int q = 1; q < 16; ++q) { const float4 b = part[q][n][c]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; } \
    *(float4*)(Y + ((size_t)n * gridDim.x + reg) * dim + slab * COLS + c * 4) = a;                             \
  }

__attribute__((target("packed-fp32-ops"))) __global__ __launch_bounds__(256) void contract_pk(
    const float* __restrict__ X, const float4* __restrict__ W, float* __restrict__ Y, int dim) { CONTRACT_BODY }
__attribute__((target("no-packed-fp32-ops"))) __global__ __launch_bounds__(256) void contract_ref(
    const float* __restrict__ X, const float4* __restrict__ W, float* __restrict__ Y, int dim) { CONTRACT_BODY }

template <bool BF>
__global__ __launch_bounds__(256) void hog(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (BF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}

__global__ void compare(const unsigned* __restrict__ y, const unsigned* __restrict__ ref, int n, unsigned* counters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned bad = 0;
  for (; i < n; i += gridDim.x * blockDim.x) bad += y[i] != ref[i];
  if (bad) { atomicAdd(&counters[0], bad); atomicOr(&counters[1], 1u); }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 2000;
  const int R = 64, dim = 512;
  float *X, *Y, *Yref, *sink; float4* W; unsigned* cnt;
  hipMalloc(&X, (size_t)R * P * dim * 4); hipMalloc(&W, (size_t)R * P * 16);
  hipMalloc(&Y, (size_t)KM * R * dim * 4); hipMalloc(&Yref, (size_t)KM * R * dim * 4);
  hipMalloc(&sink, 64); hipMalloc(&cnt, 8);
  std::vector<float> hx((size_t)R * P * dim), hw((size_t)R * P * 4);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto& v : hx) v = rnd() * 4.f;
  for (auto& v : hw) v = rnd() * 0.02f + 0.007f;
  hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  const dim3 grid(R, dim / COLS);
  contract_ref<<<grid, 256, 0, sb>>>(X, W, Yref, dim);
  hipStreamSynchronize(sb);
  const int n = KM * R * dim;
  printf("contraction acc[n] += w[n] * x (KM = %d, P = %d rows, %d x %d blocks), %d launches per cell; differing launches / differing "
         "elements against contract_ref alone\n", KM, P, R, dim / COLS, launches);
  for (int variant = 0; variant < 2; ++variant) {
    for (int beside = 0; beside < 5; ++beside) {       // 0 alone, 1 bf16 hog 1 wave/SIMD, 2 bf16 hog 2 waves/SIMD, 3 fp32 hog, 4 bf16 hog 4 blocks/CU
      unsigned bad_launches = 0, bad_elems = 0;
      const int chunk = 50;                             // the hog is relaunched every `chunk` contractions (~1 ms each)
      for (int l0 = 0; l0 < launches; l0 += chunk) {
        if (beside == 1) hog<true><<<256, 256, 0, sa>>>(sink, 60000);
        if (beside == 2) hog<true><<<512, 256, 0, sa>>>(sink, 30000);
        if (beside == 3) hog<false><<<256, 256, 0, sa>>>(sink, 30000);
        if (beside == 4) hog<true><<<1024, 256, 0, sa>>>(sink, 15000);
        for (int l = l0; l < l0 + chunk && l < launches; ++l) {
          hipMemsetAsync(cnt, 0, 8, sb);
          hipMemsetAsync(Y, 0xFF, (size_t)n * 4, sb);
          if (variant == 0) contract_pk<<<grid, 256, 0, sb>>>(X, W, Y, dim);
          else contract_ref<<<grid, 256, 0, sb>>>(X, W, Y, dim);
          compare<<<64, 256, 0, sb>>>((const unsigned*)Y, (const unsigned*)Yref, n, cnt);
          unsigned h[2];
          hipMemcpyAsync(h, cnt, 8, hipMemcpyDeviceToHost, sb);
          hipStreamSynchronize(sb);
          bad_launches += h[1];
          bad_elems += h[0];
        }
        hipStreamSynchronize(sa);
      }
      const char* names[5] = {"alone", "beside bf16-MFMA hog, 1 wave/SIMD", "beside bf16-MFMA hog, 2 waves/SIMD",
                              "beside fp32-MFMA hog, 1 wave/SIMD", "beside bf16-MFMA hog, 4 waves/SIMD"};
      printf("  %-13s %-36s %6u / %d launches differ, %u elements\n", variant == 0 ? "contract_pk" : "contract_ref", names[beside],
             bad_launches, launches, bad_elems);
      fflush(stdout);
    }
  }
  if (hipGetLastError() != hipSuccess) printf("HIP error\n");
  return 0;
}
