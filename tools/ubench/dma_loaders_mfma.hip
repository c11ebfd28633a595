// dma_loaders_mfma.hip -- round 6 (review item 1b): what does the LDS-DMA operand stream of rmsa_pair16 deliver per CU as a
// function of HOW MANY waves issue it, with bf16-MFMA consumers running beside / inside the issuing waves?
// One block of 16 waves per CU (256 blocks), each streaming rmsa_pair16's operand set of a (region pair, head) item at
// P = 144 -- per K tile (64 elements) 2 x 144 U rows + 192 W rows of 128 bytes = 60 one-KiB pieces, 8 K tiles, 491 KB per
// block, L2-resident after the first launch -- through a two-stage ring, one barrier per K tile.
//   mode 0  "specialised": waves [0, NL) only issue pieces, waves [NL, 16) only multiply (30 MFMAs + 16 fragment reads per
//           wave and K tile: pair16's consumer loop);
//   mode 1  "mixed": waves [0, NL) issue AND multiply, the others only multiply (NL = 16 is rmsa_pair16's phase 1);
//   mode 2  "no MFMA": waves [0, NL) issue, nobody multiplies (the bare stream).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_loaders_mfma.hip -o tools/_abl/dma_loaders_mfma && tools/_abl/dma_loaders_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
constexpr int BM2 = 288, BNW = 192, ROWS = BM2 + BNW, NPIECE = ROWS / 8, STAGE = ROWS * 128, D = 512, NK = 8;

template <int NL, int MODE>
__global__ __launch_bounds__(1024, 1) void k(const char* __restrict__ U, const char* __restrict__ W, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, pair = (b >> 3) / 8 * 8 + (b & 7), head = (b >> 3) % 8;
  const unsigned lds_b = lds_addr_of(smem);
  const bool loader = wave < NL, consumer = MODE == 1 || (MODE == 0 && wave >= NL);
  auto issue = [&](int kt, int slot) {
    for (int p = wave; p < NPIECE; p += NL) {
      const int row = p * 8 + (lane >> 3), s = lane & 7, sw = s ^ ((row >> 1) & 7);
      const bool isw = row >= BM2;
      const int r = row - BM2;
      const size_t base = isw ? (size_t)((r >> 6) * D + head * 64 + (r & 63)) * D * 2 : (size_t)(pair * BM2 + row) * D * 2;
      dma16s((isw ? W : U) + (size_t)kt * 128, (unsigned)base + (unsigned)(sw << 4), lds_b + slot * STAGE + p * 1024);
    }
  };
  if (loader) issue(0, 0);
  f32x4 acc[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lr = lane & 15, lg = lane >> 4, cw = wave & 3, rsel = (wave >> 3) & 1;
  for (int kt = 0; kt < NK; ++kt) {
    if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (loader && kt + 1 < NK) issue(kt + 1, (kt + 1) & 1);
    if (consumer) {
      const char* As = smem + (kt & 1) * STAGE + rsel * 144 * 128;
      const char* Bs = smem + (kt & 1) * STAGE + BM2 * 128;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a8[5], b8[3];
        const int cs = 4 * kk + lg;
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int row = 64 * j + 16 * cw + lr; b8[j] = *(const bf16x8*)(Bs + row * 128 + ((cs ^ ((row >> 1) & 7)) << 4)); }
#pragma unroll
        for (int i = 0; i < 5; ++i) { const int row = (i * 16 + lr) % 144; a8[i] = *(const bf16x8*)(As + row * 128 + ((cs ^ ((row >> 1) & 7)) << 4)); }
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i * 3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], b8[j], acc[i * 3 + j], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 15; ++i) s += acc[i][0];
  if (s == 123.456f) sink[0] = s;
}

template <int NL, int MODE>
void run(const char* U, const char* W, float* sink) {
  auto kern = k<NL, MODE>;
  const size_t lds = 2 * STAGE;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) kern<<<256, 1024, lds>>>(U, W, sink);
  hipEventRecord(a);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) kern<<<256, 1024, lds>>>(U, W, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps, bytes = (double)NK * STAGE;
  const char* names[] = {"specialised", "mixed", "no MFMA"};
  printf("  %-11s  %2d issuing wave(s) of 16: %6.2f us per launch  %5.1f GB/s per CU  %5.1f B/clk/CU at 2.4 GHz  chip %5.2f TB/s%s\n",
         names[MODE], NL, us, bytes / us * 1e-3, bytes / (us * 2400.0), bytes * 256 / us * 1e-6,
         hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
  char *U, *W; float* sink;
  hipMalloc(&U, (size_t)32 * BM2 * D * 2); hipMalloc(&W, (size_t)3 * D * D * 2); hipMalloc(&sink, 64);
  hipMemset(U, 0x11, (size_t)32 * BM2 * D * 2); hipMemset(W, 0x22, (size_t)3 * D * D * 2);
  printf("LDS-DMA operand stream of rmsa_pair16 (491 KB per block, 256 blocks of 16 waves, one per CU, L2-warm), by issuing waves\n");
  run<1, 2>(U, W, sink); run<2, 2>(U, W, sink); run<4, 2>(U, W, sink); run<8, 2>(U, W, sink); run<16, 2>(U, W, sink);
  run<1, 0>(U, W, sink); run<2, 0>(U, W, sink); run<4, 0>(U, W, sink); run<8, 0>(U, W, sink);
  run<1, 1>(U, W, sink); run<2, 1>(U, W, sink); run<4, 1>(U, W, sink); run<8, 1>(U, W, sink); run<16, 1>(U, W, sink);
  printf("MFMA floor of the consumer loop: 16 waves x 8 K tiles x 30 MFMAs x 16 cycles / 4 SIMDs = 15.4 K cycles = 6.4 us at 2.4 GHz\n");
  return 0;
}
