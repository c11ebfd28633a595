// dma_mixed.hip -- can the two routes into LDS be ADDED?  tools/ubench/dma_rows.hip says the LDS-DMA route
// (global_load_lds_dwordx4) delivers ~14 / 21 / 30 B/clk/CU with 4 / 8 / 16 issuing waves, the instruction blocking its wave
// while the piece is accepted.  The register route (global_load_dwordx4 -> VGPRs -> ds_write_b128) does not block at issue.
// If the two have different bottlenecks, some waves on each should stream more bytes per clock than either alone.
// Same stream as dma_rows (the 16-bit R-MSA projection's operands: a region's U panel + a head's W slice per block, K tiles
// of 128-byte rows, two-stage ring, one barrier per K tile), pieces of 1 KiB dealt round-robin to NDMA + NREG waves.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_mixed.hip -o tools/_abl/dma_mixed && tools/_abl/dma_mixed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// waves 0 .. NDMA-1 issue LDS-DMA, waves NDMA .. NDMA+NREG-1 load into registers and write to LDS one K tile later
template <int NDMA, int NREG>
__global__ __launch_bounds__(64 * (NDMA + NREG)) void mixed_kernel(const char* __restrict__ U, const char* __restrict__ W, int BM,
                                                                    int D, int heads, unsigned* sink) {
  constexpr int NW = NDMA + NREG;
  constexpr int LP = (64 + NW - 1) / NW;               // pieces per wave per stage, at most (rows <= 512)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int xcd = b & 7, idx = b >> 3, grp = idx / heads;
  const int reg = grp * 8 + xcd, head = idx - grp * heads;
  const int rows = BM + 192, npieces = rows / 8, stage_b = rows * 128, nk = D * 2 / 128;
  const unsigned lds_b = lds_addr_of(smem);
  unsigned off[LP];                                    // per-lane byte offset of this wave's q-th piece (from U or W)
  bool isw[LP], live[LP];
#pragma unroll
  for (int q = 0; q < LP; ++q) {
    const int p = q * NW + wave;
    live[q] = p < npieces;
    const int pc = live[q] ? p : 0;
    const int row = pc * 8 + (lane >> 3), s = lane & 7, sw = s ^ ((row >> 1) & 7);
    isw[q] = pc * 8 >= BM;                            // wave-uniform (a piece never straddles: BM % 8 == 0)
    size_t base;
    if (row < BM) base = (size_t)(reg * BM + row) * D * 2;
    else { const int r = row - BM; base = (size_t)((r >> 6) * D + head * 64 + (r & 63)) * D * 2; }
    off[q] = (unsigned)base + (unsigned)(sw << 4);
  }
  uint4 r[LP];
  auto issue = [&](const int kt, const int slot) {
#pragma unroll
    for (int q = 0; q < LP; ++q) {
      if (!live[q]) continue;                          // wave-uniform
      const char* base = (isw[q] ? W : U) + (size_t)kt * 128;
      if (wave < NDMA) dma16s(base, off[q], lds_b + slot * stage_b + (q * NW + wave) * 1024);
      else r[q] = *(const uint4*)(base + off[q]);
    }
  };
  issue(0, 0);
  unsigned acc = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int slot = kt & 1;
    if (wave < NDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
#pragma unroll
      for (int q = 0; q < LP; ++q)
        if (live[q]) *(uint4*)(smem + slot * stage_b + (q * NW + wave) * 1024 + lane * 16) = r[q];   // waits for the loads
    }
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1, slot ^ 1);
    acc += *(const unsigned*)(smem + slot * stage_b + ((tid * 16) % stage_b));
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NDMA, int NREG>
void run(const char* name, const char* U, const char* W, int R, int BM, int D, int heads, unsigned* sink, double mb) {
  auto k = mixed_kernel<NDMA, NREG>;
  const size_t lds = (size_t)2 * (BM + 192) * 128;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 20;
  for (int i = 0; i < 3; ++i) k<<<R * heads, 64 * (NDMA + NREG), lds>>>(U, W, BM, D, heads, sink);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) k<<<R * heads, 64 * (NDMA + NREG), lds>>>(U, W, BM, D, heads, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  const float us = ms * 1e3f / reps;
  printf("  %-44s %7.1f us  %6.1f B/clk/CU (2.4 GHz, 256 CUs)\n", name, us, mb * 1e6 / (us * 1e-6) / 2.4e9 / 256);
}

int main() {
  const int D = 512, heads = 8;
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int R = cfg == 0 ? 64 : 256, BM = cfg == 0 ? 144 : 128;
    char *U, *W; unsigned* sink;
    hipMalloc(&U, (size_t)R * BM * D * 2); hipMalloc(&W, (size_t)3 * D * D * 2); hipMalloc(&sink, 64);
    hipMemset(U, 1, (size_t)R * BM * D * 2); hipMemset(W, 1, (size_t)3 * D * D * 2);
    const double mb = (double)R * heads * (BM + 192) * D * 2 / 1e6;
    printf("R=%d regions x %d heads, BM=%d: %.0f MB into LDS per launch, one block per CU, two-stage ring\n", R, heads, BM, mb);
    run<8, 0>("8 DMA waves", U, W, R, BM, D, heads, sink, mb);
    run<16, 0>("16 DMA waves", U, W, R, BM, D, heads, sink, mb);
    run<0, 8>("8 register waves", U, W, R, BM, D, heads, sink, mb);
    run<0, 16>("16 register waves", U, W, R, BM, D, heads, sink, mb);
    run<4, 4>("4 DMA + 4 register waves", U, W, R, BM, D, heads, sink, mb);
    run<8, 8>("8 DMA + 8 register waves", U, W, R, BM, D, heads, sink, mb);
    run<12, 4>("12 DMA + 4 register waves", U, W, R, BM, D, heads, sink, mb);
    run<4, 12>("4 DMA + 12 register waves", U, W, R, BM, D, heads, sink, mb);
    hipFree(U); hipFree(W); hipFree(sink);
  }
  return 0;
}
