// dma_rows.hip -- what does the LDS-DMA path (global_load_lds_dwordx4) deliver for the fused 16-bit R-MSA kernel's operand
// stream, and how does it depend on (a) the width of the row segment one K tile takes (128 B = 64 elements, or 64 B = 32
// elements: twice the requests for the same bytes), (b) one or two resident blocks per CU, (c) 4 or 8 issuing waves?
// The kernel is the projection loop of rmsa_fused16 with the MFMAs removed: block = (region, head), streams the region's
// U panel (BM rows x D 16-bit) and the head's W slice (192 rows x D) through a ring of NSTG stages, one barrier per K tile.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_rows.hip -o tools/_abl/dma_rows && tools/_abl/dma_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// SEG: bytes of a row one K tile takes (128 or 64); NW: issuing waves (of 8); pad_lds: extra dynamic LDS to force 1 block/CU
template <int SEG, int NW>
__global__ __launch_bounds__(512, 2) void stream_kernel(const char* __restrict__ U, const char* __restrict__ W, int BM, int D,
                                                        int heads, int nstg, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int n_regions = gridDim.x / heads;
  const int xcd = b & 7, idx = b >> 3, grp = idx / heads;
  const int reg = grp * 8 + xcd, head = idx - grp * heads;
  (void)n_regions;
  constexpr int LPR = SEG / 16;                 // lanes per row
  constexpr int RPP = 64 / LPR;                 // rows per 1-KiB piece
  const int rows = BM + 192;
  const int npieces = rows / RPP;
  const int stage_b = rows * SEG;
  const int nk = D * 2 / SEG;
  const unsigned lds_b = lds_addr_of(smem);
  // per-lane source offset of piece p: row = p * RPP + lane / LPR, slot = lane % LPR
  auto src_off = [&](int p) -> unsigned {
    const int row = p * RPP + lane / LPR, s = lane % LPR;
    const int sw = SEG == 128 ? (s ^ ((row >> 1) & 7)) : (s ^ ((row >> 2) & 3));
    size_t base;
    if (row < BM) base = (size_t)(reg * BM + row) * D * 2;                                       // U panel row
    else { const int r = row - BM; base = (size_t)((r >> 6) * D + head * 64 + (r & 63)) * D * 2; }   // W_h row (q | k | v)
    return (unsigned)(base & 0xFFFFFFFFu) + (unsigned)(sw << 4);
  };
  // split pointers: U rows use base U, W rows base W (piece never straddles: BM % RPP == 0)
  auto issue = [&](int kt, int slot) {
    for (int p = wave; p < npieces; p += NW) {
      const bool isw = p * RPP >= BM;
      const char* base = (isw ? W : U) + (size_t)kt * SEG;
      dma16s(base, src_off(p), lds_b + slot * stage_b + p * 1024);
    }
  };
  if (wave < NW) {
    issue(0, 0);
    if (nstg == 3 && nk > 1) issue(1, 1);
  }
  unsigned acc = 0;
  int slot_next = nstg == 3 ? 2 : 1, slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (wave < NW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");   // simple: everything issued so far landed
    __syncthreads();
    if (wave < NW && kt + nstg - 1 < nk) issue(kt + nstg - 1, slot_next);
    slot_next = slot_next + 1 == nstg ? 0 : slot_next + 1;
    // touch the stage (one ds_read per lane) so the data path is real
    acc += *(const unsigned*)(smem + slot * stage_b + ((tid * 16) % stage_b));
    slot = slot + 1 == nstg ? 0 : slot + 1;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}


// The same stream through REGISTERS: global_load_dwordx4 (fire and forget: the wave goes on) + ds_write_b128 one K tile
// later.  The LDS-DMA instruction blocks its wave until the piece is accepted (~300-500 cycles per KiB at these rates): a
// wave that also has MFMAs to issue serialises the two.  (Statically indexed staging registers: a first version indexed
// them with kt & 1 and the compiler put them in scratch -- 5 B/clk.)  DEPTH is ignored (1 tile ahead).
template <int DEPTH>
__global__ __launch_bounds__(512, 2) void stream_reg_kernel(const char* __restrict__ U, const char* __restrict__ W, int BM, int D,
                                                           int heads, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int xcd = b & 7, idx = b >> 3, grp = idx / heads;
  const int reg = grp * 8 + xcd, head = idx - grp * heads;
  const int rows = BM + 192, npieces = rows / 8, stage_b = rows * 128, nk = D * 2 / 128;
  constexpr int LP = 8;                              // pieces per wave per stage, at most (rows <= 512)
  const char* src[LP];
  int dst[LP];
#pragma unroll
  for (int q = 0; q < LP; ++q) {
    int p = q * 8 + wave;
    p = p < npieces ? p : npieces - 1;               // (ragged tail: re-load the last piece, same destination)
    const int row = p * 8 + (lane >> 3), s = lane & 7, sw = s ^ ((row >> 1) & 7);
    if (row < BM) src[q] = U + (size_t)(reg * BM + row) * D * 2 + (sw << 4);
    else { const int r = row - BM; src[q] = W + (size_t)((r >> 6) * D + head * 64 + (r & 63)) * D * 2 + (sw << 4); }
    dst[q] = p * 1024 + lane * 16;
  }
  uint4 r[LP];
#pragma unroll
  for (int q = 0; q < LP; ++q) r[q] = *(const uint4*)(src[q]);
  unsigned acc = 0;
  for (int kt = 0; kt < nk; ++kt) {
    char* buf = smem + (kt & 1) * stage_b;
#pragma unroll
    for (int q = 0; q < LP; ++q) *(uint4*)(buf + dst[q]) = r[q];       // waits for the loads of tile kt
    if (kt + 1 < nk) {
#pragma unroll
      for (int q = 0; q < LP; ++q) r[q] = *(const uint4*)(src[q] + (size_t)(kt + 1) * 128);
    }
    __syncthreads();
    acc += *(const unsigned*)(buf + ((tid * 16) % stage_b));
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH>
float run_reg(const char* U, const char* W, int R, int BM, int D, int heads, size_t lds, unsigned* sink, int reps) {
  auto k = stream_reg_kernel<DEPTH>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k<<<R * heads, 512, lds>>>(U, W, BM, D, heads, sink);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) k<<<R * heads, 512, lds>>>(U, W, BM, D, heads, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / reps;
}

template <int SEG, int NW>
float run(const char* U, const char* W, int R, int BM, int D, int heads, int nstg, size_t lds, unsigned* sink, int reps) {
  auto k = stream_kernel<SEG, NW>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k<<<R * heads, 512, lds>>>(U, W, BM, D, heads, nstg, sink);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) k<<<R * heads, 512, lds>>>(U, W, BM, D, heads, nstg, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / reps;
}

int main() {
  const int D = 512, heads = 8;
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int R = cfg == 0 ? 64 : 256, BM = cfg == 0 ? 144 : 128;
    char *U, *W; unsigned* sink;
    hipMalloc(&U, (size_t)R * BM * D * 2); hipMalloc(&W, (size_t)3 * D * D * 2); hipMalloc(&sink, 64);
    hipMemset(U, 1, (size_t)R * BM * D * 2); hipMemset(W, 1, (size_t)3 * D * D * 2);
    const double mb = (double)R * heads * (BM + 192) * D * 2 / 1e6;
    printf("R=%d regions x %d heads, BM=%d: %.0f MB through LDS-DMA per launch\n", R, heads, BM, mb);
    const size_t s128 = (size_t)(BM + 192) * 128, s64 = (size_t)(BM + 192) * 64;
    struct { const char* name; int seg, nw, nstg; size_t lds; } v[] = {
      {"128B rows, 4 loaders, 3 stages, 1 block/CU (today)", 128, 4, 3, 3 * s128},
      {"128B rows, 8 loaders, 3 stages, 1 block/CU", 128, 8, 3, 3 * s128},
      {"128B rows, 8 loaders, 2 stages, 1 block/CU", 128, 8, 2, 100 * 1024},
      {" 64B rows, 8 loaders, 3 stages, 1 block/CU", 64, 8, 3, 100 * 1024},
      {" 64B rows, 8 loaders, 3 stages, 2 blocks/CU", 64, 8, 3, 3 * s64},
      {" 64B rows, 4 loaders, 3 stages, 2 blocks/CU", 64, 4, 3, 3 * s64},
      {"128B rows, 8 loaders, 2 stages, 2 blocks/CU (LDS 2x, if it fit)", 128, 8, 2, 2 * s128 > 80 * 1024 ? 80 * 1024 : 2 * s128},
    };
    for (auto& c : v) {
      float us;
      if (c.seg == 128 && c.nw == 4) us = run<128, 4>(U, W, R, BM, D, heads, c.nstg, c.lds, sink, 20);
      else if (c.seg == 128) us = run<128, 8>(U, W, R, BM, D, heads, c.nstg, c.lds, sink, 20);
      else if (c.nw == 8) us = run<64, 8>(U, W, R, BM, D, heads, c.nstg, c.lds, sink, 20);
      else us = run<64, 4>(U, W, R, BM, D, heads, c.nstg, c.lds, sink, 20);
      printf("  %-62s %7.1f us  %6.1f B/clk/CU (2.4 GHz, 256 CUs)\n", c.name, us, mb * 1e6 / (us * 1e-6) / 2.4e9 / 256);
    }
    {
      float us = run_reg<1>(U, W, R, BM, D, heads, 100 * 1024, sink, 20);
      printf("  %-62s %7.1f us  %6.1f B/clk/CU\n", "registers (global_load + ds_write), 1 tile ahead, 1 block/CU", us, mb * 1e6 / (us * 1e-6) / 2.4e9 / 256);
      us = run_reg<1>(U, W, R, BM, D, heads, 2 * s128, sink, 20);
      printf("  %-62s %7.1f us  %6.1f B/clk/CU\n", "registers, 1 tile ahead, 2 blocks/CU if they fit", us, mb * 1e6 / (us * 1e-6) / 2.4e9 / 256);
    }
    hipFree(U); hipFree(W); hipFree(sink);
  }
  return 0;
}
