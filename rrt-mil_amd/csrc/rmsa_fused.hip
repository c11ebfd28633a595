// rmsa_fused.hip -- R-MSA in ONE kernel per (region, head): qkv projection + EPEG + attention.
//
// Replaces InnerAttention.forward up to (not including) proj, modules/rmsa.py:100-122:
//     qkv = Linear(D, 3D)(U) ; q *= hd^-0.5 ; S = q k^T ; S += dwconv(S) ; softmax ; O = A v
// The unfused path (linear_f32.hip -> region_attn.hip) writes the [Np, 3D] qkv tensor to HBM and
// reads it back: 113 MB of traffic per bag at the north star, a store burst at the end of every
// GEMM tile (measured ~1.2K cycles to issue one store while all blocks write at once), the
// end-of-kernel L2 write-back of 56 MB, one more launch, and ~7K cycles per attention wave waiting
// for its first K/V DMA.  Here a block owns one region x one head:
//   phase 1  C[P x 192] = U_r[P x D] . W_h[192 x D]^T on the fp32 (or bf16/f16) matrix cores --
//            the same 16x16x4 MFMA / LDS-DMA / XOR-swizzle machinery as linear_ws_kernel
//            (2 loader waves + 4 compute waves, 144 x 192 tile, BK = 32, double buffered);
//   phase 2  the accumulators (+bias, q*scale) go to LDS as XOR-swizzled Q, K and V tiles,
//            aliasing the now dead staging buffers -- qkv never exists in HBM;
//   phase 3  EPEG as a sliding-window stencil over the Q tile: thread = (16-byte column slot, run of
//            6 consecutive query rows) reads 6 + k - 1 rows once instead of k rows per output
//            (the per-lane version was LDS-bandwidth bound: 8.4K cycles per wave), x log2(e);
//   phase 4  softmax(Q~ K^T) V per 16-query tile with every key tile of the region resident
//            (single pass, no online-softmax rescale), scores transposed exactly as in
//            region_attn.hip; only O [P x 64] is written.
// LDS: max(2 x 43 KB staging, Q/K/V 3 x 36.9 KB) = 108 KiB (Q~ overwrites Q) -> one block per CU, with
// 52 KiB left for a co-resident kernel of another bag; 512 blocks at N = 9000 = 2 per CU.
// Requires head dim 64 and P <= 16*MT <= 208 (MT = 13: 3 x 52 KB tiles = 156 KiB of the 160 KiB LDS).
//
// PROJ (round 4; fp32, inference, bags of >= two rounds of (region, head) items): the same launch also runs the layer's
// out-projection + region_reverse + un-pad + residual (modules/rmsa.py:131, :41-54, :227-228; rrt.py:125) -- block b does
// item b and then the 64-column projection slab b - lag of a region whose items finished a round earlier (proj_slab below;
// DESIGN.md section 3, "K2+K3+K4 in one launch").  Bit-identical to this kernel followed by linear_ws_kernel<.., UNPART>.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>

#include "internal.h"

namespace {

constexpr int BK = 32;
constexpr int HD = 64;
constexpr int BN = 3 * HD;          // q | k | v columns of one head
constexpr float NEG_BIG = -3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool WT>
__device__ __forceinline__ void store_o(f32x2* p, f32x2 v) {
  if constexpr (WT) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");   // (64 bits: no store-data hazard)
  else *p = v;
}
#define RRT_STORE_O(ptr, val) store_o<PROJ>((ptr), (val))   // (non-temporal stores measured: no gain)

// O leaves the block through an ordinary store -- or, when the out-projection runs as a later phase of OTHER blocks of
// the same launch (PROJ, below), through a write-through one (sc0 sc1: the line reaches memory, not just this XCD's L2;
// the eight XCDs' L2s are not coherent with each other inside a launch -- crmsa.hip's hand-over note)
template <bool WT>
__device__ __forceinline__ void store_o(f32x4* p, f32x4 v) {
  // (+ the wait states a > 64-bit VMEM store needs before a VALU write may reuse its data registers: the compiler's hazard
  // recogniser does not see through inline asm -- tools/experiments/inner_msa.hip found that the hard way)
  if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *p = v;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2 };

template <int PREC>
struct Frag8;
template <>
struct Frag8<PREC_BF16> {
  using type = bf16x8;
  static __device__ __forceinline__ type pack(float4 a, float4 b) {
    type r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
  }
  static __device__ __forceinline__ f32x4 mfma(type a, type b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Frag8<PREC_F16> {
  using type = f16x8;
  static __device__ __forceinline__ type pack(float4 a, float4 b) {
    type r;
    r[0] = (_Float16)a.x; r[1] = (_Float16)a.y; r[2] = (_Float16)a.z; r[3] = (_Float16)a.w;
    r[4] = (_Float16)b.x; r[5] = (_Float16)b.y; r[6] = (_Float16)b.z; r[7] = (_Float16)b.w;
    return r;
  }
  static __device__ __forceinline__ f32x4 mfma(type a, type b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate): n <= 31 here
__device__ __forceinline__ void wait_vmcnt(const int n) {
#define RRT_VMC(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
  switch (n) {
    RRT_VMC(1) RRT_VMC(2) RRT_VMC(3) RRT_VMC(4) RRT_VMC(5) RRT_VMC(6) RRT_VMC(7) RRT_VMC(8) RRT_VMC(9) RRT_VMC(10)
    RRT_VMC(11) RRT_VMC(12) RRT_VMC(13) RRT_VMC(14) RRT_VMC(15) RRT_VMC(16) RRT_VMC(17) RRT_VMC(18) RRT_VMC(19) RRT_VMC(20)
    RRT_VMC(21) RRT_VMC(22) RRT_VMC(23) RRT_VMC(24) RRT_VMC(25) RRT_VMC(26) RRT_VMC(27) RRT_VMC(28) RRT_VMC(29) RRT_VMC(30)
    RRT_VMC(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef RRT_VMC
}

// block -> (region, head) for n_items = n_regions * heads blocks (see the kernel's comment on the XCD-aware order)
__device__ __forceinline__ void map_item(const int b, const int n_items, const int heads, int& reg, int& head) {
  const int n_regions = n_items / heads;
  const int full = (n_regions >> 3) * 8 * heads;            // blocks of complete 8-region groups
  if (b < full) {
    const int xcd = b & 7, idx = b >> 3;
    const int grp = idx / heads;                             // group of 8 regions, one per XCD
    reg = grp * 8 + xcd;
    head = idx - grp * heads;
  } else {                                                   // ragged tail: plain order (bijective)
    const int rem = b - full;
    reg = (n_regions >> 3) * 8 + rem / heads;
    head = rem % heads;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// PROJ phase: one 64-column slab of one region's out-projection, by the whole block (after its own item, if any).
//     Y[P x 64] = O_r[P x D] . Wp[64 c .. 64 c + 63, :]^T + b ;  out[token(slot)] = resid[token(slot)] + Y[slot]
// (proj nn.Linear, region_reverse, un-pad, residual: modules/rmsa.py:131, :41-54, :227-228; rrt.py:125).
// Why here and not as the next launch: as a kernel of its own the projection's blocks move in lockstep -- every block of
// the chip waits for its first (cold) DMA round trip at the same moment and later bursts its un-partition epilogue at
// the same moment: 45 us for 31 us of MFMA.  As a phase of the fused launch's blocks the
// slabs drift apart, O_r is an item old (in this XCD's L2) and the launch ramp is paid once.
// Machinery: the projection loop's -- four loader waves feed a two-stage LDS-DMA ring of 128-byte rows (XOR-swizzled
// by the source address), four compute waves (one 16-column tile each, all MT row tiles) run v_mfma_f32_16x16x4_f32
// with the operand roles swapped; both halves' fragments are double-buffered in registers, the tile barrier sits
// between the halves (stage kt is in registers everywhere -> its buffer is free for stage kt + 2).  The K-sum order
// per accumulator (k tile, half, component) is linear_ws_kernel's: the result is bit-identical to the separate launch.
#ifdef RRT_TRACE
#define RRT_SLAB_TRACE_ARG , WaveTrace& _tr
#define RRT_SLAB_TRACE_PASS , _tr
#else
#define RRT_SLAB_TRACE_ARG
#define RRT_SLAB_TRACE_PASS
#endif
#ifndef RRT_PROJ_RQA
#define RRT_PROJ_RQA 6
#endif
template <int MT, int KM>     // KM: CR-MSA representatives the by-product has registers for (0: none; k <= KM)
__device__ __forceinline__ void proj_slab(const float* __restrict__ /*U*/, const float* __restrict__ O, const int n_rows,
                                          const int P, const int D, const int heads_rt, const FusedProj& pj,
                                          char* smem RRT_SLAB_TRACE_ARG) {
  constexpr int BM = 16 * MT;
  constexpr int SST = (BM + HD) * BK;              // floats per stage: A tile (BM rows) + B tile (64 rows of Wp)
  constexpr int NA = BM / 8, NB = HD / 8;          // 1 KiB DMA pieces per stage
  constexpr int LA = (NA + 3) / 4, LB = NB / 4;
  constexpr bool PREF = MT <= 9;                   // residual rows requested under the K loop (registers allow it)
  constexpr int LDS_MAIN_F = (2 * (BM + BN) * BK > 3 * BM * HD ? 2 * (BM + BN) * BK : 3 * BM * HD);   // the kernel's LDS, floats
  constexpr int NS = LDS_MAIN_F / SST >= 4 ? 4 : LDS_MAIN_F / SST;     // ring stages that fit it (MT = 6, 7: three)
  static_assert(NS >= 3 && NS * SST <= LDS_MAIN_F, "slab ring");
  const int b = (int)blockIdx.x;
  if (b < pj.lag) return;
  const int sidx = b - pj.lag;                     // < n_items by the grid size
  int reg, col;
  map_item(sidx, pj.n_items, heads_rt, reg, col);
  float* lds = (float*)smem;
  const unsigned lds_b = lds_addr_of(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int row0 = reg * P;
  const int nk = D / BK;
  RRT_TRACE_MARK();                                 // slab [1] entry
  // the region's `heads` items have arrived (their O rows are in memory): blocks with lower indices, dispatched earlier.
  // The wait rests on in-order workgroup dispatch (an item this slab waits for has a lower block index, so it is running
  // or done) -- an assumption about the dispatcher, not an architectural guarantee.  It is therefore BOUNDED: after
  // pj.spin_limit sleeps of ~0.2 us (product: 2^22, about a second -- five orders of magnitude over an item) the block
  // gives up, raises the process's hand-over error word (pinned host memory: the next C-ABI call that could launch this
  // kernel returns RRT_E_HANDOVER instead of queueing behind a wedged GPU) and leaves WITHOUT writing its slab.
  int* const s_abort = (int*)(smem + LDS_MAIN_F * 4 + 2048);      // (behind the tap / bias tables + the slab's gamma * phi table)
  if (tid == 0) {
    int spins = 0, bad = 0;
    while (__hip_atomic_load(pj.cnt + reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pj.wait_for) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > pj.spin_limit) { bad = 1; break; }
    }
    if (bad && pj.err != nullptr) __hip_atomic_store(pj.err, 1 + reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *s_abort = bad;
  }
  __syncthreads();
  if (*s_abort) return;                             // (block-uniform)
  RRT_TRACE_MARK();                                 // slab [2] region complete
  if (wave >= 4) {
    const int lw = wave - 4;
    unsigned aoff[LA], boff[LB];
#pragma unroll
    for (int qi = 0; qi < LA; ++qi) {
      const int S = (qi * 4 + lw) * 64 + lane;
      const int row = S >> 3, p = S & 7;
      // rows past the region (P < BM) re-read the region's LAST row (their accumulators are never stored): the slab has
      // waited for cnt[reg] only, so it must not touch O rows of region reg + 1 -- those may not be written yet, and a
      // line fetched early would sit stale in this XCD's L2 when that region's own slab (ragged tails: any XCD) reads it
      int gr = row < P ? row0 + row : row0 + P - 1;
      gr = gr < n_rows ? gr : n_rows - 1;
      aoff[qi] = (unsigned)gr * (unsigned)D * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int qi = 0; qi < LB; ++qi) {
      const int S = (qi * 4 + lw) * 64 + lane;
      const int row = S >> 3, p = S & 7;             // row in [0, 64): output column 64 col + row
      boff[qi] = (unsigned)(col * HD + row) * (unsigned)D * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
    auto stage = [&](int kt, unsigned buf) {
#pragma unroll
      for (int qi = 0; qi < LA; ++qi)
        if (qi * 4 + lw < NA) dma16s(O + kt * BK, aoff[qi], buf + (qi * 4 + lw) * 1024);
#pragma unroll
      for (int qi = 0; qi < LB; ++qi) dma16s(pj.Wp + kt * BK, boff[qi], buf + BM * BK * 4 + (qi * 4 + lw) * 1024);
    };
    // NS-stage ring: a slab's K tile is 72 MFMAs per wave (2.3 K cycles) -- less than a DMA round trip, and the block
    // is alone on its CU -- so the stages of the next NS - 1 K tiles are always in flight (with two stages every K tile
    // waited for its operands: the slab took as long as the separate launch's tile).  Barrier B_j = "stage j is in
    // registers everywhere, stage j + 1 has landed": stage j + NS goes into buffer j % NS right behind it.
    if constexpr (KM > 0) {
      // gamma[c] * phi[c, n] of this slab's 64 columns (zero for n >= k), for the compute waves' epilogue: [KM][64] floats
      // behind the ring (the tap / bias tables of the item phases, dead now), published by the barriers of the K loop
      float* const gpl = (float*)(smem + LDS_MAIN_F * 4);
      for (int e = tid - 256; e < 64 * KM; e += 256) {
        const int n = e >> 6, c = col * HD + (e & 63);
        gpl[e] = n < pj.k ? pj.ln_g[c] * pj.phi[(size_t)c * pj.k + n] : 0.f;
      }
    }
    const int mine = (NA - lw + 3) / 4 + LB;        // DMA pieces this wave issues per stage (vmcnt counts them)
    // Issuing a stage takes a wave ~1.1 K cycles (6 or 7 pieces that block it ~160 cycles each), so the ring is filled
    // as the loop goes: stages 0 and 1, K tile 0 published, stage 2, and behind barrier B_kt stage kt + 3 -- into buffer
    // (kt + 3) % NS, which B_kt (NS = 3) or B_kt-1 (NS = 4) freed.  Before B_kt the wave waits for stage kt + 1 with
    // stage kt + 2 still in flight (vmcnt counts this wave's own pieces).
    stage(0, lds_b);
    if (nk > 1) stage(1, lds_b + SST * 4);
    RRT_TRACE_MARK();                               // loader slab [3] first stages issued
    wait_vmcnt(nk > 1 ? mine : 0);
    RRT_TRACE_MARK();                               // loader slab [4] K tile 0 landed
    __syncthreads();                                // publishes K tile 0
    if (nk > 2) stage(2, lds_b + 2 * SST * 4);
    for (int kt = 0; kt < nk; ++kt) {
      wait_vmcnt(kt + 2 < nk ? mine : 0);           // K tile kt + 1 has landed
      if (kt == 0 || kt == 7 || kt == 14) RRT_TRACE_MARK();   // loader slab [5,6,7] K tile 1 / 8 / 15 landed
      __syncthreads();                              // B_kt
      if (kt + 3 < nk) stage(kt + 3, lds_b + ((kt + 3) % NS) * SST * 4);
    }
    return;
  }
  // ------------------------------------------------------------------ compute waves: column tile `wave`
  const int ncol = col * HD + wave * 16 + 4 * lg;   // this lane's four output columns
  // un-partition map of this lane's MT rows (region_reverse, rmsa.py:41-54: the region is known, one division per row);
  // computed where it is used (twice) rather than held in MT registers across the K loop
  const int tbase = [&] {
    const int ri = fdiv(reg, pj.g.rs, pj.g.inv_rs), rj = reg - ri * pj.g.rs;
    return ri * pj.g.s * pj.g.H + rj * pj.g.s;
  }();
  auto token_of = [&](const int i) {
    const int m = i * 16 + lr;
    const int pi = fdiv(m, pj.g.s, pj.g.inv_s), pjj = m - pi * pj.g.s;
    const int t = tbase + pi * pj.g.H + pjj;
    return (m < P && t < pj.g.L) ? t : -1;          // rows past the region / pad slots: nothing to write
  };
  // element offset of each row's four columns in resid / out, -1 = not written; MT registers held across the K loop
  // (recomputed at both uses the index arithmetic was ~2 K cycles of the slab's VALU time)
  int toff[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int t = token_of(i);
    toff[i] = t < 0 ? -1 : t * D + ncol;
  }
  float4 rq[PREF ? MT : 1];
  const float4 bias = pj.bias ? *(const float4*)(pj.bias + ncol) : make_float4(0.f, 0.f, 0.f, 0.f);
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 af[2][MT], bfr[2];
  auto frags = [&](const float* As, const int kk, float4 (&a)[MT], float4& bb) {
    {
      const int row = wave * 16 + lr;
      bb = *(const float4*)(As + BM * BK + row * BK + (((4 * kk + lg) ^ ((row >> 1) & 7)) << 2));
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = i * 16 + lr;
      a[i] = *(const float4*)(As + row * BK + (((4 * kk + lg) ^ ((row >> 1) & 7)) << 2));
    }
  };
  // one half K tile: 4 MT MFMAs on the fragments (a, bb) while the OTHER set's MT + 1 fragment reads (na, nb from
  // nAs / nkk; nAs == nullptr: none) are issued between them -- one read behind every third MFMA (a wave issues in
  // order: ten reads in a row drain the matrix pipe, ~150 cycles per half), the last one six or more MFMAs before the
  // half ends, so that the lgkmcnt(0) of the barrier behind a first half does not wait
  auto half = [&](const float4 (&a)[MT], const float4& bb, const bool load, const float* nAs, const int nkk,
                  float4 (&na)[MT], float4& nb) {
    if (load) frags(nAs, nkk, na, nb);
#pragma unroll
    for (int comp = 0; comp < 4; ++comp) {
      const float bv = comp == 0 ? bb.x : comp == 1 ? bb.y : comp == 2 ? bb.z : bb.w;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float av = comp == 0 ? a[i].x : comp == 1 ? a[i].y : comp == 2 ? a[i].z : a[i].w;
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i], 0, 0, 0);
      }
    }
    if (load) {
#pragma unroll
      for (int r = 0; r < MT + 1; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // three MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - 3 * (MT + 1), 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // (barriers of the compute side: LDS reads retired + s_barrier, nothing to do with vmcnt)
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  RRT_TRACE_MARK();                                 // slab [3] ready for K tile 0
  lds_barrier();                                    // K tile 0 published
  RRT_TRACE_MARK();                                 // slab [4] K tile 0 published
  frags(lds, 0, af[0], bfr[0]);
  __builtin_amdgcn_sched_barrier(0);
  auto tile = [&](const int kt) {                   // K tile kt < nk - 1: both halves, the next tile's first fragments
    half(af[0], bfr[0], true, lds + (kt % NS) * SST, 1, af[1], bfr[1]);   // first half; the second half's fragments under it
    lds_barrier();                                  // B_kt: stage kt is in registers everywhere, stage kt + 1 published
    if (kt == 0 || kt == 7 || kt == 14) RRT_TRACE_MARK();   // slab [5,6,7] B_0, B_7, B_14
    half(af[1], bfr[1], true, lds + ((kt + 1) % NS) * SST, 0, af[0], bfr[0]);
  };
  // The residual rows are requested under the last two K tiles (a request takes 3-4 K cycles to come back on a busy
  // chip, a K tile 2.5 K): the first RQA of them in front of the second-last tile, the rest in front of the last one --
  // as early as the register file of the item phases leaves room for (held from the start of the slab they cost the
  // kernel 8 VGPRs over what the item phases need -- and a co-resident kernel of another bag its place).
  // Unconditional (rows that are not written re-read row 0): loads under per-row branches left the compiler's
  // wait-count bookkeeping with "anything may be outstanding" at every store of the epilogue.
  constexpr int RQA = PREF ? (RRT_PROJ_RQA < MT ? RRT_PROJ_RQA : MT) : 0;
  for (int kt = 0; kt + 2 < nk; ++kt) tile(kt);
  if constexpr (PREF) {
#pragma unroll
    for (int i = 0; i < RQA; ++i) rq[i] = ld_row<NT_RESID>(pj.resid + (toff[i] < 0 ? ncol : toff[i]));
  }
  if (nk >= 2) tile(nk - 2);
  if constexpr (PREF) {
#pragma unroll
    for (int i = RQA; i < MT; ++i) rq[i] = ld_row<NT_RESID>(pj.resid + (toff[i] < 0 ? ncol : toff[i]));
  }
  half(af[0], bfr[0], true, lds + ((nk - 1) % NS) * SST, 1, af[1], bfr[1]);
  lds_barrier();                                    // (the loader side counts one barrier per K tile)
  half(af[1], bfr[1], false, nullptr, 0, af[0], bfr[0]);
  RRT_TRACE_MARK();                                 // slab [8] last MFMA issued
  // epilogue: + bias, un-partition, + residual (plain stores: the next launch reads them)
  if constexpr (PREF) {
    // every residual row is "consumed" here, in front of the first store: the compiler's waits for them are then
    // counted against loads only (long landed) -- placed at each row's use they count the stores issued in between as
    // well (vmcnt(9 - i) in front of store i: the last stores went out one at a time)
#pragma unroll
    for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(rq[i].x), "+v"(rq[i].y), "+v"(rq[i].z), "+v"(rq[i].w));
  }
  if constexpr (KM == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float4 q;
      if constexpr (PREF) q = rq[i];
      else q = ld_row<NT_RESID>(pj.resid + (toff[i] < 0 ? ncol : toff[i]));
      const float4 v = make_float4(acc[i][0] + bias.x + q.x, acc[i][1] + bias.y + q.y, acc[i][2] + bias.z + q.z, acc[i][3] + bias.w + q.w);
      if (toff[i] >= 0) *(float4*)(pj.out + toff[i]) = v;
    }
    RRT_TRACE_MARK();                               // slab [9] stores issued
  } else {
    // ---- CR-MSA's first pass as a by-product (round 5): this block holds x1[token(m), 64 col .. 64 col + 63] of every row
    // m of its region in registers -- the rows LayerNorm 2 and the logits <LN2(x1), phi_n> of modules/rmsa.py:303-307 are
    // about to re-read from memory.  Per row and slab it leaves a record  (mean_64, M2_64, d_0 .. d_k-1):  mean and centred
    // sum of squares of the row's 64 values here, and d_n = sum_c x1[c] * (gamma[c] phi[c, n]) -- merged over the D / 64
    // slabs of a row (Chan et al.: exact in real arithmetic, no E[x^2] - E[x]^2 cancellation) they give mean, rstd and
    //   logit_n = rstd * (sum_slabs d_n - mean * G_n) + B_n ,  G_n = sum_c gamma_c phi_cn ,  B_n = sum_c beta_c phi_cn
    // (crmsa_combine_parts_kernel, crmsa.hip).  Lane (lr, lg) of column tile `wave` holds 4 values of row 16 i + lr: it
    // writes (mean_4, M2_4, d_n over its 4 columns) to LDS -- the ring is dead: every fragment of the last K tile was in
    // registers before the last barrier -- and one thread per row merges the row's 16 four-column parts in column order.
    // (First form: lane-swap merges over lg in registers, (2 + k) x 2 v_permlane swaps per row with their wait states and
    // a run-time k: 8.4 K cycles per slab, traced; this form: ~2.5 K.)
    constexpr int NF = KM <= 2 ? 1 : KM <= 6 ? 2 : 3;   // float4s per part record: (m, q, d_0, d_1) (d_2 .. d_5) (d_6, d_7)
    float4* const rec = (float4*)lds;                  // [16 parts][BM rows][NF]
    const float* gpl = (const float*)(smem + LDS_MAIN_F * 4);      // [KM][64] gamma * phi of this slab's columns (loader waves)
    float4 gp[KM];
#pragma unroll
    for (int n = 0; n < KM; ++n) gp[n] = *(const float4*)(gpl + n * 64 + wave * 16 + 4 * lg);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float4 q;
      if constexpr (PREF) q = rq[i];
      else q = ld_row<NT_RESID>(pj.resid + (toff[i] < 0 ? ncol : toff[i]));
      const float4 v = make_float4(acc[i][0] + bias.x + q.x, acc[i][1] + bias.y + q.y, acc[i][2] + bias.z + q.z, acc[i][3] + bias.w + q.w);
      if (toff[i] >= 0) *(float4*)(pj.out + toff[i]) = v;
      const float m = ((v.x + v.y) + (v.z + v.w)) * 0.25f;
      const float a = v.x - m, b = v.y - m, c = v.z - m, d = v.w - m;
      float o[4 * NF];
      o[0] = m;
      o[1] = (a * a + b * b) + (c * c + d * d);
#pragma unroll
      for (int n = 0; n < 4 * NF - 2; ++n)
        o[2 + n] = n < KM ? (v.x * gp[n < KM ? n : 0].x + v.y * gp[n < KM ? n : 0].y) + (v.z * gp[n < KM ? n : 0].z + v.w * gp[n < KM ? n : 0].w) : 0.f;
      float4* const r = rec + ((size_t)(wave * 4 + lg) * BM + i * 16 + lr) * NF;
#pragma unroll
      for (int f = 0; f < NF; ++f) r[f] = make_float4(o[4 * f], o[4 * f + 1], o[4 * f + 2], o[4 * f + 3]);
    }
    RRT_TRACE_MARK();                               // slab [9] stores issued, part records in LDS
    lds_barrier();                                  // (the four compute waves: the loader waves have left)
    const int m_ = tid;                             // compute waves are threads 0 .. 255 >= BM rows
    if (m_ < P) {
      const int pi = fdiv(m_, pj.g.s, pj.g.inv_s), pjj = m_ - pi * pj.g.s;
      const int t = tbase + pi * pj.g.H + pjj;
      if (t < pj.g.L) {
        float o[4 * NF];
        {
          const float4* r0 = rec + (size_t)m_ * NF;
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const float4 x = r0[f];
            o[4 * f] = x.x; o[4 * f + 1] = x.y; o[4 * f + 2] = x.z; o[4 * f + 3] = x.w;
          }
        }
#pragma unroll
        for (int j = 1; j < 16; ++j) {                // Chan et al.: (4 j values) + (4 values), column order
          const float4* rj = rec + ((size_t)j * BM + m_) * NF;
          float x[4 * NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const float4 y = rj[f];
            x[4 * f] = y.x; x[4 * f + 1] = y.y; x[4 * f + 2] = y.z; x[4 * f + 3] = y.w;
          }
          const float dl = x[0] - o[0], w = 1.0f / (float)(j + 1);
          o[0] += dl * w;
          o[1] += x[1] + dl * dl * (4.0f * (float)j * w);
#pragma unroll
          for (int n = 2; n < 4 * NF; ++n) o[n] += x[n];
        }
        // the record leaves as float4s (stride: 2 + k rounded up to a multiple of 4 floats; the pad is never read)
        const int nf = (2 + pj.k + 3) >> 2;
        float4* dst = (float4*)(pj.part + ((size_t)t * pj.n_slabs + col) * (4 * nf));
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        if constexpr (NF > 1) if (nf > 1) dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        if constexpr (NF > 2) if (nf > 2) dst[2] = make_float4(o[8], o[9], o[10], o[11]);
      }
    }
    RRT_TRACE_MARK();                               // slab [10] row records stored
  }
}

template <int MT, int PREC, bool PROJ, int KM = 0>
__global__ __launch_bounds__(512, 1) void rmsa_fused_kernel(const float* __restrict__ U,
                                                            const float* __restrict__ Wqkv,
                                                            const float* __restrict__ bqkv,
                                                            const float* __restrict__ pe_w,
                                                            float* __restrict__ O, int n_rows, int P,
                                                            int D, int heads_rt, int epeg_k, float q_scale,
                                                            float* __restrict__ stash, FusedProj pj) {
  static_assert(!PROJ || PREC == PREC_F32, "the projection phase is fp32 only");
  constexpr int BM = 16 * MT;
  constexpr int STAGE = (BM + BN) * BK;            // floats per pipeline stage
  constexpr int TILE = BM * HD;                    // floats of one Q / K / V tile
  constexpr int NA = BM / 8, NB = BN / 8;          // DMA wave-instructions per A / B stage
  constexpr int LA = (NA + 3) / 4, LB = NB / 4;    // per loader wave (four of them)
  constexpr int NT = 3;                            // 16-column tiles per compute wave (4 x 48 = 192)
  constexpr int LDS_MAIN = (2 * STAGE > 3 * TILE ? 2 * STAGE : 3 * TILE) * 4;   // bytes: staging ring / Q, K, V tiles
  constexpr bool PIPE = PREC == PREC_F32 && MT <= 11;   // software-pipelined projection loop (phase 1)
  constexpr bool SPLIT_LAST = MT == 9;             // nine tiles on eight waves: the ninth is shared out (phase 4)
  constexpr int RUN = (BM * 16 + 511) / 512;       // query rows per stencil thread: 5 for BM = 144
  constexpr int TAP_OFF = 12 + RUN - 1;            // tap t lives at taps[t + TAP_OFF]; taps[12..] is 16-byte aligned
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = (float*)smem;
  float* Qs = lds;                                 // after phase 1 (aliases the staging ring)
  float* Ks = lds + TILE;
  float* Vs = lds + 2 * TILE;
  float* Qt = Qs;                                  // Q~ overwrites Q in place (phase 3)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const unsigned lds_b = lds_addr_of(lds);
#ifndef RRT_FUSED_PRIO
#define RRT_FUSED_PRIO 3
#endif
  // this launch's waves go ahead of a co-resident wave of another bag's kernel at the SIMD's issue arbiter: the small
  // kernels are latency-bound and take the slots the matrix pipe leaves (round 5, four bags in flight: 5363 / 5365 ->
  // 5369 / 5381 slides/s, tools/experiments/ab_lib.sh; -DRRT_FUSED_PRIO=0 rebuilds the other side)
  if (RRT_FUSED_PRIO > 0) __builtin_amdgcn_s_setprio(RRT_FUSED_PRIO);
  // XCD-aware block -> (region, head) map.  Hardware places block b on XCD b % 8 (speed heuristic
  // only): the 8 head-blocks of a region are given consecutive slots of ONE XCD, so the region's
  // U panel (P x D fp32 = 295 KB) is fetched from HBM once and served to the other 7 heads from
  // that XCD's L2 (the naive head-fastest order put the 8 heads on 8 different XCDs: 8x the reads).
  const int n_items = PROJ ? pj.n_items : (int)gridDim.x;
  auto item_of = [&](const int b, int& reg_, int& head_) { map_item(b, n_items, heads_rt, reg_, head_); };
  // PROJ: block b runs item b (b < n_items) and then the projection slab b - lag (b >= lag): the slab's region finished
  // its items a whole item (~150 K cycles) earlier, on blocks with lower indices -- dispatched before this one, so the
  // wait below is on work that is already running or done (no deadlock), and in practice never spins.  lag % 8 == 0
  // keeps slab (r, c) on the XCD whose L2 holds O_r.
  const bool has_item = !PROJ || (int)blockIdx.x < n_items;
  int head = 0, reg = 0;
  if (has_item) item_of(blockIdx.x, reg, head);
  if constexpr (PROJ) {
    if (pj.zero64 != nullptr && blockIdx.x == 0 && threadIdx.x < 64) pj.zero64[threadIdx.x] = 0;
  }
  const int row0 = reg * P;                        // first token row of this region
  const int nk = D / BK;
  RRT_TRACE_INIT(blockIdx.x * 8 + wave);
  if (has_item) {
  // EPEG taps as the stencil wants them -- log2(e) * (w[t] + [t == k/2]), zero outside [0, k) -- in a 128-entry LDS
  // table behind the tiles (index t + TAP_OFF).  (Fetching w[t] from global inside the stencil loop put one
  // dependent vector load on every source row: the trace showed 9.2K cycles for a phase with ~2K cycles of work.)
  float* const taps = (float*)(smem + LDS_MAIN);
  float* const bias_l = taps + 128;                // q | k | v bias of this head (192 floats), read in phase 2
  if (tid >= 128 && tid < 128 + BN) {
    const int n = tid - 128;
    bias_l[n] = bqkv ? bqkv[(n >> 6) * D + head * HD + (n & 63)] : 0.f;
  }
  if (tid < 128) {
    const int t = tid - TAP_OFF;
    float wt = (pe_w != nullptr && t >= 0 && t < epeg_k) ? pe_w[head * epeg_k + t] : 0.f;
    if (t == (epeg_k >> 1)) wt += 1.0f;
    taps[tid] = wt * LOG2E;
  }
  RRT_TRACE_MARK();                                 // [1] entry

  // ================================================================== phase 1: projection
  if (wave >= 4) {
    const int lw = wave - 4;
    unsigned aoff[LA], boff[LB];
#pragma unroll
    for (int qi = 0; qi < LA; ++qi) {
      int S = (qi * 4 + lw) * 64 + lane;
      int row = S >> 3, p = S & 7;
      int gr = row0 + row;
      gr = gr < n_rows ? gr : n_rows - 1;          // rows past the last region: re-read (never used)
      aoff[qi] = (unsigned)gr * (unsigned)D * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int qi = 0; qi < LB; ++qi) {
      int S = (qi * 4 + lw) * 64 + lane;
      int row = S >> 3, p = S & 7;                  // row in [0,192): c = row/64 picks q / k / v
      int wr = (row >> 6) * D + head * HD + (row & 63);
      boff[qi] = (unsigned)wr * (unsigned)D * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
    auto stage = [&](int kt, unsigned buf) {
#pragma unroll
      for (int qi = 0; qi < LA; ++qi)
        if (qi * 4 + lw < NA) dma16s(U + kt * BK, aoff[qi], buf + (qi * 4 + lw) * 1024);
#pragma unroll
      for (int qi = 0; qi < LB; ++qi) dma16s(Wqkv + kt * BK, boff[qi], buf + BM * BK * 4 + (qi * 4 + lw) * 1024);
    };
    stage(0, lds_b);
    if constexpr (PIPE) {
      // fp32: the compute waves pass barrier B_kt in the MIDDLE of k tile kt, once both halves of stage kt are in
      // their registers -- so stage kt + 1 is published (and its first fragments fetched) half a tile before it is
      // needed, and buffer kt & 1 is free for stage kt + 2 from that point on
      wait_vm0();
      __syncthreads();                              // publishes K tile 0
      if (nk > 1) stage(1, lds_b + STAGE * 4);
      for (int kt = 0; kt < nk; ++kt) {
        wait_vm0();                                 // K tile kt + 1 has landed
        __syncthreads();                            // B_kt
        if (kt + 2 < nk) stage(kt + 2, lds_b + (kt & 1) * STAGE * 4);
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        wait_vm0();
        __syncthreads();                            // publishes K tile kt
        if (kt + 1 < nk) stage(kt + 1, lds_b + ((kt + 1) & 1) * STAGE * 4);
      }
    }
  } else {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE) {
      // Software pipeline over half k tiles (16 columns = one float4 slot per lane group, 4 MFMA k steps): the
      // fragments of the next half are always in flight under the 108 MFMAs of the current one.  With one compute
      // wave per SIMD nothing else covers an LDS round trip: fetching a stage's first fragments right after its
      // barrier left the matrix pipe idle ~580 of every 7.5 K cycles (traced).
      // Register budget: the A fragments are single-buffered and refilled IN PLACE -- row tile i's fragment of the next
      // half is read right behind the 12 MFMAs that were the last users of the current one (a full half, ~3 K cycles,
      // before it is needed), the three B fragments alternate between two sets.  36 VGPRs less than two full fragment
      // sets: the kernel allocates 208 instead of 240, which leaves a SIMD room for a wave of ANOTHER kernel next to
      // its two (the other bag's LayerNorm / CR-MSA kernels no longer wait for a block to retire).
      float4 af[MT], bfr[2][NT];
      auto a_frag = [&](const float* As, int kk, int i) {
        const int row = i * 16 + lr;
        return *(const float4*)(As + row * BK + (((4 * kk + lg) ^ ((row >> 1) & 7)) << 2));
      };
      auto b_frag = [&](const float* As, int kk, int j) {
        const int row = wave * (16 * NT) + j * 16 + lr;
        return *(const float4*)(As + BM * BK + row * BK + (((4 * kk + lg) ^ ((row >> 1) & 7)) << 2));
      };
      // A step = one row tile of one half k tile = 12 MFMAs (k step outer, column tile inner: an accumulator comes round
      // every third MFMA) + one refill read, issued in the shadow of the step's last MFMA (a wave issues in order).
      // The refill lags ONE step: behind step i goes the next-half fragment of row tile i - 1, whose registers the
      // matrix pipe finished reading a step ago (refilling row tile i itself made the compiler rotate accumulator and
      // fragment registers and pay the MFMA-source hazard in s_nops: ~120 cycles per k tile).
      auto mm = [&](const int i, const float4 (&bc)[NT]) {
#pragma unroll
        for (int comp = 0; comp < 4; ++comp)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float a = comp == 0 ? af[i].x : comp == 1 ? af[i].y : comp == 2 ? af[i].z : af[i].w;
            const float b = comp == 0 ? bc[j].x : comp == 1 ? bc[j].y : comp == 2 ? bc[j].z : bc[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i][j], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      };
      // first half of a k tile (stage As; B set 0), refilling towards its second half (B set 1); owe = row tile MT - 1's
      // first-half fragment has not been read yet (it is owed by the previous second half)
      auto first_half = [&](const float* As, const bool owe) {
        mm(0, bfr[0]);
        if (owe) af[MT - 1] = a_frag(As, 0, MT - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 1; i < MT; ++i) {
          mm(i, bfr[0]);
          af[i - 1] = a_frag(As, 1, i - 1);
          if (i <= NT) bfr[1][i - 1] = b_frag(As, 1, i - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // second half up to the barrier: two steps; the first one carries the last read of the stage
      auto second_head = [&](const float* As) {
        mm(0, bfr[1]);
        af[MT - 1] = a_frag(As, 1, MT - 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(1, bfr[1]);
      };
      // ... and behind it, refilling towards the next k tile's first half (stage nAs)
      auto second_tail = [&](const float* nAs, const bool refill) {
        if (refill) {
          af[0] = a_frag(nAs, 0, 0);
          bfr[0][0] = b_frag(nAs, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 2; i < MT; ++i) {
          mm(i, bfr[1]);
          if (refill) {
            af[i - 1] = a_frag(nAs, 0, i - 1);
            if (i <= NT) bfr[0][i - 1] = b_frag(nAs, 0, i - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      static_assert(MT > NT, "the refill schedule needs more row tiles than column tiles");
      __syncthreads();                              // K tile 0 published
      RRT_TRACE_MARK();                             // [2]
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = a_frag(lds, 0, i);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[0][j] = b_frag(lds, 0, j);
      // Barrier B_kt (stage kt is in registers everywhere, stage kt + 1 published) sits two steps into the second half
      // of k tile kt: the last read of stage kt went out 12 MFMAs earlier, so its lgkmcnt(0) does not wait (directly
      // behind the first half it idled ~150 cycles per k tile) -- and the loop is rotated so that its back edge is right
      // behind the barrier, where the conservative lgkmcnt(0) the compiler puts at a loop header is free too.
      first_half(lds, false);
      second_head(lds);
      __syncthreads();                                            // B_0
      RRT_TRACE_MARK();                                           // [3] B_0
      for (int kt = 1; kt < nk; ++kt) {
        const float* As = lds + (kt & 1) * STAGE;
        __builtin_amdgcn_sched_barrier(0);
        second_tail(As, true);                                    // rest of k tile kt - 1
        first_half(As, true);
        second_head(As);
        __syncthreads();                                          // B_kt
        if (kt == 7) RRT_TRACE_MARK();                            // [4] B_7
      }
      second_tail(nullptr, false);                                // rest of the last k tile
    } else {
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      if (kt == 0 || kt == 1 || kt == 8) RRT_TRACE_MARK();   // [2,3,4] barrier kt passed
      const float* As = lds + (kt & 1) * STAGE;
      const float* Bs = As + BM * BK;
      if constexpr (PREC == PREC_F32) {              // MT = 13: no registers for a second fragment set
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          float4 af[MT], bf[NT];
          const int cslot = 4 * kk + lg;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            int row = wave * (16 * NT) + j * 16 + lr;
            bf[j] = *(const float4*)(Bs + row * BK + ((cslot ^ ((row >> 1) & 7)) << 2));
          }
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            int row = i * 16 + lr;
            af[i] = *(const float4*)(As + row * BK + ((cslot ^ ((row >> 1) & 7)) << 2));
          }
#pragma unroll
          for (int comp = 0; comp < 4; ++comp)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                const float a = comp == 0 ? af[i].x : comp == 1 ? af[i].y : comp == 2 ? af[i].z : af[i].w;
                const float b = comp == 0 ? bf[j].x : comp == 1 ? bf[j].y : comp == 2 ? bf[j].z : bf[j].w;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i][j], 0, 0, 0);
              }
        }
      } else {
        using F = Frag8<PREC>;
        typename F::type a8[MT], b8[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = wave * (16 * NT) + j * 16 + lr, f = (row >> 1) & 7;
          b8[j] = F::pack(*(const float4*)(Bs + row * BK + (((2 * lg) ^ f) << 2)),
                          *(const float4*)(Bs + row * BK + (((2 * lg + 1) ^ f) << 2)));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = i * 16 + lr, f = (row >> 1) & 7;
          a8[i] = F::pack(*(const float4*)(As + row * BK + (((2 * lg) ^ f) << 2)),
                          *(const float4*)(As + row * BK + (((2 * lg + 1) ^ f) << 2)));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = F::mfma(b8[j], a8[i], acc[i][j]);
      }
    }
    }
    RRT_TRACE_MARK();                               // [5] last projection MFMA issued
    // ================================================================ phase 2: Q / K / V tiles -> LDS
    __syncthreads();                                // every wave is done with the staging ring
    // transposed accumulators: reg r of lane (lr, lg) is C[m = 16i + lr][n = 48*wave + 16j + 4lg + r]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * (16 * NT) + j * 16 + 4 * lg;        // 0..191, multiple of 4
      const int c = n >> 6, d = n & 63;                         // q / k / v and the head-dim column
      const float4 b = *(const float4*)(bias_l + n);   // (a global load here sat on the critical path between two phases)
      const float sc = c == 0 ? q_scale : 1.0f;
      float* dstm = c == 0 ? Qs : (c == 1 ? Ks : Vs);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = i * 16 + lr;
        const int slot = (d >> 2) ^ (m & 15);     // all three tiles XOR-swizzled: conflict-free row-per-lane writes
        float4 v = make_float4((acc[i][j][0] + b.x) * sc, (acc[i][j][1] + b.y) * sc,
                               (acc[i][j][2] + b.z) * sc, (acc[i][j][3] + b.w) * sc);
        *(float4*)(dstm + m * HD + (slot << 2)) = v;
      }
    }
  }
  if (wave >= 4) __syncthreads();                   // loader side of the "staging ring is dead" barrier
  __syncthreads();                                  // Q / K / V tiles complete
  RRT_TRACE_MARK();                                 // [6] Q/K/V in LDS
  // Training forward (rrt_encoder_forward_train): the backward kernels read q (scaled, + bias), k and v rows from a
  // [rows, 3 D] stash -- written here from the three tiles (the only time qkv reaches HBM: 256-byte row segments,
  // fire and forget) instead of running the unfused projection + attention pair for the sake of that tensor.
  if (stash != nullptr) {
    for (int idx = tid; idx < BM * 16 * 3; idx += 512) {
      const int c = idx / (BM * 16), rem = idx - c * (BM * 16);
      const int m = rem >> 4, sl = rem & 15;
      if (m < P) {
        const float4 v = *(const float4*)(lds + c * TILE + m * HD + ((sl ^ (m & 15)) << 2));
        *(float4*)(stash + (size_t)(row0 + m) * (3 * D) + c * D + head * HD + 4 * sl) = v;
      }
    }
  }

  // ================================================================== phase 3: EPEG stencil -> Q~
  // thread = (slot s of 16, run g of RUN consecutive query rows); all eight waves take part
  {
    // In place: every thread first gathers its outputs in registers, a barrier retires all reads of
    // the Q tile, then Q~ is written over it.
    static_assert(RUN - 1 <= TAP_OFF && (TAP_OFF - (RUN - 1)) % 4 == 0 && 2 * RUN + 90 < 128, "tap table range / alignment");
    const int half = epeg_k >> 1;
    const int s = tid & 15, g = tid >> 4;
    const int r0 = g * RUN;
    f32x2 out[RUN][2];                              // packed pairs: v_pk_fma_f32 does two lanes' worth per issue slot
#pragma unroll
    for (int o = 0; o < RUN; ++o) out[o][0] = out[o][1] = (f32x2){0.f, 0.f};
    if (r0 < BM) {
      // source row j of the run (j = 0 .. RUN + k - 2, row r0 - k/2 + j) meets output o with tap t = j - o: the RUN
      // weights slide by one per source row.  Four rows per trip, their loads issued together (a row per trip was
      // a chain of dependent LDS round trips); rows outside the region [0, P) read as zero (zero padding), taps
      // outside [0, k) are zero in the table.
      const int nsrc = RUN + 2 * half;
      // taps tp[j0 - RUN + 1 .. j0 + 3] of a four-row group: one broadcast read (the address is the same in every
      // lane), then compile-time indices -- row u meets output o with T[u + RUN - 1 - o]
      constexpr int NTV = (RUN + 3 + 3) / 4;
      const float4* tp4 = (const float4*)(taps + TAP_OFF - (RUN - 1));
      // epeg_k <= 15 (every published configuration but the NSCLC one): ALL source rows of the run and every tap requested
      // at once -- one LDS round trip instead of five dependent four-row trips (round 5; traced 5.75 K -> 5.28 K cycles: the
      // phase is bound by its 200 v_pk_fma_f32 per thread -- 8 cycles each on this chip --, not by the trips)
      constexpr int NFAST = RUN + 14;
      if (nsrc <= NFAST) {
        constexpr int NTT = (NFAST + RUN - 1 + 3) / 4;
        float4 v[NFAST];
        float T[4 * NTT];
#pragma unroll
        for (int u = 0; u < NFAST; ++u) {
          const int rr = r0 - half + u;
          const bool ok = rr >= 0 && rr < P;
          const int rc = ok ? rr : 0;
          v[u] = *(const float4*)(Qs + rc * HD + ((s ^ (rc & 15)) << 2));
          if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NTT; ++q) {
          const float4 t4 = tp4[q];
          T[4 * q] = t4.x; T[4 * q + 1] = t4.y; T[4 * q + 2] = t4.z; T[4 * q + 3] = t4.w;
        }
#pragma unroll
        for (int u = 0; u < NFAST; ++u) {               // (rows u >= nsrc meet taps >= k: zero in the table)
          const f32x2 lo = {v[u].x, v[u].y}, hi = {v[u].z, v[u].w};
#pragma unroll
          for (int o = 0; o < RUN; ++o) {
            const float wt = T[u + RUN - 1 - o];
            const f32x2 w2 = {wt, wt};
            out[o][0] = __builtin_elementwise_fma(w2, lo, out[o][0]);
            out[o][1] = __builtin_elementwise_fma(w2, hi, out[o][1]);
          }
        }
      } else
      for (int j0 = 0; j0 < nsrc; j0 += 4) {
        float4 v[4];
        float T[4 * NTV];
#pragma unroll
        for (int q = 0; q < NTV; ++q) {
          const float4 t4 = tp4[(j0 >> 2) + q];
          T[4 * q] = t4.x; T[4 * q + 1] = t4.y; T[4 * q + 2] = t4.z; T[4 * q + 3] = t4.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = r0 - half + j0 + u;
          const bool ok = rr >= 0 && rr < P;
          const int rc = ok ? rr : 0;
          v[u] = *(const float4*)(Qs + rc * HD + ((s ^ (rc & 15)) << 2));
          if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x2 lo = {v[u].x, v[u].y}, hi = {v[u].z, v[u].w};
#pragma unroll
          for (int o = 0; o < RUN; ++o) {
            const float wt = T[u + RUN - 1 - o];
            const f32x2 w2 = {wt, wt};
            out[o][0] = __builtin_elementwise_fma(w2, lo, out[o][0]);
            out[o][1] = __builtin_elementwise_fma(w2, hi, out[o][1]);
          }
        }
      }
    }
    __syncthreads();                                // all reads of Q done
    if (r0 < BM) {
#pragma unroll
      for (int o = 0; o < RUN; ++o) {
        const int m = r0 + o;
        if (m < BM)
          *(float4*)(Qt + m * HD + ((s ^ (m & 15)) << 2)) = make_float4(out[o][0][0], out[o][0][1], out[o][1][0], out[o][1][1]);
      }
    }
  }
  __syncthreads();
  RRT_TRACE_MARK();                                 // [7] Q~ built

  // ================================================================== phase 4: attention from LDS
  // All eight waves, two per SIMD (wave w and w + 4), a 16-query tile each per pass: tile t -> wave t % 8.  One wave
  // per SIMD left every LDS round trip and the whole softmax exposed (traced: 13.5 K cycles per tile for 9.2 K of
  // MFMA); two waves with a tile each keep the matrix pipe ~100 % busy between them (18.2 K cycles for two tiles).
  if constexpr (SPLIT_LAST) {
    // MT = 8 + 1: eight whole tiles, one per wave, and the ninth SHARED OUT by key tile -- wave w also runs the
    // ninth tile's queries against key tile w (wave 0: and key tile 8), inside its own MFMA streams: the K and V
    // fragments are the ones its own tile needs anyway, so the extra costs 32 MFMAs and no LDS traffic.  The nine
    // partial (max, sum, O) are merged like an online softmax.  (As a separate stage -- four key quarters on four
    // waves, before or after the whole tiles -- the same work ran latency-bound: 8.8 K cycles for 2.3 K of MFMA.)
    // Wave w visits the key tiles in the order w, w + 1, ... (mod MT) so that "its" key tile is at a compile-time
    // position of the unrolled loops.
    constexpr int XT = MT - 1;
    // partial O of key tile w: over wave w's OWN Q~ rows (tile w of the Q~ image, dead once its fragments are in
    // registers -- nobody else reads them); key tile 8's and the (max, sum) pairs behind the tap / bias tables
    float* const pstat = (float*)(smem + LDS_MAIN + 1280);      // [MT][32]: max [16], sum [16]
    float* const plast = pstat + MT * 32;                       // [16][HD]
    const int i0 = wave * 16;
    int tk[MT];                                                  // key tile at position j (scalar registers)
#pragma unroll
    for (int j = 0; j < MT; ++j) tk[j] = j + wave < MT ? j + wave : j + wave - MT;
    float4 bq[4], bx[4];
    {
      const int m = i0 + lr, mx = XT * 16 + lr;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bq[c] = *(const float4*)(Qt + m * HD + (((4 * c + lg) ^ lr) << 2));
        bx[c] = *(const float4*)(Qt + mx * HD + (((4 * c + lg) ^ lr) << 2));
      }
    }
    f32x4 s[MT], sx = {0.f, 0.f, 0.f, 0.f}, sy = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MT; ++j) s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 a[MT];
#pragma unroll
      for (int j = 0; j < MT; ++j) a[j] = *(const float4*)(Ks + (tk[j] * 16 + lr) * HD + (((4 * c + lg) ^ lr) << 2));
#pragma unroll
      for (int comp = 0; comp < 4; ++comp) {
        const float qb = comp == 0 ? bq[c].x : comp == 1 ? bq[c].y : comp == 2 ? bq[c].z : bq[c].w;
        const float xb = comp == 0 ? bx[c].x : comp == 1 ? bx[c].y : comp == 2 ? bx[c].z : bx[c].w;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const float ka = comp == 0 ? a[j].x : comp == 1 ? a[j].y : comp == 2 ? a[j].z : a[j].w;
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka, qb, s[j], 0, 0, 0);
          if (j == 0) sx = __builtin_amdgcn_mfma_f32_16x16x4f32(ka, xb, sx, 0, 0, 0);
        }
      }
    }
    if (wave == 0) {                                             // key tile 8 of the shared-out tile
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 a8 = *(const float4*)(Ks + (XT * 16 + lr) * HD + (((4 * c + lg) ^ lr) << 2));
        sy = __builtin_amdgcn_mfma_f32_16x16x4f32(a8.x, bx[c].x, sy, 0, 0, 0);
        sy = __builtin_amdgcn_mfma_f32_16x16x4f32(a8.y, bx[c].y, sy, 0, 0, 0);
        sy = __builtin_amdgcn_mfma_f32_16x16x4f32(a8.z, bx[c].z, sy, 0, 0, 0);
        sy = __builtin_amdgcn_mfma_f32_16x16x4f32(a8.w, bx[c].w, sy, 0, 0, 0);
      }
    }
    RRT_TRACE_MARK();                               // tile: S^T issued
    // fp32 MFMA and the vector ALU are the same lanes on this chip (measured, tools/ubench/mfma_valu_overlap.hip: no
    // overlap, not even across waves), so every VALU instruction here is time taken from the matrix work: packed
    // sub / add / mul, key masks only when the region does not fill its tiles.
    float cmax = NEG_BIG, xmax = NEG_BIG, ymax = NEG_BIG;
    if (P < BM) {
#pragma unroll
      for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (tk[j] * 16 + 4 * lg + r >= P) s[j][r] = NEG_BIG;   // keys past the region
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (tk[0] * 16 + 4 * lg + r >= P) sx[r] = NEG_BIG;
        if (XT * 16 + 4 * lg + r >= P) sy[r] = NEG_BIG;
      }
    }
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cmax = fmaxf(cmax, s[j][r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xmax = fmaxf(xmax, sx[r]);
      ymax = fmaxf(ymax, sy[r]);
    }
    cmax = max_xor32(max_xor16(cmax));
    xmax = max_xor32(max_xor16(xmax));
    ymax = max_xor32(max_xor16(ymax));
    float a4[4] = {0.f, 0.f, 0.f, 0.f};              // four independent partial sums (plain scalar ops: the packed
#pragma unroll                                       // forms cost more in register moves than they saved)
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[j][r] - cmax);
        s[j][r] = e;
        a4[r] += e;
      }
    float psum = (a4[0] + a4[1]) + (a4[2] + a4[3]), xsum = 0.f, ysum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sx[r] = __builtin_amdgcn_exp2f(sx[r] - xmax);
      sy[r] = __builtin_amdgcn_exp2f(sy[r] - ymax);
      xsum += sx[r];
      ysum += sy[r];
    }
    psum = sum_xor32(sum_xor16(psum));
    xsum = sum_xor32(sum_xor16(xsum));
    ysum = sum_xor32(sum_xor16(ysum));
    // 1 / sum of query 4 lg + r for the output registers: a general lane permute (ds_bpermute), issued here and
    // consumed after the P.V products -- its trip through the LDS queue is hidden behind them
    float ir[4];
    {
      const float inv = 1.0f / psum;
#pragma unroll
      for (int r = 0; r < 4; ++r) ir[r] = __shfl(inv, 4 * lg + r);
    }
    RRT_TRACE_MARK();                               // tile: softmax done
    f32x4 oacc[4], ox[4], oy[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) oacc[c] = ox[c] = oy[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      // V rows PD groups ahead of their four MFMAs (the compiler's own schedule fetched one group = 128 cycles ahead,
      // less than an LDS round trip with eight waves reading)
      constexpr int G = 4 * MT, PD = 5;
      float4 vq[G];
      auto vload = [&](int g) {
        const int rr = 4 * lg + (g & 3);
        return *(const float4*)(Vs + (tk[g >> 2] * 16 + rr) * HD + ((lr ^ rr) << 2));
      };
#pragma unroll
      for (int g = 0; g < PD; ++g) vq[g] = vload(g);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + PD < G) vq[g + PD] = vload(g + PD);
        const float4 v = vq[g];
        const float p = s[g >> 2][g & 3];
        oacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.x, oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.y, oacc[1], 0, 0, 0);
        oacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.z, oacc[2], 0, 0, 0);
        oacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.w, oacc[3], 0, 0, 0);
        if (g < 4) {                                 // position 0 = key tile `wave`: the shared-out tile's partial
          const float px = sx[g];
          ox[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(px, v.x, ox[0], 0, 0, 0);
          ox[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(px, v.y, ox[1], 0, 0, 0);
          ox[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(px, v.z, ox[2], 0, 0, 0);
          ox[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(px, v.w, ox[3], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * lg + r;
        const float4 v = *(const float4*)(Vs + (XT * 16 + rr) * HD + ((lr ^ rr) << 2));
        const float py = sy[r];
        oy[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(py, v.x, oy[0], 0, 0, 0);
        oy[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(py, v.y, oy[1], 0, 0, 0);
        oy[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(py, v.z, oy[2], 0, 0, 0);
        oy[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(py, v.w, oy[3], 0, 0, 0);
      }
    }
    RRT_TRACE_MARK();                               // tile: PV issued
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * lg + r;
      if (i < P)
        RRT_STORE_O((f32x4*)(O + (size_t)(row0 + i) * D + head * HD + (lr << 2)),
                    ((f32x4){oacc[0][r] * ir[r], oacc[1][r] * ir[r], oacc[2][r] * ir[r], oacc[3][r] * ir[r]}));
    }
    // partials of the shared-out tile: O [query 4 lg + r][d = 4 lr + c] unnormalised, the query's max and sum
    {
      float* mine = Qt + wave * 16 * HD;            // slot = key tile
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *(float4*)(mine + (4 * lg + r) * HD + 4 * lr) = make_float4(ox[0][r], ox[1][r], ox[2][r], ox[3][r]);
      if (lg == 0) {
        pstat[wave * 32 + lr] = xmax;
        pstat[wave * 32 + 16 + lr] = xsum;
      }
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *(float4*)(plast + (4 * lg + r) * HD + 4 * lr) = make_float4(oy[0][r], oy[1][r], oy[2][r], oy[3][r]);
        if (lg == 0) {
          pstat[XT * 32 + lr] = ymax;
          pstat[XT * 32 + 16 + lr] = ysum;
        }
      }
    }
    RRT_TRACE_MARK();                               // tile: O and partials stored
    __syncthreads();
    RRT_TRACE_MARK();                               // partials published
    {
      // merge: thread = (query q, two columns); 512 threads cover the 16 x 64 tile
      const int q = tid >> 5, col = (tid & 31) * 2;
      float mw[MT], M = NEG_BIG;
#pragma unroll
      for (int w = 0; w < MT; ++w) {
        mw[w] = pstat[w * 32 + q];
        M = fmaxf(M, mw[w]);
      }
      float L = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
      for (int w = 0; w < MT; ++w) {
        const float sc = __builtin_amdgcn_exp2f(mw[w] - M);
        L += sc * pstat[w * 32 + 16 + q];
        const float2 a = *(const float2*)((w < XT ? Qt + w * 16 * HD : plast) + q * HD + col);
        o0 += sc * a.x;
        o1 += sc * a.y;
      }
      const float invL = 1.0f / L;
      if (XT * 16 + q < P)
        RRT_STORE_O((f32x2*)(O + (size_t)(row0 + XT * 16 + q) * D + head * HD + col), ((f32x2){o0 * invL, o1 * invL}));
      RRT_TRACE_MARK();                             // merged tile stored
    }
  } else {
  for (int t = wave; t < MT; t += 8) {
    const int i0 = t * 16;
    if (i0 >= P) break;
    float4 bq[4];
    {
      const int m = i0 + lr;
#pragma unroll
      for (int c = 0; c < 4; ++c) bq[c] = *(const float4*)(Qt + m * HD + (((4 * c + lg) ^ (m & 15)) << 2));
    }
    f32x4 s[MT];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 a[MT];
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) {
        const int row = jt * 16 + lr;
        a[jt] = *(const float4*)(Ks + row * HD + (((4 * c + lg) ^ (row & 15)) << 2));
      }
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].x, bq[c].x, s[jt], 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].y, bq[c].y, s[jt], 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].z, bq[c].z, s[jt], 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].w, bq[c].w, s[jt], 0, 0, 0);
    }
    RRT_TRACE_MARK();                               // tile: S^T issued
    float cmax = NEG_BIG;
    if (P < BM) {
#pragma unroll
      for (int jt = 0; jt < MT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (jt * 16 + 4 * lg + r >= P) s[jt][r] = NEG_BIG;     // keys past the region
    }
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) cmax = fmaxf(cmax, s[jt][r]);
    cmax = max_xor32(max_xor16(cmax));
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[jt][r] - cmax);
        s[jt][r] = e;
        a4[r] += e;
      }
    const float psum = sum_xor32(sum_xor16((a4[0] + a4[1]) + (a4[2] + a4[3])));
    float ir[4];                                     // 1 / sum of query 4 lg + r: permute issued here, consumed after P.V
    {
      const float inv = 1.0f / psum;
#pragma unroll
      for (int r = 0; r < 4; ++r) ir[r] = __shfl(inv, 4 * lg + r);
    }
    RRT_TRACE_MARK();                               // tile: softmax done
    f32x4 oacc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = jt * 16 + 4 * lg + r;
        const float4 v = *(const float4*)(Vs + row * HD + ((lr ^ (row & 15)) << 2));
        const float p = s[jt][r];
        oacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.x, oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.y, oacc[1], 0, 0, 0);
        oacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.z, oacc[2], 0, 0, 0);
        oacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.w, oacc[3], 0, 0, 0);
      }
    RRT_TRACE_MARK();                               // tile: PV issued
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * lg + r;
      if (i < P)
        RRT_STORE_O((f32x4*)(O + (size_t)(row0 + i) * D + head * HD + (lr << 2)),
                    ((f32x4){oacc[0][r] * ir[r], oacc[1][r] * ir[r], oacc[2][r] * ir[r], oacc[3][r] * ir[r]}));
    }
    RRT_TRACE_MARK();                               // tile: O stored
  }
  }
  // ---------------------------------------------------------------- PROJ: this item has arrived
  if constexpr (PROJ) {
    wait_vm0();                                     // this thread's write-through stores of O are in memory ...
    __syncthreads();                                // ... and everybody's; the tiles in LDS are dead
    RRT_TRACE_MARK();                               // item: O in memory
    if (tid == 0) __hip_atomic_fetch_add(pj.cnt + reg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }   // has_item
  if constexpr (PROJ) proj_slab<MT, KM>(U, O, n_rows, P, D, heads_rt, pj, smem RRT_SLAB_TRACE_PASS);
}

// Blocks between an item and the slab of the same index: one block per CU runs at a time (LDS), so a lag of one
// "round" (the CU count, a multiple of 8 so that a slab stays on its region's XCD) puts a slab a whole item behind the
// items it waits for.  Never more than the items themselves (the launch is n_items + lag blocks).
int proj_lag(int n_items) {
  static int cus[64] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  d &= 63;
  if (cus[d] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    cus[d] = n;
  }
  int lag = cus[d] & ~7;
  if (lag > n_items) lag = n_items & ~7;
  // tuning build: RRT_PROJ_LAG=n (e.g. 250: every slab then runs on ANOTHER XCD than the items it reads -- the test that the
  // hand-over does not depend on the placement heuristic)
  static const char* force = rrt_tune_env("RRT_PROJ_LAG");
  if (force != nullptr && atoi(force) > 0) lag = atoi(force);
  return lag;
}

template <int MT, int PREC>
hipError_t launch_mt(const float* U, const float* Wqkv, const float* bqkv, const float* pe_w, float* O,
                     int n_regions, int P, int D, int heads, int epeg_k, float* stash, hipStream_t st,
                     const FusedProj* proj) {
  constexpr int BM = 16 * MT;
  constexpr size_t STG = (size_t)2 * (BM + BN) * BK * 4, QKV = (size_t)3 * BM * HD * 4;
  // staging ring / Q, K, V; tap table (512 B) + bias (768 B); shared-out tile (MT = 9): (max, sum) pairs + one partial O
  // (PROJ: the slab's gamma * phi table [8][64] + its abort flag use the first 2052 bytes behind the ring)
  constexpr size_t LDS = (STG > QKV ? STG : QKV) + (MT == 9 ? 1280 + MT * 32 * 4 + 16 * HD * 4 : 2064);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  const float q_scale = 1.0f / sqrtf((float)HD);
  if constexpr (PREC == PREC_F32) {
    if (proj != nullptr) {
      FusedProj pj = *proj;
      pj.n_items = heads * n_regions;
      if (pj.lag <= 0) pj.lag = proj_lag(pj.n_items);
      if (pj.wait_for <= 0) pj.wait_for = heads;
      pj.n_slabs = D / HD;
      if (pj.part != nullptr && (pj.ln_g == nullptr || pj.phi == nullptr || pj.k < 1 || pj.k > RRT_MAX_CRMSA_K))
        return hipErrorInvalidValue;
      if (pj.spin_limit <= 0) pj.spin_limit = 1 << 22;
      if (pj.err == nullptr) pj.err = handover_err_device();
#define RRT_LAUNCH_PROJ(KM_)                                                                                    \
  do {                                                                                                           \
    auto kern = rmsa_fused_kernel<MT, PREC_F32, true, KM_>;                                                      \
    static OncePerDevice once;                                                                                   \
    if (once.first())                                                                                            \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);        \
    kern<<<dim3(pj.n_items + pj.lag), dim3(512), LDS, st>>>(U, Wqkv, bqkv, pe_w, O, n_regions * P, P, D, heads,  \
                                                            pe_w ? epeg_k : 0, q_scale, nullptr, pj);            \
  } while (0)
      if constexpr (MT >= 6) {                       // (the merged launch starts at regions of > 64 tokens)
        if (pj.part != nullptr && pj.k <= 4) RRT_LAUNCH_PROJ(4);
        else if (pj.part != nullptr) RRT_LAUNCH_PROJ(8);
        else RRT_LAUNCH_PROJ(0);
      } else {
        if (pj.part != nullptr) return hipErrorInvalidValue;
        RRT_LAUNCH_PROJ(0);
      }
#undef RRT_LAUNCH_PROJ
      return hipGetLastError();
    }
  }
  auto kern = rmsa_fused_kernel<MT, PREC, false>;
  static OncePerDevice once;
  if (once.first())
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
  kern<<<dim3(heads * n_regions), dim3(512), LDS, st>>>(U, Wqkv, bqkv, pe_w, O, n_regions * P, P, D, heads,
                                                      pe_w ? epeg_k : 0, q_scale, stash, FusedProj{});
  return hipGetLastError();
}

}  // namespace

#ifdef RRT_TRACE
RRT_TRACE_DEFINE_READER(rrt_debug_trace_fused)
#endif

// ---- hand-over error word: 64 bytes of pinned host memory mapped into the device's address space, one per process.
// A slab that gives up its bounded wait stores into it with system scope; the host reads it without any device sync.
namespace {
// (published once through std::call_once: a second host thread launching its first merged forward at the same moment
//  either runs the initialiser or waits for it, and then sees both pointers)
int* g_err_host = nullptr;
int* g_err_dev = nullptr;
std::once_flag g_err_once;
void handover_err_init() {
  void* h = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && h != nullptr) {
    memset(h, 0, 64);
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d != nullptr) {
      g_err_dev = (int*)d;
      __atomic_store_n(&g_err_host, (int*)h, __ATOMIC_RELEASE);   // (handover_err_peek reads it without the once-flag)
    }
  } else {
    (void)hipGetLastError();     // no pinned memory: the wait stays bounded, the error is just not reported
  }
}
}  // namespace
int* handover_err_device() {
  std::call_once(g_err_once, handover_err_init);
  return g_err_dev;
}
int handover_err_peek(bool clear) {
  int* const eh = __atomic_load_n(&g_err_host, __ATOMIC_ACQUIRE);
  if (eh == nullptr) return 0;
  const int v = __atomic_load_n(eh, __ATOMIC_RELAXED);
  if (clear && v != 0) __atomic_store_n(eh, 0, __ATOMIC_RELAXED);
  return v;
}

bool rmsa_fused_supported(int P, int D, int heads, int epeg_k) {
  static const bool off = rrt_tune_env("RRT_NO_FUSED") != nullptr;
  if (off) return false;
  // one block holds a whole region: Q, K and V tiles of 16*MT rows in LDS -> P <= 208; MT in
  // {4, 6, 7, 8, 9, 11, 13} covers the region sizes s*s, s = 7..14, of bags of ~2.3k..12.5k tokens at
  // region_num = 8 (4x that at 16); smaller / larger regions take the unfused kernels
  return heads > 0 && D == heads * HD && D % BK == 0 && P > 48 && P <= 208 && epeg_k >= 0 && epeg_k <= 63;
}

bool rmsa_fused_supported_rows(long n_rows, int D) {   // 32-bit DMA byte offsets into U
  return n_rows * (long)D * 4 < 4000000000L;
}

// The projection as a phase of the fused launch pays when the launch has at least two rounds of items (a slab then
// runs a whole item behind the blocks it waits for, and the last round of slabs is as wide as the chip) -- and the wait
// is deadlock-free when every item a slab waits for has a LOWER block index than the slab's block: a region's items
// span 8 * heads consecutive indices, lag >= that.  fp32 exact arithmetic only.
bool rmsa_fused_proj_supported(int n_regions, int P, int D, int heads, int epeg_k, int prec) {
  static const bool off = rrt_tune_env("RRT_NO_FUSED_PROJ") != nullptr;
  if (off || prec != PREC_F32 || !rmsa_fused_supported(P, D, heads, epeg_k)) return false;
  const int n_items = n_regions * heads, lag = proj_lag(n_items);
  // regions of <= 64 tokens (MT = 4: bags of <= ~4.1 k tokens at region_num = 8) keep the two launches: their items are
  // short (the slab is a third of a block's life, not a quarter) and the separate projection's small tiles fill the chip
  // well -- measured 0.125 vs 0.122 ms per bag (profiles/r04_final_sweep_n_merged_vs_pair.txt); from 81 tokens on the phase wins
  return D % 4 == 0 && P > 64 && n_items >= 2 * lag && lag >= 8 * heads;
}

hipError_t launch_rmsa_fused(const float* U, const float* Wqkv, const float* bqkv, const float* pe_w,
                             float* O, int n_regions, int P, int D, int heads, int epeg_k, int prec,
                             hipStream_t st, float* stash, const FusedProj* proj) {
  if (proj != nullptr && (stash != nullptr || !rmsa_fused_proj_supported(n_regions, P, D, heads, epeg_k, prec)))
    return hipErrorInvalidValue;
  // a caller-chosen lag (rrt_debug_rmsa_fused_proj_f32: slabs on other XCDs than their items) keeps the deadlock-freedom
  // rule -- every item a slab waits for has a lower block index -- and the "never more blocks than two per item" bound
  if (proj != nullptr && proj->lag > 0 && (proj->lag < 8 * heads || proj->lag > n_regions * heads)) return hipErrorInvalidValue;
#define RRT_FUSED(MT_)                                                                              \
  switch (prec) {                                                                                   \
    case 1: return launch_mt<MT_, PREC_BF16>(U, Wqkv, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, stash, st, proj); \
    case 2: return launch_mt<MT_, PREC_F16>(U, Wqkv, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, stash, st, proj);  \
    default: return launch_mt<MT_, PREC_F32>(U, Wqkv, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, stash, st, proj); \
  }
  if (P > 176) { RRT_FUSED(13) }
  if (P > 144) { RRT_FUSED(11) }
  if (P > 128) { RRT_FUSED(9) }
  if (P > 112) { RRT_FUSED(8) }
  if (P > 96) { RRT_FUSED(7) }
  if (P > 64) { RRT_FUSED(6) }
  RRT_FUSED(4)
#undef RRT_FUSED
}
