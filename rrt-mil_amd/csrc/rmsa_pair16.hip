// rmsa_pair16.hip -- the fused 16-bit R-MSA core with TWO regions x one head per block (round 3).
//
// Same arithmetic interface as rmsa_fused16.hip (modules/rmsa.py:100-122 under the reference's --amp path): 16-bit U and
// W in, 16-bit O out, qkv / scores / probabilities never leave the CU.  What changed is what bounds it.  Measured
// (tools/ubench/dma_rows.hip = rmsa_fused16's projection loop with the MFMAs removed): streaming a block's operands --
// the region's U panel and the head's W slice, 344 KB at P = 144 -- through the LDS-DMA path takes 20 of the kernel's
// 28 us; a CU gets ~14 B/clk with four issuing waves, ~21 with eight, ~30 at best.  So:
//   * one block owns a PAIR of regions and one head: the W slice (57 % of the bytes) is staged once for both regions'
//     rows -- 245 KB per (region, head) instead of 344;
//   * all eight waves issue DMA pieces AND multiply: waves 0-3 own region A's rows, waves 4-7 region B's, each wave one
//     16-column tile of Q, of K and of V as before (no idle loader waves; the 2-stage ring of (2 BM + 192) x 128 B is as
//     deep as a 160 KiB LDS allows);
//   * the EPEG stencil runs on the matrix cores.  Along the query axis it is a banded Toeplitz product
//     Q~[q, d] = sum_src T[q, src] Q[src, d],  T[q, src] = w_h[src - q + k/2] + [src == q]:  a 16-query tile meets its
//     16 + 2 (k/2) source rows in one to three 32-wide MFMA steps.  Q leaves the projection TRANSPOSED (Q^T [d][token],
//     16-bit, zero halo columns = the convolution's zero padding), so the A operand is one ds_read_b128; the B operand
//     -- the band of T -- is the same for every tile and sits in registers as a (hi, lo) pair of 16-bit values (two
//     MFMAs per step: the taps keep 16 significant bits); the result lands as four consecutive head-dim columns of a
//     query = the row-major Q~ tile the score product reads.  5.6 K cycles of fp32 VALU work -> < 1 K.
// Rounding points (oracle: forward_f64(lowp=LowP(.., attn=True, stencil16=True))): U, W; Q log2(e) hd^-0.5 (BEFORE the
// stencil: one rounding more than rmsa_fused16, which ran the stencil in fp32), Q~, K, V, exp2(S - max), O.
// Regions: any P <= 176 (MT <= 11 row tiles), head dim 64, at least 8 regions; others take rmsa_fused16.
//
// PROJ (round 6; sixteen-wave form, an even number of row tiles, bags of >= two rounds of (pair, head) items -- BASELINE
// configs[3]: N = 30000 at region_num = 16 is 1024 items, four rounds of the chip): the same launch CAN also run the layer's
// out-projection + region_reverse + un-pad + residual (modules/rmsa.py:131, :41-54, :227-228; rrt.py:125), as
// rmsa_fused_kernel<.., PROJ> does in fp32: block b runs item b and then the 64-column projection SLAB b - lag of a pair
// whose eight head items finished a round earlier (pair_slab below).  Arithmetic and summation order of a slab are
// linear_ws_kernel<.., IN16>'s (K tiles ascending, two 32-wide halves each; (acc + bias) + residual): bit-identical to the
// two launches (tests/test_hip_parity.py::test_rmsa_pair16_proj).  MEASURED, AND NOT USED BY THE FORWARD: 109 us (two-stage
// slab ring) / 116 us (three-stage) against 64.5 + 35.6 us for the two launches at N = 30000.  The in-kernel timeline
// (profiles/r06_trace_pair16_proj.txt): an item is 39.2 K cycles, a slab adds 25 K -- 1.1 K waiting for the counter, 6.9 K
// until its first two stages are issued (cold code, sixteen waves' first DMA pieces), 13 K of K loop (1.4 K per K tile:
// the DMA-issue bound of sixteen issuing waves -- the separate projection's bound too), 2.8 K of epilogue -- and the item
// itself waits 4 K for its write-through O stores before it may count as arrived.  In fp32 the merged launch won because
// the separate projection ran in lockstep at 2/3 of its MFMA bound; the 16-bit projection already runs at its (DMA-issue)
// bound with two blocks per CU covering each other's prologue and epilogue, and a slab inside a block that owns its CU has
// nothing to overlap with.  The code stays as the measured answer to "merge the 16-bit launches" (round-5 review, item 1a).
#include <stdio.h>
#include <stdlib.h>

#include "fused16.h"

namespace {
using namespace f16k;

// write-through 16-byte store (sc0 sc1: the line reaches memory, not just this XCD's L2 -- the slab that reads it may in
// principle run on another XCD; + the wait states a > 64-bit inline-asm store needs before its data registers are reused)
typedef unsigned p16_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt16(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
  const p16_u32x4 v = {a, b, c, d};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// PROJ phase: one 64-column slab of one region PAIR's out-projection, by the whole block (sixteen waves):
//     Y[2P x 64] = O16[rows of the pair, D] . Wp16[64 c .. 64 c + 63, :]^T ;  out[token] = resid[token] + (Y + b)
// All sixteen waves issue DMA pieces and multiply (as in the item's projection phase): wave (rg, cw) owns the row tiles
// 4 rg' .. of its quarter of the 2 MT row tiles and the 16-column tile cw.  Two-stage ring of (32 MT + 64) rows x 128 B in
// the item's (dead) LDS.  A slab's loader never reads O rows outside its own pair: rows past 2 P re-read the pair's last row.
#ifdef RRT_TRACE
#define RRT_PSLAB_TRACE_ARG , WaveTrace& _tr
#define RRT_PSLAB_TRACE_PASS , _tr
#else
#define RRT_PSLAB_TRACE_ARG
#define RRT_PSLAB_TRACE_PASS
#endif
template <int MT, int PREC>
__device__ __forceinline__ void pair_slab(const uint16_t* __restrict__ O, const int n_rows, const int P, const int D,
                                          const int heads_rt, const PairProj& pj, char* smem, int* s_abort RRT_PSLAB_TRACE_ARG) {
  using H = H16<PREC>;
  using Frag = typename H::frag;
  static_assert(MT % 2 == 0, "pair_slab: the 2 MT row tiles are shared out over four wave groups");
  constexpr int MT2 = 2 * MT, RT = MT2 / 4;           // row tiles of the pair; per wave
  constexpr int BM2 = 16 * MT2;
  constexpr int STG_B = (BM2 + HD) * ROWB;            // bytes per stage
  constexpr int NA = BM2 / 8, NBp = HD / 8, NP = NA + NBp, LP = (NP + 15) / 16;
  // the ring lives in the item's (dead) tiles: three stages where they fit (MT = 6, 8: they do).  With two, every K tile
  // waited for the DMA round trip of a stage issued one short compute step earlier: a slab took 22 K cycles, as long as the
  // separate launch's tile, and the merged launch was SLOWER than the two (109 against 100 us at N = 30000)
  constexpr int MAIN_B = (2 * (2 * 16 * MT * ROWB + 64 * VT_PITCH) > 4 * 16 * MT * ROWB + 2 * BN * ROWB)
                             ? 2 * (2 * 16 * MT * ROWB + 64 * VT_PITCH) : 4 * 16 * MT * ROWB + 2 * BN * ROWB;
  constexpr int NS = 3 * STG_B <= MAIN_B ? 3 : 2;
  const int b = (int)blockIdx.x;
  if (b < pj.lag) return;
  const int sidx = b - pj.lag;                        // < n_items by the grid size
  RRT_TRACE_MARK();                                   // slab [1] entry
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  int pair, col;
  {
    const int xcd = sidx & 7, idx = sidx >> 3;        // (the item map of the kernel; PROJ launches have whole groups of 8 pairs)
    const int grp = idx / heads_rt;
    pair = grp * 8 + xcd;
    col = idx - grp * heads_rt;
  }
  // the pair's `heads` items have arrived (their O rows are in memory): blocks with lower indices.  Bounded wait, as
  // rmsa_fused.hip::proj_slab (in-order workgroup dispatch is an assumption, not a guarantee): a slab that gives up raises
  // the process's hand-over error word and writes nothing.
  if (tid == 0) {
    int spins = 0, bad = 0;
    while (__hip_atomic_load(pj.cnt + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pj.wait_for) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > pj.spin_limit) { bad = 1; break; }
    }
    if (bad && pj.err != nullptr) __hip_atomic_store(pj.err, 1 + 2 * pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *s_abort = bad;
  }
  __syncthreads();
  if (*s_abort) return;                               // (block-uniform)
  RRT_TRACE_MARK();                                   // slab [2] the pair's items have arrived
  const int row0 = 2 * pair * P;                      // first O row of the pair; its 2 P rows are contiguous
  const int nrows = 2 * P;
  const unsigned lds_b = lds_addr_of(smem);
  unsigned off[LP];
#pragma unroll
  for (int q = 0; q < LP; ++q) {
    const int piece = q * 16 + wave;
    const int pp = lane & 7;
    if (piece < NA) {
      const int srow = piece * 8 + (lane >> 3);
      int gr = row0 + (srow < nrows ? srow : nrows - 1);
      gr = gr < n_rows ? gr : n_rows - 1;
      off[q] = (unsigned)gr * (unsigned)D * 2u + (unsigned)((pp ^ ((srow >> 1) & 7)) << 4);
    } else {
      const int wrow = (piece - NA) * 8 + (lane >> 3);          // [0, 64): output column 64 col + wrow
      off[q] = (unsigned)(col * HD + wrow) * (unsigned)D * 2u + (unsigned)((pp ^ ((wrow >> 1) & 7)) << 4);
    }
  }
  auto stage = [&](int kt, unsigned buf) {
    const char* ob = (const char*)O + kt * ROWB;
    const char* wb = (const char*)pj.Wp + kt * ROWB;
#pragma unroll
    for (int q = 0; q < LP; ++q) {
      const int piece = q * 16 + wave;
      if (piece < NA) dma16s(ob, off[q], buf + piece * 1024);
      else if (piece < NP) dma16s(wb, off[q], buf + BM2 * ROWB + (piece - NA) * 1024);
    }
  };
  const int nk = D / 64;
  const int cw = wave & 3, rg = wave >> 2;            // 16-column tile; quarter of the row tiles
  const int ncol = col * HD + 16 * cw + 4 * lg;       // this lane's four output columns
  // un-partition map of this lane's RT rows (region_reverse, rmsa.py:41-54) and their residual rows, requested FIRST: they
  // are older than every DMA piece, so the loop's counted waits (loads return in order) cover them, and they land under it
  int toff[RT];
  float4 rq[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int m = (rg * RT + i) * 16 + lr;            // row of the pair
    const int rsel = m >= P ? 1 : 0;
    const int mm = m - rsel * P;
    const int reg = 2 * pair + rsel;
    const int ri = fdiv(reg, pj.g.rs, pj.g.inv_rs), rj = reg - ri * pj.g.rs;
    const int pi = fdiv(mm, pj.g.s, pj.g.inv_s), pjj = mm - pi * pj.g.s;
    const int t = (ri * pj.g.s + pi) * pj.g.H + rj * pj.g.s + pjj;
    toff[i] = (m < nrows && t < pj.g.L) ? t * D + ncol : -1;
    rq[i] = *(const float4*)(pj.resid + (toff[i] < 0 ? 0 : toff[i]));
  }
  const float4 bias = pj.bias ? *(const float4*)(pj.bias + ncol) : make_float4(0.f, 0.f, 0.f, 0.f);
  f32x4 acc[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  asm volatile("" ::: "memory");                      // (the residual requests stay in front of the DMA pieces)
  // pieces this wave issues per stage: the loop waits until only the NEXT stage's are still in flight
  int mine = 0;
#pragma unroll
  for (int q = 0; q < LP; ++q) mine += (q * 16 + wave < NP) ? 1 : 0;
  auto wait_mine = [&] {                              // s_waitcnt vmcnt(mine), mine in 1 .. 4 (wave-uniform)
    if (mine <= 1) wait_vmcnt<1>(); else if (mine == 2) wait_vmcnt<2>(); else if (mine == 3) wait_vmcnt<3>(); else wait_vmcnt<4>();
  };
  stage(0, lds_b);
  if (NS == 3 && nk > 1) stage(1, lds_b + STG_B);
  RRT_TRACE_MARK();                                   // slab [3] first stages issued
  for (int kt = 0; kt < nk; ++kt) {
    if (NS == 3 && kt + 1 < nk) wait_mine(); else wait_vm0();
    if (kt == 0 || kt == 4) RRT_TRACE_MARK();         // slab [4,6] K tile 0 / 4 landed
    lds_sync();                                       // K tile kt is complete; everyone is done with K tile kt - 1
    if (kt == 0 || kt == 4) RRT_TRACE_MARK();         // slab [5,7] barrier passed
    {
      const int nx = kt + NS - 1;                     // into the buffer K tile kt - 1 just left
      if (nx < nk) stage(nx, lds_b + (nx % NS) * STG_B);
    }
    const char* As = smem + (kt % NS) * STG_B;
    const char* Bs = As + BM2 * ROWB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int cslot = 4 * kk + lg;
      Frag a8[RT], b8;
      {
        const int row = 16 * cw + lr;
        b8 = *(const Frag*)(Bs + row * ROWB + ((cslot ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int row = (rg * RT + i) * 16 + lr;
        a8[i] = *(const Frag*)(As + row * ROWB + ((cslot ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < RT; ++i) acc[i] = H::mfma(b8, a8[i], acc[i]);   // reg r = C[m = .. + lr][n = 16 cw + 4 lg + r]
    }
  }
  RRT_TRACE_MARK();                                   // slab [8] last MFMA issued
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    if (toff[i] < 0) continue;
    const float4 q = rq[i];
    float4 v;
    v.x = (acc[i][0] + bias.x) + q.x; v.y = (acc[i][1] + bias.y) + q.y;
    v.z = (acc[i][2] + bias.z) + q.z; v.w = (acc[i][3] + bias.w) + q.w;
    *(float4*)(pj.out + toff[i]) = v;
  }
  RRT_TRACE_MARK();                                   // slab [9] stores issued
}

constexpr int VQ_B = 64 * VT_PITCH;
constexpr bool PAIR16_NH2_DEFAULT = true;    // sixteen-wave form for 6..9 row tiles: see launch_rmsa_pair16     // per region: V^T [64][512 B]; before that, Q^T [64][512 B] for the stencil

// NH = 1: eight waves, wave (region, c) owns all row tiles of its region for its 16-column tile of Q / K / V.
// NH = 2: sixteen waves (<= 128 VGPRs), wave (region, row half, c) owns half the row tiles: twice the issuing waves for the
//         DMA pieces (the stream's rate grows with them) and eighteen query tiles on sixteen waves instead of eight.
template <int MT, int PREC, int NH, bool PROJ = false>
__global__ __launch_bounds__(512 * NH, 1) void rmsa_pair16_kernel(const uint16_t* __restrict__ U,
                                                                  const uint16_t* __restrict__ W,
                                                                  const float* __restrict__ bqkv,
                                                                  const float* __restrict__ pe_w,
                                                                  uint16_t* __restrict__ O, int n_rows, int n_regions, int P,
                                                                  int D, int heads_rt, int epeg_k, float q_scale,
                                                                  int qs_pitch, int H8, int nstep, const PairProj pj) {
  static_assert(!PROJ || (NH == 2 && MT % 2 == 0), "the projection phase: sixteen waves, an even number of row tiles");
  using H = H16<PREC>;
  using Frag = typename H::frag;
  using E = typename H::elem;
  constexpr int BM = 16 * MT;
  constexpr int MTP = (MT + 1) & ~1;                 // key tiles rounded up to whole 32-key MFMA blocks
  constexpr int NWV = 8 * NH;                        // waves per block
  constexpr int MTH = (MT + NH - 1) / NH;            // row tiles a wave owns (the last half may own one fewer)
  // two rings: U tiles (region A rows | region B rows) two stages deep, W tiles (192 rows) THREE deep where the LDS
  // allows and every wave issues the same number of W pieces (NH = 1; measured equal to two: the stream is bound by
  // how many waves issue, not by bytes in flight -- tools/ubench/dma_rows.hip)
  constexpr int USTG_B = 2 * BM * ROWB, WSTG_B = BN * ROWB;
  constexpr int NSW = (NH == 1 && 2 * USTG_B + 3 * WSTG_B + 512 <= 160 * 1024) ? 3 : 2;
  constexpr int NA = 2 * BM / 8, NB = BN / 8;        // 1-KiB DMA pieces (8 rows) per stage
  constexpr int NP = NA + NB, LP = (NP + NWV - 1) / NWV;   // pieces per stage / per wave
  static_assert(NB == 24, "with eight waves every wave issues exactly three W pieces per stage (the counted wait below)");
  constexpr int QT_B = BM * ROWB, KS_B = BM * ROWB;
  constexpr int REG_B = QT_B + KS_B + VQ_B;          // per region: Q~ | K | V^T (Q^T before the stencil)
  constexpr int RING_B = 2 * USTG_B + NSW * WSTG_B, TILES_B = 2 * REG_B;
  constexpr int LDS_MAIN = RING_B > TILES_B ? RING_B : TILES_B;
  static_assert(16 * MTP * 2 <= VT_PITCH, "V^T row");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef RRT_PAIR16_PRIO
  __builtin_amdgcn_s_setprio(RRT_PAIR16_PRIO);      // (experiment: as rmsa_fused_kernel's RRT_FUSED_PRIO)
#endif
  const int rsel = wave / (4 * NH), rh = (wave >> 2) % NH, cw = wave & 3;   // region of the pair; row half; 16-column tile
  const int i_base = rh * MTH;                       // first row tile of this wave
  const int nt = (MT - i_base) < MTH ? (MT - i_base) : MTH;   // row tiles it owns
  const int lr = lane & 15, lg = lane >> 4;
  const unsigned lds_b = lds_addr_of(smem);
  // XCD-aware block -> (pair, head) map: the head-blocks of a pair sit on ONE XCD (its U panels are fetched from HBM
  // once and served to the other heads from that XCD's L2)
  __shared__ int s_abort;
  const int n_items = PROJ ? pj.n_items : (int)gridDim.x;
  const bool has_item = !PROJ || (int)blockIdx.x < n_items;
  if constexpr (PROJ) {
    if (pj.zero64 != nullptr && blockIdx.x == 0 && threadIdx.x < 64) pj.zero64[threadIdx.x] = 0;
  }
  int head = 0, pair = 0;
  if (has_item) {
    const int b = blockIdx.x;
    const int n_pairs = n_items / heads_rt;
    const int full = (n_pairs >> 3) * 8 * heads_rt;
    if (b < full) {
      const int xcd = b & 7, idx = b >> 3;
      const int grp = idx / heads_rt;
      pair = grp * 8 + xcd;
      head = idx - grp * heads_rt;
    } else {
      const int rem = b - full;
      pair = (n_pairs >> 3) * 8 + rem / heads_rt;
      head = rem % heads_rt;
    }
  }
  RRT_TRACE_INIT(blockIdx.x * NWV + wave);
  if (has_item) {
  const int reg = 2 * pair + rsel;
  const bool valid = reg < n_regions;                // (odd region count: the last pair's second half is a dummy)
  const int row0 = (valid ? reg : 2 * pair) * P;
  const int nk = D / 64;
  const int half = epeg_k >> 1;
  // the stencil's band as ready-made MFMA B operands, built once per block while the first K tiles are in flight:
  // tfrag[s][hi | lo][lane] (16 bytes each) = the eight taps tap(32 s + 8 lg + e - H8 - lr + k/2), e = 0..7, of lane
  // (lr, lg) -- tap(i) = w_h[i] + [i == k/2] for i in [0, k), else 0 -- split into a 16-bit value and its 16-bit remainder
  Frag* const tfrag = (Frag*)(smem + LDS_MAIN);
  if (tid < 192) {
    const int s_ = tid >> 6, l_ = tid & 63, lr_ = l_ & 15, lg_ = l_ >> 4;
    Frag hi8, lo8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = 32 * s_ + 8 * lg_ + e - H8 - lr_ + half;
      float tv = (pe_w != nullptr && t >= 0 && t < epeg_k && s_ < nstep) ? pe_w[head * epeg_k + t] : 0.f;
      if (t == half && s_ < nstep) tv += 1.0f;
      const E hi = (E)tv;
      hi8[e] = hi;
      lo8[e] = (E)(tv - (float)hi);
    }
    tfrag[(s_ * 2) * 64 + l_] = hi8;
    tfrag[(s_ * 2 + 1) * 64 + l_] = lo8;
  }
  RRT_TRACE_MARK();                                 // [1] entry

  // ================================================================== phase 1: projection, both regions against one W tile
  unsigned off[LP];
#pragma unroll
  for (int q = 0; q < LP; ++q) {
    const int piece = q * NWV + wave;
    const int p = lane & 7;
    if (piece < NA) {
      const int srow = piece * 8 + (lane >> 3);     // row of the stage's A part: [0, BM) region A, [BM, 2 BM) region B
      const int rs = srow >= BM ? 1 : 0;
      const int rr = (2 * pair + rs < n_regions) ? 2 * pair + rs : 2 * pair;
      int gr = rr * P + (srow - rs * BM);
      gr = gr < n_rows ? gr : n_rows - 1;           // rows past the last region: re-read (finite, never used)
      off[q] = (unsigned)gr * (unsigned)D * 2u + (unsigned)((p ^ ((srow >> 1) & 7)) << 4);
    } else {
      const int wrow = (piece - NA) * 8 + (lane >> 3);    // [0, 192): wrow / 64 picks q / k / v
      const int wr = (wrow >> 6) * D + head * HD + (wrow & 63);
      off[q] = (unsigned)wr * (unsigned)D * 2u + (unsigned)((p ^ ((wrow >> 1) & 7)) << 4);
    }
  }
  auto stage_u = [&](int kt, unsigned buf) {
    const char* ub = (const char*)U + kt * ROWB;
#pragma unroll
    for (int q = 0; q < LP; ++q) {
      const int piece = q * NWV + wave;
#ifdef RRT_NT_A16_PAIR
      if (piece < NA) dma16s_nt(ub, off[q], buf + piece * 1024);
#else
      if (piece < NA) dma16s(ub, off[q], buf + piece * 1024);
#endif
    }
  };
  auto stage_w = [&](int kt, unsigned buf) {
    const char* wb = (const char*)W + kt * ROWB;
#pragma unroll
    for (int q = 0; q < LP; ++q) {
      const int piece = q * NWV + wave;
      if (piece >= NA && piece < NP) dma16s(wb, off[q], buf + (piece - NA) * 1024);
    }
  };
  const unsigned wring = lds_b + 2 * USTG_B;
  f32x4 acc[MTH][3];
#pragma unroll
  for (int i = 0; i < MTH; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  stage_u(0, lds_b);
  stage_w(0, wring);
  if (NSW == 3 && nk > 1) stage_w(1, wring + WSTG_B);
  // bias of this lane's columns: lands under the K loop
  const int dk = 16 * cw + 4 * lg;                  // first of the lane's 4 K columns
  const int dc = 16 * cw + lr;                      // the lane's Q / V column
  float4 bk4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float bq1 = 0.f, bv1 = 0.f;
  if (bqkv) {
    bq1 = bqkv[head * HD + dc];
    bk4 = *(const float4*)(bqkv + D + head * HD + dk);
    bv1 = bqkv[2 * D + head * HD + dc];
  }
  RRT_TRACE_MARK();                                 // [2] first stages issued
  int wslot = 0;                                    // W ring slot of K tile kt
  for (int kt = 0; kt < nk; ++kt) {
    // U tile kt and W tile kt landed; with three W stages the W tile kt + 1 -- this wave's LAST three pieces, issued
    // after U tile kt -- may still be in flight
    if (NSW == 3 && kt + 1 < nk) wait_vmcnt<3>(); else wait_vm0();
    __syncthreads();                                // K tile kt is complete; everyone is done with K tile kt - 1
    if (kt == 0 || kt == 1 || kt == 4) RRT_TRACE_MARK();   // [3,5,7]
    if (kt + 1 < nk) stage_u(kt + 1, lds_b + ((kt + 1) & 1) * USTG_B);
    {
      const int nw = kt + NSW - 1;                  // W tile to issue now: into the slot K tile kt - 1 just left
      int ns = wslot + NSW - 1;
      ns = ns >= NSW ? ns - NSW : ns;
      if (nw < nk) stage_w(nw, wring + ns * WSTG_B);
    }
    if (kt == 0 || kt == 1 || kt == 4) RRT_TRACE_MARK();   // [4,6,8] next stages issued
    const char* As = smem + (kt & 1) * USTG_B + (rsel * BM + i_base * 16) * ROWB;
    const char* Bs = smem + 2 * USTG_B + wslot * WSTG_B;
    wslot = wslot + 1 == NSW ? 0 : wslot + 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      Frag a8[MTH], b8[3];
      const int cslot = 4 * kk + lg;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int row = 64 * j + 16 * cw + lr;
        b8[j] = *(const Frag*)(Bs + row * ROWB + ((cslot ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MTH; ++i) {
        if (NH > 1 && i >= nt) continue;            // (wave-uniform: the short half has one row tile fewer)
        const int row = i * 16 + lr;                // BM and 16 i_base are multiples of 16: swizzle key of the stage row
        a8[i] = *(const Frag*)(As + row * ROWB + ((cslot ^ (((rsel * BM + i_base * 16 + row) >> 1) & 7)) << 4));
      }
      // Q and V with the U fragment in the A slot: reg r of lane (lr, lg) = C[token 16 i + 4 lg + r][d = 16 cw + lr]
      // (four consecutive TOKENS of one column: what the transposed images Q^T and V^T want); K with the roles
      // swapped: reg r = K[token 16 i + lr][d = 16 cw + 4 lg + r] (the row-major K image)
#pragma unroll
      for (int i = 0; i < MTH; ++i) {
        if (NH > 1 && i >= nt) continue;
        acc[i][0] = H::mfma(a8[i], b8[0], acc[i][0]);
        acc[i][1] = H::mfma(b8[1], a8[i], acc[i][1]);
        acc[i][2] = H::mfma(a8[i], b8[2], acc[i][2]);
      }
    }
  }
  RRT_TRACE_MARK();                                 // [9] last projection MFMA issued
  __syncthreads();                                  // the staging ring is dead
  RRT_TRACE_MARK();                                 // [10]

  // ================================================================== phases 2 + 3: Q^T -> stencil -> Q~, K, V^T
  // Wave (region, half, c) owns head-dim columns 16 c .. 16 c + 15 of its region for ITS row tiles: it writes those rows
  // of Q^T, runs the stencil for its query tiles on those rows, and then overwrites them with its part of V^T (same
  // 512-byte pitch, same place).  NH = 1: everything a wave reads it wrote itself -- no barrier until the tiles are
  // complete.  NH = 2: the stencil's sources cross the row halves, so the two halves meet at a block barrier before the
  // stencil and before V^T goes over Q^T.
  char* const RB = smem + rsel * REG_B;
  char* const QT = RB;                              // Q~ [BM] x 128 B, slot XOR ((row >> 1) & 7)
  char* const KS = RB + QT_B;                       // K  [BM] x 128 B, same swizzle
  char* const VQ = RB + QT_B + KS_B;                // [64][512 B]: Q^T (slot XOR (d & 15)) now, V^T after the stencil
  {
    const float qs = q_scale * LOG2E;
    char* const qrow = VQ + dc * VT_PITCH;
    // 8 bytes at token position pos (a multiple of 4) of this lane's Q^T row
    auto qaddr = [&](int pos) -> char* { return qrow + ((((pos >> 3) ^ lr) & 31) << 4) + ((pos & 4) << 1); };
    // halo: token positions [0, H8) and [H8 + BM, BM - 16 + 32 nstep) read as zero (region edge = zero padding)
    const uint2 z = make_uint2(0u, 0u);
    if (rh == 0)
      for (int c = lg; c < (H8 >> 2); c += 4) *(uint2*)qaddr(4 * c) = z;
    if (rh == NH - 1) {
      const int right0 = H8 + BM, rightn = (32 * nstep - 16 - H8) >> 2;
      for (int c = lg; c < rightn; c += 4) *(uint2*)qaddr(right0 + 4 * c) = z;
    }
#pragma unroll
    for (int i = 0; i < MTH; ++i) {
      if (NH > 1 && i >= nt) continue;
      const int ig = i_base + i;                    // row tile of the region
      const int tok = 16 * ig + 4 * lg;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (acc[i][0][r] + bq1) * qs;
      if (ig >= MT - 2) {                           // only the last two row tiles can hold rows past the region: zero them
#pragma unroll                                      // (NH = 1: decided at compile time; NH = 2: a wave-uniform branch)
        for (int r = 0; r < 4; ++r) v[r] = (tok + r < P) ? v[r] : 0.f;
      }
      *(uint2*)qaddr(H8 + tok) = pack4<PREC>(v[0], v[1], v[2], v[3]);
      const int m = ig * 16 + lr;
      *(uint2*)(KS + m * ROWB + (((dk >> 3) ^ ((m >> 1) & 7)) << 4) + ((dk & 4) << 1)) =
          pack4<PREC>(acc[i][1][0] + bk4.x, acc[i][1][1] + bk4.y, acc[i][1][2] + bk4.z, acc[i][1][3] + bk4.w);
    }
    if (NH > 1) __syncthreads();                    // the other half's Q^T columns (the stencil reads across the halves)
    RRT_TRACE_MARK();                               // [11] Q^T, K written
    // band of T for this lane (B-slot row = query lr of the tile, k-slice = sources 32 s + 8 lg + e of the window that
    // starts H8 rows before the tile), prepared at the top of the kernel
    Frag thi[3], tlo[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      thi[s] = tfrag[(s * 2) * 64 + lane];
      tlo[s] = tfrag[(s * 2 + 1) * 64 + lane];
    }
    // A operand: rows d = 16 cw + lr, sources 16 t + 32 s + 8 lg .. + 7 (positions; 16-byte slots).
    // (every tile of the wave, also one that lies wholly past the region: its Q^T rows are zeros, its Q~ rows are never
    //  read as queries -- a data-dependent exit here would keep the compiler from overlapping the tiles' reads and MFMAs)
#pragma unroll
    for (int i = 0; i < MTH; ++i) {
      if (NH > 1 && i >= nt) continue;
      const int t = i_base + i;
      f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; ++s)
        if (s < nstep) {
          const int slot = 2 * t + 4 * s + lg;      // (16 t + 32 s + 8 lg) * 2 bytes / 16
          const Frag a = *(const Frag*)(qrow + (((slot ^ lr) & 31) << 4));
          o = H::mfma(a, thi[s], o);
          o = H::mfma(a, tlo[s], o);
        }
      // o[r] = Q~[query 16 t + lr][d = 16 cw + 4 lg + r]
      const int m = 16 * t + lr;
      *(uint2*)(QT + m * ROWB + (((dk >> 3) ^ ((m >> 1) & 7)) << 4) + ((dk & 4) << 1)) = pack4<PREC>(o[0], o[1], o[2], o[3]);
    }
    if (NH > 1) __syncthreads();                    // every read of Q^T is done (both halves): its place becomes V^T
    RRT_TRACE_MARK();                               // [12] stencil done
    // V^T over the Q^T rows (NH = 1: the wave's own reads of them are complete, LDS operations of a wave execute in order)
    char* const VT = VQ;
#pragma unroll
    for (int i = 0; i < MTH; ++i) {
      if (NH > 1 && i >= nt) continue;
      const int ig = i_base + i;
      // tokens 16 ig + 4 lg + r, r = 0..3 -> positions 32 (ig / 2) + 8 lg + 4 (ig % 2) + r of V^T row dc: 8 bytes
      const int vslot = 4 * (ig >> 1) + lg;
      *(uint2*)(VT + dc * VT_PITCH + ((vslot ^ vt_swz(dc)) << 4) + ((ig & 1) << 3)) =
          pack4<PREC>(acc[i][2][0] + bv1, acc[i][2][1] + bv1, acc[i][2][2] + bv1, acc[i][2][3] + bv1);
    }
    if ((MT & 1) && rh == NH - 1) {
      // odd tile count: the second half of the last 32-key block has no keys; its V^T columns meet P = 0 in the MFMA
      // and must hold finite numbers -> zeros.  The wave's 64 lanes = its 16 rows x 4 groups of 4 positions.
      const int dd = 16 * cw + (lane >> 2), g = lane & 3;
      const int vslot = 4 * (MT >> 1) + g;
      *(uint2*)(VT + dd * VT_PITCH + ((vslot ^ vt_swz(dd)) << 4) + 8) = make_uint2(0u, 0u);
    }
  }
  __syncthreads();
  RRT_TRACE_MARK();                                 // [13] tiles complete

  // ================================================================== phase 4: attention from LDS (as rmsa_fused16)
  const char* const VT = VQ;
  for (int t = rh * 4 + cw; t < MT; t += 4 * NH) {
    const int i0 = t * 16;
    if (i0 >= P) break;
    Frag bq[2];
    {
      const int m = i0 + lr;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) bq[kk] = *(const Frag*)(QT + m * ROWB + (((4 * kk + lg) ^ ((m >> 1) & 7)) << 4));
    }
    f32x4 s[MTP];
#pragma unroll
    for (int jt = 0; jt < MTP; ++jt) s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      constexpr int CH = NH > 1 ? (MT + 1) / 2 : MT;  // K fragments in flight (sixteen waves: 128 VGPRs)
#pragma unroll
      for (int j0 = 0; j0 < MT; j0 += CH) {
        Frag a[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int jt = j0 + u < MT ? j0 + u : MT - 1;
          const int row = jt * 16 + lr;
          a[u] = *(const Frag*)(KS + row * ROWB + (((4 * kk + lg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (j0 + u < MT) s[j0 + u] = H::mfma(a[u], bq[kk], s[j0 + u]);
      }
    }
    RRT_TRACE_MARK();                               // tile: S^T issued
    // s[jt][r] = log2e * score(query i0 + lr, key 16 jt + 4 lg + r)
    float cmax = NEG_BIG;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
      if ((jt + 1) * 16 > P) {                                    // (wave-uniform) tiles with keys past the region
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (jt * 16 + 4 * lg + r >= P) s[jt][r] = NEG_BIG;
      }
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) cmax = fmaxf(cmax, s[jt][r]);
    cmax = max_xor32(max_xor16(cmax));
    float psum = 0.f;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[jt][r] - cmax);
        s[jt][r] = p;
        psum += p;
      }
    psum = sum_xor32(sum_xor16(psum));
    const float inv = 1.0f / psum;                  // of THIS lane's query (lr): the four lg lanes agree
    asm volatile("" :: "v"(inv));
    RRT_TRACE_MARK();                               // tile: softmax done
    f32x4 oacc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < MTP / 2; ++b) {
      const Frag pb = pack8<PREC>(s[2 * b], s[2 * b + 1]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd = 4 * lr + c;
        const Frag vf = *(const Frag*)(VT + dd * VT_PITCH + (((4 * b + lg) ^ ((lr ^ (c << 2)) & 15)) << 4));
        oacc[c] = H::mfma(vf, pb, oacc[c]);
      }
    }
    RRT_TRACE_MARK();                               // tile: PV issued
    // oacc[c][r] = O[query i0 + lr][d = 16 lg + 4 r + c]
    const int i = i0 + lr;
    if (i < P && valid) {
      uint2 q0 = pack4<PREC>(oacc[0][0] * inv, oacc[1][0] * inv, oacc[2][0] * inv, oacc[3][0] * inv);
      uint2 q1 = pack4<PREC>(oacc[0][1] * inv, oacc[1][1] * inv, oacc[2][1] * inv, oacc[3][1] * inv);
      uint2 q2 = pack4<PREC>(oacc[0][2] * inv, oacc[1][2] * inv, oacc[2][2] * inv, oacc[3][2] * inv);
      uint2 q3 = pack4<PREC>(oacc[0][3] * inv, oacc[1][3] * inv, oacc[2][3] * inv, oacc[3][3] * inv);
      uint16_t* dst = O + (size_t)(row0 + i) * D + head * HD + 16 * lg;
      if constexpr (PROJ) {
        store_wt16(dst, q0.x, q0.y, q1.x, q1.y);
        store_wt16(dst + 8, q2.x, q2.y, q3.x, q3.y);
      } else {
        *(uint4*)dst = make_uint4(q0.x, q0.y, q1.x, q1.y);
        *(uint4*)(dst + 8) = make_uint4(q2.x, q2.y, q3.x, q3.y);
      }
    }
    RRT_TRACE_MARK();                               // tile: O stored
  }
  // ---------------------------------------------------------------- PROJ: this item has arrived
  if constexpr (PROJ) {
    wait_vm0();                                     // this thread's write-through stores of O are in memory ...
    __syncthreads();                                // ... and everybody's; the tiles in LDS are dead
    RRT_TRACE_MARK();                               // item: O in memory
    if (tid == 0) __hip_atomic_fetch_add(pj.cnt + pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }   // has_item
  if constexpr (PROJ) pair_slab<MT, PREC>(O, n_rows, P, D, heads_rt, pj, smem, &s_abort RRT_PSLAB_TRACE_PASS);
}

// stencil geometry for a tap count: zero halo H8 (k/2 rounded up to 8), 32-wide MFMA steps, bytes of a Q^T row in use
struct StencilGeo { int H8, nstep, pitch; };
StencilGeo stencil_geo(int BM, int epeg_k) {
  StencilGeo g;
  const int half = epeg_k >> 1;
  g.H8 = (half + 7) & ~7;
  g.nstep = (16 + 2 * g.H8 + 31) / 32;
  const int qw = BM - 16 + 32 * g.nstep;            // token positions a Q^T row holds
  g.pitch = qw * 2;                                 // bytes of a Q^T row in use (rows sit at the V^T pitch, 512 B)
  return g;
}

template <int MT, int PREC, int NH>
hipError_t launch_pair(const uint16_t* U, const uint16_t* W, const float* bqkv, const float* pe_w, uint16_t* O,
                       int n_regions, int P, int D, int heads, int epeg_k, hipStream_t st) {
  constexpr int BM = 16 * MT;
  constexpr size_t USTG = (size_t)2 * BM * ROWB, WSTG = (size_t)BN * ROWB;
  constexpr size_t RING = 2 * USTG + ((NH == 1 && 2 * USTG + 3 * WSTG + 512 <= 160 * 1024) ? 3 : 2) * WSTG;
  constexpr size_t TILES = (size_t)2 * (2 * BM * ROWB + VQ_B);
  constexpr size_t LDS = (RING > TILES ? RING : TILES) + 6 * 1024;     // + the stencil's operand table
  static_assert(LDS <= 160 * 1024, "LDS budget");
  const int ek = pe_w ? epeg_k : 0;
  const StencilGeo g = stencil_geo(BM, ek);
  if (g.pitch > VT_PITCH || g.nstep > 3) return hipErrorInvalidValue;
  auto kern = rmsa_pair16_kernel<MT, PREC, NH>;
  static OncePerDevice once;
  if (once.first())
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
  const float q_scale = 1.0f / sqrtf((float)HD);
  const int pairs = (n_regions + 1) / 2;
  kern<<<dim3(heads * pairs), dim3(512 * NH), LDS, st>>>(U, W, bqkv, pe_w, O, n_regions * P, n_regions, P, D, heads, ek, q_scale,
                                                        g.pitch, g.H8, g.nstep, PairProj{});
  return hipGetLastError();
}

// one "round" of the chip between an item and the slab of the same index (one block per CU: LDS); a multiple of 8 so that
// a slab stays on its pair's XCD
int pair_proj_lag(int n_items) {
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
  return lag;
}

template <int MT, int PREC>
hipError_t launch_pair_proj(const uint16_t* U, const uint16_t* W, const float* bqkv, const float* pe_w, uint16_t* O,
                            int n_regions, int P, int D, int heads, int epeg_k, PairProj pj, hipStream_t st) {
  constexpr int NH = 2, BM = 16 * MT;
  constexpr size_t USTG = (size_t)2 * BM * ROWB, WSTG = (size_t)BN * ROWB;
  constexpr size_t RING = 2 * USTG + 2 * WSTG;
  constexpr size_t TILES = (size_t)2 * (2 * BM * ROWB + VQ_B);
  constexpr size_t SLAB = (size_t)2 * (2 * BM + HD) * ROWB;          // the slab's two-stage ring
  constexpr size_t MAIN = RING > TILES ? RING : TILES;
  static_assert(SLAB <= MAIN, "the slab's ring lives in the item's LDS");
  constexpr size_t LDS = MAIN + 6 * 1024;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  const int ek = pe_w ? epeg_k : 0;
  const StencilGeo g = stencil_geo(BM, ek);
  if (g.pitch > VT_PITCH || g.nstep > 3) return hipErrorInvalidValue;
  auto kern = rmsa_pair16_kernel<MT, PREC, NH, true>;
  static OncePerDevice once;
  if (once.first())
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
  const float q_scale = 1.0f / sqrtf((float)HD);
  kern<<<dim3(pj.n_items + pj.lag), dim3(512 * NH), LDS, st>>>(U, W, bqkv, pe_w, O, n_regions * P, n_regions, P, D, heads, ek,
                                                              q_scale, g.pitch, g.H8, g.nstep, pj);
  return hipGetLastError();
}

}  // namespace

#ifdef RRT_TRACE
RRT_TRACE_DEFINE_READER(rrt_debug_trace_pair16)
#endif

bool rmsa_pair16_supported(int n_regions, int P, int D, int heads, int epeg_k) {
  static const bool off = rrt_tune_env("RRT_NO_PAIR16") != nullptr;
  if (off) return false;
  if (!(heads > 0 && D == heads * HD && D % 64 == 0 && P > 16 && P <= 176 && epeg_k >= 0 && epeg_k <= 63 && n_regions >= 8))
    return false;
  const int MT = P > 144 ? 11 : P > 128 ? 9 : P > 112 ? 8 : P > 96 ? 7 : P > 64 ? 6 : P > 32 ? 4 : 2;
  const StencilGeo g = stencil_geo(16 * MT, epeg_k);
  return g.pitch <= VT_PITCH && g.nstep <= 3;
}

hipError_t launch_rmsa_pair16(const uint16_t* U, const uint16_t* W, const float* bqkv, const float* pe_w, uint16_t* O,
                              int n_regions, int P, int D, int heads, int epeg_k, int prec, hipStream_t st) {
  if (prec != 1 && prec != 2) return hipErrorInvalidValue;
  // sixteen waves (two row halves per region) where a wave still has >= 3 row tiles; RRT_PAIR16_NH=1|2 in a tuning build
  static const int nh_env = rrt_tune_env("RRT_PAIR16_NH") ? atoi(rrt_tune_env("RRT_PAIR16_NH")) : 0;
#define RRT_PAIR16(MT_)                                                                                       \
  if ((nh_env ? nh_env == 2 : PAIR16_NH2_DEFAULT) && MT_ >= 6 && MT_ <= 9)                                               \
    return prec == 1 ? launch_pair<MT_, 1, 2>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, st)         \
                     : launch_pair<MT_, 2, 2>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, st);        \
  return prec == 1 ? launch_pair<MT_, 1, 1>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, st)           \
                   : launch_pair<MT_, 2, 1>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, st);
  if (P > 144) { RRT_PAIR16(11) }
  if (P > 128) { RRT_PAIR16(9) }
  if (P > 112) { RRT_PAIR16(8) }
  if (P > 96) { RRT_PAIR16(7) }
  if (P > 64) { RRT_PAIR16(6) }
  if (P > 32) { RRT_PAIR16(4) }
  RRT_PAIR16(2)
#undef RRT_PAIR16
}

// The projection as a phase of the pair launch: where the launch has at least two rounds of items (a slab then runs a whole
// item behind the blocks it waits for) and every item a slab waits for has a LOWER block index (whole groups of 8 pairs:
// a pair's items span 8 * heads consecutive indices, lag >= that).
bool rmsa_pair16_proj_supported(int n_regions, int P, int D, int heads, int epeg_k) {
  static const bool off = rrt_tune_env("RRT_NO_PAIR16_PROJ") != nullptr;
  if (off || !rmsa_pair16_supported(n_regions, P, D, heads, epeg_k)) return false;
  const bool mt6 = P > 64 && P <= 96, mt8 = P > 112 && P <= 128;
  if (!(mt6 || mt8) || (n_regions & 15) != 0 || D % 64 != 0) return false;
  const int n_items = (n_regions / 2) * heads, lag = pair_proj_lag(n_items);
  return n_items >= 2 * lag && lag >= 8 * heads && (long)n_regions * P * D * 2 < 4000000000L;
}

hipError_t launch_rmsa_pair16_proj(const uint16_t* U, const uint16_t* W, const float* bqkv, const float* pe_w, uint16_t* O,
                                   int n_regions, int P, int D, int heads, int epeg_k, int prec, const PairProj& proj,
                                   hipStream_t st) {
  if ((prec != 1 && prec != 2) || !rmsa_pair16_proj_supported(n_regions, P, D, heads, pe_w ? epeg_k : 0)) return hipErrorInvalidValue;
  if (proj.Wp == nullptr || proj.resid == nullptr || proj.out == nullptr || proj.cnt == nullptr) return hipErrorInvalidValue;
  PairProj pj = proj;
  pj.n_items = (n_regions / 2) * heads;
  if (pj.lag <= 0) pj.lag = pair_proj_lag(pj.n_items);
  if (pj.lag < 8 * heads || pj.lag > pj.n_items || (pj.lag & 7)) return hipErrorInvalidValue;
  if (pj.wait_for <= 0) pj.wait_for = heads;
  if (pj.spin_limit <= 0) pj.spin_limit = 1 << 22;
  if (pj.err == nullptr) pj.err = handover_err_device();
  if (P > 112)
    return prec == 1 ? launch_pair_proj<8, 1>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, pj, st)
                     : launch_pair_proj<8, 2>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, pj, st);
  return prec == 1 ? launch_pair_proj<6, 1>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, pj, st)
                   : launch_pair_proj<6, 2>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, pj, st);
}
