// rmsa_fused16.hip -- the fused R-MSA core (rmsa_fused.hip) for the reduced-precision modes, on 16-bit data end to end.
//
// Replaces InnerAttention.forward up to (not including) proj, modules/rmsa.py:100-122, under the reference's --amp
// path (main.py:101-102,439: nn.Linear, q k^T and attn v run in bf16 / fp16 under autocast).  One block per
// (region, head), 8 waves:
//   phase 1  C[P x 192] = U16_r[P x D] . W16_h[192 x D]^T on the 16-bit matrix cores (v_mfma_f32_16x16x32_bf16/f16,
//            fp32 accumulate).  U16 (LayerNorm output, cast16.hip) and W16 arrive in 16 bits: a K tile is 64
//            elements = 128-byte rows, so the LDS image, the XOR swizzle and the DMA addressing are those of the
//            fp32 kernel with half the bytes per element, and a 16-byte LDS slot IS one MFMA operand -- nothing is
//            converted in the loop.  With MFMAs 16x faster than fp32 the loop is bound by how fast 1-KiB DMA pieces
//            can be issued, so FOUR loader waves (one per SIMD, next to the four compute waves) feed the ring.
//   phase 2  accumulators (+bias, q*scale) -> LDS: Q as fp32 (the EPEG stencil runs in fp32), K as 16-bit rows,
//            V as its 16-bit TRANSPOSE with the keys of each 32-key block permuted into MFMA operand order
//            (pos = 32 b + 8 g + 4 h + r for key = 32 b + 16 h + 4 g + r), so that the P.V operand is one
//            ds_read_b128 and P^T never leaves the registers it was computed in.  (The V column tiles are
//            accumulated with the MFMA operand roles NOT swapped, so a lane holds four consecutive tokens of one
//            column: V^T goes out as 8-byte stores, no cross-lane shuffle.);
//   phase 3  EPEG sliding-window stencil over the fp32 Q tile (as in rmsa_fused.hip), x log2(e), rounded ONCE to
//            16 bits into the Q~ tile that overwrites Q;
//   phase 4  S^T = K Q~^T (scores transposed: lane = query, registers = keys), row softmax in fp32 registers,
//            O^T = V^T P^T with V^T rows taken in the order d = 4 a + c so that a lane ends up with 16 consecutive
//            head-dim columns of ITS query: two 16-byte stores per lane, 128 contiguous bytes per query row.
// Rounding points (restated in oracle/rrt_oracle.py::forward_f64(lowp=...)): U, W (inputs); Q~ * log2 e, K, V,
// exp2(S - max) (operands of the two attention products); O (output).  Accumulation, bias, scale, stencil, softmax
// statistics and normalisation are fp32.
// LDS: max(3 x (16 MT + 192) x 128 B ring, Q 16 MT x 256 B + K 16 MT x 128 B + V^T 64 x 512 B) = 126 KiB at MT = 9.
#include <stdio.h>
#include <stdlib.h>

#include "fused16.h"

namespace {
using namespace f16k;

template <int MT, int PREC>
__global__ __launch_bounds__(512, 2) void rmsa_fused16_kernel(const uint16_t* __restrict__ U,
                                                              const uint16_t* __restrict__ W,
                                                              const float* __restrict__ bqkv,
                                                              const float* __restrict__ pe_w,
                                                              uint16_t* __restrict__ O, int n_rows, int P, int D,
                                                              int heads_rt, int epeg_k, float q_scale) {
  using H = H16<PREC>;
  using Frag = typename H::frag;
  constexpr int BM = 16 * MT;
  constexpr int MTP = (MT + 1) & ~1;                 // key tiles rounded up to whole 32-key MFMA blocks
  constexpr int STAGE_B = (BM + BN) * ROWB;          // bytes per pipeline stage
  constexpr int NA = BM / 8, NB = BN / 8;            // 1-KiB DMA pieces (8 rows) per A / B stage
  constexpr int LA = (NA + 3) / 4, LB = NB / 4;      // per loader wave
  constexpr int NT = 3;                              // 16-column tiles per compute wave (4 x 48 = 192)
  constexpr int QF_B = BM * 256, KS_B = BM * ROWB;   // fp32 Q tile, 16-bit K tile
  static_assert(16 * MTP * 2 <= VT_PITCH, "V^T row");
  // three ring stages while they fit (MT <= 13); regions of 209..256 tokens (MT = 15, 16) get two: their tiles alone
  // are 125 / 131 KiB, and a K tile there carries enough MFMAs (15-16 row tiles) to cover one DMA round trip
  constexpr int NSTG = MT <= 13 ? 3 : 2;
  constexpr int RING_B = NSTG * STAGE_B, TILES_B = QF_B + KS_B + 64 * VT_PITCH;
  constexpr int LDS_MAIN = RING_B > TILES_B ? RING_B : TILES_B;
  constexpr int RUN = (BM + 31) / 32;                // query rows per stencil thread (512 threads = 32 runs x 16 slots)
  constexpr int TAP_OFF = 12 + RUN - 1;              // tap t lives at taps[t + TAP_OFF]; taps[12..] is 16-byte aligned
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const QF = smem;                             // fp32 Q [BM][64], slot XOR (row & 15); later Q~ 16-bit rows
  char* const KS = smem + QF_B;                      // K [BM] x 128 B, slot XOR ((row >> 1) & 7)
  char* const VT = smem + QF_B + KS_B;               // V^T [64] x 512 B, slot XOR ((d >> 2) & 15)
  char* const QT = smem;                             // Q~ [BM] x 128 B (phase 3 writes it over Q)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const unsigned lds_b = lds_addr_of(smem);
  // XCD-aware block -> (region, head) map (as rmsa_fused.hip): the 8 head-blocks of a region sit on ONE XCD so
  // that the region's U panel is fetched from HBM once and served to the other heads from that XCD's L2
  int head, reg;
  {
    const int b = blockIdx.x;
    const int n_regions = gridDim.x / heads_rt;
    const int full = (n_regions >> 3) * 8 * heads_rt;
    if (b < full) {
      const int xcd = b & 7, idx = b >> 3;
      const int grp = idx / heads_rt;
      reg = grp * 8 + xcd;
      head = idx - grp * heads_rt;
    } else {
      const int rem = b - full;
      reg = (n_regions >> 3) * 8 + rem / heads_rt;
      head = rem % heads_rt;
    }
  }
  const int row0 = reg * P;
  const int nk = D / 64;
  // EPEG taps as the stencil wants them -- log2(e) * (w[t] + [t == k/2]), zero outside [0, k) -- in a 128-entry LDS
  // table behind the tiles (index t + TAP_OFF): no global load inside the stencil loop
  float* const taps = (float*)(smem + LDS_MAIN);
  if (tid < 128) {
    const int t = tid - TAP_OFF;
    float wt = (pe_w != nullptr && t >= 0 && t < epeg_k) ? pe_w[head * epeg_k + t] : 0.f;
    if (t == (epeg_k >> 1)) wt += 1.0f;
    taps[tid] = wt * LOG2E;
  }
  RRT_TRACE_INIT(blockIdx.x * 8 + wave);
  RRT_TRACE_MARK();                                 // [1] entry

  // ================================================================== phase 1: projection
  // 3-stage ring: stage kt + 2 is issued while stage kt + 1 lands and stage kt feeds the MFMAs (with two stages the
  // loop period was DMA issue + landing latency, 1550 cycles per K tile against 920 of MFMA)
  if (wave >= 4) {
    const int lw = wave - 4;
    unsigned aoff[LA], boff[LB];
#pragma unroll
    for (int qi = 0; qi < LA; ++qi) {
      const int row = (qi * 4 + lw) * 8 + (lane >> 3), p = lane & 7;
      int gr = row0 + row;
      gr = gr < n_rows ? gr : n_rows - 1;          // rows past the last region: re-read (finite, never used)
      aoff[qi] = (unsigned)gr * (unsigned)D * 2u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int qi = 0; qi < LB; ++qi) {
      const int row = (qi * 4 + lw) * 8 + (lane >> 3), p = lane & 7;   // row in [0,192): row / 64 picks q / k / v
      const int wr = (row >> 6) * D + head * HD + (row & 63);
      boff[qi] = (unsigned)wr * (unsigned)D * 2u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
    auto stage = [&](int kt, unsigned buf) {
      const char* ub = (const char*)U + kt * ROWB;
      const char* wb = (const char*)W + kt * ROWB;
#pragma unroll
      for (int qi = 0; qi < LA; ++qi)
        if (qi * 4 + lw < NA) dma16s(ub, aoff[qi], buf + (qi * 4 + lw) * 1024);
#pragma unroll
      for (int qi = 0; qi < LB; ++qi) dma16s(wb, boff[qi], buf + BM * ROWB + (qi * 4 + lw) * 1024);
    };
    const bool full = (LA - 1) * 4 + lw < NA;       // this loader issues LA (else LA - 1) A pieces per stage
    stage(0, lds_b);
    if (NSTG == 3 && nk > 1) stage(1, lds_b + STAGE_B);
    RRT_TRACE_MARK();                               // loader [2] first stages issued
    if constexpr (NSTG == 3) {
      int slot = 2;                                 // ring slot of stage kt + 2
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {                          // stage kt landed; stage kt + 1 may still be in flight
          if (full) wait_vmcnt<LA + LB>(); else wait_vmcnt<LA - 1 + LB>();
        } else {
          wait_vm0();
        }
        if (kt == 0 || kt == 4) RRT_TRACE_MARK();   // loader [3,5] K tile 0 / 4 landed
        __syncthreads();                            // publishes K tile kt; everyone is done with tile kt - 1
        if (kt + 2 < nk) stage(kt + 2, lds_b + slot * STAGE_B);
        slot = slot == 2 ? 0 : slot + 1;
        if (kt == 0 || kt == 4) RRT_TRACE_MARK();   // loader [4,6] next stage issued
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        wait_vm0();                                 // stage kt landed
        if (kt == 0 || kt == 4) RRT_TRACE_MARK();
        __syncthreads();                            // publishes K tile kt; everyone is done with tile kt - 1
        if (kt + 1 < nk) stage(kt + 1, lds_b + ((kt + 1) & 1) * STAGE_B);
        if (kt == 0 || kt == 4) RRT_TRACE_MARK();
      }
    }
    __syncthreads();                                // "the staging ring is dead"
    RRT_TRACE_MARK();                               // loader [7]
    if (MT & 1) {
      // odd tile count: the second half of the last 32-key block has no keys; its V^T columns meet P = 0 in the
      // MFMA and must hold finite numbers -> zeros.  256 loader threads = 64 rows x 4 groups of 4 positions.
      const int t2 = tid - 256, dd = t2 >> 2, g = t2 & 3;
      const int vslot = 4 * (MT >> 1) + g;          // pos = 32 b + 8 g + 4 .. + 7, b = (MT - 1) / 2
      *(uint2*)(VT + dd * VT_PITCH + ((vslot ^ vt_swz(dd)) << 4) + 8) = make_uint2(0u, 0u);
    }
  } else {
    // Wave w owns ONE 16-column tile of each of Q, K and V (W_h rows 16 w .., 64 + 16 w .., 128 + 16 w ..): the four
    // waves do the same work in phase 2.  Q and K tiles are accumulated with the operand roles swapped (A slot = W
    // fragment): reg r of lane (lr, lg) is C[token 16 i + lr][d = 16 w + 4 lg + r] -- four consecutive columns of a
    // token row, what the row-major Q / K images want.  The V tile keeps the roles (A slot = U fragment): reg r is
    // V[token 16 i + 4 lg + r][d = 16 w + lr] -- four consecutive TOKENS of one column, what V^T wants.
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias of this lane's columns: lands under the K loop
    const int dq = 16 * wave + 4 * lg;              // first of the lane's 4 Q / K columns
    const int dv = 16 * wave + lr;                  // the lane's V column
    float4 bq4 = make_float4(0.f, 0.f, 0.f, 0.f), bk4 = bq4;
    float bv1 = 0.f;
    if (bqkv) {
      bq4 = *(const float4*)(bqkv + head * HD + dq);
      bk4 = *(const float4*)(bqkv + D + head * HD + dq);
      bv1 = bqkv[2 * D + head * HD + dv];
    }
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      if (kt == 0 || kt == 1 || kt == 4) RRT_TRACE_MARK();   // compute [2,3,4] barrier kt passed
      const char* As = smem + slot * STAGE_B;
      const char* Bs = As + BM * ROWB;
      slot = slot == NSTG - 1 ? 0 : slot + 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        Frag a8[MT], b8[NT];
        const int cslot = 4 * kk + lg;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = 64 * j + 16 * wave + lr;
          b8[j] = *(const Frag*)(Bs + row * ROWB + ((cslot ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = i * 16 + lr;
          a8[i] = *(const Frag*)(As + row * ROWB + ((cslot ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[i][0] = H::mfma(b8[0], a8[i], acc[i][0]);
          acc[i][1] = H::mfma(b8[1], a8[i], acc[i][1]);
          acc[i][2] = H::mfma(a8[i], b8[2], acc[i][2]);
        }
      }
    }
    RRT_TRACE_MARK();                               // compute [5] last projection MFMA issued
    // ================================================================ phase 2: Q (fp32), K, V^T (16-bit) -> LDS
    __syncthreads();                                // every wave is done with the staging ring
    RRT_TRACE_MARK();                               // compute [6]
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = i * 16 + lr;
      *(float4*)(QF + m * 256 + (((dq >> 2) ^ (m & 15)) << 4)) =
          make_float4((acc[i][0][0] + bq4.x) * q_scale, (acc[i][0][1] + bq4.y) * q_scale,
                      (acc[i][0][2] + bq4.z) * q_scale, (acc[i][0][3] + bq4.w) * q_scale);
      *(uint2*)(KS + m * ROWB + (((dq >> 3) ^ ((m >> 1) & 7)) << 4) + ((dq & 4) << 1)) =
          pack4<PREC>(acc[i][1][0] + bk4.x, acc[i][1][1] + bk4.y, acc[i][1][2] + bk4.z, acc[i][1][3] + bk4.w);
      // tokens 16 i + 4 lg + r, r = 0..3 -> positions 32 (i / 2) + 8 lg + 4 (i % 2) + r of V^T row dv: 8 bytes
      const int vslot = 4 * (i >> 1) + lg;
      *(uint2*)(VT + dv * VT_PITCH + ((vslot ^ vt_swz(dv)) << 4) + ((i & 1) << 3)) =
          pack4<PREC>(acc[i][2][0] + bv1, acc[i][2][1] + bv1, acc[i][2][2] + bv1, acc[i][2][3] + bv1);
    }
  }
  __syncthreads();                                  // Q / K / V^T tiles complete
  RRT_TRACE_MARK();                                 // compute [7] / loader [8]: tiles in LDS

  // ================================================================== phase 3: EPEG stencil -> Q~ (16-bit)
  // thread = (fp32 slot s of 16 = 4 head-dim columns, run g of RUN consecutive query rows); all 8 waves.
  // Source row j of a run (j = 0 .. RUN + k - 2, row r0 - k/2 + j) meets output o with tap t = j - o: the RUN
  // weights slide by one per source row.  Rows are taken four at a time with their loads issued together (a row
  // per trip was a chain of dependent LDS round trips: 5.5K cycles for ~1.5K of work); rows outside the region
  // [0, P) read as zero (the EPEG convolution zero-pads at the region edge), taps outside [0, k) are zero in the
  // table.
  {
    static_assert(RUN - 1 <= TAP_OFF && (TAP_OFF - (RUN - 1)) % 4 == 0 && 2 * RUN + 90 < 128, "tap table range / alignment");
    const int half = epeg_k >> 1;
    const int s = tid & 15, g = tid >> 4;
    const int r0 = g * RUN;
    float4 out[RUN];
#pragma unroll
    for (int o = 0; o < RUN; ++o) out[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < BM) {
      const int nsrc = RUN + 2 * half;
      // taps tp[j0 - RUN + 1 .. j0 + 3] of a four-row group: one broadcast read (the address is the same in every
      // lane), then compile-time indices -- row u meets output o with T[u + RUN - 1 - o]
      constexpr int NTV = (RUN + 3 + 3) / 4;
      const float4* tp4 = (const float4*)(taps + TAP_OFF - (RUN - 1));
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
          v[u] = *(const float4*)(QF + rc * 256 + ((s ^ (rc & 15)) << 4));
          if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int o = 0; o < RUN; ++o) {
            const float wt = T[u + RUN - 1 - o];
            out[o].x += wt * v[u].x; out[o].y += wt * v[u].y; out[o].z += wt * v[u].z; out[o].w += wt * v[u].w;
          }
      }
    }
    __syncthreads();                                // all reads of Q done
    if (r0 < BM) {
#pragma unroll
      for (int o = 0; o < RUN; ++o) {
        const int m = r0 + o;
        if (m < BM)
          *(uint2*)(QT + m * ROWB + (((s >> 1) ^ ((m >> 1) & 7)) << 4) + ((s & 1) << 3)) =
              pack4<PREC>(out[o].x, out[o].y, out[o].z, out[o].w);
      }
    }
  }
  __syncthreads();
  RRT_TRACE_MARK();                                 // compute [8] / loader [9]: Q~ built

  // ================================================================== phase 4: attention from LDS
  for (int t = wave; t < MT; t += 8) {
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
      Frag a[MT];
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) {
        const int row = jt * 16 + lr;
        a[jt] = *(const Frag*)(KS + row * ROWB + (((4 * kk + lg) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) s[jt] = H::mfma(a[jt], bq[kk], s[jt]);
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
    cmax = max_xor32(max_xor16(cmax));             // VALU lane swaps (common.h), not ds_bpermute
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
    // O^T = V^T P^T: A = V^T rows d = 4 a + c (a = lr), B = P^T straight from the score registers
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
    if (i < P) {
      uint4 lo, hi;
      uint2 q0 = pack4<PREC>(oacc[0][0] * inv, oacc[1][0] * inv, oacc[2][0] * inv, oacc[3][0] * inv);
      uint2 q1 = pack4<PREC>(oacc[0][1] * inv, oacc[1][1] * inv, oacc[2][1] * inv, oacc[3][1] * inv);
      uint2 q2 = pack4<PREC>(oacc[0][2] * inv, oacc[1][2] * inv, oacc[2][2] * inv, oacc[3][2] * inv);
      uint2 q3 = pack4<PREC>(oacc[0][3] * inv, oacc[1][3] * inv, oacc[2][3] * inv, oacc[3][3] * inv);
      lo = make_uint4(q0.x, q0.y, q1.x, q1.y);
      hi = make_uint4(q2.x, q2.y, q3.x, q3.y);
      uint16_t* dst = O + (size_t)(row0 + i) * D + head * HD + 16 * lg;
      *(uint4*)dst = lo;
      *(uint4*)(dst + 8) = hi;
    }
    RRT_TRACE_MARK();                               // tile: O stored
  }
}

template <int MT, int PREC>
hipError_t launch_mt(const uint16_t* U, const uint16_t* W, const float* bqkv, const float* pe_w, uint16_t* O,
                     int n_regions, int P, int D, int heads, int epeg_k, hipStream_t st) {
  constexpr int BM = 16 * MT;
  constexpr size_t RING = (size_t)(MT <= 13 ? 3 : 2) * (BM + BN) * ROWB, TILES = (size_t)BM * (256 + ROWB) + 64 * VT_PITCH;
  constexpr size_t LDS = (RING > TILES ? RING : TILES) + 512;          // + the EPEG tap table
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = rmsa_fused16_kernel<MT, PREC>;
  static OncePerDevice once;
  if (once.first())
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
  const float q_scale = 1.0f / sqrtf((float)HD);
  kern<<<dim3(heads * n_regions), dim3(512), LDS, st>>>(U, W, bqkv, pe_w, O, n_regions * P, P, D, heads,
                                                       pe_w ? epeg_k : 0, q_scale);
  return hipGetLastError();
}

}  // namespace

#ifdef RRT_TRACE
RRT_TRACE_DEFINE_READER(rrt_debug_trace_fused16)
#endif

bool rmsa_fused16_supported(int P, int D, int heads, int epeg_k) {
  static const bool off = rrt_tune_env("RRT_NO_FUSED16") != nullptr;
  if (off) return false;
  // one block holds a whole region (P <= 256 -> MT <= 16); 64-element K tiles; 32-bit DMA byte offsets
  return heads > 0 && D == heads * HD && D % 64 == 0 && P > 16 && P <= 256 && epeg_k >= 0 && epeg_k <= 63;
}

hipError_t launch_rmsa_fused16(const uint16_t* U, const uint16_t* W, const float* bqkv, const float* pe_w,
                               uint16_t* O, int n_regions, int P, int D, int heads, int epeg_k, int prec,
                               hipStream_t st) {
  if (prec != 1 && prec != 2) return hipErrorInvalidValue;
  if (rmsa_pair16_supported(n_regions, P, D, heads, pe_w ? epeg_k : 0))
    return launch_rmsa_pair16(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, prec, st);
#define RRT_FUSED16(MT_)                                                                                \
  return prec == 1 ? launch_mt<MT_, 1>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, st)          \
                   : launch_mt<MT_, 2>(U, W, bqkv, pe_w, O, n_regions, P, D, heads, epeg_k, st);
  if (P > 240) { RRT_FUSED16(16) }
  if (P > 208) { RRT_FUSED16(15) }
  if (P > 176) { RRT_FUSED16(13) }
  if (P > 144) { RRT_FUSED16(11) }
  if (P > 128) { RRT_FUSED16(9) }
  if (P > 112) { RRT_FUSED16(8) }
  if (P > 96) { RRT_FUSED16(7) }
  if (P > 64) { RRT_FUSED16(6) }
  if (P > 32) { RRT_FUSED16(4) }
  RRT_FUSED16(2)
#undef RRT_FUSED16
}
