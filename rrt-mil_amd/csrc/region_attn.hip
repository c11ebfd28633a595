// region_attn.hip -- per-region multi-head attention with the EPEG term, fp32 MFMA.
//
// Replaces InnerAttention.forward's core, modules/rmsa.py:103-122:
//     attn = (q*scale) @ k^T ; attn += Conv2d_dw(k x 1)(attn) ; softmax ; attn @ v
//
// Two identities remove the [R,h,P,P] score tensor the reference materialises:
//  (1) EPEG is a depth-wise 1-D convolution along the QUERY axis of S = Q K^T
//      (kernel (k,1), zero padded at the region edge, modules/rmsa.py:84,106-108).
//      A stencil over the row index of S acts on the left factor only:
//          S + T S = (Q + T Q) K^T ,  (T Q)[i,:] = sum_t w[t] Q[i+t-k/2,:]  (rows outside [0,P) = 0)
//      so it is applied to the [P,64] Q tile (15 taps x 64 dims per query) instead
//      of the [P,P] score tile.
//  (2) the conv bias adds one constant per head to every score of that head; row
//      softmax is shift invariant, so it drops out.
// What is left is plain softmax(Q~ K^T) V per (region, head), done flash-style:
//   * one wave = 16 queries; scores are computed TRANSPOSED (S^T = K Q~^T, A = K tile,
//     B = Q~ fragments held in registers) with v_mfma_f32_16x16x4_f32, so lane l holds,
//     for query (l&15), keys 4*(l>>4)+reg of every 16-key tile: the row softmax needs
//     only 2 cross-lane steps (xor 16, xor 32) and P^T is already in the A-operand
//     layout for P.V (no LDS round trip);
//   * K/V stream through LDS in chunks of 16*TC keys (16-B DMA, double buffered,
//     one barrier per chunk), online softmax across chunks (any P);
//   * K image XOR-swizzled (slot ^ (row&15)) via the DMA source address so the
//     row-per-lane ds_read_b128 is bank-conflict free; V is read row-contiguous;
//   * a lane's float4 of V covers 4 head-dim tiles (d = 4*(l&15)+c), so the O tile
//     comes out as one float4 per row: coalesced 256-B row stores.
// Head dim 64 only (dim/heads == 64); other head dims use region_attn_generic.
#include <stdio.h>
#include <stdlib.h>

#include "internal.h"

namespace {

constexpr int HD = 64;
constexpr float NEG_BIG = -3.0e38f;

// DIRECT = true: no K/V ring in LDS -- every wave loads its K and V MFMA fragments straight from
// global memory (L2-resident: the K/V of a (region, head) is P x 512 B and shared by P/16 waves).
// The chunk loop then has no DMA, no barrier and no LDS traffic; LDS only stages the Q rows.
template <int TC, bool DIRECT>
__global__ __launch_bounds__(256) void region_attn_kernel(const float* __restrict__ qkv,
                                                          const float* __restrict__ pe_w,
                                                          float* __restrict__ o, int P, int dim,
                                                          int epeg_k) {
  constexpr int KC = 16 * TC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = (float*)smem;   // [2][K: KC*64 | V: KC*64]
  constexpr int STAGE = 2 * KC * HD;
  constexpr float LOG2E = 1.4426950408889634f;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  const int head = blockIdx.y, reg = blockIdx.z;
  const int ld = 3 * dim;
  const size_t rbase = (size_t)reg * P;
  const float* qbase = qkv + rbase * ld + head * HD;
  const float* kbase = qbase + dim;
  const float* vbase = qbase + 2 * dim;

  const int ib0 = blockIdx.x * nw * 16;           // first query of this block
  const int i0 = ib0 + wave * 16;                 // first query of this wave
  const bool active = i0 < P;
  const int qi = i0 + (lane & 15);
  const int lg = lane >> 4;

  RRT_TRACE_INIT(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * nw + wave);
  RRT_TRACE_MARK();                                   // [1] kernel entry
  const unsigned lds_b = lds_addr_of(lds);
  auto stage = [&](int ch, unsigned buf) {
    const int j0 = ch * KC;
    for (int q = wave; q < KC / 4; q += nw) {          // 64 slots = 4 rows per wave-instruction
      int S = q * 64 + lane;
      int row = S >> 4, p = S & 15;
      int j = j0 + row;
      j = j < P ? j : P - 1;                            // tail keys: re-read last row, masked below
      dma16(kbase + (size_t)j * ld + ((p ^ (row & 15)) << 2), buf + q * 1024);
      dma16(vbase + (size_t)j * ld + (p << 2), buf + (KC * HD + q * 256) * 4);
    }
  };
  if (!DIRECT) stage(0, lds_b);

  // ---- Q~ fragments (B operand): bq[c] = log2(e) * Q~[qi][16c + 4*lg .. +3] -------------
  // Q~ = Q + EPEG stencil over the query axis (see header).  The Q rows the block needs
  // ([ib0 - k/2, ib0 + 16*nw + k/2) clipped to the region) are staged once through LDS --
  // in the second K/V buffer, which is idle until chunk 1 is prefetched -- with the same
  // XOR swizzle as K, so the 15 taps are conflict-free ds_read_b128 instead of global loads.
  float4 bq[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bq[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (epeg_k <= 0) {   // no EPEG (e.g. the CR-MSA inner attention): Q fragments straight from global
    if (active && qi < P) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bq[c] = *(const float4*)(qbase + (size_t)qi * ld + 16 * c + 4 * lg);
        bq[c].x *= LOG2E; bq[c].y *= LOG2E; bq[c].z *= LOG2E; bq[c].w *= LOG2E;
      }
    }
  } else {
    const int half = epeg_k >> 1;
    const int r_lo = max(ib0 - half, 0);
    const int r_hi = min(ib0 + nw * 16 + half, P);          // exclusive
    const int nrows = r_hi - r_lo;                           // <= 2*KC (checked at launch)
    const unsigned qbuf = DIRECT ? lds_b : lds_b + STAGE * 4;
    for (int q = wave; q * 4 < nrows; q += nw) {
      int S = q * 64 + lane;
      int row = S >> 4, p = S & 15;
      int r = r_lo + row;
      r = r < P ? r : P - 1;
      dma16(qbase + (size_t)r * ld + ((p ^ (row & 15)) << 2), qbuf + q * 1024);
    }
    wait_vm0();
    __syncthreads();
    RRT_TRACE_MARK();                                 // [2] Q rows (+chunk 0) landed
    if (active && qi < P) {
      const float* Qs = DIRECT ? lds : lds + STAGE;
      const int lrow = qi - r_lo;                            // this lane's query row in the LDS image
#pragma unroll
      for (int c = 0; c < 4; ++c)
        bq[c] = *(const float4*)(Qs + lrow * HD + (((4 * c + lg) ^ (lrow & 15)) << 2));
      if (epeg_k > 0) {
        const float* w = pe_w + head * epeg_k;
        for (int t = 0; t < epeg_k; ++t) {
          const int r = qi + t - half;
          if (r >= 0 && r < P) {
            const float wt = w[t];
            const int rr = r - r_lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float4 v = *(const float4*)(Qs + rr * HD + (((4 * c + lg) ^ (rr & 15)) << 2));
              bq[c].x += wt * v.x; bq[c].y += wt * v.y; bq[c].z += wt * v.z; bq[c].w += wt * v.w;
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {   // scores in log2 units: softmax via exp2
        bq[c].x *= LOG2E; bq[c].y *= LOG2E; bq[c].z *= LOG2E; bq[c].w *= LOG2E;
      }
    }
    // the barrier at the top of chunk 0 orders these reads before chunk 1 overwrites the buffer
  }

  RRT_TRACE_MARK();                                   // [3] Q~ fragments built
  float m_run = NEG_BIG, l_run = 0.f;
  f32x4 oacc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nch = (P + KC - 1) / KC;
  if (DIRECT && !active) return;
  for (int ch = 0; ch < nch; ++ch) {
    const float* Ks = nullptr;
    const float* Vs = nullptr;
    if (!DIRECT) {
      wait_vm0();
      __syncthreads();
      float* cur = lds + (ch & 1) * STAGE;
      if (ch + 1 < nch) stage(ch + 1, lds_b + ((ch + 1) & 1) * STAGE * 4);
      if (!active) continue;
      Ks = cur;
      Vs = cur + KC * HD;
    }
    const int j0 = ch * KC;
    RRT_TRACE_MARK();                                 // [4+5c] chunk barrier passed

    // S^T tiles: s[jt][r] = log2e * score(query lane&15, key j0 + 16*jt + 4*lg + r).
    // jt innermost: consecutive MFMAs hit different accumulators (40-cycle dependent latency)
    f32x4 s[TC];
#pragma unroll
    for (int jt = 0; jt < TC; ++jt) s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 a[TC];
#pragma unroll
      for (int jt = 0; jt < TC; ++jt) {
        const int row = jt * 16 + (lane & 15);
        if (DIRECT) {
          int j = j0 + row;
          j = j < P ? j : P - 1;                      // tail keys: masked below
          a[jt] = *(const float4*)(kbase + (size_t)j * ld + 16 * c + 4 * lg);
        } else {
          a[jt] = *(const float4*)(Ks + row * HD + (((4 * c + lg) ^ (row & 15)) << 2));
        }
      }
#pragma unroll
      for (int jt = 0; jt < TC; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].x, bq[c].x, s[jt], 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < TC; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].y, bq[c].y, s[jt], 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < TC; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].z, bq[c].z, s[jt], 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < TC; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jt].w, bq[c].w, s[jt], 0, 0, 0);
    }
    RRT_TRACE_MARK();                                 // [5+5c] S^T MFMAs issued
    if (j0 + KC > P) {   // tail chunk only: mask keys >= P
#pragma unroll
      for (int jt = 0; jt < TC; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + jt * 16 + 4 * lg + r >= P) s[jt][r] = NEG_BIG;
    }
    float cmax = NEG_BIG;
#pragma unroll
    for (int jt = 0; jt < TC; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) cmax = fmaxf(cmax, s[jt][r]);
    cmax = max_xor32(max_xor16(cmax));             // VALU lane swaps (common.h), not ds_bpermute
    asm volatile("" :: "v"(cmax));
    RRT_TRACE_MARK();                                 // [6+5c] scores complete + row max reduced
    const float m_new = fmaxf(m_run, cmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int jt = 0; jt < TC; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __builtin_amdgcn_exp2f(s[jt][r] - m_new);
        s[jt][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;   // per-lane partial (this lane's keys); lanes of a query share alpha
    if (ch > 0) {
      // rescale O: row r' of the O tile is query 4*lg + r', whose alpha lives in lane (4*lg + r')
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float ar = __shfl(alpha, 4 * lg + r);
#pragma unroll
        for (int c = 0; c < 4; ++c) oacc[c][r] *= ar;
      }
    }
    asm volatile("" :: "v"(l_run), "v"(oacc[0][0]));
    RRT_TRACE_MARK();                                 // [7+5c] softmax + rescale done
    // O += P V   (A = P^T regs, B = V rows; float4 of V = 4 head-dim tiles)
#pragma unroll
    for (int jt = 0; jt < TC; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = jt * 16 + 4 * lg + r;
        float4 v;
        if (DIRECT) {
          int j = j0 + row;
          j = j < P ? j : P - 1;                      // p = 0 for masked keys
          v = *(const float4*)(vbase + (size_t)j * ld + ((lane & 15) << 2));
        } else {
          v = *(const float4*)(Vs + row * HD + ((lane & 15) << 2));
        }
        const float p = s[jt][r];
        oacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.x, oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.y, oacc[1], 0, 0, 0);
        oacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.z, oacc[2], 0, 0, 0);
        oacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.w, oacc[3], 0, 0, 0);
      }
    RRT_TRACE_MARK();                                 // [8+5c] PV MFMAs issued
  }
  if (!active) return;
  RRT_TRACE_MARK();                                   // last PV MFMAs issued
  const float l_tot = sum_xor32(sum_xor16(l_run));
  const float inv = 1.0f / l_tot;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ir = __shfl(inv, 4 * lg + r);
    const int i = i0 + 4 * lg + r;
    if (i < P) {
      float4 out = make_float4(oacc[0][r] * ir, oacc[1][r] * ir, oacc[2][r] * ir, oacc[3][r] * ir);
      *(float4*)(o + (rbase + i) * dim + head * HD + ((lane & 15) << 2)) = out;
    }
  }
}

// ---- regions of 209..256 tokens (bags of ~12.6-16 k tokens at region_num = 8), whole region resident ------------------
// The streaming kernel above re-stages K and V once per 64 queries and rescales its accumulators at every 48-key chunk
// behind a block-wide barrier (P = 256: 99 us, 0.55 of the fp32 MFMA rate, the DMA round trip of the next chunk not
// covered by one chunk's 96 MFMAs).  Here a block = one (region, head), sixteen waves = sixteen 16-query tiles, and the
// tiles live in two 64 KiB halves of LDS:  Q | K  ->  (Q~ fragments to registers, barrier)  ->  V | K.  The scores of a
// query tile against ALL keys stay in registers (s[MT] = 64 VGPRs), so the softmax is single-pass, there is no
// rescaling and no barrier inside the products; the V image lands over Q while S^T = K Q~^T runs.  One block per CU,
// four waves per SIMD (<= 128 VGPRs).  Same transposed-score layout, swizzle and DMA path as above.
template <int MT>
__global__ __launch_bounds__(1024) void region_attn_resident_kernel(const float* __restrict__ qkv,
                                                                   const float* __restrict__ pe_w,
                                                                   float* __restrict__ o, int P, int dim, int epeg_k) {
  constexpr int BM = 16 * MT;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const Xs = (float*)smem;                  // Q, then V (row-contiguous, not swizzled)
  float* const Ks = Xs + BM * HD;                  // K, XOR-swizzled 16-byte slots
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int head = blockIdx.x, reg = blockIdx.y;
  const int ld = 3 * dim;
  const size_t rbase = (size_t)reg * P;
  const float* qbase = qkv + rbase * ld + head * HD;
  const float* kbase = qbase + dim;
  const float* vbase = qbase + 2 * dim;
  const unsigned xs_b = lds_addr_of(Xs), ks_b = lds_addr_of(Ks);
  // one wave-instruction = 4 rows of 256 bytes; rows >= P re-read the last row (masked / never used below)
  for (int q = wave; q < BM / 4; q += 16) {
    const int S = q * 64 + lane, row = S >> 4, p = S & 15;
    const int j = row < P ? row : P - 1;
    dma16(qbase + (size_t)j * ld + ((p ^ (row & 15)) << 2), xs_b + q * 1024);
    dma16(kbase + (size_t)j * ld + ((p ^ (row & 15)) << 2), ks_b + q * 1024);
  }
  wait_vm0();
  __syncthreads();
  const int i0 = wave * 16;
  const bool active = wave < MT && i0 < P;
  const int qi = i0 + lr;
  float4 bq[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bq[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active && qi < P) {
#pragma unroll
    for (int c = 0; c < 4; ++c) bq[c] = *(const float4*)(Xs + qi * HD + (((4 * c + lg) ^ (qi & 15)) << 2));
    if (epeg_k > 0) {
      const float* w = pe_w + head * epeg_k;
      const int half = epeg_k >> 1;
      for (int t = 0; t < epeg_k; ++t) {
        const int r = qi + t - half;
        if (r >= 0 && r < P) {
          const float wt = w[t];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float4 v = *(const float4*)(Xs + r * HD + (((4 * c + lg) ^ (r & 15)) << 2));
            bq[c].x += wt * v.x; bq[c].y += wt * v.y; bq[c].z += wt * v.z; bq[c].w += wt * v.w;
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { bq[c].x *= LOG2E; bq[c].y *= LOG2E; bq[c].z *= LOG2E; bq[c].w *= LOG2E; }
  }
  __syncthreads();                                  // every wave has its Q~ fragments: the Q image is dead
  for (int q = wave; q < BM / 4; q += 16) {        // V over Q, rows as they lie in memory
    const int S = q * 64 + lane, row = S >> 4, p = S & 15;
    const int j = row < P ? row : P - 1;
    dma16(vbase + (size_t)j * ld + (p << 2), xs_b + q * 1024);
  }
  // S^T tiles against every key tile: s[jt][r] = score(query lr, key 16 jt + 4 lg + r), log2 units
  f32x4 s[MT];
#pragma unroll
  for (int jt = 0; jt < MT; ++jt) s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (active) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int j0 = 0; j0 < MT; j0 += 4) {          // four key tiles' fragments in flight (16 VGPRs)
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = (j0 + u < MT ? j0 + u : MT - 1) * 16 + lr;
          a[u] = *(const float4*)(Ks + row * HD + (((4 * c + lg) ^ (row & 15)) << 2));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j0 + u < MT) s[j0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, bq[c].x, s[j0 + u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j0 + u < MT) s[j0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, bq[c].y, s[j0 + u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j0 + u < MT) s[j0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, bq[c].z, s[j0 + u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j0 + u < MT) s[j0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, bq[c].w, s[j0 + u], 0, 0, 0);
      }
    }
    if (P < BM) {
#pragma unroll
      for (int jt = 0; jt < MT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (jt * 16 + 4 * lg + r >= P) s[jt][r] = NEG_BIG;     // keys past the region
    }
  }
  float inv = 0.f;
  if (active) {
    float cmax = NEG_BIG;
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
    inv = 1.0f / sum_xor32(sum_xor16((a4[0] + a4[1]) + (a4[2] + a4[3])));
  }
  wait_vm0();
  __syncthreads();                                  // V image complete
  if (!active) return;
  f32x4 oacc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jt = 0; jt < MT; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = jt * 16 + 4 * lg + r;
      const float4 v = *(const float4*)(Xs + row * HD + (lr << 2));
      const float p = s[jt][r];
      oacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.x, oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.y, oacc[1], 0, 0, 0);
      oacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.z, oacc[2], 0, 0, 0);
      oacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v.w, oacc[3], 0, 0, 0);
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ir = __shfl(inv, 4 * lg + r);
    const int i = i0 + 4 * lg + r;
    if (i < P)
      *(float4*)(o + (rbase + i) * dim + head * HD + (lr << 2)) =
          make_float4(oacc[0][r] * ir, oacc[1][r] * ir, oacc[2][r] * ir, oacc[3][r] * ir);
  }
}

// Generic head-dim fallback (hd != 64: e.g. crmsa_heads=1 -> hd=dim, or dim=64 -> hd=8).
// One wave per query; VALU only (the published TCGA-BRCA-R50 / NSCLC-PLIP configs run CR-MSA's inner attention
// with crmsa_heads=1: 3 x 64 queries of head dim 512).
__global__ __launch_bounds__(256) void region_attn_generic_kernel(const float* __restrict__ qkv,
                                                                  const float* __restrict__ pe_w,
                                                                  float* __restrict__ o, int P,
                                                                  int dim, int hd, int epeg_k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Ppad = (P + 3) & ~3;
  float* qs = (float*)smem + wave * (hd + Ppad);   // q~ [hd]
  float* ps = qs + hd;                               // scores / probabilities [P]
  const int head = blockIdx.y, reg = blockIdx.z;
  const int qi = blockIdx.x * 4 + wave;
  if (qi >= P) return;
  const int ld = 3 * dim;
  const size_t rbase = (size_t)reg * P;
  const float* qbase = qkv + rbase * ld + head * hd;
  const float* kbase = qbase + dim;
  const float* vbase = qbase + 2 * dim;
  const int half = epeg_k >> 1;
  for (int d = lane; d < hd; d += 64) {
    float a = qbase[(size_t)qi * ld + d];
    for (int t = 0; t < epeg_k; ++t) {
      int r = qi + t - half;
      if (r >= 0 && r < P) a += pe_w[head * epeg_k + t] * qbase[(size_t)r * ld + d];
    }
    qs[d] = a;
  }
  __builtin_amdgcn_wave_barrier();
  float mx = NEG_BIG;
  if ((hd & 3) == 0) {
    // keys serially, the head dim across the lanes in 16-byte pieces: every K row is read as contiguous
    // 256-float runs (the lane-per-key form touched 64 rows per load instruction); 8 keys = up to 16
    // independent 16-byte loads in flight per trip (the loop is a chain of L2 round trips otherwise)
    for (int j0 = 0; j0 < P; j0 += 8) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = 0.f;
      for (int d = lane * 4; d < hd; d += 256) {
        const float4 q4 = *(const float4*)(qs + d);
        float4 k4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u < P ? j0 + u : P - 1;
          k4[u] = *(const float4*)(kbase + (size_t)j * ld + d);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += (q4.x * k4[u].x + q4.y * k4[u].y) + (q4.z * k4[u].z + q4.w * k4[u].w);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float v = wave_sum(a[u]);                  // wave-uniform
        if (j0 + u < P) {
          if (lane == 0) ps[j0 + u] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  } else {
    for (int j = lane; j < P; j += 64) {
      const float* kr = kbase + (size_t)j * ld;
      float a = 0.f;
      for (int d = 0; d < hd; ++d) a += qs[d] * kr[d];
      ps[j] = a;
      mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
  }
  float sum = 0.f;
  for (int j = lane; j < P; j += 64) {
    float p = __expf(ps[j] - mx);
    ps[j] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  __builtin_amdgcn_wave_barrier();
  if ((hd & 3) == 0) {
    // 4 head-dim columns per lane, 16 value rows in flight per trip
    for (int d = lane * 4; d < hd; d += 256) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j0 = 0; j0 < P; j0 += 16) {
        float4 v4[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = j0 + u < P ? j0 + u : P - 1;
          v4[u] = *(const float4*)(vbase + (size_t)j * ld + d);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float p = j0 + u < P ? ps[j0 + u] : 0.f;
          acc.x += p * v4[u].x; acc.y += p * v4[u].y; acc.z += p * v4[u].z; acc.w += p * v4[u].w;
        }
      }
      *(float4*)(o + (rbase + qi) * dim + head * hd + d) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
    return;
  }
  for (int d = lane; d < hd; d += 64) {
    float a = 0.f;
    for (int j = 0; j < P; ++j) a += ps[j] * vbase[(size_t)j * ld + d];
    o[(rbase + qi) * dim + head * hd + d] = a * inv;
  }
}

// P = 64 without EPEG (the MSA over CR-MSA's 64 region representatives, modules/rmsa.py:322): nothing goes through LDS and
// there is no barrier -- a wave owns 16 queries and issues every load of its Q, K and V fragments up front, straight into
// the MFMA operand layouts (K: a float4 of head dim per lane and 16-dim chunk, its four elements feed four consecutive
// MFMAs, which permutes the reduction identically on both sides; V: one key row element per lane), so the kernel is one
// memory round trip + 128 MFMAs: 6.4 us against 7.2 for the K/V-ring kernel on these 24 small problems.
// grid (heads, regions), 4 waves; q already scaled by the projection's epilogue.
// Register budget (round 4): with every fragment requested up front the kernel held 164 VGPRs, more than the 144 a SIMD has
// left beside two waves of another bag's fused R-MSA kernel -- its blocks then WAITED for a fused block to retire and took the
// CU slot of that kernel's next block (measured beside the fused kernel: 12.4 us of its time per launch, twice this kernel's
// own 6.1 us).  Now the second half of V is requested behind S^T (into the registers K leaves): <= 128 VGPRs.
__global__ __launch_bounds__(256, 4) void region_attn64_kernel(const float* __restrict__ qkv, float* __restrict__ o, int dim) {
  const int LDQ = 3 * dim;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int head = blockIdx.x, reg = blockIdx.y;
  const float* base = qkv + (size_t)reg * 64 * LDQ + head * HD;
  float4 qf[4], kf[4][4];
  float vf[4][4][4];                                       // [kt][j][dt]: V[key 16 kt + 4 g + j][d 16 dt + r]
#pragma unroll
  for (int i = 0; i < 4; ++i) qf[i] = *(const float4*)(base + (size_t)(16 * wave + r) * LDQ + 16 * i + 4 * g);
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int i = 0; i < 4; ++i) kf[kt][i] = *(const float4*)(base + (size_t)(16 * kt + r) * LDQ + dim + 16 * i + 4 * g);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vf[kt][j][dt] = base[(size_t)(16 * kt + 4 * g + j) * LDQ + 2 * dim + 16 * dt + r];
  __builtin_amdgcn_sched_barrier(0);                      // these loads are issued before the first MFMA waits
  // S^T = K q^T: st[kt][j] = score(query 16 wave + r, key 16 kt + 4 g + j)
  f32x4 st[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][i].x, qf[i].x, st[kt], 0, 0, 0);
      st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][i].y, qf[i].y, st[kt], 0, 0, 0);
      st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][i].z, qf[i].z, st[kt], 0, 0, 0);
      st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][i].w, qf[i].w, st[kt], 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kt = 2; kt < 4; ++kt)                          // second half of V: in flight under the softmax
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vf[kt][j][dt] = base[(size_t)(16 * kt + 4 * g + j) * LDQ + 2 * dim + 16 * dt + r];
  __builtin_amdgcn_sched_barrier(0);
  float mx = NEG_BIG;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, st[kt][j]);
  mx = max_xor32(max_xor16(mx));
  float se = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st[kt][j] = __expf(st[kt][j] - mx);
      se += st[kt][j];
    }
  se = sum_xor32(sum_xor16(se));
  const float inv = 1.0f / se;
  // O^T = V^T P^T: a = V[key][d 16 dt + r], b = this lane's probability for that key (query r)
  f32x4 ot[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ot[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][j][dt], st[kt][j], ot[dt], 0, 0, 0);
  float* orow = o + (size_t)(reg * 64 + 16 * wave + r) * dim + head * HD;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    *(float4*)(orow + 16 * dt + 4 * g) = make_float4(ot[dt][0] * inv, ot[dt][1] * inv, ot[dt][2] * inv, ot[dt][3] * inv);
}

}  // namespace

#ifdef RRT_TRACE
RRT_TRACE_DEFINE_READER(rrt_debug_trace_attn)
#endif

hipError_t launch_region_attention(const float* qkv, const float* pe_w, float* o, int n_regions,
                                   int P, int dim, int heads, int epeg_k, hipStream_t st) {
  const int hd = dim / heads;
  if (pe_w == nullptr) epeg_k = 0;
  if (hd != HD) {
    const int Ppad = (P + 3) & ~3;
    size_t lds = (size_t)4 * (hd + Ppad) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)region_attn_generic_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    region_attn_generic_kernel<<<dim3((P + 3) / 4, heads, n_regions), 256, lds, st>>>(
        qkv, pe_w, o, P, dim, hd, epeg_k);
    return hipGetLastError();
  }
  static const bool no64 = rrt_tune_env("RRT_NO_ATTN64") != nullptr;
  if (P == 64 && epeg_k == 0 && !no64) {
    region_attn64_kernel<<<dim3(heads, n_regions), 256, 0, st>>>(qkv, o, dim);
    return hipGetLastError();
  }
  static const bool no_resident = rrt_tune_env("RRT_NO_ATTN_RESIDENT") != nullptr;
  if (P > 208 && P <= 256 && !no_resident) {        // whole region resident, single-pass softmax
    if (P > 240) {
      auto kern = region_attn_resident_kernel<16>;
      constexpr int LDS = 2 * 256 * HD * 4;
      static OncePerDevice once;
      if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      kern<<<dim3(heads, n_regions), 1024, LDS, st>>>(qkv, pe_w, o, P, dim, epeg_k);
    } else {
      auto kern = region_attn_resident_kernel<15>;
      constexpr int LDS = 2 * 240 * HD * 4;
      static OncePerDevice once;
      if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      kern<<<dim3(heads, n_regions), 1024, LDS, st>>>(qkv, pe_w, o, P, dim, epeg_k);
    }
    return hipGetLastError();
  }
  const int ntiles = (P + 15) / 16;
  // waves per block: avoid idle waves (P=144 -> 9 tiles -> 3 waves x 3 blocks)
  int nw = 4;
  if (ntiles < 4) nw = ntiles;
  else if (ntiles % 4 != 0 && ntiles % 3 == 0) nw = 3;
  // the block's Q rows (16*nw queries + EPEG halo) are staged in one K/V stage buffer (2*KC rows)
  {
    const int kc = P <= 16 ? 16 : (P <= 32 ? 32 : 48);
    while (nw > 1 && (P < 16 * nw + epeg_k - 1 ? P : 16 * nw + epeg_k - 1) > 2 * kc) --nw;
    if ((P < 16 * nw + epeg_k - 1 ? P : 16 * nw + epeg_k - 1) > 2 * kc) return hipErrorInvalidValue;
  }
  int tc_force = 0;
  if (const char* e = rrt_tune_env("RRT_ATTN_CFG")) {   // tuning hook: "tc,nw" (rejected if the Q rows do not fit)
    int a = 0, b = 0;
    if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && a <= 4 && b >= 1 && b <= 4) {
      const int need = P < 16 * b + epeg_k - 1 ? P : 16 * b + epeg_k - 1;
      if (epeg_k <= 0 || need <= 2 * 16 * a) { tc_force = a; nw = b; }
    }
  }
  const int nqb = (ntiles + nw - 1) / nw;
  dim3 grid(nqb, heads, n_regions), block(nw * 64);
  static const int direct = rrt_tune_env("RRT_ATTN_DIRECT") ? atoi(rrt_tune_env("RRT_ATTN_DIRECT")) : 0;
  if (direct) {
    // LDS = the Q staging rows only (16*nw queries + halo), 4-row granularity
    const int qrows = epeg_k > 0 ? (((P < 16 * nw + epeg_k - 1 ? P : 16 * nw + epeg_k - 1) + 3) & ~3) : 4;
    const size_t lds = (size_t)qrows * HD * 4;
    if (direct == 1) region_attn_kernel<1, true><<<grid, block, lds, st>>>(qkv, pe_w, o, P, dim, epeg_k);
    else if (direct == 2) region_attn_kernel<2, true><<<grid, block, lds, st>>>(qkv, pe_w, o, P, dim, epeg_k);
    else region_attn_kernel<3, true><<<grid, block, lds, st>>>(qkv, pe_w, o, P, dim, epeg_k);
  } else if (tc_force == 4) {
    auto kern = region_attn_kernel<4, false>;
    static OncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * HD * 4);
    kern<<<grid, block, 2 * 2 * 64 * HD * 4, st>>>(qkv, pe_w, o, P, dim, epeg_k);
  } else if (tc_force == 2) {
    region_attn_kernel<2, false><<<grid, block, 2 * 2 * 32 * HD * 4, st>>>(qkv, pe_w, o, P, dim, epeg_k);
  } else if (tc_force == 1) {
    region_attn_kernel<1, false><<<grid, block, 2 * 2 * 16 * HD * 4, st>>>(qkv, pe_w, o, P, dim, epeg_k);
  } else if (P <= 16) {
    region_attn_kernel<1, false><<<grid, block, 2 * 2 * 16 * HD * 4, st>>>(qkv, pe_w, o, P, dim, epeg_k);
  } else if (P <= 32) {
    region_attn_kernel<2, false><<<grid, block, 2 * 2 * 32 * HD * 4, st>>>(qkv, pe_w, o, P, dim, epeg_k);
  } else {
    region_attn_kernel<3, false><<<grid, block, 2 * 2 * 48 * HD * 4, st>>>(qkv, pe_w, o, P, dim, epeg_k);
  }
  return hipGetLastError();
}
