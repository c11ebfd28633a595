// attn_bwd.hip -- backward of the region attention core (row f2 building block).
//
// Forward (modules/rmsa.py:103-122, per region and head, Identity 1 of DESIGN.md):
//     Q~ = (I + T_w) q ,  S = Q~ K^T ,  A = softmax_rows(S) ,  O = A V          q = scale * q_raw (as stashed)
// Backward, one block per (region, head) with three tiles resident in LDS -- Q~, K and a third one that holds V
// during pass A and dO during pass B (3 x 53 KB at P = 208; the fragments a wave needs from the absent tile come
// from global / L2) --, the probabilities recomputed (nothing of size P x P is ever stored, forward or backward):
//     D_i  = <dO_i, O_i>                       (= rowsum(dA o A))
//     dV   = A^T dO        dA = dO V^T        dS = A o (dA - D)
//     dQ~  = dS K          dK = dS^T Q~
//     dq_raw = scale * (I + T_w)^T dQ~         (the same sliding-window stencil with the taps flipped)
//     dw_h[t] = sum_i <dQ~_i, q_{i + t - k/2}>  ;  the conv bias has an exactly-zero gradient (Identity 2)
// Pass A (a wave owns 16-query tiles, scores transposed exactly as in the forward kernels) produces the row
// statistics, D and dQ~ (parked in the dq columns of the output); pass B (a wave owns 16-key tiles; the same two
// MFMA helpers with the roles of Q~ and K exchanged) produces dK and dV.  Scores are in base-2 units (log2(e)
// folded into Q~).  Requires head dim 64 and P <= 208.
#include <stdlib.h>
#include <type_traits>

#include "internal.h"

namespace {

constexpr int HD = 64;
constexpr float NEG_BIG = -3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// rows of a swizzled [BM][64] tile as MFMA fragments: lane (lr, lg) <- tile[row0 + lr][4*(4c + lg) .. +3]
__device__ __forceinline__ void load_frags(const float* tile, int row, int lg, float4 (&f)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) f[c] = *(const float4*)(tile + row * HD + (((4 * c + lg) ^ (row & 15)) << 2));
}

// s[t][r] = sum_d X[16 t + 4 lg + r][d] * f[d]  for the fragment rows f (lane lr = one row of the other operand)
template <int MT>
__device__ __forceinline__ void tile_scores(const float* X, const float4 (&f)[4], int lr, int lg, f32x4 (&s)[MT]) {
  constexpr int CH = MT > 7 ? (MT + 1) / 2 : MT;     // row fragments in flight: at most 7 tiles (28 VGPRs)
#pragma unroll
  for (int t = 0; t < MT; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int t0 = 0; t0 < MT; t0 += CH) {
      float4 a[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int row = (t0 + u < MT ? t0 + u : MT - 1) * 16 + lr;
        a[u] = *(const float4*)(X + row * HD + (((4 * c + lg) ^ (row & 15)) << 2));
      }
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (t0 + u < MT) s[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, f[c].x, s[t0 + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (t0 + u < MT) s[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, f[c].y, s[t0 + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (t0 + u < MT) s[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, f[c].z, s[t0 + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (t0 + u < MT) s[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, f[c].w, s[t0 + u], 0, 0, 0);
    }
  }
}

// o[c][r] = sum_{t, rows} p[t][r'] * X[16 t + 4 lg + r'][4 lr + c]   ->  o[c][r] belongs to (row 4 lg + r of the
// p-operand's lane index, column 4 lr + c)
template <int MT>
__device__ __forceinline__ void tile_apply(const float* X, const f32x4 (&p)[MT], int lr, int lg, f32x4 (&o)[4]) {
  // two tiles' rows in flight (32 VGPRs): fully unrolled, the scheduler hoists all 4 MT row loads and spills
#pragma unroll 2
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = t * 16 + 4 * lg + r;
      const float4 v = *(const float4*)(X + row * HD + ((lr ^ (row & 15)) << 2));
      const float w = p[t][r];
      o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, v.x, o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, v.y, o[1], 0, 0, 0);
      o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, v.z, o[2], 0, 0, 0);
      o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, v.w, o[3], 0, 0, 0);
    }
}

// the same two products over a run of N consecutive row tiles starting at tile t0: the passes below walk the "other" operand in
// runs of three tiles so that only a run's scores are live next to the accumulators (nine tiles at once: 72 VGPRs of scores in
// pass B, and the register allocator serialised every LDS read with its use)
template <int N>
__device__ __forceinline__ void run_scores(const float* X, const float4 (&f)[4], int lr, int lg, int t0, f32x4 (&s)[N]) {
#pragma unroll
  for (int u = 0; u < N; ++u) s[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float4 a[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
      const int row = (t0 + u) * 16 + lr;
      a[u] = *(const float4*)(X + row * HD + (((4 * c + lg) ^ (row & 15)) << 2));
    }
#pragma unroll
    for (int u = 0; u < N; ++u) s[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, f[c].x, s[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < N; ++u) s[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, f[c].y, s[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < N; ++u) s[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, f[c].z, s[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < N; ++u) s[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, f[c].w, s[u], 0, 0, 0);
  }
}
template <int N>
__device__ __forceinline__ void run_apply(const float* X, const f32x4 (&p)[N], int lr, int lg, int t0, f32x4 (&o)[4]) {
#pragma unroll
  for (int u = 0; u < N; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (t0 + u) * 16 + 4 * lg + r;
      const float4 v = *(const float4*)(X + row * HD + ((lr ^ (row & 15)) << 2));
      const float w = p[u][r];
      o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, v.x, o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, v.y, o[1], 0, 0, 0);
      o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, v.z, o[2], 0, 0, 0);
      o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, v.w, o[3], 0, 0, 0);
    }
}

// NW waves per block: 6 (the forward fused kernel's schedule) up to 9 row tiles; 4 for 11 / 13 row tiles -- one wave
// per SIMD owns the whole 512-entry register file, and the two [MT]-long score arrays no longer spill.
template <int MT, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_bwd_kernel(const float* __restrict__ qkv,
                                                          const float* __restrict__ pe_w,
                                                          const float* __restrict__ O,
                                                          const float* __restrict__ dO,
                                                          float* __restrict__ dqkv, float* __restrict__ dpe_part,
                                                          int P, int D, int heads, int epeg_k, float q_scale) {
  constexpr int BM = 16 * MT;
  constexpr int TILE = BM * HD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qt = (float*)smem;             // q, then Q~ (x log2 e) in place
  float* Ks = Qt + TILE;
  float* Xs = Ks + TILE;                // V (pass A), dO (pass B), dQ~ (stencil adjoint)
  float* lse = Xs + TILE;               // [BM] row log-sum-exp, base 2
  float* dd = lse + BM;                 // [BM] D_i
  float* wred = dd + BM;                // [9][64]: the stencils' tap tables (256 floats)
  // Nine row tiles on four SIMDs are 3 : 2 : 2 : 2 whichever wave takes the ninth (traced: both passes waited ~20 K cycles
  // for SIMD 0).  HELP (MT = 9, NW = 12): waves 0..7 own tiles 0..7, and the ninth is SHARED OUT to waves 8..11 -- one per
  // SIMD -- by key tiles in pass A (its row statistics merged through LDS behind one extra barrier) and by query tiles in
  // pass B; the four partial dQ~ / dK / dV tiles are summed in wave order (fixed order: bit-reproducible).
  constexpr bool HELP = MT == 9 && NW == 12;
  constexpr int XT = MT - 1;
  float* const hstat = wred + 9 * 64;   // [4][32]: local max [16], local sum [16] of the shared-out tile's queries
  float* const hpart = hstat + 128;     // [4][2][16 x 64] partial tiles

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int reg = blockIdx.x / heads, head = blockIdx.x - reg * heads;
  const size_t row0 = (size_t)reg * P;
  const int ld = 3 * D;
  RRT_TRACE_INIT(blockIdx.x * 16 + wave);            // (up to twelve waves per block; 512 blocks fill the trace buffer)
  RRT_TRACE_MARK();                                 // [1] entry

  // Tap tables of the two stencils in LDS (index t + RUN - 1; zero outside [0, k)): forward log2(e) (w[t] + [t == k/2]), adjoint
  // q_scale (w[k - 1 - t] + [t == k/2]).  (The stencils fetched w[t] from global inside their row loops -- one dependent vector
  // load per source row, as the forward kernel once did: 8.7 K and 7.6 K cycles for two phases with ~3 K of work each.)
  constexpr int RUN = (BM * 16 + NW * 64 - 1) / (NW * 64);
  static_assert(2 * RUN + 62 < 128, "tap table range");
  const int half = epeg_k >> 1;
  const float* w = pe_w + head * epeg_k;
  float* const tapsF = wred;
  float* const tapsA = wred + 128;
  if (tid < 128) {
    const int t = tid - (RUN - 1);
    const bool in = epeg_k > 0 && t >= 0 && t < epeg_k;
    const float id = t == half ? 1.0f : 0.f;
    tapsF[tid] = ((in ? w[t] : 0.f) + id) * LOG2E;
    tapsA[tid] = ((in ? w[epeg_k - 1 - t] : 0.f) + id) * q_scale;
  }
  // ---- phase 0: q, k, v tiles -> LDS (XOR-swizzled 16-byte slots, rows >= P are zeros)
  for (int idx = tid; idx < BM * 16; idx += NW * 64) {
    const int m = idx >> 4, s = idx & 15;
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4, v4 = q4;
    if (m < P) {
      const float* src = qkv + (row0 + m) * ld + head * HD + 4 * s;
      q4 = *(const float4*)src;
      k4 = *(const float4*)(src + D);
      v4 = *(const float4*)(src + 2 * D);
    }
    const int off = m * HD + ((s ^ (m & 15)) << 2);
    *(float4*)(Qt + off) = q4;
    *(float4*)(Ks + off) = k4;
    *(float4*)(Xs + off) = v4;
  }
  __syncthreads();
  RRT_TRACE_MARK();                                 // [2] q, k, v in LDS
  // The first tile's dO and O fragment rows (every schedule gives wave w tile w first) are requested HERE, under the stencil
  // (at the head of pass A all waves of the block waited ~9 K cycles for them together, traced; in front of the LDS fill they
  // only delayed it: vector-memory loads return in order).
  float4 fg0[4], fo0[4];
  {
    const int m = (HELP && wave >= XT ? XT : wave) * 16 + lr;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool ok = (HELP || wave < MT) && m < P;
      fg0[c] = ok ? *(const float4*)(dO + (row0 + m) * D + head * HD + 4 * (4 * c + lg)) : make_float4(0.f, 0.f, 0.f, 0.f);
      fo0[c] = ok ? *(const float4*)(O + (row0 + m) * D + head * HD + 4 * (4 * c + lg)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }

  // ---- phase 1: Q~ = log2(e) * (I + T_w) q, in place (the forward's sliding-window stencil)
  {
    auto tap = [&](int t) { return tapsF[t + RUN - 1]; };
    const int s = tid & 15, g = tid >> 4;
    const int r0 = g * RUN;
    float4 out[RUN];
#pragma unroll
    for (int o = 0; o < RUN; ++o) out[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < BM) {
      const int lo = max(r0 - half, 0), hi = min(r0 + RUN - 1 + half, P - 1);
      float wr[RUN];
#pragma unroll
      for (int o = 0; o < RUN; ++o) wr[o] = tap(lo - r0 - o + half);
#pragma unroll 4
      for (int rr = lo; rr <= hi; ++rr) {
        const float4 v = *(const float4*)(Qt + rr * HD + ((s ^ (rr & 15)) << 2));
        const float wnext = tap(rr + 1 - r0 + half);
#pragma unroll
        for (int o = 0; o < RUN; ++o) {
          out[o].x += wr[o] * v.x; out[o].y += wr[o] * v.y; out[o].z += wr[o] * v.z; out[o].w += wr[o] * v.w;
        }
#pragma unroll
        for (int o = RUN - 1; o > 0; --o) wr[o] = wr[o - 1];
        wr[0] = wnext;
      }
    }
    __syncthreads();
    if (r0 < BM) {
#pragma unroll
      for (int o = 0; o < RUN; ++o) {
        const int m = r0 + o;
        if (m < BM) *(float4*)(Qt + m * HD + ((s ^ (m & 15)) << 2)) = (m < P) ? out[o] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __syncthreads();
  RRT_TRACE_MARK();                                 // [3] Q~ built

  // tile -> wave schedule of the forward fused kernel (balanced per SIMD; at most three tiles per wave)
  auto tile_of = [&](int pass) {
    if (HELP) return (pass == 0 && wave < XT) ? wave : MT;   // waves 8..11: the shared-out tile, below
    if (NW == MT) return pass == 0 ? wave : MT;     // a wave per tile: no second pass
    if (NW == 4) return wave + 4 * pass;            // 4 waves, one per SIMD: tiles round-robin
    if (pass == 0) return wave;
    if (pass == 1) return wave >= 2 ? wave + 4 : (wave == 0 ? 12 : MT);
    return (wave == 2 || wave == 3) ? wave + 8 : MT;
  };
  constexpr int NPASS = (HELP || NW == MT) ? 1 : NW == 4 ? (MT + 3) / 4 : 3;
  // fragment rows of a [rows, D]-strided global tensor: lane (lr, lg) <- row[4*(4c + lg) .. +3]
  auto global_frags = [&](const float* base, size_t stride, int m, float4 (&f)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      f[c] = m < P ? *(const float4*)(base + (row0 + m) * stride + head * HD + 4 * (4 * c + lg))
                   : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  float* dq_park = dqkv + head * HD;    // dQ~ rows are parked in the dq columns until the stencil adjoint

  // ---- pass A: query tiles.  Row statistics, D, dQ~
#pragma unroll 1
  for (int ps = 0; ps < NPASS; ++ps) {
    const int t = tile_of(ps);
    const int i0 = t * 16;
    if (t >= MT || i0 >= P) break;
    const int m = i0 + lr;
    float4 fq[4], fg[4];
    load_frags(Qt, m, lg, fq);
    // D_i = <dO_i, O_i>: this lane's 16 of the 64 dims, then across the 4 lane groups
    float dsum = 0.f;
    {
      float4 fo[4];
      if (ps == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { fg[c] = fg0[c]; fo[c] = fo0[c]; }
      } else {
        global_frags(dO, (size_t)D, m, fg);
        global_frags(O, (size_t)D, m, fo);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        dsum += (fg[c].x * fo[c].x + fg[c].y * fo[c].y) + (fg[c].z * fo[c].z + fg[c].w * fo[c].w);
    }
    dsum = sum_xor32(sum_xor16(dsum));     // VALU lane swaps (common.h), not ds_bpermute
    if (ps == 0) RRT_TRACE_MARK();                  // [4] first tile: dO / O rows here, D_i
    f32x4 s[MT];
    tile_scores<MT>(Ks, fq, lr, lg, s);            // s[jt][r] = S2[query lr][key 16 jt + 4 lg + r]
    float cmax = NEG_BIG;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (jt * 16 + 4 * lg + r >= P) s[jt][r] = NEG_BIG;
        cmax = fmaxf(cmax, s[jt][r]);
      }
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
    psum = sum_xor32(sum_xor16(psum));     // VALU lane swaps (common.h), not ds_bpermute
    const float inv = 1.0f / psum;
    if (lg == 0) {
      lse[m] = cmax + __builtin_amdgcn_logf(psum);   // v_log_f32 = log2
      dd[m] = dsum;
    }
    if (ps == 0) RRT_TRACE_MARK();                  // [5] first tile: scores + softmax
    if (HELP) __syncthreads();                      // barrier X: the shared-out tile's local statistics are in LDS
    {
      // dA[query lr][key] = dO . V^T in runs of three key tiles (Xs = V), folded into dS as each run arrives
      auto krun = [&](auto nc, const int t0) {
        constexpr int N = decltype(nc)::value;
        f32x4 da[N];
        run_scores<N>(Xs, fg, lr, lg, t0, da);
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t0 + u][r] = s[t0 + u][r] * inv * (da[u][r] - dsum);   // dS (masked keys: A = 0)
      };
#pragma unroll
      for (int t0 = 0; t0 + 3 <= MT; t0 += 3) krun(std::integral_constant<int, 3>{}, t0);
      if constexpr (MT % 3 == 1) krun(std::integral_constant<int, 1>{}, MT - 1);
      if constexpr (MT % 3 == 2) krun(std::integral_constant<int, 2>{}, MT - 2);
    }
    if (ps == 0) RRT_TRACE_MARK();                  // [6] first tile: dA, dS
    f32x4 dqt[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dqt[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tile_apply<MT>(Ks, s, lr, lg, dqt);             // dQ~[query 4 lg + r][d = 4 lr + c]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * lg + r;
      if (i < P) *(float4*)(dq_park + (row0 + i) * ld + (lr << 2)) = make_float4(dqt[0][r], dqt[1][r], dqt[2][r], dqt[3][r]);
    }
    if (ps == 0) RRT_TRACE_MARK();                  // [7] first tile: dQ~ parked
  }
  if constexpr (HELP) {
    if (wave >= XT) {                               // the ninth query tile against this wave's key tiles
      const int hq = wave - XT;
      const int m = XT * 16 + lr;
      float4 fq[4];
      load_frags(Qt, m, lg, fq);
      float dsum = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        dsum += (fg0[c].x * fo0[c].x + fg0[c].y * fo0[c].y) + (fg0[c].z * fo0[c].z + fg0[c].w * fo0[c].w);
      dsum = sum_xor32(sum_xor16(dsum));
      auto share = [&](auto nc, const int t0) {
        constexpr int N = decltype(nc)::value;
        f32x4 sc[N];
        run_scores<N>(Ks, fq, lr, lg, t0, sc);
        float cmax = NEG_BIG;
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if ((t0 + u) * 16 + 4 * lg + r >= P) sc[u][r] = NEG_BIG;
            cmax = fmaxf(cmax, sc[u][r]);
          }
        cmax = max_xor32(max_xor16(cmax));
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(sc[u][r] - cmax);
            sc[u][r] = e;
            psum += e;
          }
        psum = sum_xor32(sum_xor16(psum));
        if (lg == 0) {
          hstat[hq * 32 + lr] = cmax;
          hstat[hq * 32 + 16 + lr] = psum;
        }
        __syncthreads();                            // barrier X
        float M = NEG_BIG;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) M = fmaxf(M, hstat[w4 * 32 + lr]);
        float L = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) L += hstat[w4 * 32 + 16 + lr] * __builtin_amdgcn_exp2f(hstat[w4 * 32 + lr] - M);
        if (hq == 0 && lg == 0) {
          lse[m] = M + __builtin_amdgcn_logf(L);    // v_log_f32 = log2
          dd[m] = dsum;
        }
        const float scale = __builtin_amdgcn_exp2f(cmax - M) / L;
        f32x4 da[N];
        run_scores<N>(Xs, fg0, lr, lg, t0, da);     // dA[query lr][key]  (Xs = V)
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[u][r] = sc[u][r] * scale * (da[u][r] - dsum);
        f32x4 dqt[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) dqt[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        run_apply<N>(Ks, sc, lr, lg, t0, dqt);
        float* const mine = hpart + hq * 2048;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *(float4*)(mine + (4 * lg + r) * HD + 4 * lr) = make_float4(dqt[0][r], dqt[1][r], dqt[2][r], dqt[3][r]);
      };
      if (hq == 0) share(std::integral_constant<int, 3>{}, 0);
      else share(std::integral_constant<int, 2>{}, 1 + 2 * hq);
    }
  }
  RRT_TRACE_MARK();                                 // [8] pass A done (this wave)
  float4 fv0[4];                                    // pass B's first tile: its V rows, requested under the barrier + dO refill
  {
    const int m = (HELP && wave >= XT ? XT : wave) * 16 + lr;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      fv0[c] = ((HELP || wave < MT) && m < P) ? *(const float4*)(qkv + 2 * D + (row0 + m) * ld + head * HD + 4 * (4 * c + lg))
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();                                  // V is dead: the third tile becomes dO
  if constexpr (HELP) {                             // the shared-out tile's dQ~: four partials in wave order, parked
    if (tid < 256) {
      const int row = tid >> 4, sl = tid & 15;
      if (XT * 16 + row < P) {
        float4 a = *(const float4*)(hpart + row * HD + 4 * sl);
#pragma unroll
        for (int w4 = 1; w4 < 4; ++w4) {
          const float4 b = *(const float4*)(hpart + w4 * 2048 + row * HD + 4 * sl);
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *(float4*)(dq_park + (row0 + XT * 16 + row) * ld + 4 * sl) = a;
      }
    }
  }
  for (int idx = tid; idx < BM * 16; idx += NW * 64) {
    const int m = idx >> 4, s = idx & 15;
    const float4 g4 = m < P ? *(const float4*)(dO + (row0 + m) * D + head * HD + 4 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(Xs + m * HD + ((s ^ (m & 15)) << 2)) = g4;
  }
  __syncthreads();
  RRT_TRACE_MARK();                                 // [9] dO tile in LDS

  // ---- pass B: key tiles.  dV = A^T dO, dK = dS^T Q~ (Q~ carries log2 e: x ln 2)
#pragma unroll 1
  for (int ps = 0; ps < NPASS; ++ps) {
    const int t = tile_of(ps);
    const int j0 = t * 16;
    if (t >= MT || j0 >= P) break;
    const int m = j0 + lr;
    float4 fk[4], fv[4];
    load_frags(Ks, m, lg, fk);
    if (ps == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) fv[c] = fv0[c];
    } else {
      global_frags(qkv + 2 * D, (size_t)ld, m, fv);
    }
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dv[c] = dk[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // query tiles in runs of three: S^T and dA^T of the run, A and dS from them, the run's share of dV and dK
    auto qrun = [&](auto nc, const int t0) {
      constexpr int N = decltype(nc)::value;
      f32x4 a[N], ds[N];
      run_scores<N>(Qt, fk, lr, lg, t0, a);         // a[u][r] = S2[query 16 (t0 + u) + 4 lg + r][key lr]
      run_scores<N>(Xs, fv, lr, lg, t0, ds);        // dA[query][key lr]  (Xs = dO)
#pragma unroll
      for (int u = 0; u < N; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = (t0 + u) * 16 + 4 * lg + r;
          // rows past the region: lse / dd were never written there (0 x garbage would be NaN)
          const float p = q < P ? __builtin_amdgcn_exp2f(a[u][r] - lse[q]) : 0.f;
          a[u][r] = p;
          ds[u][r] = q < P ? p * (ds[u][r] - dd[q]) : 0.f;
        }
      run_apply<N>(Xs, a, lr, lg, t0, dv);          // dV[key 4 lg + r][d = 4 lr + c]
      run_apply<N>(Qt, ds, lr, lg, t0, dk);
    };
#pragma unroll
    for (int t0 = 0; t0 + 3 <= MT; t0 += 3) qrun(std::integral_constant<int, 3>{}, t0);
    if constexpr (MT % 3 == 1) qrun(std::integral_constant<int, 1>{}, MT - 1);
    if constexpr (MT % 3 == 2) qrun(std::integral_constant<int, 2>{}, MT - 2);
    if (ps == 0) RRT_TRACE_MARK();                  // [10] first key tile: S^T, dA^T ... (runs)
    if (ps == 0) RRT_TRACE_MARK();                  // [11]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = j0 + 4 * lg + r;
      if (key < P) {
        float* dst = dqkv + (row0 + key) * ld + head * HD + (lr << 2);
        *(float4*)(dst + D) = make_float4(dk[0][r] * LN2, dk[1][r] * LN2, dk[2][r] * LN2, dk[3][r] * LN2);
        *(float4*)(dst + 2 * D) = make_float4(dv[0][r], dv[1][r], dv[2][r], dv[3][r]);
      }
    }
    if (ps == 0) RRT_TRACE_MARK();                  // [12] first key tile: dV, dK stored
  }
  if constexpr (HELP) {
    if (wave >= XT) {                               // the ninth key tile against this wave's query tiles
      const int hq = wave - XT;
      const int m = XT * 16 + lr;
      float4 fk[4];
      load_frags(Ks, m, lg, fk);
      auto share = [&](auto nc, const int t0) {
        constexpr int N = decltype(nc)::value;
        f32x4 a[N], ds[N];
        run_scores<N>(Qt, fk, lr, lg, t0, a);
        run_scores<N>(Xs, fv0, lr, lg, t0, ds);
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = (t0 + u) * 16 + 4 * lg + r;
            const float p = q < P ? __builtin_amdgcn_exp2f(a[u][r] - lse[q]) : 0.f;
            a[u][r] = p;
            ds[u][r] = q < P ? p * (ds[u][r] - dd[q]) : 0.f;
          }
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) dv[c] = dk[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        run_apply<N>(Xs, a, lr, lg, t0, dv);
        run_apply<N>(Qt, ds, lr, lg, t0, dk);
        float* const mine = hpart + hq * 2048;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          *(float4*)(mine + (4 * lg + r) * HD + 4 * lr) = make_float4(dv[0][r], dv[1][r], dv[2][r], dv[3][r]);
          *(float4*)(mine + 1024 + (4 * lg + r) * HD + 4 * lr) = make_float4(dk[0][r], dk[1][r], dk[2][r], dk[3][r]);
        }
      };
      if (hq == 0) share(std::integral_constant<int, 3>{}, 0);
      else share(std::integral_constant<int, 2>{}, 1 + 2 * hq);
    }
  }
  RRT_TRACE_MARK();                                 // [13] pass B done (this wave)
  __syncthreads();                                  // every read of the dO tile is done; parked dQ~ rows are visible
  if constexpr (HELP) {                             // the shared-out tile's dV / dK: four partials in wave order
    if (tid < 512) {
      const int which = tid >> 8, row = (tid & 255) >> 4, sl = tid & 15;
      const int key = XT * 16 + row;
      if (key < P) {
        float4 a = *(const float4*)(hpart + which * 1024 + row * HD + 4 * sl);
#pragma unroll
        for (int w4 = 1; w4 < 4; ++w4) {
          const float4 b = *(const float4*)(hpart + w4 * 2048 + which * 1024 + row * HD + 4 * sl);
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float* dst = dqkv + (row0 + key) * ld + head * HD + 4 * sl;
        if (which == 0) *(float4*)(dst + 2 * D) = a;
        else *(float4*)(dst + D) = make_float4(a.x * LN2, a.y * LN2, a.z * LN2, a.w * LN2);
      }
    }
  }

  // ---- parked dQ~ rows -> LDS over the dead dO tile (rows >= P: zeros)
  // ... and the stashed q rows back over the dead Q~ tile: the tap gradients pair dQ~ rows with q rows k/2 apart
  float* Gs = Xs;
  for (int idx = tid; idx < BM * 16; idx += NW * 64) {
    const int m = idx >> 4, s = idx & 15;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), q4 = g4;
    if (m < P) {
      g4 = *(const float4*)(dq_park + (row0 + m) * ld + 4 * s);
      if (epeg_k > 0) q4 = *(const float4*)(qkv + (row0 + m) * ld + head * HD + 4 * s);
    }
    const int off = m * HD + ((s ^ (m & 15)) << 2);
    *(float4*)(Gs + off) = g4;
    *(float4*)(Qt + off) = q4;
  }
  __syncthreads();
  RRT_TRACE_MARK();                                 // [14] dQ~ tile in LDS

  // ---- dq_raw = q_scale * (I + T_w)^T dQ~ : the stencil with flipped taps; rows stay inside the region
  {
    auto tapf = [&](int t) { return tapsA[t + RUN - 1]; };   // weight of source row j for output row i, t = j - i + half (flipped taps)
    const int s = tid & 15, g = tid >> 4;
    const int r0 = g * RUN;
    if (r0 < P) {
      float4 out[RUN];
#pragma unroll
      for (int o = 0; o < RUN; ++o) out[o] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int lo = max(r0 - half, 0), hi = min(r0 + RUN - 1 + half, P - 1);
      float wr[RUN];
#pragma unroll
      for (int o = 0; o < RUN; ++o) wr[o] = tapf(lo - r0 - o + half);
#pragma unroll 4
      for (int rr = lo; rr <= hi; ++rr) {
        const float4 v = *(const float4*)(Gs + rr * HD + ((s ^ (rr & 15)) << 2));
        const float wnext = tapf(rr + 1 - r0 + half);
#pragma unroll
        for (int o = 0; o < RUN; ++o) {
          out[o].x += wr[o] * v.x; out[o].y += wr[o] * v.y; out[o].z += wr[o] * v.z; out[o].w += wr[o] * v.w;
        }
#pragma unroll
        for (int o = RUN - 1; o > 0; --o) wr[o] = wr[o - 1];
        wr[0] = wnext;
      }
#pragma unroll
      for (int o = 0; o < RUN; ++o) {
        const int i = r0 + o;
        if (i < P) *(float4*)(dqkv + (row0 + i) * ld + head * HD + 4 * s) = out[o];
      }
    }
  }

  RRT_TRACE_MARK();                                 // [15] dq written
  // ---- tap gradients: dw[t] = sum_i <dQ~_i, q_{i + t - half}>, both tiles in LDS.  Thread = (16-byte slot s, run of RUN
  // rows) as in the stencils: its dQ~ rows stay in registers, the q rows r0 - k/2 .. slide past them once, and source
  // row jj meets output o with tap t = jj - o -- compile-time indices into a 16-tap register window (taps in chunks
  // of 16).  Per chunk: lane sums over the 4 row groups of a wave (lane swaps), one [16 slots][16 taps] record per
  // wave in the dead K tile, 16 threads per tap add the NW x 16 partials in a fixed order.
  // (First version: a serial loop over the taps with q re-read from global / L2 and a wave reduction per tap --
  // 49.6 K of a block's 238 K cycles, traced.)
  if (epeg_k > 0) {
    const int s = tid & 15, g = tid >> 4;
    const int r0 = g * RUN;
    float* const rec = Ks;                          // [NW][16 slots][16 taps]
    float4 gq[RUN];
#pragma unroll
    for (int o = 0; o < RUN; ++o) {
      const int i = r0 + o;
      gq[o] = i < BM ? *(const float4*)(Gs + i * HD + ((s ^ (i & 15)) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t0 = 0; t0 < epeg_k; t0 += 16) {
      float acc[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] = 0.f;
#pragma unroll
      for (int jj = 0; jj < 16 + RUN - 1; ++jj) {   // source row r0 + t0 - half + jj
        const int j = r0 + t0 - half + jj;
        const bool ok = j >= 0 && j < P;
        const int jc = ok ? j : 0;
        float4 q4 = *(const float4*)(Qt + jc * HD + ((s ^ (jc & 15)) << 2));
        if (!ok) q4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < RUN; ++o) {
          const int u = jj - o;                     // tap t0 + u
          if (u >= 0 && u < 16)
            acc[u] += (gq[o].x * q4.x + gq[o].y * q4.y) + (gq[o].z * q4.z + gq[o].w * q4.w);
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] = sum_xor32(sum_xor16(acc[u]));
      if (t0 > 0) __syncthreads();                  // the previous chunk's records have been read
      if (lane < 16) {
#pragma unroll
        for (int u = 0; u < 16; u += 4)
          *(float4*)(rec + (wave * 16 + lane) * 16 + u) = make_float4(acc[u], acc[u + 1], acc[u + 2], acc[u + 3]);
      }
      __syncthreads();
      if (tid < 256) {                              // thread = (tap u, slot group of NW records)
        const int u = tid >> 4, sl = tid & 15;
        float a = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) a += rec[(wv * 16 + sl) * 16 + u];
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
        if (sl == 0 && t0 + u < epeg_k) dpe_part[((size_t)reg * heads + head) * epeg_k + t0 + u] = a;
      }
    }
  }
  RRT_TRACE_MARK();                                 // [16] tap gradients
}

template <int MT>
hipError_t launch_bwd_mt(const float* qkv, const float* pe_w, const float* O, const float* dO, float* dqkv,
                         float* dpe_part, int n_regions, int P, int D, int heads, int epeg_k, hipStream_t st) {
  constexpr size_t LDS = ((size_t)3 * 16 * MT * HD + 2 * 16 * MT + 9 * 64 + (MT == 9 ? 128 + 4 * 2048 : 0)) * sizeof(float);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  const float q_scale = 1.0f / sqrtf((float)HD);
  static const bool six = rrt_tune_env("RRT_ATTN_BWD_NW6") != nullptr;     // tuning hook (A/B of the two schedules)
  static const bool nine = rrt_tune_env("RRT_ATTN_BWD_NW9") != nullptr;      // tuning hook: a wave per tile, no helpers
  if (MT == 9 && !six && !nine) {   // eight tile waves + four helper waves sharing the ninth tile (three waves per SIMD)
    auto kern = attn_bwd_kernel<MT, MT == 9 ? 12 : 6>;
    static OncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    kern<<<dim3(n_regions * heads), dim3(768), LDS, st>>>(qkv, pe_w, O, dO, dqkv, dpe_part, P, D, heads,
                                                          pe_w ? epeg_k : 0, q_scale);
  } else if (MT == 9 && !six) {     // a wave per tile, three waves per SIMD at <= 168 VGPRs
    auto kern = attn_bwd_kernel<MT, MT == 9 ? 9 : 6>;
    static OncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    kern<<<dim3(n_regions * heads), dim3(576), LDS, st>>>(qkv, pe_w, O, dO, dqkv, dpe_part, P, D, heads,
                                                          pe_w ? epeg_k : 0, q_scale);
  } else if ((MT == 7 || MT == 8 || MT == 11) && !six && !nine) {   // a wave per tile (P = 121 / 128: 146 -> ? us with eight waves, two per SIMD)
    constexpr int NWT = (MT == 7 || MT == 8 || MT == 11) ? MT : 6;
    auto kern = attn_bwd_kernel<MT, NWT>;
    static OncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    kern<<<dim3(n_regions * heads), dim3(NWT * 64), LDS, st>>>(qkv, pe_w, O, dO, dqkv, dpe_part, P, D, heads,
                                                               pe_w ? epeg_k : 0, q_scale);
  } else if (MT >= 11 && !six) {           // measured: 9 tiles 221 (6 waves) vs 241 us; 11: 350 vs 291; 13: 516 vs 400
    auto kern = attn_bwd_kernel<MT, 4>;
    static OncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    kern<<<dim3(n_regions * heads), dim3(256), LDS, st>>>(qkv, pe_w, O, dO, dqkv, dpe_part, P, D, heads,
                                                          pe_w ? epeg_k : 0, q_scale);
  } else {
    auto kern = attn_bwd_kernel<MT, 6>;
    static OncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    kern<<<dim3(n_regions * heads), dim3(384), LDS, st>>>(qkv, pe_w, O, dO, dqkv, dpe_part, P, D, heads,
                                                          pe_w ? epeg_k : 0, q_scale);
  }
  return hipGetLastError();
}

// ======================================================================================= streaming variant
// Regions of any size (P > 208: bags beyond ~12.5 k tokens at region_num = 8): the same mathematics with the
// "other" operand streamed through LDS in 128-row chunks instead of resident.  Four kernels:
//   stencil:  q~ = log2(e) (I + T_w) q            -> a [rows, D] buffer (global)
//   q pass:   block = (region, head, 6 query tiles); sweep 1 over key chunks: online row max / sum -> lse;
//             sweep 2: A, dA, dS, dQ~ += dS K.  Writes dQ~ rows (parked in dq), lse, D.
//   kv pass:  block = (region, head, 6 key tiles); one sweep over query chunks (Q~, dO chunks resident) with
//             A = exp2(S - lse): dV += A^T dO, dK += dS^T Q~.
//   adjoint:  dq = scale (I + T_w)^T dQ~ and the tap gradients, block = (region, head).
constexpr int CK = 128, CT = CK / 16;     // chunk rows / row tiles per chunk

__global__ __launch_bounds__(256) void attn_stencil_kernel(const float* __restrict__ qkv, const float* __restrict__ pe_w,
                                                           float* __restrict__ qt, int P, int D, int heads,
                                                           int epeg_k, long n_rows) {
  // thread = (row, 16-byte slot of one head): qt[row, head*64 + 4s ..] ; rows stay inside their region
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int slots = D / 4;
  if (idx >= n_rows * slots) return;
  const long row = idx / slots;
  const int c = (int)(idx - row * slots) * 4, head = c / HD;
  const int i = (int)(row % P);
  const int half = epeg_k >> 1;
  const float* w = pe_w + head * epeg_k;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t0 = -half; t0 <= half; t0 += 8) {       // 8 independent loads in flight
    float4 v[8];
    float wt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u, j = i + t;
      const bool ok = t <= half && j >= 0 && j < P;
      wt[u] = ok ? ((epeg_k > 0 ? w[t + half] : 0.f) + (t == 0 ? 1.0f : 0.f)) : 0.f;
      v[u] = ok ? *(const float4*)(qkv + (size_t)(row + t) * 3 * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc.x += wt[u] * v[u].x; acc.y += wt[u] * v[u].y; acc.z += wt[u] * v[u].z; acc.w += wt[u] * v[u].w;
    }
  }
  *(float4*)(qt + (size_t)row * D + c) = make_float4(acc.x * LOG2E, acc.y * LOG2E, acc.z * LOG2E, acc.w * LOG2E);
}

// rows [r0, r0 + CK) of a [.., stride]-strided tensor (head columns) -> swizzled LDS chunk; rows >= P: zeros
__device__ __forceinline__ void load_chunk(float* dst, const float* src, size_t stride, size_t row0, int r0, int P,
                                           int tid) {
  for (int idx = tid; idx < CK * 16; idx += 384) {
    const int m = idx >> 4, s = idx & 15;
    const float4 v = (r0 + m < P) ? *(const float4*)(src + (row0 + r0 + m) * stride + 4 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(dst + m * HD + ((s ^ (m & 15)) << 2)) = v;
  }
}

__global__ __launch_bounds__(384, 3) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ qt,
                                                         const float* __restrict__ O, const float* __restrict__ dO,
                                                         float* __restrict__ dqkv, float* __restrict__ lse_g,
                                                         float* __restrict__ dd_g, int P, int D, int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ks = (float*)smem;
  float* Vs = Ks + CK * HD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int head = blockIdx.x, reg = blockIdx.y;
  const size_t row0 = (size_t)reg * P;
  const int ld = 3 * D;
  const int i0 = (blockIdx.z * 6 + wave) * 16;
  const bool active = i0 < P;
  const int m = i0 + lr;
  float4 fq[4], fg[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool ok = active && m < P;
    fq[c] = ok ? *(const float4*)(qt + (row0 + m) * D + head * HD + 4 * (4 * c + lg)) : make_float4(0.f, 0.f, 0.f, 0.f);
    fg[c] = ok ? *(const float4*)(dO + (row0 + m) * D + head * HD + 4 * (4 * c + lg)) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float dsum = 0.f;
  if (active && m < P) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 o4 = *(const float4*)(O + (row0 + m) * D + head * HD + 4 * (4 * c + lg));
      dsum += (fg[c].x * o4.x + fg[c].y * o4.y) + (fg[c].z * o4.z + fg[c].w * o4.w);
    }
  }
  dsum = sum_xor32(sum_xor16(dsum));     // VALU lane swaps (common.h), not ds_bpermute
  // sweep 1: online row max / sum over the key chunks
  float mrun = NEG_BIG, lrun = 0.f;
  for (int r0 = 0; r0 < P; r0 += CK) {
    __syncthreads();
    load_chunk(Ks, qkv + D + head * HD, (size_t)ld, row0, r0, P, tid);
    __syncthreads();
    if (active) {
      f32x4 s[CT];
      tile_scores<CT>(Ks, fq, lr, lg, s);
      float cmax = NEG_BIG;
#pragma unroll
      for (int jt = 0; jt < CT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r0 + jt * 16 + 4 * lg + r >= P) s[jt][r] = NEG_BIG;
          cmax = fmaxf(cmax, s[jt][r]);
        }
      cmax = max_xor32(max_xor16(cmax));
      const float mnew = fmaxf(mrun, cmax);
      float psum = 0.f;
#pragma unroll
      for (int jt = 0; jt < CT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) psum += __builtin_amdgcn_exp2f(s[jt][r] - mnew);
      psum = sum_xor32(sum_xor16(psum));     // VALU lane swaps (common.h), not ds_bpermute
      lrun = lrun * __builtin_amdgcn_exp2f(mrun - mnew) + psum;
      mrun = mnew;
    }
  }
  const float lse = mrun + __builtin_amdgcn_logf(lrun);
  // sweep 2: A, dA, dS, dQ~
  f32x4 dqt[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) dqt[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int r0 = 0; r0 < P; r0 += CK) {
    __syncthreads();
    load_chunk(Ks, qkv + D + head * HD, (size_t)ld, row0, r0, P, tid);
    load_chunk(Vs, qkv + 2 * D + head * HD, (size_t)ld, row0, r0, P, tid);
    __syncthreads();
    if (active) {
      // the chunk's key tiles in runs of three (as in the resident kernel: a run's scores are all that is live)
      auto krun = [&](auto nc, const int t0) {
        constexpr int N = decltype(nc)::value;
        f32x4 sc[N], da[N];
        run_scores<N>(Ks, fq, lr, lg, t0, sc);
        run_scores<N>(Vs, fg, lr, lg, t0, da);
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = r0 + (t0 + u) * 16 + 4 * lg + r < P;
            const float p = ok ? __builtin_amdgcn_exp2f(sc[u][r] - lse) : 0.f;
            sc[u][r] = p * (da[u][r] - dsum);
          }
        run_apply<N>(Ks, sc, lr, lg, t0, dqt);
      };
      static_assert(CT == 8, "runs below assume eight row tiles per chunk");
      krun(std::integral_constant<int, 3>{}, 0);
      krun(std::integral_constant<int, 3>{}, 3);
      krun(std::integral_constant<int, 2>{}, 6);
    }
  }
  if (active) {
    if (lg == 0 && m < P) {
      lse_g[(row0 + m) * heads + head] = lse;
      dd_g[(row0 + m) * heads + head] = dsum;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * lg + r;
      if (i < P)
        *(float4*)(dqkv + (row0 + i) * ld + head * HD + (lr << 2)) = make_float4(dqt[0][r], dqt[1][r], dqt[2][r], dqt[3][r]);
    }
  }
}

__global__ __launch_bounds__(384) void attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ qt,
                                                          const float* __restrict__ dO, const float* __restrict__ lse_g,
                                                          const float* __restrict__ dd_g, float* __restrict__ dqkv,
                                                          int P, int D, int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = (float*)smem;
  float* Gs = Qs + CK * HD;
  float* lse = Gs + CK * HD;            // [CK]
  float* dd = lse + CK;                 // [CK]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int head = blockIdx.x, reg = blockIdx.y;
  const size_t row0 = (size_t)reg * P;
  const int ld = 3 * D;
  const int j0 = (blockIdx.z * 6 + wave) * 16;
  const bool active = j0 < P;
  const int m = j0 + lr;
  float4 fk[4], fv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool ok = active && m < P;
    const float* src = qkv + (row0 + m) * ld + head * HD + 4 * (4 * c + lg);
    fk[c] = ok ? *(const float4*)(src + D) : make_float4(0.f, 0.f, 0.f, 0.f);
    fv[c] = ok ? *(const float4*)(src + 2 * D) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f32x4 dv[4], dk[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) dv[c] = dk[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int r0 = 0; r0 < P; r0 += CK) {
    __syncthreads();
    load_chunk(Qs, qt + head * HD, (size_t)D, row0, r0, P, tid);
    load_chunk(Gs, dO + head * HD, (size_t)D, row0, r0, P, tid);
    if (tid < CK) {
      const bool ok = r0 + tid < P;
      lse[tid] = ok ? lse_g[(row0 + r0 + tid) * heads + head] : 0.f;
      dd[tid] = ok ? dd_g[(row0 + r0 + tid) * heads + head] : 0.f;
    }
    __syncthreads();
    if (active) {
      auto qrun = [&](auto nc, const int t0) {
        constexpr int N = decltype(nc)::value;
        f32x4 a[N], ds[N];
        run_scores<N>(Qs, fk, lr, lg, t0, a);
        run_scores<N>(Gs, fv, lr, lg, t0, ds);
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = (t0 + u) * 16 + 4 * lg + r;
            const bool ok = r0 + q < P;
            const float p = ok ? __builtin_amdgcn_exp2f(a[u][r] - lse[q]) : 0.f;
            a[u][r] = p;
            ds[u][r] = ok ? p * (ds[u][r] - dd[q]) : 0.f;
          }
        run_apply<N>(Gs, a, lr, lg, t0, dv);
        run_apply<N>(Qs, ds, lr, lg, t0, dk);
      };
      qrun(std::integral_constant<int, 3>{}, 0);
      qrun(std::integral_constant<int, 3>{}, 3);
      qrun(std::integral_constant<int, 2>{}, 6);
    }
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = j0 + 4 * lg + r;
      if (key < P) {
        float* dst = dqkv + (row0 + key) * ld + head * HD + (lr << 2);
        *(float4*)(dst + D) = make_float4(dk[0][r] * LN2, dk[1][r] * LN2, dk[2][r] * LN2, dk[3][r] * LN2);
        *(float4*)(dst + 2 * D) = make_float4(dv[0][r], dv[1][r], dv[2][r], dv[3][r]);
      }
    }
  }
}

// dq = scale (I + T_w)^T dQ~ (dQ~ parked in the dq columns; staged through `tmp` so the in-place update never reads
// a row it has already overwritten) and the tap-gradient partials.  Block = (region, head).
__global__ __launch_bounds__(256) void attn_adjoint_kernel(const float* __restrict__ qkv, const float* __restrict__ pe_w,
                                                           float* __restrict__ dqkv, float* __restrict__ tmp,
                                                           float* __restrict__ dpe_part, int P, int D, int heads,
                                                           int epeg_k, float q_scale) {
  __shared__ float wred[4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x, reg = blockIdx.y;
  const size_t row0 = (size_t)reg * P;
  const int ld = 3 * D, half = epeg_k >> 1;
  const float* w = pe_w + head * epeg_k;
  const int s = tid & 15;
  // copy dQ~ rows of this (region, head) to tmp [rows, D] (head columns)
  for (int i = tid >> 4; i < P; i += 16)
    *(float4*)(tmp + (row0 + i) * D + head * HD + 4 * s) = *(const float4*)(dqkv + (row0 + i) * ld + head * HD + 4 * s);
  __syncthreads();
  for (int i = tid >> 4; i < P; i += 16) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = -half; t0 <= half; t0 += 8) {    // dq_i = sum_j wt[i - j + half] dQ~_j ,  j = i + t ; 8 loads in flight
      float4 v[8];
      float wt[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u, j = i + t;
        const bool ok = t <= half && j >= 0 && j < P;
        wt[u] = ok ? ((epeg_k > 0 ? w[half - t] : 0.f) + (t == 0 ? 1.0f : 0.f)) : 0.f;
        v[u] = ok ? *(const float4*)(tmp + (row0 + j) * D + head * HD + 4 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc.x += wt[u] * v[u].x; acc.y += wt[u] * v[u].y; acc.z += wt[u] * v[u].z; acc.w += wt[u] * v[u].w;
      }
    }
    *(float4*)(dqkv + (row0 + i) * ld + head * HD + 4 * s) =
        make_float4(acc.x * q_scale, acc.y * q_scale, acc.z * q_scale, acc.w * q_scale);
  }
  if (epeg_k > 0) {
    // every thread walks its rows once and keeps all tap partials in registers (one pass over dQ~ and k
    // independent loads of q per row; the tap-by-tap loop was a chain of k passes: 140 us at P = 144)
    float tacc[64];
#pragma unroll
    for (int t = 0; t < 64; ++t) tacc[t] = 0.f;
    for (int i = tid >> 4; i < P; i += 16) {
      const float4 g4 = *(const float4*)(tmp + (row0 + i) * D + head * HD + 4 * s);
#pragma unroll
      for (int t = 0; t < 64; ++t) {
        if (t < epeg_k) {
          const int j = i + t - half;
          if (j >= 0 && j < P) {
            const float4 q4 = *(const float4*)(qkv + (row0 + j) * ld + head * HD + 4 * s);
            tacc[t] += (g4.x * q4.x + g4.y * q4.y) + (g4.z * q4.z + g4.w * q4.w);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 64; ++t) {
      if (t < epeg_k) {
        const float a = wave_sum(tacc[t]);
        if (lane == 0) wred[wave * 64 + t] = a;
      }
    }
    __syncthreads();
    if (tid < epeg_k)
      dpe_part[((size_t)reg * heads + head) * epeg_k + tid] =
          (wred[tid] + wred[64 + tid]) + (wred[128 + tid] + wred[192 + tid]);
  }
}

// The same adjoint step with both tiles in LDS, for regions of up to 256 tokens (the streaming path's P = 225 / 256: bags of
// 12.6-16 k tokens at region_num = 8): block = (region, head), eight waves; dQ~ rows (parked in the dq columns) and the stashed
// q rows staged once as XOR-swizzled [BM][64] tiles, the stencil's flipped taps from an LDS table, the tap gradients in one
// sliding pass with a fixed-order reduction -- the resident kernel's last two phases.  (attn_adjoint_kernel above walks global
// memory: every output row re-reads its k source rows, every tap its q row: 170 us at P = 256, 21 % of that backward.)
__global__ __launch_bounds__(512) void attn_adjoint_lds_kernel(const float* __restrict__ qkv, const float* __restrict__ pe_w,
                                                               float* __restrict__ dqkv, float* __restrict__ dpe_part, int P,
                                                               int D, int heads, int epeg_k, float q_scale) {
  constexpr int BM = 256, NW = 8, RUN = BM * 16 / (NW * 64);      // 8 rows per thread
  static_assert(2 * RUN + 62 < 128, "tap table range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const Gs = (float*)smem;                   // dQ~ tile
  float* const Qs = Gs + BM * HD;                   // q tile (as stashed: scaled, + bias)
  float* const tapsA = Qs + BM * HD;                // [128]
  float* const rec = tapsA + 128;                   // [NW][16 slots][16 taps]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x, reg = blockIdx.y;
  const size_t row0 = (size_t)reg * P;
  const int ld = 3 * D, half = epeg_k >> 1;
  if (tid < 128) {
    const int t = tid - (RUN - 1);
    const bool in = epeg_k > 0 && t >= 0 && t < epeg_k;
    tapsA[tid] = ((in ? pe_w[head * epeg_k + epeg_k - 1 - t] : 0.f) + (t == half ? 1.0f : 0.f)) * q_scale;
  }
  for (int idx = tid; idx < BM * 16; idx += NW * 64) {
    const int m = idx >> 4, sl = idx & 15;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), q4 = g4;
    if (m < P) {
      g4 = *(const float4*)(dqkv + (row0 + m) * ld + head * HD + 4 * sl);
      if (epeg_k > 0) q4 = *(const float4*)(qkv + (row0 + m) * ld + head * HD + 4 * sl);
    }
    const int off = m * HD + ((sl ^ (m & 15)) << 2);
    *(float4*)(Gs + off) = g4;
    *(float4*)(Qs + off) = q4;
  }
  __syncthreads();
  const int s = tid & 15, g = tid >> 4;
  const int r0 = g * RUN;
  // dq_raw = q_scale (I + T_w)^T dQ~
  if (r0 < P) {
    float4 out[RUN];
#pragma unroll
    for (int o = 0; o < RUN; ++o) out[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int lo = max(r0 - half, 0), hi = min(r0 + RUN - 1 + half, P - 1);
    float wr[RUN];
#pragma unroll
    for (int o = 0; o < RUN; ++o) wr[o] = tapsA[lo - r0 - o + half + RUN - 1];
#pragma unroll 4
    for (int rr = lo; rr <= hi; ++rr) {
      const float4 v = *(const float4*)(Gs + rr * HD + ((s ^ (rr & 15)) << 2));
      const float wnext = tapsA[rr + 1 - r0 + half + RUN - 1];
#pragma unroll
      for (int o = 0; o < RUN; ++o) {
        out[o].x += wr[o] * v.x; out[o].y += wr[o] * v.y; out[o].z += wr[o] * v.z; out[o].w += wr[o] * v.w;
      }
#pragma unroll
      for (int o = RUN - 1; o > 0; --o) wr[o] = wr[o - 1];
      wr[0] = wnext;
    }
#pragma unroll
    for (int o = 0; o < RUN; ++o) {
      const int i = r0 + o;
      if (i < P) *(float4*)(dqkv + (row0 + i) * ld + head * HD + 4 * s) = out[o];
    }
  }
  // tap gradients: dw[t] = sum_i <dQ~_i, q_{i + t - half}>
  if (epeg_k > 0) {
    float4 gq[RUN];
#pragma unroll
    for (int o = 0; o < RUN; ++o) {
      const int i = r0 + o;
      gq[o] = *(const float4*)(Gs + i * HD + ((s ^ (i & 15)) << 2));     // rows >= P are zero in the tile
    }
    for (int t0 = 0; t0 < epeg_k; t0 += 16) {
      float acc[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] = 0.f;
#pragma unroll
      for (int jj = 0; jj < 16 + RUN - 1; ++jj) {
        const int j = r0 + t0 - half + jj;
        const bool ok = j >= 0 && j < P;
        const int jc = ok ? j : 0;
        float4 q4 = *(const float4*)(Qs + jc * HD + ((s ^ (jc & 15)) << 2));
        if (!ok) q4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < RUN; ++o) {
          const int u = jj - o;
          if (u >= 0 && u < 16)
            acc[u] += (gq[o].x * q4.x + gq[o].y * q4.y) + (gq[o].z * q4.z + gq[o].w * q4.w);
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] = sum_xor32(sum_xor16(acc[u]));
      if (t0 > 0) __syncthreads();
      if (lane < 16) {
#pragma unroll
        for (int u = 0; u < 16; u += 4)
          *(float4*)(rec + (wave * 16 + lane) * 16 + u) = make_float4(acc[u], acc[u + 1], acc[u + 2], acc[u + 3]);
      }
      __syncthreads();
      if (tid < 256) {
        const int u = tid >> 4, sl = tid & 15;
        float a = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) a += rec[(wv * 16 + sl) * 16 + u];
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
        if (sl == 0 && t0 + u < epeg_k) dpe_part[((size_t)reg * heads + head) * epeg_k + t0 + u] = a;
      }
    }
  }
}

// ---- any head dim (crmsa_heads = 1 -> head dim = dim), no EPEG, short sequences: CR-MSA's inner attention over
// the k x 64 representatives.  VALU only: one block per (sequence, head), a wave per query / key row, the head dim
// across the lanes; A and dS [P, P] in LDS.  (Published TCGA-BRCA-R50 / NSCLC-PLIP configs: 3 x 64 rows.)
__global__ __launch_bounds__(256) void attn_bwd_generic_kernel(const float* __restrict__ qkv,
                                                               const float* __restrict__ O,
                                                               const float* __restrict__ dO,
                                                               float* __restrict__ dqkv, int P, int D, int hd,
                                                               float q_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* A = (float*)smem;              // [P][P] probabilities
  float* dS = A + (size_t)P * P;        // [P][P]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = blockIdx.x, reg = blockIdx.y;
  const int ld = 3 * D;
  const size_t row0 = (size_t)reg * P;
  const float* qb = qkv + row0 * ld + head * hd;
  const float* kb = qb + D;
  const float* vb = qb + 2 * D;
  const float* ob = O + row0 * D + head * hd;
  const float* gb = dO + row0 * D + head * hd;
  // rows of S, softmax, D_i, rows of dA -> A, dS
  for (int i = wave; i < P; i += 4) {
    float dsum = 0.f;
    for (int d = lane * 4; d < hd; d += 256) {
      const float4 g = *(const float4*)(gb + (size_t)i * D + d), o = *(const float4*)(ob + (size_t)i * D + d);
      dsum += (g.x * o.x + g.y * o.y) + (g.z * o.z + g.w * o.w);
    }
    dsum = wave_sum(dsum);
    float mx = NEG_BIG;
    for (int j0 = 0; j0 < P; j0 += 4) {
      float s[4] = {0.f, 0.f, 0.f, 0.f}, da[4] = {0.f, 0.f, 0.f, 0.f};
      for (int d = lane * 4; d < hd; d += 256) {
        const float4 q4 = *(const float4*)(qb + (size_t)i * ld + d), g4 = *(const float4*)(gb + (size_t)i * D + d);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + u < P ? j0 + u : P - 1;
          const float4 k4 = *(const float4*)(kb + (size_t)j * ld + d), v4 = *(const float4*)(vb + (size_t)j * ld + d);
          s[u] += (q4.x * k4.x + q4.y * k4.y) + (q4.z * k4.z + q4.w * k4.w);
          da[u] += (g4.x * v4.x + g4.y * v4.y) + (g4.z * v4.z + g4.w * v4.w);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float sv = wave_sum(s[u]), dv = wave_sum(da[u]);
        if (j0 + u < P) {
          if (lane == 0) { A[i * P + j0 + u] = sv; dS[i * P + j0 + u] = dv; }
          mx = fmaxf(mx, sv);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    float sum = 0.f;
    for (int j = lane; j < P; j += 64) {
      const float p = __expf(A[i * P + j] - mx);
      A[i * P + j] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < P; j += 64) {
      const float p = A[i * P + j] * inv;
      A[i * P + j] = p;
      dS[i * P + j] = p * (dS[i * P + j] - dsum);
    }
  }
  __syncthreads();
  // dq_i = scale * sum_j dS[i,j] k_j ; dk_j = sum_i dS[i,j] q_i ; dv_j = sum_i A[i,j] dO_i
  for (int r = wave; r < P; r += 4) {
    for (int d = lane * 4; d < hd; d += 256) {
      float4 aq = make_float4(0.f, 0.f, 0.f, 0.f), ak = aq, av = aq;
      for (int j0 = 0; j0 < P; j0 += 4) {
        float4 k4[4], q4[4], g4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + u < P ? j0 + u : P - 1;
          k4[u] = *(const float4*)(kb + (size_t)j * ld + d);
          q4[u] = *(const float4*)(qb + (size_t)j * ld + d);
          g4[u] = *(const float4*)(gb + (size_t)j * D + d);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j0 + u >= P) break;
          const float w1 = dS[r * P + j0 + u], w2 = dS[(j0 + u) * P + r], w3 = A[(j0 + u) * P + r];
          aq.x += w1 * k4[u].x; aq.y += w1 * k4[u].y; aq.z += w1 * k4[u].z; aq.w += w1 * k4[u].w;
          ak.x += w2 * q4[u].x; ak.y += w2 * q4[u].y; ak.z += w2 * q4[u].z; ak.w += w2 * q4[u].w;
          av.x += w3 * g4[u].x; av.y += w3 * g4[u].y; av.z += w3 * g4[u].z; av.w += w3 * g4[u].w;
        }
      }
      float* dst = dqkv + (row0 + r) * ld + head * hd + d;
      *(float4*)dst = make_float4(aq.x * q_scale, aq.y * q_scale, aq.z * q_scale, aq.w * q_scale);
      *(float4*)(dst + D) = ak;
      *(float4*)(dst + 2 * D) = av;
    }
  }
}

}  // namespace

#ifdef RRT_TRACE
RRT_TRACE_DEFINE_READER(rrt_debug_trace_attn_bwd)
#endif

// MFMA path: head dim 64, P <= 208.  Generic path: any head dim that is a multiple of 4, no EPEG, P <= 128.
bool attn_bwd_supported(int P, int D, int heads, int epeg_k) {
  if (heads <= 0 || P <= 0 || D % heads) return false;
  const int hd = D / heads;
  if (hd == HD) return epeg_k >= 0 && epeg_k <= 63;            // P <= 208 resident kernel, larger: streaming
  return epeg_k == 0 && P <= 128 && hd % 4 == 0;
}

// tap partials [n_regions, heads, k]; streaming variant (P > 208, head dim 64): + q~ [rows, D], tmp [rows, D],
// lse and D [rows, heads]
size_t attn_bwd_workspace(int n_regions, int P, int D, int heads, int epeg_k) {
  size_t b = ((size_t)n_regions * heads * (epeg_k > 0 ? epeg_k : 1) * sizeof(float) + 255) / 256 * 256;
  if (D == heads * HD && P > 48) {    // (streaming buffers also when the tuning hook forces that variant)
    const size_t rows = (size_t)n_regions * P;
    b += (2 * rows * D + 2 * rows * heads) * sizeof(float) + 1024;
  }
  return b;
}

// dqkv [n_regions*P, 3D] (gradient w.r.t. the qkv linear's raw output); dpe [heads, epeg_k] or null
hipError_t launch_attention_backward(const float* qkv, const float* pe_w, const float* O, const float* dO,
                                     float* dqkv, float* dpe, float* dpe_part, int n_regions, int P, int D,
                                     int heads, int epeg_k, hipStream_t st, ReduceJobs* defer, float* defer_part) {
  if (pe_w == nullptr) epeg_k = 0;
  hipError_t e;
  float* const ws_base = dpe_part;                          // the streaming variant's buffers sit behind the partials' slot
  if (defer && defer_part) dpe_part = defer_part;           // partials that must outlive this call (summed at the end)
  else defer = nullptr;
  if (D / heads != HD) {
    const size_t lds = (size_t)2 * P * P * sizeof(float);
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)attn_bwd_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attn_bwd_generic_kernel<<<dim3(heads, n_regions), 256, lds, st>>>(qkv, O, dO, dqkv, P, D, D / heads,
                                                                      1.0f / sqrtf((float)(D / heads)));
    return hipGetLastError();
  }
  static const bool force_stream = rrt_tune_env("RRT_ATTN_BWD_STREAM") != nullptr;   // tuning hook
  if (P > 208 || (force_stream && P > 48)) {
    const size_t rows = (size_t)n_regions * P;
    char* base = (char*)ws_base + ((size_t)n_regions * heads * (epeg_k > 0 ? epeg_k : 1) * sizeof(float) + 255) / 256 * 256;
    float* qt = (float*)base;
    float* tmp = qt + rows * D;
    float* lse_g = tmp + rows * D;
    float* dd_g = lse_g + rows * heads;
    const float q_scale = 1.0f / sqrtf((float)HD);
    const long n4 = (long)rows * (D / 4);
    attn_stencil_kernel<<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, st>>>(qkv, pe_w, qt, P, D, heads, epeg_k, (long)rows);
    const int groups = (P + 95) / 96;
    const size_t lq = (size_t)2 * CK * HD * sizeof(float), lkv = lq + 2 * CK * sizeof(float);
    static OncePerDevice once;
    if (once.first()) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lq);
      (void)hipFuncSetAttribute((const void*)attn_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lkv);
    }
    attn_bwd_q_kernel<<<dim3(heads, n_regions, groups), 384, lq, st>>>(qkv, qt, O, dO, dqkv, lse_g, dd_g, P, D, heads);
    attn_bwd_kv_kernel<<<dim3(heads, n_regions, groups), 384, lkv, st>>>(qkv, qt, dO, lse_g, dd_g, dqkv, P, D, heads);
    if (P <= 256) {
      constexpr int LADJ = (2 * 256 * HD + 128 + 8 * 256) * sizeof(float);
      static OncePerDevice once2;
      if (once2.first())
        (void)hipFuncSetAttribute((const void*)attn_adjoint_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LADJ);
      attn_adjoint_lds_kernel<<<dim3(heads, n_regions), 512, LADJ, st>>>(qkv, pe_w, dqkv, dpe_part, P, D, heads, epeg_k, q_scale);
    } else {
      attn_adjoint_kernel<<<dim3(heads, n_regions), 256, 0, st>>>(qkv, pe_w, dqkv, tmp, dpe_part, P, D, heads, epeg_k,
                                                                   q_scale);
    }
    e = hipGetLastError();
  } else if (P > 176) e = launch_bwd_mt<13>(qkv, pe_w, O, dO, dqkv, dpe_part, n_regions, P, D, heads, epeg_k, st);
  else if (P > 144) e = launch_bwd_mt<11>(qkv, pe_w, O, dO, dqkv, dpe_part, n_regions, P, D, heads, epeg_k, st);
  else if (P > 128) e = launch_bwd_mt<9>(qkv, pe_w, O, dO, dqkv, dpe_part, n_regions, P, D, heads, epeg_k, st);
  else if (P > 112) e = launch_bwd_mt<8>(qkv, pe_w, O, dO, dqkv, dpe_part, n_regions, P, D, heads, epeg_k, st);
  else if (P > 96) e = launch_bwd_mt<7>(qkv, pe_w, O, dO, dqkv, dpe_part, n_regions, P, D, heads, epeg_k, st);
  else if (P > 64) e = launch_bwd_mt<6>(qkv, pe_w, O, dO, dqkv, dpe_part, n_regions, P, D, heads, epeg_k, st);
  else e = launch_bwd_mt<4>(qkv, pe_w, O, dO, dqkv, dpe_part, n_regions, P, D, heads, epeg_k, st);
  if (e != hipSuccess || epeg_k == 0 || dpe == nullptr) return e;
  return reduce_or_defer(defer, dpe_part, dpe, n_regions, (size_t)heads * epeg_k, st);
}
