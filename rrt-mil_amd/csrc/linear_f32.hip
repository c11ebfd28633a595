// linear_f32.hip -- C[M,N] = A[M,K] . B[N,K]^T (+bias) on the fp32 matrix cores.
//
// Replaces nn.Linear in InnerAttention (qkv: modules/rmsa.py:100, proj: :131) plus,
// in the un-partition epilogue, region_reverse + un-pad + the TransLayer residual
// (modules/rmsa.py:41-54, :227-228; modules/rrt.py:125).
//
// MFMA-bound (exact fp32: v_mfma_f32_16x16x4_f32, 157 TFLOP/s chip peak, bitwise an
// fmaf chain).  Both operands are K-contiguous ("NT"), so A and B tiles are staged
// identically:
//   * global -> LDS by 16-byte DMA (global_load_lds_dwordx4 issued from inline asm so the
//     prefetch of K-tile t+1 really overlaps the MFMAs of tile t; SGPR base + per-lane
//     offset addressing, offsets hoisted out of the K loop), double buffered, BK = 32;
//   * the LDS image is [row][8 x 16-B slots]; slot p of a row holds logical k-slot
//     p ^ ((row>>1)&7).  The XOR is applied to the per-lane *global source* address (the
//     DMA destination is lane-linear) and again on the ds_read_b128 side: each 16-lane
//     read group then touches 16 distinct 16-B bank slots (SQ_LDS_BANK_CONFLICT = 0);
//   * a lane's float4 (4 consecutive k) feeds 4 MFMAs; lane group g = lane>>4 carries
//     k-slot 4*kk+g, so one b128 read per 16-row fragment covers 16 k.  The K-sum is
//     order-free, so no transposes are needed.
// Block = 4 waves side by side along N; each wave owns a (16*MT) x (16*NT) sub-tile, the
// block tile is (16*MT) x (64*NT).
//
// PERSISTENT grid: at most 2 blocks per CU (512), each walking a static list of tiles.
// Measured on MI355X: a non-persistent grid of 768 tiles (3 per CU) runs exactly as long
// as one of 1020 (4 per CU) -- the hardware dispatcher refills freed slots greedily,
// packing the last tiles two-per-CU -- and each tile pays ~15 us of un-overlapped
// prologue (first DMA) + epilogue (store-issue bound).  Here the tile shape is chosen so
// tiles divide evenly over the resident blocks (north star: 144 x 64 tiles, 1536 = 3 per
// block for qkv, 512 = 1 per block for proj) and the first DMA of a block's next tile is
// issued before the epilogue of the current one.
// The XCD-aware schedule gives the 64 resident blocks of an XCD a contiguous run of
// tiles per round (same A row-panels, all of W) so they share that XCD's 4 MiB L2.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "internal.h"

namespace {

constexpr int BK = 32;
constexpr int MAX_GRID = 512;   // 2 resident blocks per CU

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// operand precision of the MFMA (accumulation and everything around it stay fp32)
enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2, PREC_SPLIT = 3 };   // SPLIT: operands are (hi, lo) bf16 images, cast16.hip

template <int PREC>
struct Frag8;
template <>
struct Frag8<PREC_BF16> {
  using type = bf16x8;
  static __device__ __forceinline__ type pack(float4 a, float4 b) {   // v_cvt_pk_bf16_f32 (RNE)
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
  static __device__ __forceinline__ type pack(float4 a, float4 b) {   // v_cvt_pk_f16_f32 (RNE)
    type r;
    r[0] = (_Float16)a.x; r[1] = (_Float16)a.y; r[2] = (_Float16)a.z; r[3] = (_Float16)a.w;
    r[4] = (_Float16)b.x; r[5] = (_Float16)b.y; r[6] = (_Float16)b.z; r[7] = (_Float16)b.w;
    return r;
  }
  static __device__ __forceinline__ f32x4 mfma(type a, type b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};


// Epilogue shared by the kernels below.  The MFMAs are issued with the operand roles swapped
// (A-slot = W fragment, B-slot = X fragment), i.e. each 16x16 accumulator holds the TRANSPOSED
// output tile: reg r of lane (lr, lg) is C[m = m_tile + lr][n = n_tile + 4*lg + r].  A lane thus
// owns 4 consecutive output columns of one row -> one 16-byte store per tile (9 per wave and output
// tile instead of 36 dword stores; the dword form was store-issue bound), float4 bias / residual
// loads, and one slot->token map per row.
#ifdef RRT_TRACE
#define RRT_EPI_TRACE_ARG , WaveTrace& _tr
#define RRT_EPI_TRACE_PASS , _tr
#else
#define RRT_EPI_TRACE_ARG
#define RRT_EPI_TRACE_PASS
#endif
// Epilogue flavours (template MODE): plain store; un-partition + residual; activation.
constexpr int MODE_PLAIN = 0, MODE_UNPART = 1, MODE_ACT = 2, MODE_UNPART_DROP = 3, MODE_DROP = 4;   // 3, 4: training

// Elementwise activation of the epilogue (RRT_ACT_*; callers of the hot path: patch_to_emb's ReLU/GELU
// modules/rrt.py:208-217, DAttention's hidden activation / gate modules/datten.py:14-22,52-62).
__device__ __forceinline__ float activate(float v, int act) {
  switch (act) {
    case RRT_ACT_RELU: return v > 0.f ? v : 0.f;
    case RRT_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    case RRT_ACT_TANH: return tanhf(v);
    case RRT_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

template <int MT, int NT>
__device__ __forceinline__ void store_unpart_batched(const f32x4 (&acc)[MT][NT], const float (&bias)[NT][4],
                                                     float* __restrict__ C, int M, int N, int m0, int n0, int wave,
                                                     int lr, int lg, const LinearEpilogue& ep);

template <int MT, int NT, int MODE>
__device__ __forceinline__ void store_tile(const f32x4 (&acc)[MT][NT], float* __restrict__ C, int M, int N,
                                           int m0, int n0, int wave, int lr, int lg,
                                           const LinearEpilogue& ep RRT_EPI_TRACE_ARG) {
  constexpr bool UNPART = MODE == MODE_UNPART || MODE == MODE_UNPART_DROP, ACT = MODE == MODE_ACT, DROP = MODE >= MODE_UNPART_DROP;
  const bool vec = (N & 3) == 0;
  if constexpr (MODE == MODE_UNPART) {
    if (N % (64 * NT) == 0 && ep.q_cols == 0) {            // the straight-line form (see store_unpart_batched)
      float bz[NT][4];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bz[j][r] = ep.bias ? ep.bias[n0 + wave * (16 * NT) + j * 16 + 4 * lg + r] : 0.f;
      store_unpart_batched<MT, NT>(acc, bz, C, M, N, m0, n0, wave, lr, lg, ep);
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nb = n0 + wave * (16 * NT) + j * 16 + 4 * lg;      // first of this lane's 4 columns
    float bias[4], scale[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bias[r] = (ep.bias && nb + r < N) ? ep.bias[nb + r] : 0.f;
      scale[r] = (nb + r < ep.q_cols) ? ep.q_scale : 1.0f;
    }
#ifdef RRT_TRACE
    asm volatile("" :: "v"(bias[0]), "v"(bias[3]));
    RRT_TRACE_MARK();                                 // epilogue: bias landed
#endif
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + i * 16 + lr;
      if (m >= M) continue;
      size_t row = (size_t)m;
      if (UNPART) {
        const int t = slot_to_token(m, ep.g);
        if (t >= ep.g.L) continue;               // pad slot: nothing to write
        row = (size_t)t;
      }
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (acc[i][j][r] + bias[r]) * scale[r];
      if constexpr (ACT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = activate(v[r], ep.act);
      }
      if constexpr (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = rrt_drop_keep(ep.drop_seed, (unsigned long long)m * N + nb + r, ep.drop_thresh) ? v[r] * ep.drop_scale : 0.f;
      }
      float* dst = C + row * N + nb;
      if (vec && nb + 3 < N) {
        if (UNPART) {
          const float4 q = *(const float4*)(ep.resid + row * N + nb);
          v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
        }
        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
#ifdef RRT_TRACE
        if (i == 2 || i == 5) RRT_TRACE_MARK();       // epilogue: 3 / 6 row-tiles stored
#endif
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb + r < N) dst[r] = v[r] + (UNPART ? ep.resid[row * N + nb + r] : 0.f);
      }
    }
  }
}


// One 16-row slice (row tile I) of an output tile: the deferred epilogue of linear_ws_kernel issues
// these one per K iteration of the block's NEXT tile, so the chip never sees all blocks bursting
// their whole tiles at once (measured: ~1.2K cycles to issue ONE store while HBM writes saturate).
template <int MT, int NT, int MODE, int I>
__device__ __forceinline__ void store_slice(const f32x4 (&acc)[MT][NT], const float (&bias)[NT][4],
                                            float* __restrict__ C, int M, int N, int m0, int n0, int wave,
                                            int lr, int lg, const LinearEpilogue& ep) {
  constexpr bool UNPART = MODE == MODE_UNPART || MODE == MODE_UNPART_DROP, ACT = MODE == MODE_ACT, DROP = MODE >= MODE_UNPART_DROP;
  const int m = m0 + I * 16 + lr;
  if (m >= M) return;
  size_t row = (size_t)m;
  if (UNPART) {
    const int t = slot_to_token(m, ep.g);
    if (t >= ep.g.L) return;
    row = (size_t)t;
  }
  const bool vec = (N & 3) == 0;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nb = n0 + wave * (16 * NT) + j * 16 + 4 * lg;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (acc[I][j][r] + bias[j][r]) * ((nb + r < ep.q_cols) ? ep.q_scale : 1.0f);
    if constexpr (ACT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = activate(v[r], ep.act);
    }
    if constexpr (DROP) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[r] = rrt_drop_keep(ep.drop_seed, (unsigned long long)m * N + nb + r, ep.drop_thresh) ? v[r] * ep.drop_scale : 0.f;
    }
    float* dst = C + row * N + nb;
    if (vec && nb + 3 < N) {
      if (UNPART) {
        const float4 q = *(const float4*)(ep.resid + row * N + nb);
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
      }
      *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nb + r < N) dst[r] = v[r] + (UNPART ? ep.resid[row * N + nb + r] : 0.f);
    }
  }
}

// Un-partition + residual epilogue of a whole tile in straight-line code: every row's slot -> token map first, then ALL
// residual rows requested (unconditionally: rows that are not written re-read row 0), then the stores.  store_slice's
// form -- per row: map, branch, load, add, store -- left the compiler's wait-count bookkeeping with s_waitcnt vmcnt(0)
// in front of every load and every store (seen in the ISA: the per-row branches merge "anything may be outstanding"),
// i.e. one memory round trip per 16-row slice, MT of them in a row, with every block of the chip in the same phase: the
// "store-bound epilogue" the round-3 traces show (17 K cycles for a 96 x 64 tile).  Needs full 16-byte columns inside N and
// no column scale (q_cols == 0); rows in chunks of CH so that the residual values fit 32 registers.
template <int MT, int NT>
__device__ __forceinline__ void store_unpart_batched(const f32x4 (&acc)[MT][NT], const float (&bias)[NT][4],
                                                     float* __restrict__ C, int M, int N, int m0, int n0, int wave,
                                                     int lr, int lg, const LinearEpilogue& ep) {
  constexpr int CH = NT >= 8 ? 1 : 8 / NT;
  const int ncol = n0 + wave * (16 * NT) + 4 * lg;
#pragma unroll
  for (int i0 = 0; i0 < MT; i0 += CH) {
    long off[CH];
    float4 rq[CH][NT];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (i0 + c < MT) {
        const int m = m0 + (i0 + c) * 16 + lr;
        int t = -1;
        if (m < M) {
          t = slot_to_token(m, ep.g);
          if (t >= ep.g.L) t = -1;               // pad slot: nothing to write
        }
        off[c] = t < 0 ? -1L : (long)t * N + ncol;
#pragma unroll
        for (int j = 0; j < NT; ++j) rq[c][j] = *(const float4*)(ep.resid + (off[c] < 0 ? (long)ncol : off[c]) + j * 16);
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (i0 + c < MT) {
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(rq[c][j].x), "+v"(rq[c][j].y), "+v"(rq[c][j].z), "+v"(rq[c][j].w));
      }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (i0 + c < MT) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 a = acc[i0 + c][j];
          const float4 q = rq[c][j];
          const float4 v = make_float4(a[0] + bias[j][0] + q.x, a[1] + bias[j][1] + q.y, a[2] + bias[j][2] + q.z, a[3] + bias[j][3] + q.w);
          if (off[c] >= 0) *(float4*)(C + off[c] + j * 16) = v;
        }
      }
    }
  }
}

template <int MT, int NT, int MODE, int I = 0>
__device__ __forceinline__ void store_all_slices(const f32x4 (&acc)[MT][NT], const float (&bias)[NT][4],
                                                 float* __restrict__ C, int M, int N, int m0, int n0,
                                                 int wave, int lr, int lg, const LinearEpilogue& ep) {
  if constexpr (I < MT) {
    store_slice<MT, NT, MODE, I>(acc, bias, C, M, N, m0, n0, wave, lr, lg, ep);
    store_all_slices<MT, NT, MODE, I + 1>(acc, bias, C, M, N, m0, n0, wave, lr, lg, ep);
  }
}

template <int MT, int NT, int MODE, int PREC>
__global__ __launch_bounds__(256, 2) void linear_kernel(const float* __restrict__ A,
                                                        const float* __restrict__ B,
                                                        float* __restrict__ C, int M, int N, int K,
                                                        int tiles_n, int ntiles, LinearEpilogue ep) {
  constexpr int BM = 16 * MT, BN = 64 * NT;
  constexpr int STAGE = (BM + BN) * BK;   // floats per pipeline stage
  constexpr int NA = BM / 8, NB = BN / 8; // DMA wave-instructions per A / B tile (8 rows each)
  constexpr int QA = (NA + 3) / 4, QB = NB / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = (float*)smem;              // [2][A: BM*BK | B: BN*BK]

  RRT_TRACE_INIT(1 << 30);   // (untraced kernel: null tracer so the shared epilogue compiles)
  if (ep.zero64 != nullptr && blockIdx.x == 0 && threadIdx.x < 64) ep.zero64[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const unsigned lds_b = lds_addr_of(lds);

  // static XCD-aware schedule.  Block b sits on XCD b%8 (observed placement; only speed
  // depends on it).  Round i hands XCD x the contiguous tiles [i*G + x*G/8, +G/8).
  const int G = gridDim.x;
  int first;
  {
    const int b = blockIdx.x, q = G >> 3, r = G & 7, xcd = b & 7, idx = b >> 3;
    first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }

  // per-lane DMA source offsets (bytes from the tile's first row / k = 0), hoisted
  unsigned aoff[QA], boff[QB];
  auto tile_offsets = [&](int m0, int n0) {
#pragma unroll
    for (int qi = 0; qi < QA; ++qi) {
      int S = (qi * 4 + wave) * 64 + lane;
      int row = S >> 3, p = S & 7;
      int gr = m0 + row;
      gr = gr < M ? gr : M - 1;             // tail rows: re-read the last row (never stored)
      aoff[qi] = (unsigned)(gr - m0) * (unsigned)K * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
      int S = (qi * 4 + wave) * 64 + lane;
      int row = S >> 3, p = S & 7;
      int gr = n0 + row;
      gr = gr < N ? gr : N - 1;
      boff[qi] = (unsigned)(gr - n0) * (unsigned)K * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
    }
  };
  // (gr - m0) may be negative only when m0 >= M, which never happens for a valid tile
  auto stage = [&](const float* abase, const float* bbase, unsigned buf) {
#pragma unroll
    for (int qi = 0; qi < QA; ++qi)
      if (qi * 4 + wave < NA) dma16s(abase, aoff[qi], buf + (qi * 4 + wave) * 1024);
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) dma16s(bbase, boff[qi], buf + BM * BK * 4 + (qi * 4 + wave) * 1024);
  };

  const int nk = K / BK;
  int it = 0;   // running K-tile counter: pipeline stage = it & 1
  int tile = first;
  int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  if (tile < ntiles) {
    tile_offsets(tm * BM, tn * BN);
    stage(A + (size_t)tm * BM * K, B + (size_t)tn * BN * K, lds_b);
  }

  for (; tile < ntiles; tile += G) {
    const int m0 = tm * BM, n0 = tn * BN;
    const float* abase = A + (size_t)m0 * K;
    const float* bbase = B + (size_t)n0 * K;
    const int ntile = tile + G;
    const int ntm = ntile / tiles_n, ntn = ntile - ntm * tiles_n;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt, ++it) {
      wait_vm0();
      __syncthreads();   // K-tile `it` landed for every wave; everyone is done with the other buffer
      const float* As = lds + (it & 1) * STAGE;
      const float* Bs = As + BM * BK;
      const unsigned nxt = lds_b + ((it + 1) & 1) * STAGE * 4;
      if (kt + 1 < nk) {         // prefetch the next K tile; lands under the MFMAs below
        stage(abase + (kt + 1) * BK, bbase + (kt + 1) * BK, nxt);
      } else if (ntile < ntiles) {   // last K tile: prefetch the NEXT tile's first K tile
        tile_offsets(ntm * BM, ntn * BN);
        stage(A + (size_t)ntm * BM * K, B + (size_t)ntn * BN * K, nxt);
      }
      if constexpr (PREC == PREC_F32) {
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
          for (int i = 0; i < MT; ++i)
  #pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].x, af[i].x, acc[i][j], 0, 0, 0);
  #pragma unroll
          for (int i = 0; i < MT; ++i)
  #pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].y, af[i].y, acc[i][j], 0, 0, 0);
  #pragma unroll
          for (int i = 0; i < MT; ++i)
  #pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].z, af[i].z, acc[i][j], 0, 0, 0);
  #pragma unroll
          for (int i = 0; i < MT; ++i)
  #pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].w, af[i].w, acc[i][j], 0, 0, 0);
        }
      } else {
        // autocast-class numerics: operands rounded to bf16 / fp16 (RNE) as they leave LDS, fp32
        // accumulation.  One 16x16x32 MFMA covers the whole BK = 32 tile: lane group g holds
        // k = 8g..8g+7 of its row (logical slots 2g, 2g+1).
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

    store_tile<MT, NT, MODE>(acc, C, M, N, m0, n0, wave, lr, lg, ep RRT_EPI_TRACE_PASS);
    tm = ntm;
    tn = ntn;
  }
}


// ---------------------------------------------------------------------------------------
// Warp-specialised variant (large tiles): waves 0-3 compute, waves 4-5 load.
// Why: in linear_kernel every wave both issues LDS-DMA and stores its epilogue, and the only
// vector-memory counter (vmcnt) covers loads AND stores -- the first `s_waitcnt vmcnt(0)` of a
// block's next tile therefore waits for the previous tile's 36 epilogue stores to reach L2
// (measured: ~10 us per tile that two co-resident blocks hit in lockstep).  Here the loader waves
// own every DMA and every vmcnt wait; the compute waves never wait on vector memory, so their
// stores drain underneath the next tile's MFMAs.  One s_barrier per K tile, shared by all 6 waves.
// IN16: A and B already hold 16-bit operands in HBM (bf16 / fp16 per PREC; cast16.hip produces them): a K tile is
// 64 elements = the same 128-byte rows, so the staging ring, the XOR swizzle and the DMA schedule are byte for byte
// those of the fp32 form; a 16-byte LDS slot is one 16x16x32 MFMA operand (k = 8 lg .. 8 lg + 7 of a 32-wide
// sub-tile) and nothing is converted in the loop.
// NLW: loader waves (2 for the MFMA-bound fp32 forms; 4 for 16-bit operands, whose K loop is bound by how many LDS-DMA
// pieces the CU keeps in flight -- tools/ubench/dma_rows.hip: 14 B/clk/CU with four issuing waves, 21 with eight)
template <int MT, int NT, int MODE, int PREC, bool IN16 = false, bool DEFER = true, int NLW = 2>
__global__ __launch_bounds__(64 * (4 + NLW), 2)
void linear_ws_kernel(const void* __restrict__ Av,
                                                           const void* __restrict__ Bv,
                                                           float* __restrict__ C, int M, int N, int K,
                                                           int tiles_n, int ntiles, LinearEpilogue ep) {
  static_assert(!IN16 || PREC != PREC_F32, "16-bit operands need a 16-bit MFMA");
  constexpr int BM = 16 * MT, BN = 64 * NT;
  constexpr int ES = IN16 ? 2 : 4;                       // bytes per operand element in HBM
  constexpr int BKE = 128 / ES;                          // elements per K tile (BK floats = 128 bytes either way)
  const char* const A = (const char*)Av;
  const char* const B = (const char*)Bv;
  constexpr int STAGE = (BM + BN) * BK;
  constexpr int NA = BM / 8, NB = BN / 8;                // DMA wave-instructions per A / B tile
  constexpr int LA = (NA + NLW - 1) / NLW, LB = (NB + NLW - 1) / NLW;   // per loader wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = (float*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_b = lds_addr_of(lds);

  const int G = gridDim.x;
  int first;
  {
    const int b = blockIdx.x, q = G >> 3, r = G & 7, xcd = b & 7, idx = b >> 3;
    first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  if (ep.zero64 != nullptr && blockIdx.x == 0 && threadIdx.x < 64) ep.zero64[threadIdx.x] = 0;
  if (first >= ntiles) return;
  const int nk = K / BKE;
  RRT_TRACE_INIT(blockIdx.x * (4 + NLW) + wave);
  RRT_TRACE_MARK();                                   // [1] entry

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wave - 4;
    unsigned aoff[LA], boff[LB];
    auto tile_offsets = [&](int m0, int n0) {
#pragma unroll
      for (int qi = 0; qi < LA; ++qi) {
        int S = (qi * NLW + lw) * 64 + lane;
        int row = S >> 3, p = S & 7;
        int gr = m0 + row;
        gr = gr < M ? gr : M - 1;
        aoff[qi] = (unsigned)(gr - m0) * (unsigned)K * (unsigned)ES + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
      }
#pragma unroll
      for (int qi = 0; qi < LB; ++qi) {
        int S = (qi * NLW + lw) * 64 + lane;
        int row = S >> 3, p = S & 7;
        int gr = n0 + row;
        gr = gr < N ? gr : N - 1;
        boff[qi] = (unsigned)(gr - n0) * (unsigned)K * (unsigned)ES + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
      }
    };
    auto stage = [&](const char* abase, const char* bbase, unsigned buf) {
#pragma unroll
      for (int qi = 0; qi < LA; ++qi)
        if (qi * NLW + lw < NA) {
#ifdef RRT_NT_A16_PROJ
          if (IN16 && MODE == MODE_UNPART) dma16s_nt(abase, aoff[qi], buf + (qi * NLW + lw) * 1024);
          else
#endif
          dma16s(abase, aoff[qi], buf + (qi * NLW + lw) * 1024);
        }
#pragma unroll
      for (int qi = 0; qi < LB; ++qi)
        if (qi * NLW + lw < NB) dma16s(bbase, boff[qi], buf + BM * BK * 4 + (qi * NLW + lw) * 1024);
    };
    int tile = first;
    int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    tile_offsets(tm * BM, tn * BN);
    stage(A + (size_t)tm * BM * K * ES, B + (size_t)tn * BN * K * ES, lds_b);
    RRT_TRACE_MARK();                                 // loader [2] first stage issued
    int it = 0;
    for (; tile < ntiles; tile += G) {
      const int ntile = tile + G;
      const int ntm = ntile / tiles_n, ntn = ntile - ntm * tiles_n;
      for (int kt = 0; kt < nk; ++kt, ++it) {
        wait_vm0();
        if (kt == 0 || kt == 8) RRT_TRACE_MARK();     // loader: K tile 0 / 8 landed (before barrier)
        __syncthreads();          // publishes K tile `it`; compute waves are done with the other buffer
        if (kt == 0 || kt == 8) RRT_TRACE_MARK();     // loader: barrier passed
        const unsigned nxt = lds_b + ((it + 1) & 1) * STAGE * 4;
        if (kt + 1 < nk) {
          stage(A + ((size_t)tm * BM * K + (size_t)(kt + 1) * BKE) * ES, B + ((size_t)tn * BN * K + (size_t)(kt + 1) * BKE) * ES, nxt);
        } else if (ntile < ntiles) {
          tile_offsets(ntm * BM, ntn * BN);
          stage(A + (size_t)ntm * BM * K * ES, B + (size_t)ntn * BN * K * ES, nxt);
        }
      }
      tm = ntm;
      tn = ntn;
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int lr = lane & 15, lg = lane >> 4;
  int it = 0;
  // deferred epilogue state: the previous tile's accumulators, bias and origin.  DEFER = false (the launcher picks it
  // when no block gets a second tile): the tile is stored straight from its accumulators -- 36 registers fewer at
  // MT = 9, which is what lets two blocks share a CU (traced at M = 9216, N = 512: with 142 registers the blocks ran
  // one per CU, two rounds of 49 K cycles for 37 K of MFMA each)
  f32x4 prev[DEFER ? MT : 1][NT];
  float pbias[NT][4];
  int pm0 = 0, pn0 = 0;
  bool have_prev = false;
  auto load_bias = [&](int n0_, float (&bz)[NT][4]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nb = n0_ + wave * (16 * NT) + j * 16 + 4 * lg;
#pragma unroll
      for (int r = 0; r < 4; ++r) bz[j][r] = (ep.bias && nb + r < N) ? ep.bias[nb + r] : 0.f;
    }
  };
  // deferred un-partition epilogue: the next slice's residual row, requested a K iteration ahead (full 16-byte columns
  // only; q_scale does not apply to this epilogue)
  const bool pref_ok = MODE == MODE_UNPART && N % (64 * NT) == 0 && ep.q_cols == 0;
  // Every other instantiation (plain / activation / dropout epilogues, the non-deferred form) keeps the round-3 code below
  // verbatim: restructuring the tile loop for all of them moved the plain deferred GEMM <8, 1> from 128 to 130 VGPRs and
  // cost the N = 15000 qkv projection 5 % (measured, same box).
  if constexpr (MODE == MODE_UNPART && DEFER) {
  // Branch-free form (round 4): the residual rows come through buffer loads and the slices leave through buffer stores of
  // `out` / `resid` as raw buffers of L * N floats -- a row that is not written (pad slot, no previous tile, slices used up)
  // has the byte offset OOB, which the hardware answers with zeros / drops.  With per-row branches the compiler waited
  // vmcnt(0) in front of every deferred store (one in-order counter for loads and stores; a branch merges to "anything may
  // be outstanding"): the residual request of the next slice AND the store of the previous one on the critical path of
  // every K iteration.  Here the next slice's request is issued BEFORE this slice's store, so the wait for a residual row
  // counts the NT stores and NT requests behind it instead of draining them.
  constexpr unsigned OOB = 0xFFFFFF00u;
  // the sentinel + the last slice's column offset + one 16-byte access must not wrap into the buffer (NT <= 4 today; a wider
  // instantiation has to fold 64 j into the instruction's soffset instead)
  static_assert(64u * (NT - 1) + 16u <= 0x100u, "OOB sentinel would wrap to an in-range offset");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned out_bytes = pref_ok ? (unsigned)((size_t)ep.g.L * N * 4) : 0u;
  __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (int)out_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)ep.resid, 0, (int)out_bytes, 0x00020000);
  float4 rq[NT];
  unsigned roff = OOB;
  auto slice_off = [&](int mbase, int nbase, bool valid) -> unsigned {
    const int m = mbase + lr;
    const int t = slot_to_token(m < M ? m : 0, ep.g);
    const bool ok = valid && m < M && t < ep.g.L;
    return ok ? (unsigned)(((size_t)t * N + nbase + wave * (16 * NT) + 4 * lg) * 4) : OOB;
  };
  auto resid_request = [&](unsigned off, float4 (&dst)[NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
      dst[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off + 64u * j, 0, NT_RESID ? 2 : 0));   // aux 2 = nt
  };
  // the tile loop in two instantiations: PF = the branch-free un-partition epilogue applies (decided once per launch, so
  // that neither form's memory operations sit in the other's loop and blur its wait counts)
  float4 rq2[NT];
  auto tiles = [&](auto pfc) {
  constexpr bool PF = decltype(pfc)::value;
  for (int tile = first; tile < ntiles; tile += G) {
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    float cbias[NT][4];
    load_bias(n0, cbias);                 // lands under the K loop
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto kstep = [&](int kt, auto parity) {        // parity: which of the two residual-row buffers holds THIS slice's row
      constexpr bool PAR = decltype(parity)::value;
      // compute side of the tile barrier: this wave's LDS reads of the previous K tile retired + s_barrier.  Not
      // __syncthreads(): that also drains vmcnt -- the bias row, the deferred epilogue's residual request and its stores
      // of the previous slice -- i.e. a memory round trip on the critical path of every K iteration (the loader waves
      // keep theirs: vmcnt(0) is how they know the DMA has landed)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (kt == 0 || kt == 1 || kt == 8 || kt == 9) RRT_TRACE_MARK();   // compute: barrier kt passed
      const float* As = lds + (it & 1) * STAGE;
      const float* Bs = As + BM * BK;
      if constexpr (PREC == PREC_F32) {
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
      } else if constexpr (PREC == PREC_SPLIT) {
        // fp32 emulated on the bf16 matrix cores: a = ah + al, b = bh + bl (bf16 each), a.b ~ ah.bh + ah.bl + al.bh
        // (the dropped al.bl term is 2^-16 of the product).  A tile row = [32 hi | 32 lo]: slots 0..3 / 4..7.
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = wave * (16 * NT) + j * 16 + lr, f = (row >> 1) & 7;
          bh[j] = *(const bf16x8*)(Bs + row * BK + ((lg ^ f) << 2));
          bl[j] = *(const bf16x8*)(Bs + row * BK + (((4 + lg) ^ f) << 2));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = i * 16 + lr, f = (row >> 1) & 7;
          ah[i] = *(const bf16x8*)(As + row * BK + ((lg ^ f) << 2));
          al[i] = *(const bf16x8*)(As + row * BK + (((4 + lg) ^ f) << 2));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
          }
      } else if constexpr (IN16) {
        using F = Frag8<PREC>;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          typename F::type a8[MT], b8[NT];
          const int cslot = 4 * kk + lg;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int row = wave * (16 * NT) + j * 16 + lr;
            b8[j] = *(const typename F::type*)(Bs + row * BK + ((cslot ^ ((row >> 1) & 7)) << 2));
          }
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int row = i * 16 + lr;
            a8[i] = *(const typename F::type*)(As + row * BK + ((cslot ^ ((row >> 1) & 7)) << 2));
          }
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = F::mfma(b8[j], a8[i], acc[i][j]);
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
      // one 16-row slice of the PREVIOUS tile per K iteration (behind this iteration's MFMAs)
      // (static register indexing: always slice 0, then rotate the remaining slices down)
      if constexpr (DEFER) {
        // (fp32: K / 32 iterations for MT slices -- the iterations behind the last slice skip the block: their requests
        // and stores would all be out of range, issued for nothing; an fp32 K iteration is thousands of cycles of MFMA and
        // does not notice the uniform branch's conservative wait.  16-bit operands: K / 64 = 8 iterations, no branch.)
        if constexpr (PF) {
        if (IN16 || kt <= MT) {
          // (no `have_prev && kt < MT` branch: the offsets carry it; no copy of the requested row either: the two row
          // buffers swap roles from one K iteration to the next -- a copy would wait for the request on the spot)
          float4 (&rqc)[NT] = PAR ? rq2 : rq;
          float4 (&rqnx)[NT] = PAR ? rq : rq2;
          const unsigned offn = slice_off(pm0 + (kt + 1) * 16, pn0, have_prev && kt + 1 < MT);
          resid_request(offn, rqnx);                // the next slice's residual row first ...
#pragma unroll
          for (int j = 0; j < NT; ++j) {            // ... then this slice (its row was requested a K iteration ago)
            const float4 q = rqc[j];
            // (the bias went into `prev` at the hand-over: four registers that this form does not hold across the K loop --
            // at MT = 8 the difference between two blocks per CU and one)
            const float4 v = make_float4(prev[0][j][0] + q.x, prev[0][j][1] + q.y, prev[0][j][2] + q.z, prev[0][j][3] + q.w);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, roff + 64u * j, 0, 0);
          }
          roff = offn;
#pragma unroll
          for (int i = 0; i + 1 < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) prev[i][j] = prev[i + 1][j];
        }
        } else if (have_prev && kt < MT) {
          store_slice<MT, NT, MODE, 0>(prev, pbias, C, M, N, pm0 + kt * 16, pn0, wave, lr, lg, ep);
#pragma unroll
          for (int i = 0; i + 1 < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) prev[i][j] = prev[i + 1][j];
        }
      }
        };
    if constexpr (!PF) {
      for (int kt = 0; kt < nk; ++kt, ++it) kstep(kt, std::false_type{});
    } else {
      int kt = 0;
      for (; kt + 1 < nk; kt += 2) {
        kstep(kt, std::false_type{});
        ++it;
        kstep(kt + 1, std::true_type{});
        ++it;
      }
      if (kt < nk) {                                  // odd K tile count: one copy (and its wait) per tile
        kstep(kt, std::false_type{});
        ++it;
#pragma unroll
        for (int j = 0; j < NT; ++j) rq[j] = rq2[j];
      }
    }
    if constexpr (!DEFER) {
      RRT_TRACE_MARK();                               // compute: last MFMA of the tile issued
      if (MODE == MODE_UNPART && pref_ok) store_unpart_batched<MT, NT>(acc, cbias, C, M, N, m0, n0, wave, lr, lg, ep);
      else store_all_slices<MT, NT, MODE>(acc, cbias, C, M, N, m0, n0, wave, lr, lg, ep);
      RRT_TRACE_MARK();
    }
    if constexpr (DEFER) {
    if (have_prev)                        // K shorter than MT iterations: flush what is left
      for (int i = nk; i < MT; ++i) {
        store_slice<MT, NT, MODE, 0>(prev, pbias, C, M, N, pm0 + i * 16, pn0, wave, lr, lg, ep);
#pragma unroll
        for (int q = 0; q + 1 < MT; ++q)
#pragma unroll
          for (int j = 0; j < NT; ++j) prev[q][j] = prev[q + 1][j];
      }

    RRT_TRACE_MARK();                                 // compute: last MFMA of the tile issued
    // hand the tile over to the deferred epilogue (stores are fire-and-forget: nothing in this
    // wave ever waits on vmcnt for them)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        prev[i][j] = acc[i][j];
        if constexpr (PF) {
#pragma unroll
          for (int r = 0; r < 4; ++r) prev[i][j][r] += cbias[j][r];
        }
      }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) pbias[j][r] = PF ? 0.f : cbias[j][r];
    pm0 = m0;
    pn0 = n0;
    have_prev = true;
    if constexpr (PF) {                                                 // slice 0 of the tile just handed over
      roff = slice_off(pm0, pn0, true);
      resid_request(roff, rq);
    }
    RRT_TRACE_MARK();
    }
  }
  // the block's last tile has no successor to hide behind
  if constexpr (DEFER)
    if (have_prev) {
      if (PF) store_unpart_batched<MT, NT>(prev, pbias, C, M, N, pm0, pn0, wave, lr, lg, ep);
      else store_all_slices<MT, NT, MODE>(prev, pbias, C, M, N, pm0, pn0, wave, lr, lg, ep);
    }
  };
  if constexpr (MODE == MODE_UNPART && DEFER) {     // (the non-deferred form has no K-loop epilogue: one instantiation)
    // (the buffer form addresses `out` / `resid` with 32-bit byte offsets)
    // (fp32 144 x 128 tiles: the form does not fit 256 VGPRs -- 17 spilled -- and keeps the per-slice one)
    if constexpr (IN16 || MT * NT <= 16) {
      if (pref_ok && (size_t)ep.g.L * N * 4 <= 0xFFFFF000u) tiles(std::true_type{});
      else tiles(std::false_type{});
    } else {
      tiles(std::false_type{});
    }
  } else {
    tiles(std::false_type{});
  }
  } else {
  float4 orq[NT];
  long rrow = -1;
  auto resid_prefetch = [&](int mbase, int nbase) {
    const int m = mbase + lr;
    rrow = -1;
    if (m < M) {
      const int t = slot_to_token(m, ep.g);
      if (t < ep.g.L) rrow = t;
    }
    if (rrow >= 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) orq[j] = *(const float4*)(ep.resid + (size_t)rrow * N + nbase + wave * (16 * NT) + j * 16 + 4 * lg);
    }
  };
  for (int tile = first; tile < ntiles; tile += G) {
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    float cbias[NT][4];
    load_bias(n0, cbias);                 // lands under the K loop
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto kstep = [&](int kt) {
      // compute side of the tile barrier: this wave's LDS reads of the previous K tile retired + s_barrier.  Not
      // __syncthreads(): that also drains vmcnt -- the bias row, the deferred epilogue's residual request and its stores
      // of the previous slice -- i.e. a memory round trip on the critical path of every K iteration (the loader waves
      // keep theirs: vmcnt(0) is how they know the DMA has landed)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (kt == 0 || kt == 1 || kt == 8 || kt == 9) RRT_TRACE_MARK();   // compute: barrier kt passed
      const float* As = lds + (it & 1) * STAGE;
      const float* Bs = As + BM * BK;
      if constexpr (PREC == PREC_F32) {
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
      } else if constexpr (PREC == PREC_SPLIT) {
        // fp32 emulated on the bf16 matrix cores: a = ah + al, b = bh + bl (bf16 each), a.b ~ ah.bh + ah.bl + al.bh
        // (the dropped al.bl term is 2^-16 of the product).  A tile row = [32 hi | 32 lo]: slots 0..3 / 4..7.
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = wave * (16 * NT) + j * 16 + lr, f = (row >> 1) & 7;
          bh[j] = *(const bf16x8*)(Bs + row * BK + ((lg ^ f) << 2));
          bl[j] = *(const bf16x8*)(Bs + row * BK + (((4 + lg) ^ f) << 2));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = i * 16 + lr, f = (row >> 1) & 7;
          ah[i] = *(const bf16x8*)(As + row * BK + ((lg ^ f) << 2));
          al[i] = *(const bf16x8*)(As + row * BK + (((4 + lg) ^ f) << 2));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
          }
      } else if constexpr (IN16) {
        using F = Frag8<PREC>;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          typename F::type a8[MT], b8[NT];
          const int cslot = 4 * kk + lg;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int row = wave * (16 * NT) + j * 16 + lr;
            b8[j] = *(const typename F::type*)(Bs + row * BK + ((cslot ^ ((row >> 1) & 7)) << 2));
          }
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int row = i * 16 + lr;
            a8[i] = *(const typename F::type*)(As + row * BK + ((cslot ^ ((row >> 1) & 7)) << 2));
          }
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = F::mfma(b8[j], a8[i], acc[i][j]);
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
      // one 16-row slice of the PREVIOUS tile per K iteration (behind this iteration's MFMAs)
      // (static register indexing: always slice 0, then rotate the remaining slices down)
      if constexpr (DEFER) {
        if (have_prev && kt < MT) {
          if (MODE == MODE_UNPART && pref_ok) {
            // the slice's residual row was requested one K iteration ago (the load's ~1 us round trip sat on the critical
            // path of every iteration: the wave waited for it before it could issue the store -- and the next MFMAs)
            if (rrow >= 0) {
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                const int nb = pn0 + wave * (16 * NT) + j * 16 + 4 * lg;
                const float4 q = orq[j];
                *(float4*)(C + (size_t)rrow * N + nb) = make_float4(prev[0][j][0] + pbias[j][0] + q.x, prev[0][j][1] + pbias[j][1] + q.y,
                                                                    prev[0][j][2] + pbias[j][2] + q.z, prev[0][j][3] + pbias[j][3] + q.w);
              }
            }
            if (kt + 1 < MT) resid_prefetch(pm0 + (kt + 1) * 16, pn0);
          } else {
            store_slice<MT, NT, MODE, 0>(prev, pbias, C, M, N, pm0 + kt * 16, pn0, wave, lr, lg, ep);
          }
#pragma unroll
          for (int i = 0; i + 1 < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) prev[i][j] = prev[i + 1][j];
        }
      }
        };
    for (int kt = 0; kt < nk; ++kt, ++it) kstep(kt);
    if constexpr (!DEFER) {
      RRT_TRACE_MARK();                               // compute: last MFMA of the tile issued
      if (MODE == MODE_UNPART && pref_ok) store_unpart_batched<MT, NT>(acc, cbias, C, M, N, m0, n0, wave, lr, lg, ep);
      else store_all_slices<MT, NT, MODE>(acc, cbias, C, M, N, m0, n0, wave, lr, lg, ep);
      RRT_TRACE_MARK();
    }
    if constexpr (DEFER) {
    if (have_prev)                        // K shorter than MT iterations: flush what is left
      for (int i = nk; i < MT; ++i) {
        store_slice<MT, NT, MODE, 0>(prev, pbias, C, M, N, pm0 + i * 16, pn0, wave, lr, lg, ep);
#pragma unroll
        for (int q = 0; q + 1 < MT; ++q)
#pragma unroll
          for (int j = 0; j < NT; ++j) prev[q][j] = prev[q + 1][j];
      }

    RRT_TRACE_MARK();                                 // compute: last MFMA of the tile issued
    // hand the tile over to the deferred epilogue (stores are fire-and-forget: nothing in this
    // wave ever waits on vmcnt for them)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) prev[i][j] = acc[i][j];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) pbias[j][r] = cbias[j][r];
    pm0 = m0;
    pn0 = n0;
    have_prev = true;
    if (MODE == MODE_UNPART && pref_ok) resid_prefetch(pm0, pn0);      // slice 0 of the tile just handed over
    RRT_TRACE_MARK();
    }
  }
  // the block's last tile has no successor to hide behind
  if constexpr (DEFER)
    if (have_prev) {
      if (MODE == MODE_UNPART && pref_ok) store_unpart_batched<MT, NT>(prev, pbias, C, M, N, pm0, pn0, wave, lr, lg, ep);
      else store_all_slices<MT, NT, MODE>(prev, pbias, C, M, N, pm0, pn0, wave, lr, lg, ep);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Small-M GEMMs (the 64 k representatives of CR-MSA: M = 192..512 rows): split K INSIDE the block.
// linear_kernel's K loop is a chain of K / 32 dependent steps (DMA round trip + 16 MFMAs each): at M = 192 a launch
// is 144 blocks of one 32 x 64 tile that spend ~6 us walking sixteen K tiles one after the other, on top of a ~4 us
// launch floor.  Here a block is KG groups of four waves; group g multiplies the K tiles g, g + KG, ... of the SAME
// output tile with its own two-stage ring (all groups step in lockstep, so the block barrier still works), the KG
// partial accumulators are summed through LDS in a fixed order and group 0 runs the epilogue.  Four times fewer
// dependent steps; the MFMA work per SIMD is unchanged (wave g * 4 + c sits on SIMD c).  Exact fp32, plain epilogue
// (bias, q-scale): what the inner MSA's qkv and proj need.
// DROP: the training forward's projection (round 6: it ran the generic kernel's 48 blocks of sixteen dependent K tiles, 18 us
// against 8.7 us here); the mask is the generic epilogue's, a function of (seed, element index).
template <int MT, int KG, bool DROP = false>
__global__ __launch_bounds__(256 * KG, 1) void linear_splitk_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                     float* __restrict__ C, int M, int N, int K, int tiles_n,
                                                                     LinearEpilogue ep) {
  constexpr int BM = 16 * MT, BN = 64;
  constexpr int STAGE = (BM + BN) * BK;             // floats per stage
  constexpr int NA = BM / 8, NB = BN / 8;
  constexpr int QA = (NA + 3) / 4, QB = NB / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kg = wave >> 2, cw = wave & 3;          // K group; 16-column tile
  const int lr = lane & 15, lg = lane >> 4;
  float* lds = (float*)smem + kg * 2 * STAGE;       // this group's ring
  const unsigned lds_b = lds_addr_of(lds);
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  unsigned aoff[QA], boff[QB];
#pragma unroll
  for (int qi = 0; qi < QA; ++qi) {
    const int S = (qi * 4 + cw) * 64 + lane, row = S >> 3, p = S & 7;
    int gr = m0 + row;
    gr = gr < M ? gr : M - 1;
    aoff[qi] = (unsigned)(gr - m0) * (unsigned)K * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int qi = 0; qi < QB; ++qi) {
    const int S = (qi * 4 + cw) * 64 + lane, row = S >> 3, p = S & 7;
    int gr = n0 + row;
    gr = gr < N ? gr : N - 1;
    boff[qi] = (unsigned)(gr - n0) * (unsigned)K * 4u + (unsigned)((p ^ ((row >> 1) & 7)) << 4);
  }
  const float* abase = A + (size_t)m0 * K;
  const float* bbase = B + (size_t)n0 * K;
  auto stage = [&](int kt, unsigned buf) {
#pragma unroll
    for (int qi = 0; qi < QA; ++qi)
      if (qi * 4 + cw < NA) dma16s(abase + kt * BK, aoff[qi], buf + (qi * 4 + cw) * 1024);
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) dma16s(bbase + kt * BK, boff[qi], buf + BM * BK * 4 + (qi * 4 + cw) * 1024);
  };
  const int nk = K / BK, nkg = (nk + KG - 1) / KG;  // K tiles of this group: kg, kg + KG, ...
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (kg < nk) stage(kg, lds_b);
  for (int it = 0; it < nkg; ++it) {
    const int kt = kg + it * KG;
    wait_vm0();
    __syncthreads();
    if (kt + KG < nk) stage(kt + KG, lds_b + ((it + 1) & 1) * STAGE * 4);
    if (kt < nk) {
      const float* As = lds + (it & 1) * STAGE;
      const float* Bs = As + BM * BK;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int cslot = 4 * kk + lg;
        const int brow = cw * 16 + lr;
        const float4 bf = *(const float4*)(Bs + brow * BK + ((cslot ^ ((brow >> 1) & 7)) << 2));
        float4 af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = i * 16 + lr;
          af[i] = *(const float4*)(As + row * BK + ((cslot ^ ((row >> 1) & 7)) << 2));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.x, af[i].x, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.y, af[i].y, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.z, af[i].z, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.w, af[i].w, acc[i], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();                                  // the rings are dead: they carry the partial tiles now
  f32x4* red = (f32x4*)smem;                        // [KG - 1][4 column waves][MT][64 lanes]
  if (kg > 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) red[(((kg - 1) * 4 + cw) * MT + i) * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (kg == 0) {
    f32x4 out[MT][1];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x4 a = acc[i];
#pragma unroll
      for (int g = 1; g < KG; ++g) {                // fixed order: bit-reproducible
        const f32x4 b = red[(((g - 1) * 4 + cw) * MT + i) * 64 + lane];
        a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
      }
      out[i][0] = a;
    }
    RRT_TRACE_INIT(1 << 30);
    store_tile<MT, 1, DROP ? MODE_DROP : MODE_PLAIN>(out, C, M, N, m0, n0, cw, lr, lg, ep RRT_EPI_TRACE_PASS);
  }
}

// > 64 KiB of dynamic LDS needs the attribute, once per kernel per device: the static lives at the call site (inside
// the launcher's template instantiation), one per kernel
#define RRT_ALLOW_LDS(kern, lds_bytes)                                                                          \
  do {                                                                                                          \
    static OncePerDevice once_;                                                                                 \
    if ((lds_bytes) > 64 * 1024 && once_.first())                                                               \
      (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (lds_bytes));  \
  } while (0)

#ifndef RRT_NLW16
#define RRT_NLW16 4
#endif
constexpr int NLW16 = RRT_NLW16;      // loader waves of the 16-bit-operand GEMMs
template <int MT, int NT, int MODE, int PREC>
hipError_t launch_cfg16(const void* A, const void* B, float* C, int M, int N, int K, int grid_cap,
                        const LinearEpilogue& ep, hipStream_t st) {
  constexpr int BM = 16 * MT, BN = 64 * NT;
  constexpr int LDS_BYTES = 2 * (BM + BN) * BK * 4;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  const int grid = ntiles < grid_cap ? ntiles : grid_cap;
  if (ntiles <= grid) {                                // no block gets a second tile: nothing to defer
    auto kws = linear_ws_kernel<MT, NT, MODE, PREC, true, false, NLW16>;
    RRT_ALLOW_LDS(kws, LDS_BYTES);
    kws<<<dim3(grid), dim3(64 * (4 + NLW16)), LDS_BYTES, st>>>(A, B, C, M, N, K, tiles_n, ntiles, ep);
  } else {
    auto kws = linear_ws_kernel<MT, NT, MODE, PREC, true, true, NLW16>;
    RRT_ALLOW_LDS(kws, LDS_BYTES);
    kws<<<dim3(grid), dim3(64 * (4 + NLW16)), LDS_BYTES, st>>>(A, B, C, M, N, K, tiles_n, ntiles, ep);
  }
  return hipGetLastError();
}

template <int MT, int NT, int MODE, int PREC>
hipError_t launch_cfg(const float* A, const float* B, float* C, int M, int N, int K, int grid_cap,
                      const LinearEpilogue& ep, hipStream_t st) {
  constexpr int BM = 16 * MT, BN = 64 * NT;
  constexpr int LDS_BYTES = 2 * (BM + BN) * BK * 4;
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two blocks per CU must fit the 160 KiB LDS");
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  const int grid = ntiles < grid_cap ? ntiles : grid_cap;
  if constexpr (MT >= 6) {
    static const bool use_ws = rrt_tune_env("RRT_LINEAR_NO_WS") == nullptr;
    if (use_ws) {
      static const bool force_defer = rrt_tune_env("RRT_LINEAR_DEFER") != nullptr;   // tuning hook
      if (ntiles <= grid && !force_defer) {            // no block gets a second tile: nothing to defer
        auto kws = linear_ws_kernel<MT, NT, MODE, PREC, false, false>;
        RRT_ALLOW_LDS(kws, LDS_BYTES);
        kws<<<dim3(grid), dim3(384), LDS_BYTES, st>>>(A, B, C, M, N, K, tiles_n, ntiles, ep);
      } else {
        auto kws = linear_ws_kernel<MT, NT, MODE, PREC, false, true>;
        RRT_ALLOW_LDS(kws, LDS_BYTES);
        kws<<<dim3(grid), dim3(384), LDS_BYTES, st>>>(A, B, C, M, N, K, tiles_n, ntiles, ep);
      }
      return hipGetLastError();
    }
  }
  if constexpr (PREC == PREC_SPLIT) {
    return hipErrorInvalidValue;          // the split form exists in the warp-specialised kernel only
  } else {
    auto kern = linear_kernel<MT, NT, MODE, PREC>;
    if (LDS_BYTES > 64 * 1024) {
      static OncePerDevice once;
      if (once.first())
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    kern<<<dim3(grid), dim3(256), LDS_BYTES, st>>>(A, B, C, M, N, K, tiles_n, ntiles, ep);
    return hipGetLastError();
  }
}

struct Cfg { int mt, nt, cap; };

// Tile shape + resident-block count that give the busiest CU the least MFMA work:
//   grid = min(tiles, cap), rounds = ceil(tiles / grid), blocks per CU = ceil(grid / 256)
//   cost = rounds * blocks_per_cu * MT * NT      [co-resident blocks share the CU's matrix pipes]
// Candidates are ordered by preference; a later one must be strictly cheaper to win.
Cfg choose(int M, int N, int prec = PREC_F32, int K = 0) {
  if (const char* e = rrt_tune_env("RRT_LINEAR_CFG_BIG")) {   // tuning hook for the bag-sized GEMMs only: "mt,nt,cap"
    Cfg c{};
    if (M > 1024 && sscanf(e, "%d,%d,%d", &c.mt, &c.nt, &c.cap) == 3) return c;
  }
  // The R-MSA out-projection of a bag of ~7.6-9.2 k tokens (N = 512, 640..768 tiles of 96 rows): a 96 x 64 block needs 40 KiB
  // of LDS, which fits NEXT TO a block of the other bag's fused R-MSA kernel (114 KiB of the CU's 160) -- the 144-row
  // block's 52 KiB does not, so with two bags in flight that GEMM could not start on a CU until the fused block
  // retired.  Alone the two shapes take the same time (46.5 us at N = 9000); two bags in flight: 4.79 -> 4.84 k slides/s.
  if (prec == PREC_F32 && N == 512) {
    const long t96 = (long)((M + 95) / 96) * 8;
    if (t96 >= 640 && t96 <= 768) return Cfg{6, 1, 768};
  }
  // Round 3: the out-projection of every other bag size (N = K = 512, fp32), from a sweep of all shapes x caps over bags of
  // 3 k .. 15 k tokens (tools/sweep_proj_cfg.py).  What the numbers say: a block alone on a CU is latency-bound (~19 us +
  // 1.2 us per row tile whatever its size), k co-resident blocks of one tile each take max(that, k * MT * 2.6 us), and
  // a second tile behind the first costs a whole extra round (the block runs it alone).  So: the single-round shape with the
  // least max(floor, k * MT), 64-row tiles first (32 KiB of LDS: four per CU, and the best fit next to another bag's fused
  // block); bags with more tiles than any shape can keep resident take 64-row tiles four per CU (two even rounds).
  // The rounds x blocks-per-CU estimate below picked 32-row tiles of the plain kernel at 12 k tokens (83 us; now 66)
  // and 128-row tiles at 13-15 k (95; now 76).
  if (prec == PREC_F32 && N == 512 && K == 512 && M >= 2048) {
    static const Cfg single[] = {{4, 1, 1024}, {8, 1, 768}, {6, 1, 1024}, {9, 1, 768}};
    Cfg best{4, 1, 1024};
    float best_cost = -1.f;
    for (const Cfg& c : single) {
      const long tiles = (long)((M + 16 * c.mt - 1) / (16 * c.mt)) * 8;
      if (tiles > c.cap) continue;
      const float k = (float)((tiles + 255) / 256);
      const float floor_us = 19.f + 1.2f * c.mt, busy_us = k * c.mt * 2.6f;
      const float cost = floor_us > busy_us ? floor_us : busy_us;
      if (best_cost < 0.f || cost < best_cost - 0.05f) { best = c; best_cost = cost; }
    }
    return best;
  }
  if (const char* e = rrt_tune_env("RRT_LINEAR_CFG")) {   // tuning hook: "mt,nt,cap"
    Cfg c{};
    if (sscanf(e, "%d,%d,%d", &c.mt, &c.nt, &c.cap) == 3) return c;
  }
  // operands rounded to 16 bits after LDS (fp32 bytes through the DMA path, 1/16 of the MFMA time): the K loop is bound
  // by DMA issue, and three 96-row blocks per CU keep more of it in flight than two 144-row ones (patch_to_emb at
  // N = 9000 x 1024 -> 512: 36.3 -> ~31 us)
  // (round 3, tools/sweep_linear_cfg.py: where one 144 x 128 block per CU covers the product -- patch_to_emb of a 9 k bag --
  //  that shape streams B half as often: 31.8 us against 35.1)
  if (prec != PREC_F32 && N % 128 == 0 && K >= 1024 && (long)((M + 143) / 144) * (N / 128) <= 256 && M >= 4096) return Cfg{9, 2, 256};
  if (prec != PREC_F32 && (M + 95) / 96 * ((N + 63) / 64) >= 512) return Cfg{6, 1, 768};
  static const Cfg cands[] = {{9, 1, 512}, {8, 1, 512}, {9, 2, 512}, {8, 2, 512}, {9, 2, 256},
                              {8, 2, 256}, {4, 1, 512}, {2, 1, 512}};
  Cfg best = cands[0];
  long best_cost = -1;
  for (const Cfg& c : cands) {
    long tiles = (long)((M + 16 * c.mt - 1) / (16 * c.mt)) * ((N + 64 * c.nt - 1) / (64 * c.nt));
    long grid = tiles < c.cap ? tiles : c.cap;
    long cost = ((tiles + grid - 1) / grid) * ((grid + 255) / 256) * c.mt * c.nt;
    if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
  }
  return best;
}

}  // namespace

#ifdef RRT_TRACE
RRT_TRACE_DEFINE_READER(rrt_debug_trace_linear)
#endif

// C[M,N] fp32 = A . B^T on (hi, lo) bf16 images of A and B (RRT_COMPUTE_F32X3, cast16.hip); epilogues as fp32
hipError_t launch_linear_split(const void* A, const void* B, float* C, int M, int N, int K, const LinearEpilogue& ep,
                               hipStream_t st) {
  if (K % 32 || ep.drop_on || ep.act) return hipErrorInvalidValue;
  const bool u = ep.resid != nullptr;
  Cfg c{9, 1, 512};                                // the warp-specialised kernel's tile shapes only
  {
    static const Cfg cands[] = {{9, 1, 512}, {8, 1, 512}, {9, 2, 512}, {8, 2, 512}, {9, 2, 256}, {8, 2, 256}};
    long best_cost = -1;
    for (const Cfg& q : cands) {
      long tiles = (long)((M + 16 * q.mt - 1) / (16 * q.mt)) * ((N + 64 * q.nt - 1) / (64 * q.nt));
      long grid = tiles < q.cap ? tiles : q.cap;
      long cost = ((tiles + grid - 1) / grid) * ((grid + 255) / 256) * q.mt * q.nt;
      if (best_cost < 0 || cost < best_cost) { c = q; best_cost = cost; }
    }
  }
#define RRT_SPLIT_CASE(MT_, NT_)                                                                                  \
  if (c.mt == MT_ && c.nt == NT_)                                                                                 \
    return u ? launch_cfg<MT_, NT_, MODE_UNPART, PREC_SPLIT>((const float*)A, (const float*)B, C, M, N, K, c.cap, ep, st) \
             : launch_cfg<MT_, NT_, MODE_PLAIN, PREC_SPLIT>((const float*)A, (const float*)B, C, M, N, K, c.cap, ep, st);
  RRT_SPLIT_CASE(9, 1);
  RRT_SPLIT_CASE(8, 1);
  RRT_SPLIT_CASE(9, 2);
  RRT_SPLIT_CASE(8, 2);
#undef RRT_SPLIT_CASE
  return hipErrorInvalidValue;
}

// tile shape of the 16-bit-operand GEMMs (launch_linear16)
static Cfg choose16(int M, int N, int K, bool solo) {
  Cfg best{9, 1, 512};
  {
    // (three 96-row blocks per CU first: at equal cost they keep more of the DMA-issue-bound loop in flight than two
    //  144-row ones -- 16.2 vs 17.3 us at N = 9000)
    // (the 64- and 32-row tiles: the 192-row out-projection of CR-MSA's representatives -- 24 / 48 blocks instead of 16)
    static const Cfg cands[] = {{6, 1, 768}, {9, 1, 512}, {8, 1, 512}, {9, 2, 512}, {8, 2, 512}, {9, 2, 256}, {8, 2, 256},
                                {4, 1, 512}, {2, 1, 512}};
    long best_cost = -1;
    for (const Cfg& c : cands) {
      if (c.mt < 6 && M > 1024) continue;           // small tiles re-stream B too often to pay on a bag-sized GEMM
      long tiles = (long)((M + 16 * c.mt - 1) / (16 * c.mt)) * ((N + 64 * c.nt - 1) / (64 * c.nt));
      long grid = tiles < c.cap ? tiles : c.cap;
      long cost = ((tiles + grid - 1) / grid) * ((grid + 255) / 256) * c.mt * c.nt;
      if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
    }
  }
  // Round 3: the bag-sized out-projection (N = K = 512) by rule, from the same sweep as the fp32 one
  // (SWEEP_BF16=1 tools/sweep_proj_cfg.py): the smallest tile whose blocks are all resident at once -- 64-row tiles four
  // per CU up to 8 k rows, 96-row up to 12 k, 144-row up to 13.8 k, then 128 / 144 x 128 tiles two per CU up to 18 k --
  // and 128-row tiles in rounds beyond.  (The rounds x blocks estimate above cost 10.5-12 k-token bags 28 us instead of
  // 21-23 and kept the 64-row tiles, the best shape below 8 k tokens, out of bag-sized products altogether.)
  if (N == 512 && K == 512 && M >= 2048) {
    auto tiles = [&](int mt, int nt) { return (long)((M + 16 * mt - 1) / (16 * mt)) * (8 / nt); };
    if (tiles(4, 1) <= 1024) best = Cfg{4, 1, 1024};
    else if (tiles(6, 1) <= 1024) best = Cfg{6, 1, tiles(6, 1) <= 768 ? 768 : 1024};
    else if (tiles(9, 1) <= 768) best = Cfg{9, 1, 768};
    else if (tiles(8, 2) <= 512) best = Cfg{8, 2, 512};
    else if (tiles(9, 2) <= 512) best = Cfg{9, 2, 512};
    else best = Cfg{8, 1, 512};
    // Round 6: the rule above was swept with ONE bag in flight.  With several (ep.solo == false: the executor with more than
    // one stream) a bag costs the sum of its chip-wide kernels, and this one is bound by the bytes it
    // moves through the CUs' LDS-DMA path: 128-column tiles read the A panel four times instead of eight.  Four bags in
    // flight, bf16 (tools/experiments/r06_probe13.sh, profiles/r06_proj16_tiles_in_flight.txt), slides/s against the rule
    // above: N = 7000 +2.5 %, 8000 +2.7 %, 9000 +2.5-4 % (96 x 128 tiles, all resident), 10500 +2.5 %; N = 12000 +1.6 %,
    // 15000 +2.4 %, 30000 +1-2 % (128 x 128 tiles, one per CU); below ~6.5 k rows the small tiles stay.  One bag in flight
    // the same shapes LOSE 4-8 % (0.0755 -> 0.080-0.083 ms at N = 9000): hence the hint.  The bits do not depend on the
    // tile shape (a tile's K-sum order is the same in every shape).
    if (!solo) {
      if (M > 11500) best = Cfg{8, 2, 256};
      else if (M > 6600) best = Cfg{6, 2, 512};
    }
  }
  if (const char* e = K > 512 ? rrt_tune_env("RRT_LINEAR16_CFG_KBIG") : nullptr) {   // tuning hook, K > 512 only: "mt,nt,cap"
    Cfg q{};
    if (sscanf(e, "%d,%d,%d", &q.mt, &q.nt, &q.cap) == 3) best = q;
  }
  if (const char* e = rrt_tune_env("RRT_LINEAR16_CFG")) {   // tuning hook: "mt,nt,cap"
    Cfg q{};
    if (sscanf(e, "%d,%d,%d", &q.mt, &q.nt, &q.cap) == 3) best = q;
  }
  return best;
}

// C[M,N] fp32 = A16[M,K] . B16[N,K]^T with the operands already in 16 bits (ep.prec = 1 bf16 / 2 fp16); the
// epilogues are the fp32 kernel's.  K % 64 == 0; the warp-specialised large-tile kernel only (the callers are the
// R-MSA projections of the reduced-precision path: M = Np >= 3136 rows).
hipError_t launch_linear16(const void* A, const void* B, float* C, int M, int N, int K, const LinearEpilogue& ep,
                           hipStream_t st) {
  if ((ep.prec != PREC_BF16 && ep.prec != PREC_F16) || K % 64 || ep.drop_on) return hipErrorInvalidValue;
  const bool u = ep.resid != nullptr;
  const Cfg c = choose16(M, N, K, ep.solo);
#define RRT_MODES16(MT_, NT_, P_)                                                                  \
  (u ? launch_cfg16<MT_, NT_, MODE_UNPART, P_>(A, B, C, M, N, K, c.cap, ep, st)                    \
     : ep.act ? launch_cfg16<MT_, NT_, MODE_ACT, P_>(A, B, C, M, N, K, c.cap, ep, st)              \
              : launch_cfg16<MT_, NT_, MODE_PLAIN, P_>(A, B, C, M, N, K, c.cap, ep, st))
#define RRT_CASE16(MT_, NT_)                                                                       \
  if (c.mt == MT_ && c.nt == NT_)                                                                  \
    return ep.prec == PREC_BF16 ? RRT_MODES16(MT_, NT_, PREC_BF16) : RRT_MODES16(MT_, NT_, PREC_F16);
  RRT_CASE16(9, 1);
  RRT_CASE16(8, 1);
  RRT_CASE16(9, 2);
  RRT_CASE16(8, 2);
  RRT_CASE16(6, 1);
  RRT_CASE16(4, 1);
  RRT_CASE16(2, 1);
  RRT_CASE16(6, 2);      // (round 6: several bags in flight, 6.6 k .. 11.5 k rows)
#ifdef RRT_TUNING        // wide tiles for the round-4 sweep (fewer bytes through the CU's memory pipe per output)
  RRT_CASE16(5, 4);
  RRT_CASE16(4, 4);
  RRT_CASE16(6, 4);
  RRT_CASE16(8, 4);
  RRT_CASE16(3, 4);
  RRT_CASE16(5, 2);
  RRT_CASE16(4, 2);
#endif
#undef RRT_CASE16
#undef RRT_MODES16
  return hipErrorInvalidValue;
}

hipError_t launch_linear(const float* A, const float* B, float* C, int M, int N, int K,
                         const LinearEpilogue& ep, hipStream_t st) {
  const bool u = ep.resid != nullptr;
  if (ep.drop_on && (ep.prec != PREC_F32 || ep.act)) return hipErrorInvalidValue;   // dropout: fp32 training only
  // the GEMMs of CR-MSA's representatives (M = 64 k <= 512 rows), when the forward has the GPU to itself: K split inside
  // the block (linear_splitk_kernel; its 16-wave, 80-96 KiB blocks do not fit next to another bag's fused R-MSA block)
  static const bool no_splitk = rrt_tune_env("RRT_NO_SPLITK") != nullptr;
  if (!no_splitk && ep.solo && ep.prec == PREC_F32 && !u && !ep.act && !ep.zero64 && M <= 512 && K % (4 * BK) == 0 && K >= 256) {
    constexpr int KG = 4;
    const int tiles_n = (N + 63) / 64;
    if (N >= 1024 && !ep.drop_on) {                 // qkv: 32-row tiles (6 x 24 = 144 blocks at k = 3)
      constexpr int MT = 2;
      constexpr int LDS_BYTES = KG * 2 * (16 * MT + 64) * BK * 4;
      auto kern = linear_splitk_kernel<MT, KG>;
      RRT_ALLOW_LDS(kern, LDS_BYTES);
      kern<<<dim3(((M + 16 * MT - 1) / (16 * MT)) * tiles_n), dim3(256 * KG), LDS_BYTES, st>>>(A, B, C, M, N, K, tiles_n, ep);
    } else {                                        // proj: 16-row tiles (12 x 8 = 96 blocks)
      constexpr int MT = 1;
      constexpr int LDS_BYTES = KG * 2 * (16 * MT + 64) * BK * 4;
      if (ep.drop_on) {
        auto kern = linear_splitk_kernel<MT, KG, true>;
        RRT_ALLOW_LDS(kern, LDS_BYTES);
        kern<<<dim3(((M + 16 * MT - 1) / (16 * MT)) * tiles_n), dim3(256 * KG), LDS_BYTES, st>>>(A, B, C, M, N, K, tiles_n, ep);
      } else {
        auto kern = linear_splitk_kernel<MT, KG>;
        RRT_ALLOW_LDS(kern, LDS_BYTES);
        kern<<<dim3(((M + 16 * MT - 1) / (16 * MT)) * tiles_n), dim3(256 * KG), LDS_BYTES, st>>>(A, B, C, M, N, K, tiles_n, ep);
      }
    }
    return hipGetLastError();
  }
  const Cfg c = choose(M, N, ep.prec, K);
#define RRT_CASE(MT_, NT_)                                                                          \
  if (c.mt == MT_ && c.nt == NT_) {                                                                 \
    if (ep.prec == PREC_BF16) return RRT_MODES(MT_, NT_, PREC_BF16);                                \
    if (ep.prec == PREC_F16) return RRT_MODES(MT_, NT_, PREC_F16);                                  \
    return RRT_MODES(MT_, NT_, PREC_F32);                                                           \
  }
#define RRT_MODES(MT_, NT_, P_)                                                                     \
  (ep.drop_on ? (u ? launch_cfg<MT_, NT_, MODE_UNPART_DROP, PREC_F32>(A, B, C, M, N, K, c.cap, ep, st)  \
                       : launch_cfg<MT_, NT_, MODE_DROP, PREC_F32>(A, B, C, M, N, K, c.cap, ep, st))        \
   : u ? launch_cfg<MT_, NT_, MODE_UNPART, P_>(A, B, C, M, N, K, c.cap, ep, st)                      \
     : ep.act ? launch_cfg<MT_, NT_, MODE_ACT, P_>(A, B, C, M, N, K, c.cap, ep, st)                 \
              : launch_cfg<MT_, NT_, MODE_PLAIN, P_>(A, B, C, M, N, K, c.cap, ep, st))
  RRT_CASE(9, 1);
  RRT_CASE(8, 1);
  RRT_CASE(9, 2);
  RRT_CASE(8, 2);
  RRT_CASE(6, 1);
  RRT_CASE(4, 1);
  RRT_CASE(2, 1);
#ifdef RRT_TUNING        // round 4: 16-row tiles for the representatives' GEMMs (more, lighter blocks), RRT_LINEAR_CFG=1,1,1024
  RRT_CASE(1, 1);
#endif
#undef RRT_CASE
#undef RRT_MODES
  return hipErrorInvalidValue;
}
