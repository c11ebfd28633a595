// linear_bwd.hip -- gradients of nn.Linear  Y[M,N] = X[M,K] . W[N,K]^T + b  (row f2 building block):
//     dX[M,K] = dY . W            dW[N,K] = dY^T . X            db[N] = column sums of dY
// dX reuses the forward GEMM (linear_f32.hip computes A.B^T with both operands reduction-contiguous) on a
// transposed copy of W (<= 3 MB).  dW reduces over the ROWS of both operands, so it gets its own kernel:
// a "TN" product on the fp32 matrix cores with split-K over row chunks (a 512 x 512 x 9216 product has only
// 16 output tiles; the chunks make it >= 2 blocks per CU), partial products summed by a small reduce kernel
// in a fixed order (bit-reproducible, no atomics).
#include <stdlib.h>

#include "internal.h"

namespace {

// ---- out[C,R] = in[R,C]^T, 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int R, int C) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < R) out[(size_t)c * R + r] = tile[tx][ty + 8 * i];
  }
}

// ---- several transposes in one launch (the weight matrices of a backward pass: their W^T images do not depend on the
//      gradients, so they are all made up front instead of one 5 us launch in front of every dX product)
__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeJobs jobs) {
  __shared__ float tile[32][33];
  int j = 0;
#pragma unroll
  for (int q = 1; q < TRANSPOSE_MAX_JOBS; ++q)
    if (q < jobs.n && (int)blockIdx.x >= jobs.blk0[q]) j = q;
  const float* __restrict__ in = jobs.in[j];
  float* __restrict__ out = jobs.out[j];
  const int R = jobs.R[j], C = jobs.C[j];
  const int b = blockIdx.x - jobs.blk0[j], nbx = (C + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = (b % nbx) * 32, r0 = (b / nbx) * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < R) out[(size_t)c * R + r] = tile[tx][ty + 8 * i];
  }
}

// ---- split-K TN product: part[s][n][k] = sum_{m in chunk s} A[m][n] * B[m][k]
// block = 4 waves (2 x 2), tile 128 (n) x 128 (k), 32 rows of both operands per step.  Rows are staged in LDS
// as they lie in memory (coalesced 16-byte loads); the MFMA fragments are read "down the columns":
// lane (lr, lg) takes A_s[m = 4 kk + lg][n = .. + lr] -- 16 consecutive floats per lane group, and the row
// pitch of 144 floats puts the four groups on disjoint bank ranges.
constexpr int TN_BN = 128, TN_BK = 128, TN_BM = 32, TN_PITCH = 144;

// dbpart (may be null): the blocks of the first k column also sum the columns of their A rows -- db[n] = sum_m dY[m][n]
// of nn.Linear's backward -- into dbpart[s][n] (the rows are in LDS anyway: one launch and one pass over dY fewer)
// FAST: N and K are multiples of the 128-wide tile (every nn.Linear of the encoder at dim = 128 n): plain 16-byte loads with
// the row clamped and zeroed at the LDS write -- no predicated fallback loads, whose merges made the compiler wait for
// the prefetch (s_waitcnt vmcnt(0)) in front of the MFMAs it was meant to run under.
template <bool FAST>
__global__ __launch_bounds__(256, 3) void gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ part, float* __restrict__ dbpart, int M, int N, int K,
                                                      int rows_per_chunk, int xcd_tiles_n) {
  __shared__ __attribute__((aligned(16))) float As[TN_BM * TN_PITCH];
  __shared__ __attribute__((aligned(16))) float Bs[TN_BM * TN_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  // Which (n tile, k tile, row chunk) this block owns.  xcd_tiles_n = 0: the plain 3-D grid.  Otherwise (1-D grid, the number of
  // row chunks a multiple of 8): consecutive block ids go to the 8 XCDs in turn, so chunk s = (id % 8) + 8 * ... puts ALL tiles
  // of a row chunk on one XCD -- its L2 then fetches each row of dY and X once for the tiles_n * tiles_k blocks that read it,
  // where the plain order spread a chunk's tiles over every L2 (a 128 x 128 tile reads 32 flop/B: at the fp32 matrix rate that
  // is 4.9 TB/s of operand reads, which only the L2s can serve).
  int bx = blockIdx.x, by = blockIdx.y, s = blockIdx.z;
  if (xcd_tiles_n > 0) {
    const int tiles_k = (K + TN_BK - 1) / TN_BK, T = xcd_tiles_n * tiles_k;
    const int id = blockIdx.x, j = id >> 3, t = j % T;
    s = (id & 7) + 8 * (j / T);
    bx = t % xcd_tiles_n;
    by = t / xcd_tiles_n;
  }
  const int n0 = bx * TN_BN, k0 = by * TN_BK;
  const int m_begin = s * rows_per_chunk;
  const int m_end = min(M, m_begin + rows_per_chunk);
  const bool vecA = (N & 3) == 0, vecB = (K & 3) == 0;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_db = dbpart != nullptr && by == 0;
  float colsum = 0.f;                               // column n0 + tid of this block's rows (tid < 128)

  // Stage = 32 rows x 128 columns of each operand: 1024 float4 per operand, 4 per thread.  The rows of stage s + 1 are
  // requested into registers BEFORE the 128 MFMAs of stage s and written to LDS behind them: a block covers its own
  // global-memory latency (the first version loaded, waited, computed: 0.55 of the fp32 MFMA rate on the qkv weight
  // gradient, the matrix pipe waiting for whichever co-resident block had rows).
  float4 ra[4], rb[4];
  auto fetch = [&](const int m0) {
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 5, c4 = (idx & 31) * 4;
        const int m = min(m0 + row, m_end - 1);     // rows past the chunk: re-read, zeroed in publish()
        ra[i] = *(const float4*)(A + (size_t)m * N + n0 + c4);
        rb[i] = *(const float4*)(B + (size_t)m * K + k0 + c4);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 5, c4 = (idx & 31) * 4;
      const int m = m0 + row;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (m < m_end) {
        const int n = n0 + c4, k = k0 + c4;
        if (vecA && n + 3 < N) va = *(const float4*)(A + (size_t)m * N + n);
        else {
          if (n < N) va.x = A[(size_t)m * N + n];
          if (n + 1 < N) va.y = A[(size_t)m * N + n + 1];
          if (n + 2 < N) va.z = A[(size_t)m * N + n + 2];
          if (n + 3 < N) va.w = A[(size_t)m * N + n + 3];
        }
        if (vecB && k + 3 < K) vb = *(const float4*)(B + (size_t)m * K + k);
        else {
          if (k < K) vb.x = B[(size_t)m * K + k];
          if (k + 1 < K) vb.y = B[(size_t)m * K + k + 1];
          if (k + 2 < K) vb.z = B[(size_t)m * K + k + 2];
          if (k + 3 < K) vb.w = B[(size_t)m * K + k + 3];
        }
      }
      ra[i] = va;
      rb[i] = vb;
    }
  };
  auto publish = [&](const int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 5, c4 = (idx & 31) * 4;
      if (FAST && m0 + row >= m_end) ra[i] = rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      *(float4*)(As + row * TN_PITCH + c4) = ra[i];
      *(float4*)(Bs + row * TN_PITCH + c4) = rb[i];
    }
  };
  if (m_begin < m_end) {
    fetch(m_begin);
    publish(m_begin);
  }
  __syncthreads();
  for (int m0 = m_begin; m0 < m_end; m0 += TN_BM) {
    const bool more = m0 + TN_BM < m_end;
    if (more) fetch(m0 + TN_BM);
    if (do_db && tid < TN_BN) {
      float c0 = 0.f, c1 = 0.f;
#pragma unroll
      for (int r = 0; r < TN_BM; r += 2) { c0 += As[r * TN_PITCH + tid]; c1 += As[(r + 1) * TN_PITCH + tid]; }
      colsum += c0 + c1;
    }
    // fragments of k step kk + 1 are read under the 16 MFMAs of step kk (two register sets)
    float a[2][4], b[2][4];
    auto frags = [&](const int kk, float (&fa)[4], float (&fb)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = As[(4 * kk + lg) * TN_PITCH + wn * 64 + i * 16 + lr];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = Bs[(4 * kk + lg) * TN_PITCH + wk * 64 + j * 16 + lr];
    };
    frags(0, a[0], b[0]);
#pragma unroll
    for (int kk = 0; kk < TN_BM / 4; ++kk) {
      if (kk + 1 < TN_BM / 4) frags(kk + 1, a[(kk + 1) & 1], b[(kk + 1) & 1]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                // every wave is done reading this stage
    if (more) publish(m0 + TN_BM);
    __syncthreads();
  }
  if (do_db && tid < TN_BN && n0 + tid < N) dbpart[(size_t)s * N + n0 + tid] = colsum;
  // D[i = 4 lg + r][j = lr] of every 16 x 16 tile
  float* out = part + (size_t)s * N * K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * lg + r, k = k0 + wk * 64 + j * 16 + lr;
        if (n < N && k < K) out[(size_t)n * K + k] = acc[i][j][r];
      }
}

// ---- out[i] = sum_s part[s][i], fixed order (bit-reproducible).  A block owns 256 consecutive elements; its four
//      waves split the S partials (wave w takes s = w, w + 4, ...) with up to 12 independent 16-byte loads in flight
//      each, then combine through LDS in wave order.  (The plain per-thread loop was a chain of S memory round
//      trips: 131 us for 16 x 3 MB; 8 loads in flight: 37 us; this: one or two trips.)
// Optional second segment (part2 / out2 / n2; blocks past the first segment's): the bias gradient's partials ride in the
// same launch as the weight gradient's.
// NWV waves per block: 4, or 16 when there are many partials of a short vector (LayerNorm's 512 x [2, dim], the taps'
// [regions][heads, k]: a handful of blocks, each wave walking S / 4 partials twelve at a time, was a chain of ~11
// memory round trips = 8 us for 2 MB).
template <int NWV>
__global__ __launch_bounds__(NWV * 64) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int S, size_t n, const float* __restrict__ part2,
                                                              float* __restrict__ out2, size_t n2, unsigned nb1,
                                                              float* __restrict__ out_tr = nullptr, size_t split = 0,
                                                              int tr_dim = 0, int tr_k = 0) {
  __shared__ float4 comb[NWV - 1][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned blk = blockIdx.x;
  if (blk >= nb1) { blk -= nb1; part = part2; out = out2; n = n2; }
  const size_t i = ((size_t)blk * 64 + lane) * 4;
  const bool vec = (n & 3) == 0 && i + 3 < n;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
    if (vec) {
      for (int s0 = wave; s0 < S; s0 += 12 * NWV) {
        float4 b[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
          const int s = s0 + NWV * u;
          b[u] = s < S ? *(const float4*)(part + (size_t)s * n + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
      }
    } else {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      for (int s = wave; s < S; s += NWV)
        for (int e = 0; e < 4; ++e)
          if (i + e < n) t[e] += part[(size_t)s * n + i + e];
      a = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
  if (wave > 0) comb[wave - 1][lane] = a;
  __syncthreads();
  if (wave == 0 && i < n) {
#pragma unroll
    for (int w = 0; w < NWV - 1; ++w) {
      const float4 o = comb[w][lane];
      a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    }
    if (out_tr && i >= split) {
      // the tail [tr_k, tr_dim] of the reduced vector leaves transposed, as [tr_dim, tr_k] (CR-MSA's d phi: the copy and the
      // transpose launches that used to follow were 10 us of the backward's dependent chain)
      const float t[4] = {a.x, a.y, a.z, a.w};
      for (int e = 0; e < 4; ++e)
        if (i + e < n) {
          const size_t q = i + e - split;
          out_tr[(q % tr_dim) * tr_k + q / tr_dim] = t[e];
        }
    } else if (vec) *(float4*)(out + i) = a;
    else {
      const float t[4] = {a.x, a.y, a.z, a.w};
      for (int e = 0; e < 4; ++e)
        if (i + e < n) out[i + e] = t[e];
    }
  }
}

// ---- several such reductions in one launch (ReduceJobs, internal.h).  The jobs are short vectors with hundreds of partials
//      (LayerNorm: 512 x [2, dim]; CR-MSA: 512 x [2 + k, dim]): 256 elements per block is a handful of blocks that each pull
//      0.5 MB through one CU (15 us for the default encoder's four jobs).  Here a block owns 32 consecutive elements = one
//      128-byte line of every partial: a wave reads eight partials per load (lane = 8 * partial + float4), sixteen waves 128,
//      up to twelve loads in flight; the 128 lane-sums of an element meet in LDS and are added in a fixed order.
constexpr int RJ_ELEMS = 32;
__global__ __launch_bounds__(1024) void reduce_jobs_kernel(ReduceJobs jobs) {
  __shared__ float4 comb[128][8];          // [16 waves x 8 partial slots][float4 of the line]
  __shared__ float4 comb2[16][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int j = 0;
#pragma unroll
  for (int q = 1; q < REDUCE_MAX_JOBS; ++q)
    if (q < jobs.count && blockIdx.x >= jobs.blk0[q]) j = q;
  const float* __restrict__ part = jobs.part[j];
  const size_t n = jobs.n[j];
  const int S = jobs.S[j];
  const int sub = lane >> 3, c = lane & 7;
  const size_t i = ((size_t)(blockIdx.x - jobs.blk0[j]) * 8 + c) * 4;      // this lane's first element
  const bool vec = (n & 3) == 0 && i + 3 < n;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
    for (int s0 = wave * 8 + sub; s0 < S; s0 += 12 * 128) {
      float4 b[12];
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int s = s0 + 128 * u;
        b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < S) {
          const float* src = part + (size_t)s * n + i;
          if (vec) b[u] = *(const float4*)src;
          else {
            b[u].x = src[0];
            if (i + 1 < n) b[u].y = src[1];
            if (i + 2 < n) b[u].z = src[2];
            if (i + 3 < n) b[u].w = src[3];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 12; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
    }
  }
  comb[wave * 8 + sub][c] = a;
  __syncthreads();
  if (threadIdx.x < 128) {                 // thread (w, c): the eight partial slots of wave w
    const int w = threadIdx.x >> 3, cc = threadIdx.x & 7;
    float4 t = comb[w * 8][cc];
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      const float4 o = comb[w * 8 + q][cc];
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    comb2[w][cc] = t;
  }
  __syncthreads();
  if (threadIdx.x < 8 && i < n) {          // (threads 0..7: sub = 0, c = threadIdx.x, so `i` is this thread's element)
    float4 t = comb2[0][c];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
      const float4 o = comb2[w][c];
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    float* __restrict__ out = jobs.out[j];
    float* __restrict__ out_tr = jobs.out_tr[j];
    const size_t split = jobs.split[j];
    const float v[4] = {t.x, t.y, t.z, t.w};
    if (out_tr && i >= split) {
      const int tr_dim = jobs.tr_dim[j], tr_k = jobs.tr_k[j];
      for (int e = 0; e < 4; ++e)
        if (i + e < n) {
          const size_t q = i + e - split;
          out_tr[(q % tr_dim) * tr_k + q / tr_dim] = v[e];
        }
    } else if (vec && (((size_t)out) & 15) == 0) *(float4*)(out + i) = t;
    else {
      for (int e = 0; e < 4; ++e)
        if (i + e < n) out[i + e] = v[e];
    }
  }
}

// ---- column sums of Y[M,N] over row chunks: part[c][n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ Y, float* __restrict__ part, int M,
                                                     int N, int rows_per_chunk) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int m = m0;
  for (; m + 3 < m1; m += 4) {
    a0 += Y[(size_t)m * N + n];
    a1 += Y[(size_t)(m + 1) * N + n];
    a2 += Y[(size_t)(m + 2) * N + n];
    a3 += Y[(size_t)(m + 3) * N + n];
  }
  for (; m < m1; ++m) a0 += Y[(size_t)m * N + n];
  part[(size_t)blockIdx.y * N + n] = (a0 + a1) + (a2 + a3);
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

int linear_bwd_chunks(int M, int N, int K, int* rows_per_chunk) {
  const long tiles = (long)((N + TN_BN - 1) / TN_BN) * ((K + TN_BK - 1) / TN_BK);
  static const int target = rrt_tune_env("RRT_TN_BLOCKS") ? atoi(rrt_tune_env("RRT_TN_BLOCKS")) : 768;   // ~3 blocks per CU
  int S = (int)((target + tiles - 1) / tiles);
  const int max_s = (M + 4 * TN_BM - 1) / (4 * TN_BM);      // at least 128 rows per chunk
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  int rpc = (M + S - 1) / S;
  rpc = (rpc + TN_BM - 1) / TN_BM * TN_BM;
  S = (M + rpc - 1) / rpc;
  *rows_per_chunk = rpc;
  return S;
}

size_t linear_bwd_workspace(int M, int N, int K) {
  int rpc;
  const int S = linear_bwd_chunks(M, N, K, &rpc);
  const int SB = (M + 127) / 128;
  return align256((size_t)N * K * 4) + align256((size_t)S * N * K * 4) + align256((size_t)SB * N * 4);
}

hipError_t launch_transpose(const float* in, float* out, int R, int C, hipStream_t st) {
  transpose_kernel<<<dim3((C + 31) / 32, (R + 31) / 32), 256, 0, st>>>(in, out, R, C);
  return hipGetLastError();
}

hipError_t launch_transpose_batch(TransposeJobs& jobs, hipStream_t st) {
  if (jobs.n == 0) return hipSuccess;
  int nb = 0;
  for (int j = 0; j < jobs.n; ++j) {
    jobs.blk0[j] = nb;
    nb += ((jobs.C[j] + 31) / 32) * ((jobs.R[j] + 31) / 32);
  }
  transpose_batch_kernel<<<dim3(nb), 256, 0, st>>>(jobs);
  return hipGetLastError();
}

hipError_t launch_reduce_partials(const float* part, float* out, int S, size_t n, hipStream_t st) {
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (S >= 64 && nb <= 64) reduce_partials_kernel<16><<<dim3(nb), 1024, 0, st>>>(part, out, S, n, nullptr, nullptr, 0, nb);
  else reduce_partials_kernel<4><<<dim3(nb), 256, 0, st>>>(part, out, S, n, nullptr, nullptr, 0, nb);
  return hipGetLastError();
}

// out[0, split) = the head of the reduced vector, out_tr[tr_dim, tr_k] = its tail [tr_k, tr_dim] transposed (split % 4 == 0)
hipError_t launch_reduce_partials_scatter(const float* part, float* out, int S, size_t n, size_t split, float* out_tr,
                                          int tr_dim, int tr_k, hipStream_t st) {
  if (split % 4 != 0 || split > n || (n - split) != (size_t)tr_dim * tr_k) return hipErrorInvalidValue;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (S >= 64 && nb <= 64)
    reduce_partials_kernel<16><<<dim3(nb), 1024, 0, st>>>(part, out, S, n, nullptr, nullptr, 0, nb, out_tr, split, tr_dim, tr_k);
  else reduce_partials_kernel<4><<<dim3(nb), 256, 0, st>>>(part, out, S, n, nullptr, nullptr, 0, nb, out_tr, split, tr_dim, tr_k);
  return hipGetLastError();
}

hipError_t reduce_or_defer(ReduceJobs* defer, const float* part, float* out, int S, size_t n, hipStream_t st, size_t split,
                           float* out_tr, int tr_dim, int tr_k) {
  if (out_tr && (split % 4 != 0 || split > n || (n - split) != (size_t)tr_dim * tr_k)) return hipErrorInvalidValue;
  if (defer && defer->count < REDUCE_MAX_JOBS) {
    const int j = defer->count++;
    defer->part[j] = part;
    defer->out[j] = out;
    defer->out_tr[j] = out_tr;
    defer->n[j] = n;
    defer->split[j] = split;
    defer->S[j] = S;
    defer->tr_dim[j] = tr_dim;
    defer->tr_k[j] = tr_k;
    return hipSuccess;
  }
  if (out_tr) return launch_reduce_partials_scatter(part, out, S, n, split, out_tr, tr_dim, tr_k, st);
  return launch_reduce_partials(part, out, S, n, st);
}

hipError_t launch_reduce_jobs(ReduceJobs& jobs, hipStream_t st) {
  if (jobs.count == 0) return hipSuccess;
  unsigned nb = 0;
  for (int j = 0; j < jobs.count; ++j) {
    jobs.blk0[j] = nb;
    nb += (unsigned)((jobs.n[j] + RJ_ELEMS - 1) / RJ_ELEMS);
  }
  jobs.blk0[jobs.count] = nb;
  reduce_jobs_kernel<<<dim3(nb), 1024, 0, st>>>(jobs);
  jobs.count = 0;
  return hipGetLastError();
}

// grid of the TN product: 1-D with the XCD-aware order (kernel comment) when the chunks divide over the 8 XCDs, else 3-D
static dim3 tn_grid(int N, int K, int S, int* xcd_tiles_n) {
  const int tn = (N + TN_BN - 1) / TN_BN, tk = (K + TN_BK - 1) / TN_BK;
  static const bool plain = rrt_tune_env("RRT_TN_PLAIN_GRID") != nullptr;       // (A/B, tuning build only)
  if (S % 8 == 0 && !plain) {
    *xcd_tiles_n = tn;
    return dim3((unsigned)(tn * tk * S));
  }
  *xcd_tiles_n = 0;
  return dim3(tn, tk, S);
}

// dW[N,K] = dY[M,N]^T . X[M,K] ; scratch: S * N * K floats
hipError_t launch_gemm_tn(const float* dY, const float* X, float* dW, float* scratch, int M, int N, int K,
                          hipStream_t st) {
  int rpc;
  const int S = linear_bwd_chunks(M, N, K, &rpc);
  int xt = 0;
  const dim3 grid = tn_grid(N, K, S, &xt);
  if (N % TN_BN == 0 && K % TN_BK == 0) gemm_tn_kernel<true><<<grid, 256, 0, st>>>(dY, X, S == 1 ? dW : scratch, nullptr, M, N, K, rpc, xt);
  else gemm_tn_kernel<false><<<grid, 256, 0, st>>>(dY, X, S == 1 ? dW : scratch, nullptr, M, N, K, rpc, xt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || S == 1) return e;
  return launch_reduce_partials(scratch, dW, S, (size_t)N * K, st);
}

// ... and db[N] = column sums of dY in the same two launches (dbscratch: S * N floats)
static hipError_t launch_gemm_tn_db(const float* dY, const float* X, float* dW, float* db, float* scratch, float* dbscratch, int M,
                                    int N, int K, hipStream_t st) {
  int rpc;
  const int S = linear_bwd_chunks(M, N, K, &rpc);
  int xt = 0;
  const dim3 grid = tn_grid(N, K, S, &xt);
  if (N % TN_BN == 0 && K % TN_BK == 0)
    gemm_tn_kernel<true><<<grid, 256, 0, st>>>(dY, X, S == 1 ? dW : scratch, S == 1 ? db : dbscratch, M, N, K, rpc, xt);
  else gemm_tn_kernel<false><<<grid, 256, 0, st>>>(dY, X, S == 1 ? dW : scratch, S == 1 ? db : dbscratch, M, N, K, rpc, xt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || S == 1) return e;
  const unsigned nb1 = (unsigned)(((size_t)N * K + 255) / 256), nb2 = (unsigned)((N + 255) / 256);
  reduce_partials_kernel<4><<<dim3(nb1 + nb2), 256, 0, st>>>(scratch, dW, S, (size_t)N * K, dbscratch, db, (size_t)N, nb1);
  return hipGetLastError();
}

hipError_t launch_colsum(const float* Y, float* out, float* scratch, int M, int N, hipStream_t st) {
  const int SB = (M + 127) / 128;
  colsum_kernel<<<dim3((N + 255) / 256, SB), 256, 0, st>>>(Y, SB == 1 ? out : scratch, M, N, 128);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || SB == 1) return e;
  return launch_reduce_partials(scratch, out, SB, (size_t)N, st);
}

// Full nn.Linear backward.  dX may be null (first layer).  ws: linear_bwd_workspace(M, N, K) bytes.
hipError_t launch_linear_backward(const float* dY, const float* X, const float* W, float* dX, float* dW, float* db,
                                  int M, int N, int K, int prec, void* ws, hipStream_t st, const float* WT_ready) {
  char* base = (char*)ws;
  float* WT = (float*)base;
  float* scratch = (float*)(base + align256((size_t)N * K * 4));
  int rpc;
  const int S = linear_bwd_chunks(M, N, K, &rpc);
  float* cs = (float*)((char*)scratch + align256((size_t)S * N * K * 4));
  hipError_t e;
  if (dX) {
    if (WT_ready) WT = const_cast<float*>(WT_ready);          // made up front (launch_transpose_batch)
    else {
      e = launch_transpose(W, WT, N, K, st);                  // WT [K, N]
      if (e != hipSuccess) return e;
    }
    LinearEpilogue ep{};
    ep.prec = prec;
    ep.solo = true;                                           // (a training step has the GPU to itself: small-M GEMMs split K in the block)
    e = launch_linear(dY, WT, dX, M, K, N, ep, st);           // dX[M,K] = dY[M,N] . WT[K,N]^T
    if (e != hipSuccess) return e;
  }
  if (dW && db) {                                              // the usual case: bias gradient inside the dW product
    e = launch_gemm_tn_db(dY, X, dW, db, scratch, cs, M, N, K, st);
    if (e != hipSuccess) return e;
  } else if (dW) {
    e = launch_gemm_tn(dY, X, dW, scratch, M, N, K, st);
    if (e != hipSuccess) return e;
  } else if (db) {
    e = launch_colsum(dY, db, cs, M, N, st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
