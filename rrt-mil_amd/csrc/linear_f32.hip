// linear_f32.hip -- C[M,N] = A[M,K] . B[N,K]^T (+bias) on the fp32 matrix cores.
//
// Replaces nn.Linear in InnerAttention (qkv: modules/rmsa.py:100, proj: :131) plus,
// in the un-partition epilogue, region_reverse + un-pad + the TransLayer residual
// (modules/rmsa.py:41-54, :227-228; modules/rrt.py:125).
//
// MFMA-bound (exact fp32: v_mfma_f32_32x32x2_f32, 157 TFLOP/s chip peak).  Both
// operands are K-contiguous ("NT"), so A and B tiles are staged identically:
//   * global -> LDS by 16-byte DMA (global_load_lds_dwordx4), double-buffered, BK = 32;
//   * the LDS image is [row][8 x 16-B slots]; slot p of a row holds logical k-slot
//     p ^ ((row>>1)&7).  The XOR is applied to the per-lane *global source* address
//     (the DMA destination is lane-linear) and again on the ds_read_b128 side, which
//     makes every 16-lane read group hit 16 distinct 16-B bank slots (conflict-free);
//   * a lane's float4 (4 consecutive k) feeds 4 MFMAs: lanes 0-31 carry k-slot 2*kk,
//     lanes 32-63 k-slot 2*kk+1 -- the K-sum is order-free, so no transposes.
// Each of the 4 waves owns a (TM*32) x (TN*32) sub-tile; block tile = 2x2 waves.
// 1-D grid with an XCD-aware (bijective) remap so the blocks of one XCD walk the N
// tiles of the same A row-panel (A panel + all of W stay in that XCD's 4 MiB L2).
#include "internal.h"

namespace {

constexpr int BK = 32;

template <int TM, int TN>
struct Tile {
  static constexpr int BM = 64 * TM;   // 2 waves along M
  static constexpr int BN = 64 * TN;   // 2 waves along N
  static constexpr int LDS_BYTES = 2 * (BM + BN) * BK * 4;
};

// Stage one [ROWS x BK] tile: ROWS*8 16-B slots, 64 slots per wave-instruction.
template <int ROWS>
__device__ __forceinline__ void stage_tile(const float* __restrict__ src, int ld, int row0, int nrows,
                                           int k0, float* lds, int wave, int lane) {
  constexpr int NINSTR = ROWS * 8 / 64;   // wave-instructions for the tile
#pragma unroll
  for (int q = wave; q < NINSTR; q += 4) {
    int S = q * 64 + lane;
    int row = S >> 3, p = S & 7;
    int c = p ^ ((row >> 1) & 7);
    int gr = row0 + row;
    gr = gr < nrows ? gr : nrows - 1;                 // tail rows: re-read the last row (never stored)
    dma16(src + (size_t)gr * ld + k0 + c * 4, lds + q * 256);
  }
}

template <int TM, int TN, bool UNPART>
__global__ __launch_bounds__(256, 2) void linear_kernel(const float* __restrict__ A,
                                                        const float* __restrict__ B,
                                                        float* __restrict__ C, int M, int N, int K,
                                                        int tiles_n, int nblocks, LinearEpilogue ep) {
  using T = Tile<TM, TN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = (float*)smem;   // [2][A: BM*BK | B: BN*BK]
  constexpr int STAGE = (T::BM + T::BN) * BK;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware bijective remap: hardware places block b on XCD b%8; give each XCD a
  // contiguous run of logical tiles.
  int b = blockIdx.x;
  {
    int q = nblocks >> 3, r = nblocks & 7, xcd = b & 7, idx = b >> 3;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = b / tiles_n, tn = b - tm * tiles_n;
  const int m0 = tm * T::BM, n0 = tn * T::BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK;
  stage_tile<T::BM>(A, K, m0, M, 0, lds, wave, lane);
  stage_tile<T::BN>(B, K, n0, N, 0, lds + T::BM * BK, wave, lane);

  for (int kt = 0; kt < nk; ++kt) {
    wait_vm0();
    __syncthreads();   // tile kt landed for every wave; everyone is done reading the other buffer
    float* cur = lds + (kt & 1) * STAGE;
    if (kt + 1 < nk) {
      float* nxt = lds + ((kt + 1) & 1) * STAGE;
      stage_tile<T::BM>(A, K, m0, M, (kt + 1) * BK, nxt, wave, lane);
      stage_tile<T::BN>(B, K, n0, N, (kt + 1) * BK, nxt + T::BM * BK, wave, lane);
    }
    const float* As = cur;
    const float* Bs = cur + T::BM * BK;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 af[TM], bf[TN];
      const int cslot = 2 * kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int row = wr * (32 * TM) + i * 32 + (lane & 31);
        af[i] = *(const float4*)(As + row * BK + ((cslot ^ ((row >> 1) & 7)) << 2));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int row = wc * (32 * TN) + j * 32 + (lane & 31);
        bf[j] = *(const float4*)(Bs + row * BK + ((cslot ^ ((row >> 1) & 7)) << 2));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }

  // epilogue.  32x32 C layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wc * (32 * TN) + j * 32 + (lane & 31);
    const bool n_ok = n < N;
    const float bias = (ep.bias && n_ok) ? ep.bias[n] : 0.f;
    const float scale = (n < ep.q_cols) ? ep.q_scale : 1.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * (32 * TM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && n_ok) {
          float v = (acc[i][j][r] + bias) * scale;
          if (UNPART) {
            int t = slot_to_token(m, ep.g);
            if (t < ep.g.L) C[(size_t)t * N + n] = ep.resid[(size_t)t * N + n] + v;
          } else {
            C[(size_t)m * N + n] = v;
          }
        }
      }
    }
  }
}

template <int TM, int TN, bool UNPART>
hipError_t launch_cfg(const float* A, const float* B, float* C, int M, int N, int K,
                      const LinearEpilogue& ep, hipStream_t st) {
  using T = Tile<TM, TN>;
  int tiles_m = (M + T::BM - 1) / T::BM, tiles_n = (N + T::BN - 1) / T::BN;
  int nblocks = tiles_m * tiles_n;
  auto kern = linear_kernel<TM, TN, UNPART>;
  static bool attr_done = false;   // one-time opt-in for >64 KiB dynamic LDS is not needed (<= 64 KiB)
  (void)attr_done;
  kern<<<dim3(nblocks), dim3(256), T::LDS_BYTES, st>>>(A, B, C, M, N, K, tiles_n, nblocks, ep);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_linear(const float* A, const float* B, float* C, int M, int N, int K,
                         const LinearEpilogue& ep, hipStream_t st) {
  const bool unpart = ep.resid != nullptr;
  // small-M (CR-MSA representatives: M = 64*k): 64x64 tiles fill more CUs
  const bool small = (long)((M + 127) / 128) * ((N + 127) / 128) < 128;
  if (small) {
    return unpart ? launch_cfg<1, 1, true>(A, B, C, M, N, K, ep, st)
                  : launch_cfg<1, 1, false>(A, B, C, M, N, K, ep, st);
  }
  return unpart ? launch_cfg<2, 2, true>(A, B, C, M, N, K, ep, st)
                : launch_cfg<2, 2, false>(A, B, C, M, N, K, ep, st);
}
