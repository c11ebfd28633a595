// peg.hip -- the ablation positional encoders PEG / PPEG (modules/emb_position.py:24-82; RRTEncoder(pos='peg' | 'ppeg'),
// hooked in at modules/rrt.py:181-187).  Tokens are laid on an H x H grid (H = ceil(sqrt(N)); the tail is filled by
// WRAPPING the first H*H - N tokens; PPEG additionally zero-pads a grid smaller than 7 x 7 up to 7 x 7), every
// channel gets depth-wise 2-D convolutions with zero borders and the identity:
//     PEG : y = x^ + conv_k(x^)          PPEG: y = x^ + conv_k(x^) + conv_5(x^) + conv_3(x^)
// All of it is ONE stencil per channel: W_eff = w_k (+ w_5 + w_3 centred) + delta, b_eff = sum of the biases.
// Block = 8 x 8 output tokens x 64 channels; the (8 + 2h)^2 x 64 input patch is staged in LDS once (each token row
// is read ~3x instead of k*k times), the per-channel stencil lives in registers.  HBM/L2-bound.
#include "internal.h"

namespace {

template <int KK>   // effective kernel side (odd): max(peg_k, 5 for ppeg)
__global__ __launch_bounds__(256) void peg_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                  const float* __restrict__ b0, const float* __restrict__ w1,
                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, float* __restrict__ y, int N, int C,
                                                  int H0, int H, int k, int conv_1d) {
  constexpr int HALO = KK / 2, SIDE = 8 + 2 * HALO;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = (float*)smem;                           // [SIDE * SIDE][64]
  const int tid = threadIdx.x;
  const int tiles_w = (H + 7) / 8;
  const int ti0 = (blockIdx.x / tiles_w) * 8, tj0 = (blockIdx.x % tiles_w) * 8;
  const int c0 = blockIdx.y * 64;
  // ---- stage the patch: position p = (gi, gj) of the padded grid, 16 lanes x float4 per position
  for (int idx = tid; idx < SIDE * SIDE * 16; idx += 256) {
    const int p = idx >> 4, q = (idx & 15) * 4;
    const int gi = ti0 - HALO + p / SIDE, gj = tj0 - HALO + p % SIDE;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gi >= 0 && gi < H && gj >= 0 && gj < H && c0 + q < C) {
      const int t = gi * H + gj;                        // token of the (possibly 7 x 7-padded) grid
      if (t < H0 * H0) {                                // inside the wrapped square; beyond it: PPEG's zero padding
        const int src = t < N ? t : t - N;              // x[:, :add_length] appended (emb_position.py:36,68)
        v = *(const float4*)(x + (size_t)src * C + c0 + q);
      }
    }
    *(float4*)(tile + p * 64 + q) = v;
  }
  // ---- the channel's effective stencil
  const int c = c0 + (tid & 63);
  float wk[KK * KK];
  float bias = 0.f;
#pragma unroll
  for (int i = 0; i < KK * KK; ++i) wk[i] = 0.f;
  if (c < C) {
    // static (i, j) so that the stencil stays in registers; a tap belongs to a conv if it falls inside its window
    const int kw = conv_1d ? 1 : k, off = (KK - k) / 2;
#pragma unroll
    for (int i = 0; i < KK; ++i)
#pragma unroll
      for (int j = 0; j < KK; ++j) {
        float a = (i == KK / 2 && j == KK / 2) ? 1.0f : 0.f;          // + x^ (the identity branch)
        const int colc = KK / 2;                                        // the only column of a (k, 1) kernel
        {
          const int di = i - off, dj = conv_1d ? 0 : j - off;
          if (di >= 0 && di < k && dj >= 0 && dj < kw && (!conv_1d || j == colc)) a += w0[((size_t)c * k + di) * kw + dj];
        }
        if (w1) {                                                       // PPEG: 5 x 5 (5 x 1) and 3 x 3 (3 x 1)
          const int o5 = (KK - 5) / 2, o3 = (KK - 3) / 2, k5w = conv_1d ? 1 : 5, k3w = conv_1d ? 1 : 3;
          const int d5i = i - o5, d5j = conv_1d ? 0 : j - o5, d3i = i - o3, d3j = conv_1d ? 0 : j - o3;
          if (d5i >= 0 && d5i < 5 && d5j >= 0 && d5j < k5w && (!conv_1d || j == colc)) a += w1[((size_t)c * 5 + d5i) * k5w + d5j];
          if (d3i >= 0 && d3i < 3 && d3j >= 0 && d3j < k3w && (!conv_1d || j == colc)) a += w2[((size_t)c * 3 + d3i) * k3w + d3j];
        }
        wk[i * KK + j] = a;
      }
    if (b0) bias += b0[c];
    if (w1) {
      if (b1) bias += b1[c];
      if (b2) bias += b2[c];
    }
  }
  __syncthreads();
  if (c >= C) return;
  // ---- thread = (channel, 2 of the 8 tile rows): 16 outputs
  const int r0 = (tid >> 6) * 2;
  for (int oi = r0; oi < r0 + 2; ++oi)
    for (int oj = 0; oj < 8; ++oj) {
      const int gi = ti0 + oi, gj = tj0 + oj;
      const int t = gi * H + gj;
      if (gi >= H || gj >= H || t >= N) continue;       // only the first N tokens are kept (emb_position.py:53,77)
      float acc = bias;
#pragma unroll
      for (int di = 0; di < KK; ++di)
#pragma unroll
        for (int dj = 0; dj < KK; ++dj) acc += wk[di * KK + dj] * tile[((oi + di) * SIDE + oj + dj) * 64 + (tid & 63)];
      y[(size_t)t * C + c] = acc;
    }
}

}  // namespace

// w[0..2], b[0..2]: proj (k), proj1 (5), proj2 (3); PEG: only [0].  k odd <= 11.
hipError_t launch_peg(const float* x, const float* const* w, const float* const* b, float* y, int N, int C, int k,
                      int conv_1d, int ppeg, hipStream_t st) {
  int H0 = (int)ceil(sqrt((double)N));
  while ((long)H0 * H0 < N) ++H0;
  while (H0 > 1 && (long)(H0 - 1) * (H0 - 1) >= N) --H0;
  const int H = (ppeg && H0 < 7) ? 7 : H0;
  const int KK = ppeg ? (k > 5 ? k : 5) : k;
  const int tiles = ((H + 7) / 8) * ((H + 7) / 8);
  dim3 grid(tiles, (C + 63) / 64), block(256);
#define RRT_PEG(K_)                                                                                           \
  do {                                                                                                        \
    constexpr size_t lds = (size_t)(8 + 2 * (K_ / 2)) * (8 + 2 * (K_ / 2)) * 64 * sizeof(float);              \
    auto kern = peg_kernel<K_>;                                                                               \
    if (lds > 64 * 1024)                                                                                      \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
    kern<<<grid, block, lds, st>>>(x, w[0], b[0], ppeg ? w[1] : nullptr, ppeg ? b[1] : nullptr,               \
                                   ppeg ? w[2] : nullptr, ppeg ? b[2] : nullptr, y, N, C, H0, H, k, conv_1d); \
  } while (0)
  switch (KK) {
    case 1: case 3: RRT_PEG(3); break;
    case 5: RRT_PEG(5); break;
    case 7: RRT_PEG(7); break;
    case 9: RRT_PEG(9); break;
    case 11: RRT_PEG(11); break;
    default: return hipErrorInvalidValue;
  }
#undef RRT_PEG
  return hipGetLastError();
}
