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
                                                  int H0, int H, int k, int conv_1d, int bwd) {
  // bwd: the adjoint stencil.  Source = the upstream gradient (token t < N, nothing wrapped), taps flipped, no bias,
  // and EVERY position of the wrapped square gets an output (the wrapped copies' gradients are folded afterwards).
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
      if (bwd) {
        if (t < N) v = *(const float4*)(x + (size_t)t * C + c0 + q);
      } else if (t < H0 * H0) {                         // inside the wrapped square; beyond it: PPEG's zero padding
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
        wk[bwd ? (KK - 1 - i) * KK + (KK - 1 - j) : i * KK + j] = a;
      }
    if (!bwd) {
      if (b0) bias += b0[c];
      if (w1) {
        if (b1) bias += b1[c];
        if (b2) bias += b2[c];
      }
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
      if (gi >= H || gj >= H || t >= (bwd ? H0 * H0 : N)) continue;   // forward keeps the first N tokens (emb_position.py:53,77)
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
static hipError_t launch_peg_impl(const float* x, const float* const* w, const float* const* b, float* y, int N, int C,
                                  int k, int conv_1d, int ppeg, int bwd, hipStream_t st) {
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
                                   ppeg ? w[2] : nullptr, ppeg ? b[2] : nullptr, y, N, C, H0, H, k, conv_1d, bwd); \
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

hipError_t launch_peg(const float* x, const float* const* w, const float* const* b, float* y, int N, int C, int k,
                      int conv_1d, int ppeg, hipStream_t st) {
  return launch_peg_impl(x, w, b, y, N, C, k, conv_1d, ppeg, 0, st);
}

namespace {

// dx[t] = g[t] + g[N + t] for the tokens that were wrapped into the tail of the square
__global__ __launch_bounds__(256) void peg_fold_kernel(const float* __restrict__ g, float* __restrict__ dx, int N, int C,
                                                       int HH) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (size_t)N * C) return;
  const size_t t = i / C;
  float4 v = *(const float4*)(g + i);
  if (t + N < (size_t)HH) {
    const float4 o = *(const float4*)(g + i + (size_t)N * C);
    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
  }
  *(float4*)(dx + i) = v;
}

// per (tile, 64-channel slab): part[tile][tap][c] = sum over the tile's kept outputs of dy[t,c] * x^[t + tap]
template <int KK>
__global__ __launch_bounds__(256) void peg_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ part, int N, int C, int H0, int H) {
  constexpr int HALO = KK / 2, SIDE = 8 + 2 * HALO;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = (float*)smem;                           // [SIDE * SIDE][64]
  float* red = tile + SIDE * SIDE * 64;                 // [3][KK * KK][64]
  const int tid = threadIdx.x;
  const int tiles_w = (H + 7) / 8;
  const int ti0 = (blockIdx.x / tiles_w) * 8, tj0 = (blockIdx.x % tiles_w) * 8;
  const int c0 = blockIdx.y * 64;
  for (int idx = tid; idx < SIDE * SIDE * 16; idx += 256) {
    const int p = idx >> 4, q = (idx & 15) * 4;
    const int gi = ti0 - HALO + p / SIDE, gj = tj0 - HALO + p % SIDE;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gi >= 0 && gi < H && gj >= 0 && gj < H && c0 + q < C) {
      const int t = gi * H + gj;
      if (t < H0 * H0) v = *(const float4*)(x + (size_t)(t < N ? t : t - N) * C + c0 + q);
    }
    *(float4*)(tile + p * 64 + q) = v;
  }
  __syncthreads();
  const int cl = tid & 63, c = c0 + cl, r0 = (tid >> 6) * 2;
  float acc[KK * KK];
#pragma unroll
  for (int i = 0; i < KK * KK; ++i) acc[i] = 0.f;
  if (c < C) {
    for (int oi = r0; oi < r0 + 2; ++oi)
      for (int oj = 0; oj < 8; ++oj) {
        const int gi = ti0 + oi, gj = tj0 + oj, t = gi * H + gj;
        if (gi >= H || gj >= H || t >= N) continue;
        const float g = dy[(size_t)t * C + c];
#pragma unroll
        for (int di = 0; di < KK; ++di)
#pragma unroll
          for (int dj = 0; dj < KK; ++dj) acc[di * KK + dj] += g * tile[((oi + di) * SIDE + oj + dj) * 64 + cl];
      }
  }
  const int q = tid >> 6;
  if (q > 0) {
#pragma unroll
    for (int i = 0; i < KK * KK; ++i) red[((q - 1) * KK * KK + i) * 64 + cl] = acc[i];
  }
  __syncthreads();
  if (q == 0 && c < C) {
#pragma unroll
    for (int i = 0; i < KK * KK; ++i)
      part[((size_t)blockIdx.x * KK * KK + i) * C + c] =
          acc[i] + red[i * 64 + cl] + red[(KK * KK + i) * 64 + cl] + red[(2 * KK * KK + i) * 64 + cl];
  }
}

// dWeff [KK*KK][C] -> the convs' own weight gradients [C, 1, k, kw]
__global__ __launch_bounds__(256) void peg_scatter_dw_kernel(const float* __restrict__ dweff, float* __restrict__ dw,
                                                             int C, int KK, int k, int conv_1d) {
  const int kw = conv_1d ? 1 : k, off = (KK - k) / 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= C * k * kw) return;
  const int c = idx / (k * kw), r = idx - c * k * kw, di = r / kw, dj = r - di * kw;
  dw[idx] = dweff[(size_t)((off + di) * KK + (conv_1d ? KK / 2 : off + dj)) * C + c];
}

}  // namespace

size_t peg_bwd_workspace(int N, int C, int k, int ppeg) {
  int H0 = (int)ceil(sqrt((double)N));
  while ((long)H0 * H0 < N) ++H0;
  const int H = (ppeg && H0 < 7) ? 7 : H0;
  const int KK = ppeg ? (k > 5 ? k : 5) : (k < 3 ? 3 : k);
  const size_t tiles = (size_t)((H + 7) / 8) * ((H + 7) / 8);
  return ((size_t)H0 * H0 * C + (tiles + 1) * KK * KK * C + (size_t)((N + 127) / 128 + 1) * C) * sizeof(float) + 1024;
}

// dy [N, C] -> dx [N, C], dw[i] / db[i] (i < 3; PEG: only 0; biases may be null).  x = the encoder's input of the
// forward stage.  ws: peg_bwd_workspace bytes.
hipError_t launch_peg_backward(const float* x, const float* dy, const float* const* w, float* dx, float* const* dw,
                               float* const* db, int N, int C, int k, int conv_1d, int ppeg, void* ws, hipStream_t st) {
  int H0 = (int)ceil(sqrt((double)N));
  while ((long)H0 * H0 < N) ++H0;
  while (H0 > 1 && (long)(H0 - 1) * (H0 - 1) >= N) --H0;
  const int H = (ppeg && H0 < 7) ? 7 : H0;
  const int KK = ppeg ? (k > 5 ? k : 5) : (k < 3 ? 3 : k);
  const int tiles = ((H + 7) / 8) * ((H + 7) / 8);
  float* g = (float*)ws;                                  // [H0*H0][C] adjoint-stencil output
  float* part = g + (size_t)H0 * H0 * C;                  // [tiles][KK*KK][C]
  float* dweff = part + (size_t)tiles * KK * KK * C;      // [KK*KK][C]
  float* cs = dweff + (size_t)KK * KK * C;                // column-sum scratch
  const float* nob[3] = {nullptr, nullptr, nullptr};
  hipError_t e = launch_peg_impl(dy, w, nob, g, N, C, k, conv_1d, ppeg, 1, st);
  if (e != hipSuccess) return e;
  const size_t n4 = (size_t)N * C / 4;
  peg_fold_kernel<<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, st>>>(g, dx, N, C, H0 * H0);
  dim3 grid(tiles, (C + 63) / 64);
#define RRT_PEGW(K_)                                                                                          \
  do {                                                                                                        \
    constexpr size_t lds = ((size_t)(8 + 2 * (K_ / 2)) * (8 + 2 * (K_ / 2)) * 64 + 3 * K_ * K_ * 64) * sizeof(float); \
    auto kern = peg_bwd_dw_kernel<K_>;                                                                        \
    if (lds > 64 * 1024)                                                                                      \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
    kern<<<grid, 256, lds, st>>>(x, dy, part, N, C, H0, H);                                                   \
  } while (0)
  switch (KK) {
    case 3: RRT_PEGW(3); break;
    case 5: RRT_PEGW(5); break;
    case 7: RRT_PEGW(7); break;
    case 9: RRT_PEGW(9); break;
    case 11: RRT_PEGW(11); break;
    default: return hipErrorInvalidValue;
  }
#undef RRT_PEGW
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = launch_reduce_partials(part, dweff, tiles, (size_t)KK * KK * C, st);
  if (e != hipSuccess) return e;
  const int ks[3] = {k, 5, 3};
  for (int i = 0; i < (ppeg ? 3 : 1); ++i) {
    if (!dw[i]) return hipErrorInvalidValue;
    const int kw = conv_1d ? 1 : ks[i], n = C * ks[i] * kw;
    peg_scatter_dw_kernel<<<dim3((n + 255) / 256), 256, 0, st>>>(dweff, dw[i], C, KK, ks[i], conv_1d);
    if (db[i]) {
      e = launch_colsum(dy, db[i], cs, N, C, st);
      if (e != hipSuccess) return e;
    }
  }
  return hipGetLastError();
}

