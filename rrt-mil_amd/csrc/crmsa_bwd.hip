// crmsa_bwd.hip -- backward of the CR-MSA streaming stages (row f2 building block).
//
// Forward (modules/rmsa.py:303-335; 8x8 grid g, region r, slot p, representative n < k; v = LN2(x1), 0 at pads):
//     Lg[p,n] = <v_p, phi_n>       C = softmax_p(Lg)      Dk = softmax_n(Lg)      M = (Lg - mn) / (mx - mn + 1e-8)
//     rep[n,r] = sum_p C[p,n] v_p    rep2 = InnerAttention(rep)    out_p = sum_n (M Dk)[p,n] rep2[n,r]    x2 = x1 + out
// Backward, given dx2 and d rep (the inner attention's backward sits between (1) and (2), api.hip):
//  (1) tokdot:   dWd[p,n] = <dx2_p, rep2[n,r]>            wsum: d rep2[n,r] = sum_p (M Dk)[p,n] dx2_p
//  (2) tokdot<LN>: dC[p,n] = <d rep[n,r], v_p>
//  (3) region:   d Lg = C (dC - sum_p C dC)  +  Dk (dDk - sum_n Dk dDk)  +  dM / den  (+ the arg-min / arg-max terms
//                of the min-max normaliser), dDk = dWd M, dM = dWd Dk
//  (4) dx:       dv_p = sum_n C[p,n] d rep[n,r] + d Lg[p,n] phi_n ; dx1 = dx2 + LN2'(dv) ; d gamma2, d beta2, d phi
// Pad slots carry v = 0 and drop their output row: every gradient through them is zero and they are skipped.
#include "internal.h"

namespace {

constexpr int KMAX = RRT_MAX_CRMSA_K;

// ---- out[slot, n] = <row_t, vec[n, region(slot), :]> for every padded-grid token t (pads -> 0).
//      LN: row = LN(x_t) with the stashed mean / rstd, else the raw row.
template <int NV, bool LN>
__global__ __launch_bounds__(256) void crmsa_tokdot_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ vec, float* __restrict__ out,
                                                           int dim, int k, GridDev g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int R = g.rs * g.rs;
  for (int t = blockIdx.x * 4 + wave; t < g.Np; t += gridDim.x * 4) {
    const int slot = token_to_slot(t, g);
    if (t >= g.L) {
      if (lane < k) out[(size_t)slot * k + lane] = 0.f;
      continue;
    }
    const int reg = fdiv(slot, g.P, g.inv_P);
    float4 r[NV];
    float mean = 0.f, rstd = 1.f;
    if (LN) {
      mean = mean_rstd[2 * (size_t)t];
      rstd = mean_rstd[2 * (size_t)t + 1];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      r[v] = c < dim ? *(const float4*)(x + (size_t)t * dim + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (LN && c < dim) {
        const float4 gm = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
        r[v].x = (r[v].x - mean) * rstd * gm.x + bt.x;
        r[v].y = (r[v].y - mean) * rstd * gm.y + bt.y;
        r[v].z = (r[v].z - mean) * rstd * gm.z + bt.z;
        r[v].w = (r[v].w - mean) * rstd * gm.w + bt.w;
      }
    }
    for (int n = 0; n < k; ++n) {
      const float* vp = vec + ((size_t)n * R + reg) * dim;
      float acc = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < dim) {
          const float4 q = *(const float4*)(vp + c);
          acc += (r[v].x * q.x + r[v].y * q.y) + (r[v].z * q.z + r[v].w * q.w);
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) out[(size_t)slot * k + n] = acc;
    }
  }
}

// ---- out[n, reg, c] = sum_{p real} W[slot, n] * X[token(slot), c]     (grid: regions x 64-column slabs)
__global__ __launch_bounds__(256) void crmsa_wsum_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                         float* __restrict__ out, int dim, int k, GridDev g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ws = (float*)smem;                           // [P][KMAX]
  int* tok = (int*)(Ws + (size_t)g.P * KMAX);         // [P] token or -1
  float4* part = (float4*)(tok + ((g.P + 3) & ~3));   // [16 row groups][KMAX][16 column lanes]
  const int reg = blockIdx.x, col0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  const int R = g.rs * g.rs;
  const int ri = reg / g.rs, rj = reg - ri * g.rs;
  for (int p = tid; p < g.P; p += 256) {
    const int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
    const int t = (ri * g.s + pi) * g.H + rj * g.s + pj;
    const bool real = t < g.L;
    tok[p] = real ? t : -1;
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) Ws[p * KMAX + n] = real ? W[((size_t)reg * g.P + p) * k + n] : 0.f;
  }
  __syncthreads();
  const int cl = tid & 15, rg = tid >> 4;
  const int cc = col0 + cl * 4;
  float4 acc[KMAX];
#pragma unroll
  for (int n = 0; n < KMAX; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cc < dim) {
    for (int p0 = rg; p0 < g.P; p0 += 64) {
      float4 xv[4];
      int pp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pp[u] = p0 + 16 * u;
        const int t = pp[u] < g.P ? tok[pp[u]] : -1;
        xv[u] = t >= 0 ? *(const float4*)(X + (size_t)t * dim + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < 0) pp[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (pp[u] >= 0) {
#pragma unroll
          for (int n = 0; n < KMAX; ++n)
            if (n < k) {
              const float w = Ws[pp[u] * KMAX + n];
              acc[n].x += w * xv[u].x; acc[n].y += w * xv[u].y; acc[n].z += w * xv[u].z; acc[n].w += w * xv[u].w;
            }
        }
    }
  }
#pragma unroll
  for (int n = 0; n < KMAX; ++n)
    if (n < k) part[(rg * KMAX + n) * 16 + cl] = acc[n];
  __syncthreads();
  for (int idx = tid; idx < k * 16; idx += 256) {
    const int n = idx >> 4, c = idx & 15;
    const int c4 = col0 + c * 4;
    if (c4 < dim) {
      float4 a = part[n * 16 + c];
#pragma unroll
      for (int q = 1; q < 16; ++q) {
        const float4 b = part[(q * KMAX + n) * 16 + c];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      *(float4*)(out + ((size_t)n * R + reg) * dim + c4) = a;
    }
  }
}

// ---- one block per region: d Lg and C from Lg, dC, dWd
__global__ __launch_bounds__(256) void crmsa_bwd_region_kernel(const float* __restrict__ lg,
                                                               const float* __restrict__ dC,
                                                               const float* __restrict__ dWd,
                                                               float* __restrict__ dlg, float* __restrict__ Cw,
                                                               int k, GridDev g) {
  __shared__ float s_mx[KMAX], s_mn[KMAX], s_ise[KMAX], s_sc[KMAX], s_dmn[KMAX], s_dmx[KMAX];
  __shared__ int s_amx[KMAX], s_amn[KMAX];
  __shared__ float s_red[3][KMAX][4];
  const int reg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t base = (size_t)reg * g.P * k;
  const float* L = lg + base;
  // (a) per representative: max / first arg-max, min / first arg-min, sum of exp, sum_p C dC
  for (int n = wave; n < k; n += 4) {
    float mx = -3.0e38f, mn = 3.0e38f;
    int amx = 0x7fffffff, amn = 0x7fffffff;
    for (int p = lane; p < g.P; p += 64) {
      const float v = L[(size_t)p * k + n];
      if (v > mx) { mx = v; amx = p; }
      if (v < mn) { mn = v; amn = p; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float omx = __shfl_xor(mx, off), omn = __shfl_xor(mn, off);
      const int oamx = __shfl_xor(amx, off), oamn = __shfl_xor(amn, off);
      if (omx > mx || (omx == mx && oamx < amx)) { mx = omx; amx = oamx; }
      if (omn < mn || (omn == mn && oamn < amn)) { mn = omn; amn = oamn; }
    }
    float se = 0.f;
    for (int p = lane; p < g.P; p += 64) se += __expf(L[(size_t)p * k + n] - mx);
    se = wave_sum(se);
    const float ise = 1.0f / se;
    float sc = 0.f;
    for (int p = lane; p < g.P; p += 64)
      sc += __expf(L[(size_t)p * k + n] - mx) * ise * dC[base + (size_t)p * k + n];
    sc = wave_sum(sc);
    if (lane == 0) {
      s_mx[n] = mx; s_mn[n] = mn; s_ise[n] = ise; s_sc[n] = sc; s_amx[n] = amx; s_amn[n] = amn;
    }
  }
  __syncthreads();
  // (b) per slot: the three gradient paths; per-representative sums of the min / max terms
  float dmn[KMAX], dmx[KMAX];
#pragma unroll
  for (int n = 0; n < KMAX; ++n) dmn[n] = dmx[n] = 0.f;
  for (int p = tid; p < g.P; p += 256) {
    float v[KMAX], e[KMAX];
    float m2 = -3.0e38f;
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) { v[n] = L[(size_t)p * k + n]; m2 = fmaxf(m2, v[n]); }
    float se2 = 0.f;
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) { e[n] = __expf(v[n] - m2); se2 += e[n]; }
    const float inv2 = 1.0f / se2;
    float dk[KMAX], mm[KMAX], ddk[KMAX];
    float sd = 0.f;
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) {
        const float den = s_mx[n] - s_mn[n] + 1e-8f;
        dk[n] = e[n] * inv2;
        mm[n] = (v[n] - s_mn[n]) / den;
        const float dw = dWd[base + (size_t)p * k + n];
        ddk[n] = dw * mm[n];
        sd += dk[n] * ddk[n];
      }
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) {
        const float den = s_mx[n] - s_mn[n] + 1e-8f;
        const float dw = dWd[base + (size_t)p * k + n];
        const float dM = dw * dk[n];
        const float c = __expf(v[n] - s_mx[n]) * s_ise[n];
        const float dc = dC[base + (size_t)p * k + n];
        Cw[base + (size_t)p * k + n] = c;
        dlg[base + (size_t)p * k + n] = c * (dc - s_sc[n]) + dk[n] * (ddk[n] - sd) + dM / den;
        // d/d mn = -1/den + (Lg - mn)/den^2 in THIS form (as autograd decomposes the quotient): at the arg-min slot
        // the -dM/den term cancels the direct dM/den exactly -- with one-token regions den = 1e-8 and the
        // simplified (Lg - mx - eps)/den^2 left O(10 dM) of rounding garbage behind
        const float t2 = dM * (v[n] - s_mn[n]) / (den * den);
        dmn[n] += t2 - dM / den;
        dmx[n] -= t2;
      }
  }
#pragma unroll
  for (int n = 0; n < KMAX; ++n)
    if (n < k) {
      const float a = wave_sum(dmn[n]), b = wave_sum(dmx[n]);
      if (lane == 0) {
        if (wave == 0) { s_dmn[n] = a; s_dmx[n] = b; }
        else { s_red[wave - 1][n][0] = a; s_red[wave - 1][n][1] = b; }
      }
    }
  __syncthreads();
  // (c) the normaliser's min / max receive their gradients at the (first) arg-min / arg-max slot
  if (tid < k) {
    const int n = tid;
    float a = s_dmn[n], b = s_dmx[n];
    for (int w = 0; w < 3; ++w) { a += s_red[w][n][0]; b += s_red[w][n][1]; }
    dlg[base + (size_t)s_amn[n] * k + n] += a;
    __threadfence_block();
    dlg[base + (size_t)s_amx[n] * k + n] += b;
  }
}

// ---- dx1 = dx2 + LN2'(dv), dv_t = sum_n C[slot,n] d rep[n,r] + d Lg[slot,n] phi_n ; partials of d gamma2, d beta2, d phi
constexpr int DXB_BLOCKS = 512;
// crmsa_mlp (phi = Linear(D, D/4) -> Tanh -> Linear(D/4, k), rmsa.py:248-252): th = tanh(hid),
// d hid = (d logits . W2) o (1 - th^2).  One thread per hidden element.
__global__ __launch_bounds__(256) void crmsa_mlp_bwd_hidden_kernel(const float* __restrict__ hid,
                                                                   const float* __restrict__ dlg,
                                                                   const float* __restrict__ w2, float* __restrict__ th,
                                                                   float* __restrict__ dhid, size_t rows, int hdim,
                                                                   int k) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * (size_t)hdim) return;
  const size_t r = i / hdim;
  const int j = (int)(i - r * hdim);
  const float t = tanhf(hid[i]);
  float d = 0.f;
  for (int n = 0; n < k; ++n) d += dlg[r * k + n] * w2[(size_t)n * hdim + j];
  th[i] = t;
  dhid[i] = d * (1.0f - t * t);
}

template <int NV, bool MLP>
__global__ __launch_bounds__(256) void crmsa_bwd_dx_kernel(const float* __restrict__ x1, const float* __restrict__ dx2,
                                                           const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ phi,
                                                           const float* __restrict__ Cw, const float* __restrict__ dlg,
                                                           const float* __restrict__ drep, float* __restrict__ dx1,
                                                           float* __restrict__ part, int dim, int k, GridDev g) {
  // MLP: `phi` is d v_phi [Np8, dim] (region-major rows, = d hid . W1): the logits' path arrives as a row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* phi_t = (float*)smem;                                  // [k][dim]   (matrix phi only)
  float4* red = (float4*)(phi_t + (MLP ? 0 : (size_t)k * dim)); // [3 waves][rows][NV * 64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int R = g.rs * g.rs;
  if (!MLP) {
    for (int idx = threadIdx.x; idx < dim * k; idx += 256) {
      const int d = idx / k, n = idx - d * k;
      phi_t[n * dim + d] = phi[idx];
    }
  }
  __syncthreads();
  const float inv_d = 1.0f / (float)dim;
  float4 gm[NV], bt[NV], dg[NV], db[NV], dph[KMAX][NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * 64 + lane) * 4;
    gm[v] = c < dim ? *(const float4*)(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    bt[v] = c < dim ? *(const float4*)(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[v] = db[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < KMAX; ++n) dph[n][v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int t = blockIdx.x * 4 + wave; t < g.L; t += gridDim.x * 4) {
    const int slot = token_to_slot(t, g);
    const int reg = fdiv(slot, g.P, g.inv_P);
    const float mean = mean_rstd[2 * (size_t)t], rstd = mean_rstd[2 * (size_t)t + 1];
    float cw[KMAX], dl[KMAX];
#pragma unroll
    for (int n = 0; n < KMAX; ++n)
      if (n < k) { cw[n] = Cw[(size_t)slot * k + n]; dl[n] = dlg[(size_t)slot * k + n]; }
    float4 xh[NV], dv[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      xh[v] = dv[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < dim) {
        const float4 xr = *(const float4*)(x1 + (size_t)t * dim + c);
        xh[v] = make_float4((xr.x - mean) * rstd, (xr.y - mean) * rstd, (xr.z - mean) * rstd, (xr.w - mean) * rstd);
        const float4 vv = make_float4(xh[v].x * gm[v].x + bt[v].x, xh[v].y * gm[v].y + bt[v].y,
                                      xh[v].z * gm[v].z + bt[v].z, xh[v].w * gm[v].w + bt[v].w);
        if (MLP) dv[v] = *(const float4*)(phi + (size_t)slot * dim + c);
#pragma unroll
        for (int n = 0; n < KMAX; ++n)
          if (n < k) {
            const float4 dr = *(const float4*)(drep + ((size_t)n * R + reg) * dim + c);
            dv[v].x += cw[n] * dr.x; dv[v].y += cw[n] * dr.y; dv[v].z += cw[n] * dr.z; dv[v].w += cw[n] * dr.w;
            if (!MLP) {
              const float4 ph = *(const float4*)(phi_t + n * dim + c);
              dv[v].x += dl[n] * ph.x; dv[v].y += dl[n] * ph.y; dv[v].z += dl[n] * ph.z; dv[v].w += dl[n] * ph.w;
              dph[n][v].x += dl[n] * vv.x; dph[n][v].y += dl[n] * vv.y; dph[n][v].z += dl[n] * vv.z; dph[n][v].w += dl[n] * vv.w;
            }
          }
        dg[v].x += dv[v].x * xh[v].x; dg[v].y += dv[v].y * xh[v].y; dg[v].z += dv[v].z * xh[v].z; dg[v].w += dv[v].w * xh[v].w;
        db[v].x += dv[v].x; db[v].y += dv[v].y; db[v].z += dv[v].z; db[v].w += dv[v].w;
        dv[v].x *= gm[v].x; dv[v].y *= gm[v].y; dv[v].z *= gm[v].z; dv[v].w *= gm[v].w;
        s1 += (dv[v].x + dv[v].y) + (dv[v].z + dv[v].w);
        s2 += (dv[v].x * xh[v].x + dv[v].y * xh[v].y) + (dv[v].z * xh[v].z + dv[v].w * xh[v].w);
      }
    }
    const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < dim) {
        const float4 up = *(const float4*)(dx2 + (size_t)t * dim + c);
        float4 o;
        o.x = up.x + rstd * (dv[v].x - c1 - xh[v].x * c2);
        o.y = up.y + rstd * (dv[v].y - c1 - xh[v].y * c2);
        o.z = up.z + rstd * (dv[v].z - c1 - xh[v].z * c2);
        o.w = up.w + rstd * (dv[v].w - c1 - xh[v].w * c2);
        *(float4*)(dx1 + (size_t)t * dim + c) = o;
      }
    }
  }
  // block partials: rows 0,1 = d gamma, d beta; rows 2.. = d phi^T [k][dim] (matrix phi only)
  const int rows = MLP ? 2 : 2 + k;
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      red[((wave - 1) * rows + 0) * (NV * 64) + v * 64 + lane] = dg[v];
      red[((wave - 1) * rows + 1) * (NV * 64) + v * 64 + lane] = db[v];
#pragma unroll
      for (int n = 0; n < KMAX; ++n)
        if (!MLP && n < k) red[((wave - 1) * rows + 2 + n) * (NV * 64) + v * 64 + lane] = dph[n][v];
    }
  }
  __syncthreads();
  if (wave == 0) {
    float* out = part + (size_t)blockIdx.x * rows * dim;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < dim) {
        float4 a = dg[v], b = db[v];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 p = red[(w * rows + 0) * (NV * 64) + v * 64 + lane];
          const float4 q = red[(w * rows + 1) * (NV * 64) + v * 64 + lane];
          a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
          b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
        }
        *(float4*)(out + c) = a;
        *(float4*)(out + dim + c) = b;
#pragma unroll
        for (int n = 0; n < KMAX; ++n)
          if (!MLP && n < k) {
            float4 d = dph[n][v];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
              const float4 p = red[(w * rows + 2 + n) * (NV * 64) + v * 64 + lane];
              d.x += p.x; d.y += p.y; d.z += p.z; d.w += p.w;
            }
            *(float4*)(out + (size_t)(2 + n) * dim + c) = d;
          }
      }
    }
  }
}

}  // namespace

hipError_t launch_crmsa_tokdot(const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                               const float* vec, float* out, int dim, int k, const GridDev& g, hipStream_t st) {
  const int need = (g.Np + 3) / 4;
  dim3 grid(need < 2048 ? need : 2048), block(256);
  const bool ln = mean_rstd != nullptr;
#define RRT_TD(NV)                                                                                        \
  do {                                                                                                    \
    if (ln) crmsa_tokdot_kernel<NV, true><<<grid, block, 0, st>>>(x, mean_rstd, gamma, beta, vec, out, dim, k, g); \
    else crmsa_tokdot_kernel<NV, false><<<grid, block, 0, st>>>(x, mean_rstd, gamma, beta, vec, out, dim, k, g);   \
  } while (0)
  if (dim <= 256) RRT_TD(1);
  else if (dim <= 512) RRT_TD(2);
  else if (dim <= 1024) RRT_TD(4);
  else RRT_TD(8);
#undef RRT_TD
  return hipGetLastError();
}

hipError_t launch_crmsa_wsum(const float* X, const float* W, float* out, int dim, int k, const GridDev& g,
                             hipStream_t st) {
  const size_t lds = ((size_t)g.P * KMAX + ((g.P + 3) & ~3)) * 4 + (size_t)16 * KMAX * 16 * sizeof(float4);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)crmsa_wsum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  crmsa_wsum_kernel<<<dim3(g.rs * g.rs, (dim + 63) / 64), 256, lds, st>>>(X, W, out, dim, k, g);
  return hipGetLastError();
}

hipError_t launch_crmsa_bwd_region(const float* lg, const float* dC, const float* dWd, float* dlg, float* Cw, int k,
                                   const GridDev& g, hipStream_t st) {
  crmsa_bwd_region_kernel<<<dim3(g.rs * g.rs), 256, 0, st>>>(lg, dC, dWd, dlg, Cw, k, g);
  return hipGetLastError();
}

size_t crmsa_bwd_dx_workspace(int dim, int k) { return (size_t)DXB_BLOCKS * (2 + k) * dim * sizeof(float); }

hipError_t launch_crmsa_mlp_bwd_hidden(const float* hid, const float* dlg, const float* w2, float* th, float* dhid,
                                       size_t rows, int hdim, int k, hipStream_t st) {
  const size_t n = rows * (size_t)hdim;
  crmsa_mlp_bwd_hidden_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(hid, dlg, w2, th, dhid, rows, hdim, k);
  return hipGetLastError();
}

// out_rows [2 + k, dim]: d gamma2, d beta2, d phi^T.  mlp: `phi` = d v_phi rows [Np8, dim], out_rows [2, dim]
// dphi_t (not mlp; out_rows 16-byte aligned): out_rows takes only [2, dim] and d phi leaves as phi's own [dim, k] from the
// same reduce launch
hipError_t launch_crmsa_bwd_dx(const float* x1, const float* dx2, const float* mean_rstd, const float* gamma,
                               const float* beta, const float* phi, const float* Cw, const float* dlg,
                               const float* drep, float* dx1, float* out_rows, float* part, int dim, int k,
                               const GridDev& g, bool mlp, hipStream_t st, float* dphi_t, ReduceJobs* defer) {
  if (dim > 1024) return hipErrorInvalidValue;
  const int need = (g.L + 3) / 4;
  const int blocks = need < DXB_BLOCKS ? need : DXB_BLOCKS;
  const int nvv = dim <= 256 ? 1 : (dim <= 512 ? 2 : 4);
  const int rows = mlp ? 2 : 2 + k;
  const size_t lds = (mlp ? 0 : (size_t)k * dim * 4) + (size_t)3 * rows * nvv * 64 * sizeof(float4);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
#define RRT_DX(NV)                                                                                          \
  do {                                                                                                      \
    auto kern = mlp ? crmsa_bwd_dx_kernel<NV, true> : crmsa_bwd_dx_kernel<NV, false>;                       \
    if (lds > 64 * 1024)                                                                                    \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
    kern<<<dim3(blocks), 256, lds, st>>>(x1, dx2, mean_rstd, gamma, beta, phi, Cw, dlg, drep, dx1, part, dim, k, g); \
  } while (0)
  if (nvv == 1) RRT_DX(1);
  else if (nvv == 2) RRT_DX(2);
  else RRT_DX(4);
#undef RRT_DX
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (dphi_t && !mlp)
    return reduce_or_defer(defer, part, out_rows, blocks, (size_t)rows * dim, st, (size_t)2 * dim, dphi_t, dim, k);
  return reduce_or_defer(defer, part, out_rows, blocks, (size_t)rows * dim, st);
}
