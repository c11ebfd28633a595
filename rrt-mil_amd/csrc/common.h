// common.h -- shared device helpers for the gfx950 RRTEncoder kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RRT_WAVE 64

// Non-temporal row accesses: the lines do not displace what the kernels of the other bags in flight keep in the L2 / MALL.
// Measured per site on MI355X, four bags in flight (round 5, tools/experiments/ab_lib.sh; RRT_NO_NT_DISPATCH / RRT_NT_LN1
// rebuild the other side of each comparison):
//   * the forward's LAST kernel (crmsa_dispatch_ln): x1 read for the last time, y written and not read again by this forward
//     -- non-temporal: bf16 18.5 k -> 19.6-19.9 k slides/s (the store alone: 19.6-19.7 k), configs[3] +2 %, configs[4] +1-2 %,
//     configs[2] and fp32 within noise, one bag in flight never slower;
//   * LN1's read of x (ln_partition*, cast16): x comes back as the residual 20-180 us later, and a non-temporal first read
//     costs that second one its MALL hit -- bf16 19.2 k -> 18.8 k, one bag 79 -> 81 us: plain loads.
__device__ __forceinline__ float4 ld_nt(const float* p) {
  const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_nt(float* p, const float4 v) {
  const f32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, (f32x4*)p);
}
template <bool NT> __device__ __forceinline__ float4 ld_row(const float* p) { if constexpr (NT) return ld_nt(p); else return *(const float4*)p; }
template <bool NT> __device__ __forceinline__ void st_row(float* p, const float4 v) { if constexpr (NT) st_nt(p, v); else *(float4*)p = v; }
#ifdef RRT_NT_LN1
constexpr bool NT_LN1 = true;
#else
constexpr bool NT_LN1 = false;
#endif
#ifdef RRT_NT_RESID
constexpr bool NT_RESID = true;      // (experiment) the out-projection's read of the residual rows, x's last use: no gain in
                                     // any configuration, one bag in flight 1-2 us slower -- not adopted
#else
constexpr bool NT_RESID = false;
#endif
#ifdef RRT_NO_NT_DISPATCH
constexpr bool NT_DISPATCH = false;
#else
constexpr bool NT_DISPATCH = true;
#endif

// Device copy of the region grid (modules/rmsa.py:175-202 geometry, 32-bit).
struct GridDev {
  int L;    // real tokens
  int H;    // padded grid side
  int s;    // region side
  int rs;   // regions per side
  int P;    // tokens per region (s*s)
  int Np;   // H*H
  float inv_H, inv_s, inv_rs, inv_P;   // reciprocals for the division-free index maps
  int Rt;   // CR-MSA: regions per representative row of rep / rep2 [k, Rt, D] (rs * rs; B * rs * rs when the bags of a
            // batch share one inner attention, rrt_encoder_forward_batch_f32 -- the bag's pointer is offset by the caller)
};

// n / d for 0 <= n < 2^24 via a float reciprocal + one correction step (exact; ~6 VALU
// instead of the ~40 of an integer division -- the index maps sit in GEMM epilogues)
__device__ __forceinline__ int fdiv(int n, int d, float inv_d) {
  int q = (int)((float)n * inv_d);
  int r = n - q * d;
  q += (r >= d) - (r < 0);
  return q;
}

// padded-grid token index -> region-major slot (region_partition, rmsa.py:28-39)
__device__ __forceinline__ int token_to_slot(int t, const GridDev& g) {
  int i = fdiv(t, g.H, g.inv_H), j = t - i * g.H;
  int ri = fdiv(i, g.s, g.inv_s), pi = i - ri * g.s;
  int rj = fdiv(j, g.s, g.inv_s), pj = j - rj * g.s;
  return (ri * g.rs + rj) * g.P + pi * g.s + pj;
}
// region-major slot -> padded-grid token index (region_reverse, rmsa.py:41-54)
__device__ __forceinline__ int slot_to_token(int slot, const GridDev& g) {
  int reg = fdiv(slot, g.P, g.inv_P), p = slot - reg * g.P;
  int ri = fdiv(reg, g.rs, g.inv_rs), rj = reg - ri * g.rs;
  int pi = fdiv(p, g.s, g.inv_s), pj = p - pi * g.s;
  return (ri * g.s + pi) * g.H + rj * g.s + pj;
}

// Stateless dropout mask (training, proj_drop): element (row, col) of a layer's proj output is kept iff
// hash(seed, row * ncols + col) >= p * 2^32.  Forward epilogue and backward regenerate the same mask from
// (seed, index): nothing is stored.  murmur3's 32-bit finaliser over a Weyl-scrambled index; the parity tests
// rebuild the mask in numpy (tests/hip_util.py::dropout_keep) and hand it to the oracle.
__device__ __forceinline__ bool rrt_drop_keep(unsigned seed, unsigned long long idx, unsigned thresh) {
  unsigned h = (unsigned)idx * 0x9E3779B1u + (unsigned)(idx >> 32) * 0x85EBCA77u + seed;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h >= thresh;
}

// Reproducer switches (tools/repro_guarded_ln.py; ablation builds only):
//   -DRRT_FORCE_GUARDED     the LayerNorm-type kernels keep their lane-predicated (column-guarded) form at every width
//   -DRRT_PERMLANE_NOPS=n   n extra wait states in front of every v_permlane16/32_swap of the reduction helpers
#ifdef RRT_FORCE_GUARDED
constexpr bool RRT_ALLOW_FULL = false;
#else
constexpr bool RRT_ALLOW_FULL = true;
#endif
#ifdef RRT_PERMLANE_NOPS
#define RRT_PERMLANE_PAD(a, b) asm volatile("s_nop %2" : "+v"(a), "+v"(b) : "n"(RRT_PERMLANE_NOPS - 1))
#else
#define RRT_PERMLANE_PAD(a, b)
#endif

// ---- lane ^ 16 / lane ^ 32 reductions on the VALU ------------------------------------------------
// gfx950 v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane rows between two registers.  __shfl_xor lowers to
// ds_bpermute_b32, which queues in the LDS pipe behind every wave's fragment reads (traced in the attention phases: a
// softmax with six dependent bpermutes took 3.5 K cycles next to ~1.5 K of VALU work).
static __device__ __forceinline__ float max_xor16(float v) {
  unsigned a = __float_as_uint(v), b = a;
  RRT_PERMLANE_PAD(a, b);
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
static __device__ __forceinline__ float max_xor32(float v) {
  unsigned a = __float_as_uint(v), b = a;
  RRT_PERMLANE_PAD(a, b);
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
static __device__ __forceinline__ float sum_xor16(float v) {
  unsigned a = __float_as_uint(v), b = a;
  RRT_PERMLANE_PAD(a, b);
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
static __device__ __forceinline__ float sum_xor32(float v) {
  unsigned a = __float_as_uint(v), b = a;
  RRT_PERMLANE_PAD(a, b);
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// the two values of a lane pair (lane, lane ^ 16) / (lane, lane ^ 32): a = the even row's (lower half's), b = the odd
// row's (upper half's) -- the same (a, b) in both lanes, so symmetric formulas give both lanes the same bits
static __device__ __forceinline__ void pair_xor16(float v, float& a, float& b) {
  unsigned x = __float_as_uint(v), y = x;
  RRT_PERMLANE_PAD(x, y);
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
static __device__ __forceinline__ void pair_xor32(float v, float& a, float& b) {
  unsigned x = __float_as_uint(v), y = x;
  RRT_PERMLANE_PAD(x, y);
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// Wave-wide reductions without LDS traffic: __shfl_xor lowers to ds_bpermute_b32 (an LDS-pipe round
// trip per step, 6 dependent steps per reduction); here 4 DPP row rotations reduce each 16-lane row
// in the VALU and 4 v_readlane + scalar-operand adds combine the rows.  Result is wave-uniform.
#define RRT_DPP_ROR(x, n) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), 0x120 + (n), 0xF, 0xF, false))
__device__ __forceinline__ float rrt_readlane(float v, int l) {
#ifdef RRT_READLANE_NOPS      // reproducer switch: extra wait states between the producer of v and v_readlane
  asm volatile("s_nop %1" : "+v"(v) : "n"(RRT_READLANE_NOPS - 1));
#endif
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += RRT_DPP_ROR(v, 8);
  v += RRT_DPP_ROR(v, 4);
  v += RRT_DPP_ROR(v, 2);
  v += RRT_DPP_ROR(v, 1);
#ifdef RRT_WAVE_SUM_READLANE
  return (rrt_readlane(v, 0) + rrt_readlane(v, 16)) + (rrt_readlane(v, 32) + rrt_readlane(v, 48));
#else
  return sum_xor32(sum_xor16(v));     // rows combined by lane swaps, the result in every lane (no SGPR round trip)
#endif
}
// FOUR wave totals for little more than the price of one (round 6): the first two levels of the butterfly run on lane
// swaps that fold two values into one register each -- permlane32_swap(a, b) leaves [a.lo | b.lo] and [a.hi | b.hi], whose sum
// holds a's 32-lane partials in lanes 0-31 and b's in lanes 32-63; permlane16_swap of two such registers leaves a, c, b, d in
// the four 16-lane rows -- and only the four row rotations are paid once per group: 2 + 2 + 2 + 4 x 2 instructions + four
// v_readlane for four totals where four wave_sum calls are 4 x 15.  Totals come back wave-uniform (SGPRs).  The summation
// order differs from wave_sum's (last bits).
__device__ __forceinline__ void wave_sum4(float a, float b, float c, float d, float& ta, float& tb, float& tc, float& td) {
  unsigned xa = __float_as_uint(a), xb = __float_as_uint(b), xc = __float_as_uint(c), xd = __float_as_uint(d);
  RRT_PERMLANE_PAD(xa, xb);
  const auto r0 = __builtin_amdgcn_permlane32_swap(xa, xb, false, false);
  RRT_PERMLANE_PAD(xc, xd);
  const auto r1 = __builtin_amdgcn_permlane32_swap(xc, xd, false, false);
  unsigned ab = __float_as_uint(__uint_as_float(r0[0]) + __uint_as_float(r0[1]));
  unsigned cd = __float_as_uint(__uint_as_float(r1[0]) + __uint_as_float(r1[1]));
  RRT_PERMLANE_PAD(ab, cd);
  const auto r2 = __builtin_amdgcn_permlane16_swap(ab, cd, false, false);   // rows: [a, c, b, d] twice
  float s = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
  s += RRT_DPP_ROR(s, 8);
  s += RRT_DPP_ROR(s, 4);
  s += RRT_DPP_ROR(s, 2);
  s += RRT_DPP_ROR(s, 1);
  ta = rrt_readlane(s, 0);
  tc = rrt_readlane(s, 16);
  tb = rrt_readlane(s, 32);
  td = rrt_readlane(s, 48);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, RRT_DPP_ROR(v, 8));
  v = fmaxf(v, RRT_DPP_ROR(v, 4));
  v = fmaxf(v, RRT_DPP_ROR(v, 2));
  v = fmaxf(v, RRT_DPP_ROR(v, 1));
  return fmaxf(fmaxf(rrt_readlane(v, 0), rrt_readlane(v, 16)), fmaxf(rrt_readlane(v, 32), rrt_readlane(v, 48)));
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, RRT_DPP_ROR(v, 8));
  v = fminf(v, RRT_DPP_ROR(v, 4));
  v = fminf(v, RRT_DPP_ROR(v, 2));
  v = fminf(v, RRT_DPP_ROR(v, 1));
  return fminf(fminf(rrt_readlane(v, 0), rrt_readlane(v, 16)), fminf(rrt_readlane(v, 32), rrt_readlane(v, 48)));
}

// 16-byte global -> LDS DMA (global_load_lds_dwordx4): the LDS destination is the
// wave-uniform byte address `lds_addr` (held in M0) + lane*16; the global source is per lane.
// Issued through inline asm on purpose: with the builtin, hipcc drains vmcnt(0) before the
// next ds_read (it cannot prove the DMA targets the *other* LDS buffer), which serialises
// the prefetch with the MFMA phase.  The asm form is invisible to that bookkeeping, so the
// kernels wait for it themselves: wait_vm0() before the barrier that publishes a tile.
__device__ __forceinline__ void dma16(const float* gsrc, unsigned lds_addr /* wave-uniform */) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
// Same DMA with the address split as (wave-uniform 64-bit base in SGPRs) + (per-lane 32-bit
// byte offset): the per-K-tile advance is then one scalar add instead of per-lane 64-bit math.
__device__ __forceinline__ void dma16s(const void* sbase /* wave-uniform */, unsigned voff,
                                       unsigned lds_addr /* wave-uniform */) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %2"
      :
      : "v"(voff), "s"(lds_addr), "s"(sbase)
      : "memory");
}
// ... with the non-temporal hint: rows that are read for the last time (a 16-bit intermediate on its way into the next product)
__device__ __forceinline__ void dma16s_nt(const void* sbase /* wave-uniform */, unsigned voff,
                                          unsigned lds_addr /* wave-uniform */) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %2 nt"
      :
      : "v"(voff), "s"(lds_addr), "s"(sbase)
      : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// LDS byte address of a __shared__ pointer
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// Block barrier for hand-overs through LDS only: this wave's LDS accesses retired + s_barrier.  __syncthreads() is a
// workgroup-scope release / acquire fence as well -- it waits for every global access the wave has in flight (vmcnt(0)):
// row loads that are only needed later, write-through stores whose acknowledgement only the arrival counter needs (round 4:
// a memory round trip per barrier in linear_ws_kernel's K loop and in crmsa_region4_kernel's middle phases).
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define LN_EPS 1e-5f

// ---- optional in-kernel timeline tracing (tools/build_ablation.sh trace -DRRT_TRACE) ------------
// Per wave: up to RRT_TRACE_EV 64-bit shader-clock stamps + HW_ID/XCC_ID, written to a device
// symbol that tools/trace_attn.py reads back.  Compiled out of the product build.
#ifdef RRT_TRACE
#define RRT_TRACE_EV 32
#define RRT_TRACE_WAVES 8192
// one private copy per translation unit (no -fgpu-rdc); each traced file defines its own reader with
// RRT_TRACE_DEFINE_READER(name) -> extern "C" int name(void* host, size_t bytes, int clear)
static __device__ unsigned long long g_rrt_trace[RRT_TRACE_WAVES * RRT_TRACE_EV];
#define RRT_TRACE_DEFINE_READER(NAME)                                                                  \
  extern "C" int NAME(void* host, size_t bytes, int clear) {                                           \
    size_t n = sizeof(unsigned long long) * RRT_TRACE_WAVES * RRT_TRACE_EV;                            \
    if (bytes < n) n = bytes;                                                                          \
    hipError_t e = hipDeviceSynchronize();                                                             \
    if (e == hipSuccess && host)                                                                       \
      e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rrt_trace), n, 0, hipMemcpyDeviceToHost);             \
    if (e == hipSuccess && clear) {                                                                    \
      void* d = nullptr;                                                                               \
      e = hipGetSymbolAddress(&d, HIP_SYMBOL(g_rrt_trace));                                            \
      if (e == hipSuccess) e = hipMemset(d, 0, sizeof(unsigned long long) * RRT_TRACE_WAVES * RRT_TRACE_EV); \
    }                                                                                                  \
    return (int)e;                                                                                     \
  }
struct WaveTrace {
  unsigned long long* p;
  int n;
  __device__ __forceinline__ void init(int wave_linear) {
    p = (wave_linear < RRT_TRACE_WAVES) ? g_rrt_trace + (size_t)wave_linear * RRT_TRACE_EV : nullptr;
    n = 1;
    if (p && (threadIdx.x & 63) == 0) {
      unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
      unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
      p[0] = ((unsigned long long)xcc << 32) | hw;
    }
  }
  __device__ __forceinline__ void mark() {
    if (p && n < RRT_TRACE_EV) {
      unsigned long long t = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0) p[n] = t;
    }
    ++n;
  }
};
#define RRT_TRACE_INIT(w) WaveTrace _tr; _tr.init(w)
#define RRT_TRACE_MARK() _tr.mark()
#else
#define RRT_TRACE_INIT(w)
#define RRT_TRACE_MARK()
#endif
