// inner_msa.hip -- CR-MSA's inner MSA over the 64 k region representatives in ONE launch (exact fp32).
//
// Replaces InnerAttention.forward on (k, 64, D) without EPEG, modules/rmsa.py:322 -> :100-131:
//     qkv = Linear(D, 3D)(rep) ; q *= hd^-0.5 ; A = softmax(q k^T) ; O = A v ; rep2 = Linear(D, D)(O)
// Until round 5 these were three launches (split-K qkv GEMM 7.7 us, 64-token attention 5.9 us, split-K proj GEMM 5.6 us at
// k = 3): 0.43 GFLOP, 19 us.  None of them is short of anything but time to START: a kernel that is one memory round trip
// and a hundred MFMAs costs ~5.5 us on this chip (dispatch, first wave, kernarg + first load, store, end-of-kernel release),
// so three dependent launches cannot go under ~16 us whatever their kernels do.  Here the three stages are three ROLES of
// the blocks of one launch, handed over through arrival counters like the out-projection slabs of rmsa_fused_kernel:
//   role A (blocks 0 ..): one 32 x 64 tile of qkv for one QUARTER of K (four K splits: 64 MFMAs per wave, every operand
//          fragment requested straight into registers, no LDS, no K loop) -> partial tile, write-through, then the
//          (n, head) item's counter.  24 blocks per (representative row n, head): {q, k, v} x 2 row tiles x 4 splits.
//   role B (next k * heads blocks): item (n, head).  Waits for its 24 producers (blocks with LOWER indices, dispatched
//          earlier: the PROJ argument), sums the four partials of K and V (+ bias) into LDS tiles, Q (+ bias, * scale)
//          straight into fragment registers, then the 64 x 64 attention of region_attn64_kernel from LDS; O leaves
//          write-through, then the row's counter.
//   role C (last blocks): one 16 x 32 tile of the out-projection; wave w multiplies K quarter w (the weights are requested
//          BEFORE the wait: they do not depend on anybody), the four partials meet in LDS, + bias, plain stores.
// Every role fits next to a block of another bag's fused R-MSA kernel (<= 128 VGPRs, 32 KiB LDS, 4 waves): waiting blocks
// cost a wave slot, not a CU.  Results do not depend on rrt_encoder_desc.solo (the split-K kernels' summation order did).
// Summation order: K quarters in ascending order, inside a quarter the MFMA order of the fragments -- fixed, bit-reproducible.
#include "internal.h"

namespace {

constexpr int HD = 64;
constexpr int KS = 4;                  // K splits of the qkv projection (role A) and K quarters of the out-projection (role C)
constexpr float NEG_BIG = -3.0e38f;

// write-through store: the line reaches memory (other XCDs' L2s are not coherent inside a launch).  Inline asm, so the
// compiler's hazard recogniser does not know it is a VMEM store of more than 64 bits: the wait states such a store needs
// before a VALU instruction may overwrite its data registers are part of the asm (found as components 0, 1 of lanes
// lr >= 12 of ONE of four stores holding the next value: the registers were reused for the following product)
__device__ __forceinline__ void st_wt(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

struct InnerArgs {
  const float* rep;      // [k * 64, D]
  const float* Wqkv;     // [3 D, D]
  const float* bqkv;     // [3 D] or null
  const float* Wp;       // [D, D]
  const float* bp;       // [D] or null
  float* qkv_part;       // [KS][k * 64][3 D]
  float* o_buf;          // [k * 64][D]
  float* out;            // [k * 64][D]
  int* cnt;              // [k * heads] item counters, then [k] row counters; zero at launch
  int* err;              // hand-over error word (pinned host memory) or null
  int k, heads;
  int spin_limit;
  float q_scale;
};

// bounded wait of a whole block on *c >= want (tid 0 polls); false: gave up (error word raised)
__device__ __forceinline__ bool wait_counter(const int* c, const int want, const InnerArgs& a, int* s_flag, const int code) {
  if (threadIdx.x == 0) {
    int spins = 0, bad = 0;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > a.spin_limit) { bad = 1; break; }
    }
    if (bad && a.err != nullptr) __hip_atomic_store(a.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *s_flag = bad;
  }
  __syncthreads();
  return *s_flag == 0;
}

template <int D>
__global__ __launch_bounds__(256, 4) void inner_msa_kernel(const InnerArgs a) {
  static_assert(D % (32 * KS) == 0, "K quarters are whole 32-wide tiles");
  constexpr int KQ = D / KS;                       // K elements per quarter (128 at D = 512)
  constexpr int NJ = KQ / 16;                      // float4 fragment slots per row and quarter (8)
  constexpr int LDQ = 3 * D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int heads = a.heads, M = a.k * 64;
  const int NA = a.k * heads * 6 * KS, NB = a.k * heads;
  const int b = (int)blockIdx.x;
  RRT_TRACE_INIT(blockIdx.x * 4 + wave);
  RRT_TRACE_MARK();                                 // [1] entry

  if (b < NA) {
    // ================================================================== role A: a K quarter of one 32 x 64 qkv tile
    const int item = b / (6 * KS), w_ = b - item * (6 * KS);
    const int n = item / heads, head = item - n * heads;
    const int c = w_ / (2 * KS), rt = (w_ / KS) & 1, s = w_ & (KS - 1);
    const int m0 = n * 64 + rt * 32;
    const int ncol0 = c * D + head * HD;           // the tile's first output column (q | k | v of this head)
    const float* arow = a.rep + (size_t)(m0 + lr) * D + s * KQ + 4 * lg;
    const float* brow = a.Wqkv + (size_t)(ncol0 + wave * 16 + lr) * D + s * KQ + 4 * lg;
    float4 af[2][NJ], bf[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      bf[j] = *(const float4*)(brow + 16 * j);
      af[0][j] = *(const float4*)(arow + 16 * j);
      af[1][j] = *(const float4*)(arow + (size_t)16 * D + 16 * j);
    }
    RRT_TRACE_MARK();                               // A [2] fragments requested
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].x, af[i][j].x, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].y, af[i][j].y, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].z, af[i][j].z, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].w, af[i][j].w, acc[i], 0, 0, 0);
      }
    }
    // acc[i][r] = C[m0 + 16 i + lr][ncol0 + 16 wave + 4 lg + r]
    RRT_TRACE_MARK();                               // A [3] MFMAs issued (fragments landed)
    float* dst = a.qkv_part + ((size_t)s * M + m0 + lr) * LDQ + ncol0 + wave * 16 + 4 * lg;
    // The accumulators go from the matrix pipe straight into an inline-asm store: the compiler's hazard recogniser does not
    // see a VMEM read of an MFMA result there and leaves out the wait states an 8-pass MFMA needs before its destination
    // may be read (found as component 2 of lanes lr < 4 -- the rows the last pass writes -- holding the previous value).
    // A VALU copy makes the dependency one the compiler knows about.
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 v = acc[i];
      asm volatile("s_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3"
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
      st_wt(dst + (size_t)16 * i * LDQ, v);
    }
    wait_vm0();                                    // this thread's partial is in memory ...
    RRT_TRACE_MARK();                               // A [4] partial in memory
    __syncthreads();                               // ... and everybody's
    if (tid == 0) __hip_atomic_fetch_add(a.cnt + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    RRT_TRACE_MARK();                               // A [5] counted
    return;
  }

  if (b < NA + NB) {
    // ================================================================== role B: attention of one (n, head) item
    const int item = b - NA;
    const int n = item / heads, head = item - n * heads;
    float* Ks = (float*)smem;                      // [64][64] XOR-swizzled by row (16-byte slots)
    float* Vs = Ks + 64 * HD;
    if (!wait_counter(a.cnt + item, 6 * KS, a, &s_flag, 0x10000 + item)) return;
    RRT_TRACE_MARK();                               // B [2] producers arrived
    if (tid == 0) __hip_atomic_store(a.cnt + item, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next forward
    const float* pbase = a.qkv_part + (size_t)(n * 64) * LDQ + head * HD;
    const size_t sstride = (size_t)M * LDQ;
    // K and V tiles: thread = (row r0 + 16 u, 16-byte slot); the four partials summed in ascending K order, + bias
    {
      const int slot = tid & 15, r0 = tid >> 4;
      const float4 bk = a.bqkv ? *(const float4*)(a.bqkv + D + head * HD + 4 * slot) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 bv = a.bqkv ? *(const float4*)(a.bqkv + 2 * D + head * HD + 4 * slot) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int half = 0; half < 2; ++half) {         // two rounds of 2 x 2 x KS loads: 64 VGPRs of requests in flight
        float4 pk[2][KS], pv[2][KS];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const float* p = pbase + s * sstride + (size_t)(r0 + 16 * (2 * half + u)) * LDQ + 4 * slot;
            pk[u][s] = *(const float4*)(p + D);
            pv[u][s] = *(const float4*)(p + 2 * D);
          }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = r0 + 16 * (2 * half + u);
          float4 k4 = pk[u][0], v4 = pv[u][0];
#pragma unroll
          for (int s = 1; s < KS; ++s) {
            k4.x += pk[u][s].x; k4.y += pk[u][s].y; k4.z += pk[u][s].z; k4.w += pk[u][s].w;
            v4.x += pv[u][s].x; v4.y += pv[u][s].y; v4.z += pv[u][s].z; v4.w += pv[u][s].w;
          }
          k4.x += bk.x; k4.y += bk.y; k4.z += bk.z; k4.w += bk.w;
          v4.x += bv.x; v4.y += bv.y; v4.z += bv.z; v4.w += bv.w;
          *(float4*)(Ks + row * HD + ((slot ^ (row & 15)) << 2)) = k4;
          *(float4*)(Vs + row * HD + ((slot ^ (row & 15)) << 2)) = v4;
        }
      }
    }
    // Q fragments of this wave's 16 queries, straight into the MFMA operand layout: qf[i] = q[16 wave + lr][16 i + 4 lg ..]
    float4 qf[4];
    {
      const float* qp = pbase + (size_t)(16 * wave + lr) * LDQ + 4 * lg;
      float4 pq[4][KS];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < KS; ++s) pq[i][s] = *(const float4*)(qp + s * sstride + 16 * i);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 bq = a.bqkv ? *(const float4*)(a.bqkv + head * HD + 16 * i + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 q4 = pq[i][0];
#pragma unroll
        for (int s = 1; s < KS; ++s) { q4.x += pq[i][s].x; q4.y += pq[i][s].y; q4.z += pq[i][s].z; q4.w += pq[i][s].w; }
        qf[i] = make_float4((q4.x + bq.x) * a.q_scale, (q4.y + bq.y) * a.q_scale, (q4.z + bq.z) * a.q_scale, (q4.w + bq.w) * a.q_scale);
      }
    }
    __syncthreads();                               // K, V tiles complete
    RRT_TRACE_MARK();                               // B [3] K, V tiles in LDS, Q in registers
    // S^T = K q^T: st[kt][j] = score(query 16 wave + lr, key 16 kt + 4 lg + j)   (region_attn64_kernel's arithmetic)
    f32x4 st[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = 16 * kt + lr;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 kf = *(const float4*)(Ks + row * HD + (((4 * i + lg) ^ (row & 15)) << 2));
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[i].x, st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[i].y, st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[i].z, st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[i].w, st[kt], 0, 0, 0);
      }
    }
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
    RRT_TRACE_MARK();                               // B [4] scores + softmax
    // O^T = V^T P^T: a = V[key 16 kt + 4 lg + j][d 16 dt + lr], b = this lane's probability of that key (query lr)
    f32x4 ot[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 16 * kt + 4 * lg + j;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int d = 16 * dt + lr;
          const float vv = Vs[row * HD + ((((d >> 2) ^ (row & 15)) << 2) | (d & 3))];
          ot[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, st[kt][j], ot[dt], 0, 0, 0);
        }
      }
    float* orow = a.o_buf + (size_t)(n * 64 + 16 * wave + lr) * D + head * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      st_wt(orow + 16 * dt + 4 * lg, (f32x4){ot[dt][0] * inv, ot[dt][1] * inv, ot[dt][2] * inv, ot[dt][3] * inv});
    RRT_TRACE_MARK();                               // B [5] O stores issued
    wait_vm0();
    __syncthreads();
    RRT_TRACE_MARK();                               // B [6] O in memory
    if (tid == 0) __hip_atomic_fetch_add(a.cnt + a.k * heads + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  // ==================================================================== role C: a 16 x 32 tile of the out-projection
  {
    const int cidx = b - NA - NB;
    const int tiles_n = D / 32;                    // 16 column tiles of 32
    const int per_n = 4 * tiles_n;                 // tiles per representative row n
    const int n = cidx / per_n, w_ = cidx - n * per_n;
    const int rt = w_ / tiles_n, ct = w_ - rt * tiles_n;
    const int m0 = n * 64 + rt * 16, n0 = ct * 32;
    f32x4* red = (f32x4*)smem;                     // [4 waves][2 column tiles][64 lanes]
    // wave = K quarter: the weights' fragments do not depend on anybody -- requested before the wait
    float4 bf[2][NJ];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const float* brow = a.Wp + (size_t)(n0 + 16 * jt + lr) * D + wave * KQ + 4 * lg;
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[jt][j] = *(const float4*)(brow + 16 * j);
    }
    const int* crow = a.cnt + a.k * heads + n;
    if (!wait_counter(crow, heads, a, &s_flag, 0x20000 + n)) return;
    RRT_TRACE_MARK();                               // C [2] row arrived
    float4 af[NJ];
    {
      const float* arow = a.o_buf + (size_t)(m0 + lr) * D + wave * KQ + 4 * lg;
#pragma unroll
      for (int j = 0; j < NJ; ++j) af[j] = *(const float4*)(arow + 16 * j);
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[jt][j].x, af[j].x, acc[jt], 0, 0, 0);
        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[jt][j].y, af[j].y, acc[jt], 0, 0, 0);
        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[jt][j].z, af[j].z, acc[jt], 0, 0, 0);
        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[jt][j].w, af[j].w, acc[jt], 0, 0, 0);
      }
    red[(wave * 2 + 0) * 64 + lane] = acc[0];
    red[(wave * 2 + 1) * 64 + lane] = acc[1];
    RRT_TRACE_MARK();                               // C [3] partial tile (O fragments landed)
    __syncthreads();
    // (the row counters stay at `heads` when the launch ends: the caller zeroes the counter words before every launch --
    // encoder_forward through the side job of an R-MSA kernel of the same forward, LinearEpilogue.zero64 / FusedProj.zero64)
    if (wave < 2) {
      const int jt = wave;
      f32x4 o = red[(0 * 2 + jt) * 64 + lane];
#pragma unroll
      for (int s = 1; s < KS; ++s) {               // ascending K order: bit-reproducible
        const f32x4 p = red[(s * 2 + jt) * 64 + lane];
        o[0] += p[0]; o[1] += p[1]; o[2] += p[2]; o[3] += p[3];
      }
      const int col = n0 + 16 * jt + 4 * lg;
      const float4 bb = a.bp ? *(const float4*)(a.bp + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      *(float4*)(a.out + (size_t)(m0 + lr) * D + col) = make_float4(o[0] + bb.x, o[1] + bb.y, o[2] + bb.z, o[3] + bb.w);
    }
    RRT_TRACE_MARK();                               // C [4] stored
  }
}

}  // namespace

#ifdef RRT_TRACE
RRT_TRACE_DEFINE_READER(rrt_debug_trace_inner)
#endif

// fp32 exact, dim 512 (head dim 64, crmsa_heads = 8), k <= 7 (the 9 k counters live in the 64 ints an R-MSA kernel of the
// same forward zeroes)
bool inner_msa_fused_supported(int dim, int heads, int k) {
  static const bool off = rrt_tune_env("RRT_NO_INNER_FUSED") != nullptr;
  return !off && dim == 512 && heads * HD == dim && k >= 1 && 9 * k <= 64;
}
size_t inner_msa_scratch_floats(int dim, int k) { return (size_t)KS * k * 64 * 3 * dim + (size_t)k * 64 * dim; }

hipError_t launch_inner_msa(const float* rep, const float* qkv_w, const float* qkv_b, const float* proj_w, const float* proj_b,
                            float* out, float* scratch, int* counters, int dim, int heads, int k, hipStream_t st) {
  if (!inner_msa_fused_supported(dim, heads, k)) return hipErrorInvalidValue;
  InnerArgs a{};
  a.rep = rep; a.Wqkv = qkv_w; a.bqkv = qkv_b; a.Wp = proj_w; a.bp = proj_b;
  a.qkv_part = scratch;
  a.o_buf = scratch + (size_t)KS * k * 64 * 3 * dim;
  a.out = out;
  a.cnt = counters;
  a.err = handover_err_device();
  a.k = k; a.heads = heads;
  a.spin_limit = 1 << 22;
  a.q_scale = 1.0f / sqrtf((float)HD);
  const int NA = k * heads * 6 * KS, NB = k * heads, NC = k * 4 * (dim / 32);
  const size_t lds = 2 * 64 * HD * 4;              // role B's K and V tiles (role C's partials: 8 KiB of the same)
  inner_msa_kernel<512><<<dim3(NA + NB + NC), dim3(256), lds, st>>>(a);
  return hipGetLastError();
}
