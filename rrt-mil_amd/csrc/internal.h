// internal.h -- host-side launch functions shared between the kernel files and api.cpp
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/rrt_hip.h"

GridDev to_dev(const rrt_grid& g);

// Tuning hooks (tile shapes, kernel A/B switches) are read from the environment ONLY in a -DRRT_TUNING build
// (tools/build_ablation.sh tune -DRRT_TUNING; tools/sweep_n.py and the RRT_*_CFG sweeps use it).  The product library
// never looks at the process environment: its results and kernel choices depend on its arguments alone.
#ifdef RRT_TUNING
#include <stdlib.h>
inline const char* rrt_tune_env(const char* name) { return getenv(name); }
#else
inline const char* rrt_tune_env(const char*) { return nullptr; }
#endif

// "Raise the dynamic-LDS cap of this kernel" has to happen once per DEVICE (the attribute lives in the device's
// context): a process that drives two GPUs would otherwise fail its first > 64 KiB launch on the second one.
struct OncePerDevice {
  bool seen[64] = {};
  bool first() {
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 63;
    if (seen[d]) return false;
    seen[d] = true;      // benign race: the attribute call is idempotent
    return true;
  }
};

// zero / n_zero: side job of block 0 -- n_zero ints set to 0 (arrival counters of a later kernel of the same forward)
hipError_t launch_ln_partition(const float* x, const float* gamma, const float* beta, float* u,
                               int dim, const GridDev& g, hipStream_t st, int* zero = nullptr, int n_zero = 0);

struct LinearEpilogue {
  int prec;              // MFMA operand precision: 0 f32 (exact), 1 bf16, 2 f16 (fp32 accumulate)
  const float* bias;     // [N] or null
  int q_cols;            // columns [0,q_cols) scaled by q_scale after bias
  float q_scale;
  int act;               // RRT_ACT_* applied last (0 = none)
  // training only (fp32): dropout on (acc + bias) before the residual; thresh = p * 2^32 (0 = off), the kept
  // values are scaled by drop_scale = 1 / (1 - p); mask index = A-row * N + column (common.h rrt_drop_keep)
  unsigned drop_thresh, drop_seed;
  float drop_scale;
  bool drop_on;          // selects the dropout epilogue (drop_thresh may be 0: a pure branch multiplier, stochastic depth)
  // un-partition + residual (used when resid != null): C row = token, A row = slot
  const float* resid;
  GridDev g;
  // side job: block 0 zeroes 64 ints (the CR-MSA region kernel's arrival counters: they must be 0 when it starts,
  // and some kernel earlier in the same forward has to do it -- the workspace is the caller's, uninitialised)
  int* zero64;
  // scheduling hint (rrt_encoder_desc.solo): this GEMM may use whole CUs (small-M GEMMs: split K inside a 16-wave block)
  bool solo;
};
hipError_t launch_linear(const float* A, const float* B, float* C, int M, int N, int K,
                         const LinearEpilogue& ep, hipStream_t st);

// ---- 16-bit operand path of the reduced-precision modes (cast16.hip, linear_f32.hip IN16, rmsa_fused16.hip)
constexpr int CAST16_MAX_JOBS = 2 * RRT_MAX_RMSA_LAYERS + 2;   // + CR-MSA's inner qkv / proj
struct Cast16Jobs {
  const float* src[CAST16_MAX_JOBS];
  uint16_t* dst[CAST16_MAX_JOBS];
  size_t n4[CAST16_MAX_JOBS];       // float4 groups (element count / 4)
  int count;
};
hipError_t launch_cast16(const Cast16Jobs& jobs, int prec, hipStream_t st);
hipError_t launch_ln_partition16(const float* x, const float* gamma, const float* beta, uint16_t* u, int dim,
                                 const GridDev& g, int prec, hipStream_t st, int* zero = nullptr, int n_zero = 0);
// RRT_COMPUTE_F32X3: fp32 as (hi, lo) bf16 pairs, 32-element groups [32 hi | 32 lo] (cast16.hip); jobs.dst = byte images
hipError_t launch_cast_split(const Cast16Jobs& jobs, hipStream_t st);
hipError_t launch_ln_partition_split(const float* x, const float* gamma, const float* beta, void* u, int dim,
                                     const GridDev& g, hipStream_t st);
// C fp32 = A . B^T on split images of A [M, K] and B [N, K] (3 bf16 MFMAs per product: hi.hi + hi.lo + lo.hi)
hipError_t launch_linear_split(const void* Asplit, const void* Bsplit, float* C, int M, int N, int K,
                               const LinearEpilogue& ep, hipStream_t st);
// two regions x one head per block, EPEG stencil on the matrix cores (rmsa_pair16.hip); launch_rmsa_fused16 takes it
// where it applies (>= 8 regions of <= 176 tokens)
bool rmsa_pair16_supported(int n_regions, int P, int D, int heads, int epeg_k);
hipError_t launch_rmsa_pair16(const uint16_t* U, const uint16_t* W, const float* bqkv, const float* pe_w, uint16_t* O,
                              int n_regions, int P, int D, int heads, int epeg_k, int prec, hipStream_t st);
// Round 6: the same launch also runs the layer's out-projection (16-bit operands) + region_reverse + un-pad + residual, as
// rmsa_fused_kernel<.., PROJ> does in fp32: block b runs item b, block b >= lag also the 64-column slab b - lag of a region
// PAIR once the pair's `heads` items have arrived at cnt[pair].  cnt [n_regions / 2] must be zero at launch.
struct PairProj {
  const uint16_t* Wp;    // [D, D] 16-bit
  const float* bias;     // [D] or null
  const float* resid;    // [L, D] token order
  float* out;            // [L, D]
  int* cnt;              // [n_regions / 2] arrival counters (zero at launch)
  int* zero64;           // side job of block 0 (LinearEpilogue.zero64), may be null
  int lag;               // blocks between an item and the slab of the same index (multiple of 8, >= 8 * heads); 0: the launcher's rule
  int n_items;           // heads * n_regions / 2 (filled by the launcher)
  GridDev g;
  int* err;              // the process's hand-over error word (null: the launcher's), as FusedProj
  int spin_limit;        // 0: 2^22
  int wait_for;          // 0: heads
};
// sixteen-wave form with an even number of row tiles (regions of 65..96 and 113..128 tokens), whole groups of eight pairs,
// at least two rounds of items
bool rmsa_pair16_proj_supported(int n_regions, int P, int D, int heads, int epeg_k);
hipError_t launch_rmsa_pair16_proj(const uint16_t* U, const uint16_t* W, const float* bqkv, const float* pe_w, uint16_t* O,
                                   int n_regions, int P, int D, int heads, int epeg_k, int prec, const PairProj& proj,
                                   hipStream_t st);
bool rmsa_fused_x3_supported(int P, int D, int heads, int epeg_k);
hipError_t launch_rmsa_fused_x3(const void* Usplit, const void* Wsplit, const float* bqkv, const float* pe_w,
                                void* Osplit, int n_regions, int P, int D, int heads, int epeg_k, hipStream_t st);
hipError_t launch_linear16(const void* A16, const void* B16, float* C, int M, int N, int K, const LinearEpilogue& ep,
                           hipStream_t st);
// fused R-MSA core on 16-bit operands: U16 [n_regions*P, D], Wqkv16 [3D, D] -> O16 [n_regions*P, D]
bool rmsa_fused16_supported(int P, int D, int heads, int epeg_k);
hipError_t launch_rmsa_fused16(const uint16_t* U16, const uint16_t* Wqkv16, const float* bqkv, const float* pe_w,
                               uint16_t* O16, int n_regions, int P, int D, int heads, int epeg_k, int prec,
                               hipStream_t st);

hipError_t launch_region_attention(const float* qkv, const float* pe_w, float* o, int n_regions,
                                   int P, int dim, int heads, int epeg_k, hipStream_t st);

// fused qkv projection + EPEG + attention per (region, head) (rmsa_fused.hip); P in (112,144], head dim 64
bool rmsa_fused_supported(int P, int D, int heads, int epeg_k);
bool rmsa_fused_supported_rows(long n_rows, int D);
// The out-projection as a later phase of the SAME launch (fp32, inference): block b < heads * n_regions runs the
// (region, head) item b; block b >= lag also runs the projection slab b - lag -- 64 output columns of one region,
// C = O_r . Wp[64 c .., :]^T + bias, region_reverse + un-pad + residual (rmsa.py:131, :41-54, :227-228; rrt.py:125) --
// once that region's `heads` items have arrived at cnt[region].  cnt [n_regions] must be zero at launch.
struct FusedProj {
  const float* Wp;       // [D, D]
  const float* bias;     // [D] or null
  const float* resid;    // [L, D] token order
  float* out;            // [L, D]
  int* cnt;              // [n_regions] arrival counters (zero at launch)
  int* zero64;           // side job of block 0 (LinearEpilogue.zero64), may be null
  int lag;               // blocks between an item and the slab of the same index (multiple of 8, >= 8 * heads); 0: the launcher's rule
  int n_items;           // heads * n_regions
  GridDev g;
  // the bounded hand-over wait (proj_slab): a slab waits until cnt[region] >= wait_for (0: heads) for at most spin_limit
  // sleeps of ~0.2 us (0: 2^22); a slab that gives up stores 1 + region to *err (system scope; may be null) and writes nothing
  int* err;
  int spin_limit;
  int wait_for;
  // CR-MSA's first pass as a by-product of the slab (null: off): per (token, 64-column slab) the record
  // (mean, centred sum of squares, d_0 .. d_k-1), d_n = sum_c x1[c] gamma[c] phi[c, n] -- part [L][n_slabs][2 + k], read by
  // launch_crmsa_combine_parts.  ln_g = the CR-MSA TransLayer's LayerNorm weight [D], phi [D, k]
  float* part;
  const float* ln_g;
  const float* phi;
  int k;
  int n_slabs;           // D / 64 (filled by the launcher)
};
// The process's hand-over error word (pinned, device-mapped host memory; created on first use): device pointer for the
// kernels, and the host's view of it -- 0, or 1 + the region whose projection slab gave up waiting (rrt_device_error)
int* handover_err_device();
int handover_err_peek(bool clear);
bool rmsa_fused_proj_supported(int n_regions, int P, int D, int heads, int epeg_k, int prec);
hipError_t launch_rmsa_fused(const float* U, const float* Wqkv, const float* bqkv, const float* pe_w,
                             float* O, int n_regions, int P, int D, int heads, int epeg_k, int prec,
                             hipStream_t st, float* stash = nullptr,    // stash: [rows, 3 D] q | k | v for the backward
                             const FusedProj* proj = nullptr);           // lag / n_items are filled by the launcher

// EPEG ablations (epeg_variants.hip): 2-D 'attn' EPEG over the score map; value EPEG over v's token image
size_t attn_scoremap_lds(int P, int k);
// floats of global scratch for `maps` [P, P] maps per (region, head) when they do not fit the LDS (0: they fit)
size_t attn_scoremap_scratch_floats(int n_regions, int P, int heads, int k, int maps);
hipError_t launch_attn_scoremap(const float* qkv, const float* pe_w, float* o, float* smap, int n_regions, int P, int dim,
                                int heads, int k, hipStream_t st);
hipError_t launch_value_pe(const float* qkv, const float* w, const float* bias, float* pe, int n_regions, int P, int s,
                           int dim, int heads, int k, int two_d, hipStream_t st);
hipError_t launch_add_cols(float* dst, const float* src, size_t rows, int dim, int ld, hipStream_t st);
// backward of the ablations (epeg_variants.hip)
hipError_t launch_value_pe_backward(const float* dpe, const float* qkv, const float* vsub, const float* w, float* dqkv, float* dw,
                                    float* db, int n_regions, int P, int s, int dim, int heads, int k, int two_d,
                                    hipStream_t st);
hipError_t launch_copy_cols(float* dst, const float* src, size_t rows, int dim, int ld, int off, hipStream_t st);
hipError_t launch_sub(float* dst, const float* a, const float* b, size_t n, hipStream_t st);
// C [M, N] = A [M, Kd] . W [Kd, N] (row-major W), any small Kd
hipError_t launch_small_k_matmul(const float* A, const float* W, float* Cm, int M, int N, int Kd, hipStream_t st);
hipError_t launch_attn_scoremap_backward(const float* qkv, const float* pe_w, const float* dO, float* dqkv, float* dpe_w,
                                         float* scratch, int n_regions, int P, int dim, int heads, int k, hipStream_t st);

hipError_t launch_crmsa_logits(const float* x1, const float* gamma, const float* beta,
                               const float* phi, float* mean_rstd, float* logits, int dim, int k,
                               const GridDev& g8, hipStream_t st);
bool crmsa_region_supported(int dim, int k, const GridDev& g8);
bool crmsa_region_enabled();
bool crmsa_region4_supported(int dim, int k, const GridDev& g8);
size_t crmsa_region4_scratch_floats(const GridDev& g8, int k);
// rep16 (may be null): the representatives once more as 16-bit values (prec16 = 1 bf16 / 2 fp16), the A operand of the
// inner MSA's qkv projection in the reduced-precision modes
hipError_t launch_crmsa_region4(const float* x1, const float* gamma, const float* beta, const float* phi,
                                float* mean_rstd, float* logits, float* wdisp, float* rep, uint16_t* rep16, int prec16,
                                float* part_g, int* counters, int k, const GridDev& g8, hipStream_t st);
hipError_t launch_crmsa_region(const float* x1, const float* gamma, const float* beta, const float* phi,
                               float* mean_rstd, float* logits, float* wdisp, float* rep, int k, const GridDev& g8,
                               hipStream_t st, uint16_t* rep16 = nullptr, int prec16 = 0);
// regions of more than 144 tokens: four blocks per region that stream their rows (one pass over x1); scratch and counters as
// launch_crmsa_region4
bool crmsa_stream4_supported(int dim, int k, const GridDev& g8);
hipError_t launch_crmsa_stream4(const float* x1, const float* gamma, const float* beta, const float* phi,
                                float* mean_rstd, float* logits, float* wdisp, float* rep, uint16_t* rep16, int prec16,
                                float* part_g, int* counters, int k, const GridDev& g8, hipStream_t st);
hipError_t launch_crmsa_combine(const float* x1, const float* gamma, const float* beta,
                                const float* mean_rstd, const float* logits, float* wdisp,
                                float* rep, uint16_t* rep16, int prec16, int dim, int k, const GridDev& g8, hipStream_t st);
// combine from the row records the last R-MSA layer's projection slabs left (FusedProj.part): x1 read once, no LayerNorm
// arithmetic, no hand-over between blocks (crmsa.hip, crmsa_combine_parts_kernel)
bool crmsa_combine_parts_supported(int dim, int k, const GridDev& g8);
size_t crmsa_parts_floats(long n_tokens, int dim, int k);
hipError_t launch_crmsa_combine_parts(const float* x1, const float* part, const float* gamma, const float* beta,
                                      const float* phi, float* wdisp, float* rep, uint16_t* rep16, int prec16, int dim,
                                      int k, const GridDev& g8, hipStream_t st);
// mean_rstd == nullptr: x1 is LN(x1) already, region-major [Np8, dim] (crmsa_mlp path)
hipError_t launch_crmsa_mlp_logits(const float* hid, const float* w2, float* logits, int rows, int hdim,
                                   int k, hipStream_t st);
hipError_t launch_crmsa_dispatch_ln(const float* x1, const float* x0, const float* wdisp,
                                    const float* rep2, const float* gamma,
                                    const float* beta, float* y, int dim, int k, const GridDev& g8,
                                    hipStream_t st, uint16_t* y16 = nullptr, int prec16 = 0);   // y16: the rows also in 16 bits
hipError_t launch_layernorm(const float* x1, const float* x0, const float* gamma,
                            const float* beta, float* y, int L, int dim, hipStream_t st);

// ABMIL attention pooling behind the encoder (modules/datten.py:28-38,69-83) + predictor (rrt.py:241):
// per-chunk scores and online-softmax partials, then one merge block
constexpr int POOL_CHUNK = 32;   // tokens per partial
hipError_t launch_pool_partial(const float* y, const float* hid_a, const float* hid_b, const float* wc,
                               const float* bc, float* a_raw, float* part, int N, int dim, int hid,
                               hipStream_t st);
hipError_t launch_pool_merge(const float* part, const float* a_raw, const float* pred_w, const float* pred_b,
                             float* pooled, float* logits, float* attn, int no_norm, int N, int dim,
                             int n_classes, hipStream_t st);

// backward of the pooling (mil_pool.hip): dy [N, dim], dhid_a / dhid_b [N, hid], dwcb [hid + 4] = d wc | d bc
size_t pool_backward_part_floats(int N, int hid);
hipError_t launch_pool_backward(const float* y, const float* hid_a, const float* hid_b, const float* wc, const float* attn,
                                const float* pooled, const float* d_pooled, const float* d_attn, const float* d_raw,
                                const float* c_ext, float* dy, float* dhid_a, float* dhid_b, float* dwcb, float* part, int N,
                                int dim, int hid, hipStream_t st);

// ---- row f2 building blocks (backward)
// nn.Linear backward: dX = dY W (forward GEMM on a transposed W), dW = dY^T X (split-K TN kernel), db = colsum(dY)
size_t linear_bwd_workspace(int M, int N, int K);
hipError_t launch_linear_backward(const float* dY, const float* X, const float* W, float* dX, float* dW, float* db,
                                  int M, int N, int K, int prec, void* ws, hipStream_t st, const float* WT_ready = nullptr);
hipError_t launch_transpose(const float* in, float* out, int R, int C, hipStream_t st);
constexpr int TRANSPOSE_MAX_JOBS = 2 * RRT_MAX_RMSA_LAYERS + 2;
struct TransposeJobs {           // out[j] [C, R] = in[j] [R, C]^T, one launch (blk0: filled by the launcher)
  const float* in[TRANSPOSE_MAX_JOBS];
  float* out[TRANSPOSE_MAX_JOBS];
  int R[TRANSPOSE_MAX_JOBS], C[TRANSPOSE_MAX_JOBS], blk0[TRANSPOSE_MAX_JOBS];
  int n;
};
hipError_t launch_transpose_batch(TransposeJobs& jobs, hipStream_t st);
hipError_t launch_reduce_partials(const float* part, float* out, int S, size_t n, hipStream_t st);
// Parameter-gradient reductions nothing downstream reads (LayerNorm's d gamma | d beta, the EPEG taps, CR-MSA's norm / phi rows)
// can leave the backward's dependent chain: a stage appends its job here instead of launching its reduce, and ONE launch at the
// end of the backward sums them all (each stage then needs partial buffers of its own until that launch).
constexpr int REDUCE_MAX_JOBS = 16;
struct ReduceJobs {
  const float* part[REDUCE_MAX_JOBS];
  float* out[REDUCE_MAX_JOBS];
  float* out_tr[REDUCE_MAX_JOBS];          // optional transposed tail (launch_reduce_partials_scatter), else null
  unsigned long long n[REDUCE_MAX_JOBS], split[REDUCE_MAX_JOBS];
  int S[REDUCE_MAX_JOBS], tr_dim[REDUCE_MAX_JOBS], tr_k[REDUCE_MAX_JOBS];
  unsigned blk0[REDUCE_MAX_JOBS + 1];      // filled by the launcher
  int count;
};
// defer == null or full: launches now.  (out_tr == null: plain)
hipError_t reduce_or_defer(ReduceJobs* defer, const float* part, float* out, int S, size_t n, hipStream_t st,
                           size_t split = 0, float* out_tr = nullptr, int tr_dim = 0, int tr_k = 0);
hipError_t launch_reduce_jobs(ReduceJobs& jobs, hipStream_t st);
hipError_t launch_reduce_partials_scatter(const float* part, float* out, int S, size_t n, size_t split, float* out_tr,
                                          int tr_dim, int tr_k, hipStream_t st);
hipError_t launch_gemm_tn(const float* dY, const float* X, float* dW, float* scratch, int M, int N, int K,
                          hipStream_t st);
hipError_t launch_colsum(const float* Y, float* out, float* scratch, int M, int N, hipStream_t st);
// LayerNorm backward; dgb = [2, dim] (dgamma, dbeta); g != null: dy rows are region-major slots; add: residual grad
size_t ln_bwd_workspace(int dim);
hipError_t launch_ln_backward(const float* dy, const float* x, const float* gamma, const float* add, float* dx,
                              float* dgb, float* part, int L, int dim, const GridDev* g, hipStream_t st,
                              ReduceJobs* defer = nullptr);
// region attention backward (attn_bwd.hip): head dim 64, P <= 144.  dqkv: gradient w.r.t. the qkv linear's raw
// output [n_regions*P, 3D]; dpe [heads, epeg_k] or null; dpe_part: attn_bwd_workspace bytes
bool attn_bwd_supported(int P, int D, int heads, int epeg_k);
size_t attn_bwd_workspace(int n_regions, int P, int D, int heads, int epeg_k);
hipError_t launch_attention_backward(const float* qkv, const float* pe_w, const float* O, const float* dO,
                                     float* dqkv, float* dpe, float* dpe_part, int n_regions, int P, int D,
                                     int heads, int epeg_k, hipStream_t st, ReduceJobs* defer = nullptr,
                                     float* defer_part = nullptr);   // defer_part [n_regions, heads * epeg_k]: the taps' partials
// dst [Np, dim] region-major <- src [L, dim]; pads 0; optional dropout mask (thresh != 0) regenerated per element
hipError_t launch_partition_rows(const float* src, float* dst, int dim, const GridDev& g, unsigned drop_thresh,
                                 unsigned drop_seed, float drop_scale, hipStream_t st);
hipError_t launch_apply_drop_mask(float* buf, int rows, int cols, unsigned drop_thresh, unsigned drop_seed,
                                  float drop_scale, hipStream_t st);
// CR-MSA backward stages (crmsa_bwd.hip)
hipError_t launch_crmsa_tokdot(const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                               const float* vec, float* out, int dim, int k, const GridDev& g, hipStream_t st);
hipError_t launch_crmsa_wsum(const float* X, const float* W, float* out, int dim, int k, const GridDev& g,
                             hipStream_t st);
hipError_t launch_crmsa_bwd_region(const float* lg, const float* dC, const float* dWd, float* dlg, float* Cw, int k,
                                   const GridDev& g, hipStream_t st);
size_t crmsa_bwd_dx_workspace(int dim, int k);
hipError_t launch_crmsa_bwd_dx(const float* x1, const float* dx2, const float* mean_rstd, const float* gamma,
                               const float* beta, const float* phi, const float* Cw, const float* dlg,
                               const float* drep, float* dx1, float* out_rows, float* part, int dim, int k,
                               const GridDev& g, bool mlp, hipStream_t st, float* dphi_t = nullptr,
                               ReduceJobs* defer = nullptr);
hipError_t launch_crmsa_mlp_bwd_hidden(const float* hid, const float* dlg, const float* w2, float* th, float* dhid,
                                       size_t rows, int hdim, int k, hipStream_t st);
// FFN activation as a pass (n % 4 == 0): h = act(hpre) ; dh *= act'(hpre)   (act = RRT_ACT_GELU / RRT_ACT_RELU)
// (thresh != 0: with the Mlp's dropout mask on the activation's output / its adjoint)
hipError_t launch_act_forward(const float* hpre, float* h, size_t n, int act, unsigned thresh, unsigned seed,
                              float scale, hipStream_t st);
hipError_t launch_act_backward(float* dh, const float* hpre, size_t n, int act, unsigned thresh, unsigned seed,
                               float scale, hipStream_t st);
hipError_t launch_copy_drop_mask(const float* src, float* dst, size_t n, unsigned drop_thresh, unsigned drop_seed,
                                 float drop_scale, hipStream_t st);
// PEG / PPEG positional encoders (peg.hip): y [N, C] = stencil over the wrapped H x H token grid
hipError_t launch_peg(const float* x, const float* const* w, const float* const* b, float* y, int N, int C, int k,
                      int conv_1d, int ppeg, hipStream_t st);
size_t peg_bwd_workspace(int N, int C, int k, int ppeg);
hipError_t launch_peg_backward(const float* x, const float* dy, const float* const* w, float* dx, float* const* dw,
                               float* const* db, int N, int C, int k, int conv_1d, int ppeg, void* ws, hipStream_t st);
