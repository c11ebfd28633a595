/* rrt_hip.h -- C ABI of librrt_hip.so: the MI355X (gfx950) RRTEncoder forward path.
 *
 * The reference (DearCaat/RRT-MIL) has no FFI: its boundary for this path is the
 * Python class modules/rrt.py:133-202 `RRTEncoder(nn.Module)`.  These entry points
 * are what that class's forward binds to once its internals are replaced
 * (INTEGRATION.md shows the ctypes stub); each one cites the reference lines it
 * replaces.  Plain pointers and sizes only, no torch types.  All pointers are
 * DEVICE pointers (HIP) unless marked host; `stream` is a hipStream_t passed as
 * void*.  Nothing allocates: the caller owns a workspace sized by
 * rrt_encoder_workspace_size().  Every call is asynchronous on `stream`,
 * re-entrant and stateless.
 *
 * Return value: 0 = ok; <0 = RRT_E_* (unsupported/invalid, nothing launched);
 * >0 = hipError_t from a launch.
 */
#ifndef RRT_HIP_H
#define RRT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_ABI_VERSION 27
#define RRT_MAX_RMSA_LAYERS 8
#define RRT_MAX_CRMSA_K 8

enum {
  RRT_OK = 0,
  RRT_E_INVALID = -1,       /* null pointer / non-positive size */
  RRT_E_UNSUPPORTED = -2,   /* configuration outside the HIP path (see rrt_strerror) */
  RRT_E_WORKSPACE = -3,     /* workspace too small */
  RRT_E_HANDOVER = -4       /* an earlier merged R-MSA launch gave up its in-launch hand-over wait (rrt_device_error) */
};

/* Arithmetic of the nn.Linear layers (qkv / proj of R-MSA and CR-MSA, MLP phi).  F32 is exact
 * fp32 on the fp32 matrix cores (the default; what the <=1e-3 fp32 parity claim is made on).
 * BF16 / F16 round the two MFMA operands to bf16 / fp16 and accumulate in fp32 -- the
 * autocast-class numerics of the reference's --amp path (main.py:101-102,439); LayerNorm,
 * softmax statistics, residuals and the residual stream in HBM stay fp32 in every mode. */
enum { RRT_COMPUTE_F32 = 0, RRT_COMPUTE_BF16 = 1, RRT_COMPUTE_F16 = 2, RRT_COMPUTE_F32X3 = 3 };
/* F32X3 (inference): the qkv and proj GEMMs of the R-MSA layers -- 84 % of the FLOPs -- EMULATED in fp32 on the bf16
 * matrix cores: each fp32 operand is carried as a (hi, lo) pair of bf16 values (16 significant bits) and a product is
 * three bf16 MFMAs with fp32 accumulation, hi.hi + hi.lo + lo.hi.  Attention, LayerNorm, CR-MSA and every other GEMM
 * are the F32 path's.  Measured distance to the F32 path on the encoder output: ~1e-6; regions of 49..144
 * tokens with head dim 64, anything else silently takes the exact F32 kernels. */
/* In BF16 / F16 on regions of 17..256 tokens with head dim 64 the R-MSA layers run on 16-bit data end to end
 * (rrt_ln_partition16 -> rrt_rmsa_fused16 -> rrt_linear16_f32 below): the LayerNorm output, the weights and the
 * attention output live in HBM in 16 bits, and Q~, K, V and the softmax probabilities go to the matrix cores in 16
 * bits too (fp32 accumulation, fp32 softmax statistics) -- what autocast does to nn.Linear, q k^T and attn v.
 * The residual stream x / x1 / y, LayerNorm, CR-MSA's logits / combine / dispatch stay fp32. */
enum { RRT_POS_NONE = 0, RRT_POS_PEG = 1, RRT_POS_PPEG = 2 };
enum { RRT_EPEG_ATTN = 0, RRT_EPEG_VALUE_BF = 1, RRT_EPEG_VALUE_AF = 2 };

/* Elementwise activations of the caller-side layers (patch_to_emb, DAttention). */
enum { RRT_ACT_NONE = 0, RRT_ACT_RELU = 1, RRT_ACT_GELU = 2, RRT_ACT_TANH = 3, RRT_ACT_SIGMOID = 4 };

/* Region-grid geometry: RegionAttntion.padding, modules/rmsa.py:175-202 (same body
 * CrossRegionAttntion.padding :261-288).  H = padded grid side, s = region side,
 * add = zero rows appended (H*H - L). */
typedef struct rrt_grid {
  int64_t L;
  int32_t H;
  int32_t s;
  int32_t regions_side;
  int64_t add;
} rrt_grid;

/* Constructor surface that changes the arithmetic: RRTEncoder.__init__,
 * modules/rrt.py:134-163 (kwargs that reach TransLayer/InnerAttention). */
typedef struct rrt_encoder_desc {
  int32_t dim;             /* mlp_dim */
  int32_t n_heads;         /* n_heads (R-MSA), head_dim = dim / n_heads */
  int32_t n_rmsa_layers;   /* n_layers - 1 */
  int32_t region_num;      /* R-MSA regions per side */
  int32_t region_size;     /* >0 overrides region_num */
  int32_t min_region_num;
  float   min_region_ratio;
  int32_t epeg;            /* 1: 1-D 'attn' EPEG in the R-MSA layers */
  int32_t epeg_k;
  int32_t cr_msa;          /* 1: CR-MSA layer present */
  int32_t crmsa_k;
  int32_t crmsa_heads;
  int32_t crmsa_mlp;       /* 1: phi is Linear(dim, dim/4) -> Tanh -> Linear(dim/4, k), rmsa.py:248-252 */
  int32_t all_shortcut;
  int32_t compute;         /* RRT_COMPUTE_*: operand precision of the nn.Linear layers */
  int32_t ffn;             /* 1: every TransLayer (CR-MSA's too) ends with x + Mlp(LN2(x)), rrt.py:105-106,127-129 */
  int32_t ffn_act;         /* RRT_ACT_GELU (ffn_act='gelu') or RRT_ACT_RELU (anything else), rrt.py:105 */
  int32_t ffn_hidden;      /* int(dim * mlp_ratio), a multiple of 32 */
  int32_t pos;             /* RRT_POS_NONE / RRT_POS_PEG / RRT_POS_PPEG (pos=, modules/emb_position.py:24-82) */
  int32_t pos_pos;         /* -1: before the first layer; 0: before layer index 1 (needs n_layers >= 3), rrt.py:181-187 */
  int32_t peg_k;           /* odd, <= 11 */
  int32_t peg_1d;          /* 1: (k, 1) kernels (peg_1d / conv_1d) */
  /* EPEG ablations (rmsa.py:76-85,106-129; unfused path, forward and backward): */
  int32_t epeg_2d;         /* 1: k x k kernel -- over the score map ('attn') or over v's sqrt(P) x sqrt(P) image ('value_*') */
  int32_t epeg_type;       /* RRT_EPEG_ATTN (default) / RRT_EPEG_VALUE_BF / RRT_EPEG_VALUE_AF */
  /* Reduced-precision modes (BF16 / F16 / F32X3) keep 16-bit images of the R-MSA weights at the START of the workspace
   * (4 dim^2 floats per layer, whatever n_tokens is).  With 0 EVERY call in such a mode writes them (one launch,
   * ~5 us), whether or not this bag's region size takes the 16-bit kernels -- so "an earlier successful call on this
   * workspace with the same weights, dim, n_rmsa_layers and compute" is all that 1 has to promise; the bag sizes of
   * the two calls need not match.  The library keeps no state: 0 is always right, 1 is the caller's promise
   * (rrt_mil_amd tracks parameter versions). */
  int32_t weights16_valid;
  /* Scheduling hint.  Results are deterministic for a given setting and equal between the two settings up to fp32 summation
   * order: with 1 the representatives' two small GEMMs sum K in four interleaved groups (linear_splitk), so low-order
   * bits of the output may differ from a solo = 0 forward of the same bag (both within ~1e-6 of the float64 oracle); with 1 the
   * last R-MSA layer's merged launch also leaves CR-MSA's LayerNorm statistics / logit products (RRT_PLAN_CRMSA_PARTS), which
   * are then merged slab-wise instead of summed row-wise (same distance);
   * callers that compare bits across entry points use one setting (RRTEncoder.solo; the executor uses 0 unless it has
   * exactly one stream).  0 (default): other work may share the GPU with this forward (more bags in
   * flight on other streams): every kernel of the latency-bound CR-MSA tail keeps a footprint that fits NEXT TO a block
   * of the other bag's fused R-MSA kernel (<= 4 waves, <= 40 KiB LDS).  1: the forward has the GPU to itself (one bag in
   * flight): the representatives' small GEMMs may take whole CUs (16-wave blocks, K split inside the block: 10 -> 6-7 us
   * each).  The executor sets it to (n_streams == 1) itself. */
  int32_t solo;
} rrt_encoder_desc;

/* One TransLayer's parameters: InnerAttention, modules/rmsa.py:57-89 (+ the optional FFN).  Row-major, fp32.
 * qkv_w [3*dim, dim], qkv_b [3*dim] or NULL, proj_w [dim, dim], proj_b [dim],
 * pe_w [heads, epeg_k] or NULL, pe_b [heads] or NULL (epeg_2d: pe_w [heads, k, k]; epeg_type value_*:
 * pe_w [dim, k] or [dim, k, k], pe_b [dim] -- the value variants READ the bias, it is added to v / x). */
typedef struct rrt_attn_weights {
  const float *norm_w, *norm_b;   /* the owning TransLayer's LayerNorm, modules/rrt.py:47 */
  const float *qkv_w, *qkv_b;
  const float *proj_w, *proj_b;
  const float *pe_w, *pe_b;
  /* ffn = 1 only (Mlp, modules/rrt.py:25-41): norm2 [dim]; fc1 [ffn_hidden, dim] + [ffn_hidden];
   * fc2 [dim, ffn_hidden] + [dim] */
  const float *norm2_w, *norm2_b;
  const float *fc1_w, *fc1_b;
  const float *fc2_w, *fc2_b;
} rrt_attn_weights;

typedef struct rrt_encoder_weights {
  rrt_attn_weights rmsa[RRT_MAX_RMSA_LAYERS];   /* layers.{i}.* */
  rrt_attn_weights crmsa;                         /* cr_msa.norm.*, cr_msa.attn.attn.* */
  const float *phi;                               /* cr_msa.attn.phi [dim, crmsa_k]          (crmsa_mlp = 0) */
  const float *phi0_w, *phi2_w;                   /* cr_msa.attn.phi.0.weight [dim/4, dim],
                                                     cr_msa.attn.phi.2.weight [crmsa_k, dim/4] (crmsa_mlp = 1) */
  const float *norm_w, *norm_b;                   /* final norm.* (modules/rrt.py:139,195) */
  /* pos_embedding.proj / proj1 / proj2 (depth-wise [dim, 1, k, k] or [dim, 1, k, 1]; PEG: only [0]); biases may be
   * NULL (peg_bias = False) */
  const float *pos_w[3], *pos_b[3];
  /* 0 = unknown.  Otherwise a number the caller changes whenever any weight VALUE changes (the pointers alone do not
   * tell): the executor, which owns its workspaces, skips the 16-bit weight conversion of the reduced-precision modes
   * while it stays the same.  Stateless entry points ignore it (see rrt_encoder_desc.weights16_valid). */
  uint64_t version;
} rrt_encoder_weights;

int         rrt_abi_version(void);
const char *rrt_strerror(int code);          /* static string; also explains the last RRT_E_UNSUPPORTED of this thread */

/* host-only: modules/rmsa.py:175-202 */
int rrt_region_grid(int64_t L, int32_t region_num, int32_t region_size,
                    int32_t min_region_num, float min_region_ratio, rrt_grid *out);

/* host-only: bytes of workspace rrt_encoder_forward_f32 needs for a bag of n_tokens */
int rrt_encoder_workspace_size(const rrt_encoder_desc *desc, int64_t n_tokens, size_t *bytes);

/* host-only: which kernels rrt_encoder_forward_f32 takes for the R-MSA layers of a bag of n_tokens (for measurement:
 * bench.py attributes FLOPs to launches with it).  *flags = RRT_PLAN_* bits. */
#define RRT_PLAN_FUSED      1   /* qkv projection + EPEG + attention in one kernel per (region, head) (fp32 data) */
#define RRT_PLAN_FUSED_PROJ 2   /* ... with the out-projection + un-partition + residual as a later phase of the same launch */
#define RRT_PLAN_FUSED16    4   /* the 16-bit fused kernels (bf16 / fp16 modes) */
#define RRT_PLAN_FUSED_X3   8   /* the split-bf16 fused kernel (RRT_COMPUTE_F32X3) */
#define RRT_PLAN_CRMSA_PARTS 16 /* the last layer's merged launch also leaves CR-MSA's row records (LayerNorm 2 statistics +
                                   logit dot products); CR-MSA's first pass is rrt_crmsa_combine_parts_f32 */
int rrt_encoder_plan(const rrt_encoder_desc *desc, int64_t n_tokens, int32_t *flags);

/* Whole path: RRTEncoder.forward, modules/rrt.py:165-202 (eval mode, one bag).
 * x [n_tokens, dim] -> y [n_tokens, dim]; x is not modified; y may not alias x. */
int rrt_encoder_forward_f32(const rrt_encoder_desc *desc, const rrt_encoder_weights *w,
                            const float *x, float *y, int64_t n_tokens,
                            void *workspace, size_t workspace_bytes, void *stream);

/* Same call, recording caller-owned hipEvent_t's at stage boundaries on `stream` (for
 * measurement: bench.py times the dominant kernel with these).  events[RRT_EV_COUNT], any
 * entry may be NULL.  Only the first R-MSA layer is marked. */
enum {
  RRT_EV_START = 0,        /* before the first kernel */
  RRT_EV_LN_PARTITION,     /* after LayerNorm+partition */
  RRT_EV_QKV,              /* after the R-MSA qkv linear  (the dominant kernel) */
  RRT_EV_ATTN,             /* after the region attention core */
  RRT_EV_PROJ,             /* after proj + un-partition + residual */
  RRT_EV_CR_COMBINE,       /* after CR-MSA logits + combine */
  RRT_EV_CR_INNER,         /* after the inner MSA over the representatives */
  RRT_EV_END,              /* after dispatch + final LayerNorm */
  RRT_EV_COUNT
};
int rrt_encoder_forward_events_f32(const rrt_encoder_desc *desc, const rrt_encoder_weights *w,
                                   const float *x, float *y, int64_t n_tokens,
                                   void *workspace, size_t workspace_bytes, void *stream,
                                   void **events);

/* Concurrent forwards on several streams (one bag each): a phase gate shared by those calls keeps their
 * MFMA-bound R-MSA cores from co-running (they take turns instead of time-slicing the matrix pipes).  Optional:
 * pass gate = NULL for free-running streams, which measure faster with the current kernels (the executor below
 * gates only under RRT_GATE=1).  Host-side object, one host thread.  events may be NULL (else as in
 * rrt_encoder_forward_events_f32). */
typedef struct rrt_phase_gate rrt_phase_gate;
int rrt_phase_gate_create(rrt_phase_gate **out);
int rrt_phase_gate_destroy(rrt_phase_gate *gate);
int rrt_encoder_forward_gated_f32(const rrt_encoder_desc *desc, const rrt_encoder_weights *w,
                                  const float *x, float *y, int64_t n_tokens,
                                  void *workspace, size_t workspace_bytes, void *stream,
                                  rrt_phase_gate *gate, void **events);

/* Batch > 1: RRTEncoder.forward on (B, N, D) input (modules/rrt.py:165-202), inference.  x, y: [batch, n_tokens, dim]
 * contiguous.  At B > 1 the reference's region_partition puts the regions of all bags on one axis (rmsa.py:28-39): the R-MSA
 * layers still treat the bags independently, but CR-MSA's inner attention runs over the 64 * B representatives of ALL bags
 * (rmsa.py:316-322) -- the outputs of a batch differ from the outputs of its bags taken one at a time, and this entry point
 * reproduces exactly that.  (Independent bags -- what every reference trainer feeds, batch_size = 1 -- go through
 * rrt_encoder_forward_f32 / rrt_executor_forward.)  Correctness path: the R-MSA layers bag by bag, one inner MSA for the
 * batch; workspace from rrt_encoder_batch_workspace_size. */
int rrt_encoder_batch_workspace_size(const rrt_encoder_desc *desc, int32_t batch, int64_t n_tokens, size_t *bytes);
int rrt_encoder_forward_batch_f32(const rrt_encoder_desc *desc, const rrt_encoder_weights *w,
                                  const float *x, float *y, int32_t batch, int64_t n_tokens,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* ---- stage entry points (what the fused path is built from; used by the parity tests) ---- */

/* LayerNorm (modules/rrt.py:121-123) + zero-pad (rmsa.py:199-200) + region_partition
 * (rmsa.py:28-39): x [L, dim] -> u [H*H, dim] in region-major order, pad rows = 0. */
int rrt_ln_partition_f32(const float *x, const float *gamma, const float *beta, float *u,
                         int64_t L, int32_t dim, const rrt_grid *g, void *stream);

/* nn.Linear (rmsa.py:100, :131): C[M,N] = A[M,K] . B[N,K]^T + bias[N] (bias may be NULL).
 * q_cols>0: columns [0,q_cols) are multiplied by q_scale after the bias (rmsa.py:103). */
int rrt_linear_f32(const float *A, const float *B, const float *bias, float *C,
                   int64_t M, int32_t N, int32_t K, int32_t q_cols, float q_scale,
                   int32_t compute /* RRT_COMPUTE_* */, void *stream);

/* nn.Linear + region_reverse + un-pad + residual (rmsa.py:131, :41-54, :227-228; rrt.py:125):
 * out[t] = resid[t] + (A . B^T + bias)[slot(t)] for the L real tokens. */
int rrt_linear_unpartition_residual_f32(const float *A, const float *B, const float *bias,
                                        const float *resid, float *out, int32_t N, int32_t K,
                                        const rrt_grid *g, int32_t compute, void *stream);

/* Region attention core (rmsa.py:103-122): qkv [n_regions*P, 3*dim] (q already scaled),
 * EPEG taps pe_w [heads, epeg_k] (NULL/0 = none; pe bias is softmax-invariant and not needed)
 * -> o [n_regions*P, dim] (heads merged). */
int rrt_region_attention_f32(const float *qkv, const float *pe_w, float *o,
                             int32_t n_regions, int32_t P, int32_t dim, int32_t heads,
                             int32_t epeg_k, void *stream);

/* Fused R-MSA core (rmsa.py:100-122 in one kernel per (region, head)): u [n_regions*P, dim]
 * region-major LayerNorm-ed tokens -> o [n_regions*P, dim]; qkv_w [3*dim, dim], qkv_b [3*dim] or NULL,
 * pe_w [heads, epeg_k] or NULL.  The qkv tensor never exists in HBM.  Supported when head dim is 64
 * and 48 < P <= 208 (one region's Q, K, V tiles fit a CU's LDS); otherwise RRT_E_UNSUPPORTED and the caller uses
 * rrt_linear_f32 + rrt_region_attention_f32 (rrt_encoder_forward_f32 chooses by itself). */
int rrt_rmsa_fused_f32(const float *u, const float *qkv_w, const float *qkv_b, const float *pe_w,
                       float *o, int32_t n_regions, int32_t P, int32_t dim, int32_t heads,
                       int32_t epeg_k, int32_t compute, void *stream);

/* The same kernel with the out-projection as a later phase of the SAME launch (fp32 exact): rrt_rmsa_fused_f32 followed by
 * rrt_linear_unpartition_residual_f32 in one launch, bit-identical to that pair --
 * out[t] = resid[t] + (o . proj_w^T + proj_b)[slot(t)] (rmsa.py:100-131, :41-54, :227-228; rrt.py:125).
 * u [H*H, dim] region-major (rrt_ln_partition_f32 on g); o_scratch: H*H*dim floats (the attention output, device scratch);
 * counters: regions_side^2 int32 of device scratch (zeroed by the call).  Needs regions of > 64 tokens and heads * regions >= 2 x the CU count
 * (block b of the launch runs (region, head) item b and then the 64-column projection slab b - CUs of a region whose
 * items finished a whole item earlier), otherwise RRT_E_UNSUPPORTED; rrt_encoder_forward_f32 chooses by itself. */
int rrt_rmsa_fused_proj_f32(const float *u, const float *qkv_w, const float *qkv_b, const float *pe_w,
                            const float *proj_w, const float *proj_b, const float *resid, float *out,
                            float *o_scratch, int32_t *counters, int32_t dim, int32_t heads, int32_t epeg_k,
                            const rrt_grid *g, void *stream);
/* ... and CR-MSA's first pass as a by-product (what rrt_encoder_forward_f32 does for the LAST R-MSA layer when a plain-phi
 * CR-MSA with k <= 4 follows it directly and desc.solo != 0 -- it trades ~3 us of the merged launch's matrix-pipe time for
 * ~6 us of latency-bound CR-MSA front: a gain with one bag in flight, none with four): out = x1 is CR-MSA's input, and the slabs hold its tiles in registers -- per (token t,
 * 64-column slab c) they also store the record  part[(t * dim / 64 + c) * S ..] = (mean, M2, d_0 .. d_k-1), S = 2 + k rounded
 * up to a multiple of 4:  mean and
 * centred sum of squares of x1[t, 64 c .. 64 c + 63], d_n = sum_j x1[t, j] ln2_gamma[j] phi[j, n] over those columns
 * (LayerNorm 2 = cr_msa.norm, phi [dim, k] = cr_msa.attn.phi; modules/rmsa.py:303-307, rrt.py:121).  part: n_tokens * dim / 64 *
 * S floats of device scratch (dim <= 512).  rrt_crmsa_combine_parts_f32 turns them and ONE pass over x1 into CR-MSA's combine:
 * logits / region softmax / min-max / dispatch weights wdisp [H8*H8, k] (region-major) and the representatives
 * rep [k, 64, dim] (rmsa.py:307-316) -- the job of rrt_crmsa_logits_f32 + rrt_crmsa_combine_f32, without re-reading x1 for
 * LayerNorm and without any hand-over between blocks.  g8 = rrt_region_grid(n_tokens, 8, ...). */
int rrt_rmsa_fused_proj_stats_f32(const float *u, const float *qkv_w, const float *qkv_b, const float *pe_w,
                                  const float *proj_w, const float *proj_b, const float *resid, float *out,
                                  float *o_scratch, int32_t *counters, const float *ln2_gamma, const float *phi,
                                  int32_t crmsa_k, float *part, int32_t dim, int32_t heads, int32_t epeg_k,
                                  const rrt_grid *g, void *stream);
int rrt_crmsa_combine_parts_f32(const float *x1, const float *part, const float *gamma, const float *beta,
                                const float *phi, float *wdisp, float *rep, int64_t n_tokens, int32_t dim, int32_t k,
                                const rrt_grid *g8, void *stream);
/* The in-launch hand-over of that kernel and what it assumes.  A projection slab (block b >= lag) spins on its region's
 * arrival counter until the region's `heads` items -- blocks with LOWER indices -- have stored their O rows.  Forward
 * progress rests on the GPU dispatching the workgroups of one launch in index order (so an awaited item is running or
 * done when its slab starts): observed on every gfx9 part, relied on here, but NOT an architectural guarantee.  The wait
 * is therefore bounded (2^22 sleeps of ~0.2 us, about a second): a slab that gives up writes nothing and raises the
 * process's hand-over error word (pinned host memory, no device sync needed to read it).  From then on
 * rrt_encoder_forward_* / rrt_mil_forward_f32 / rrt_executor_forward / rrt_rmsa_fused_proj_f32 return RRT_E_HANDOVER
 * without launching anything, until the caller has looked at it:
 *   rrt_device_error(clear): 0, or 1 + the region whose slab gave up; clear != 0 resets the word (after the caller has
 *   synchronised the device and discarded the outputs of the forwards in flight).  This word is the library's only state. */
int rrt_device_error(int32_t clear);
/* Test hook for that path (tests/test_hip_parity.py): rrt_rmsa_fused_proj_f32 with a caller-chosen lag (0: the rule;
 * otherwise 8 * heads <= lag <= heads * regions; a lag that is not a multiple of 8 puts every slab on ANOTHER XCD than
 * the items it reads -- the hand-over must not depend on the placement heuristic), spin limit (0: 2^22) and
 * `wait_extra` arrivals more than a region ever gets (> 0: every slab times out -- exercises the error path). */
int rrt_debug_rmsa_fused_proj_f32(const float *u, const float *qkv_w, const float *qkv_b, const float *pe_w,
                                  const float *proj_w, const float *proj_b, const float *resid, float *out,
                                  float *o_scratch, int32_t *counters, int32_t dim, int32_t heads, int32_t epeg_k,
                                  const rrt_grid *g, int32_t lag, int32_t spin_limit, int32_t wait_extra, void *stream);

/* 16-bit operand stages of the reduced-precision modes (compute = RRT_COMPUTE_BF16 / F16 selects the element type;
 * uint16_t* = raw bf16 / fp16 bits, round-to-nearest-even):
 *  cast16        : dst[i] = (16-bit) src[i], n % 4 == 0 (the nn.Linear weights, once per forward);
 *  ln_partition16: rrt_ln_partition_f32 with the normalised rows rounded to 16 bits (what autocast feeds nn.Linear);
 *  linear16      : C fp32 [M, N] = A16 [M, K] . B16 [N, K]^T + bias; with resid != NULL the un-partition + residual
 *                  epilogue of rrt_linear_unpartition_residual_f32 (M = H*H of g); K % 64 == 0 (any M: ragged
 *                  row tiles are masked);
 *  rmsa_fused16  : rrt_rmsa_fused_f32 on 16-bit u / qkv_w, attention operands in 16 bits, o in 16 bits
 *                  (rmsa.py:100-122 under autocast); head dim 64, 16 < P <= 256. */
int rrt_cast16(const float *src, uint16_t *dst, int64_t n, int32_t compute, void *stream);
int rrt_ln_partition16(const float *x, const float *gamma, const float *beta, uint16_t *u, int64_t L,
                       int32_t dim, const rrt_grid *g, int32_t compute, void *stream);
int rrt_linear16_f32(const uint16_t *A, const uint16_t *B, const float *bias, const float *resid, float *C,
                     int64_t M, int32_t N, int32_t K, const rrt_grid *g, int32_t compute, void *stream);
int rrt_rmsa_fused16(const uint16_t *u, const uint16_t *qkv_w, const float *qkv_b, const float *pe_w,
                     uint16_t *o, int32_t n_regions, int32_t P, int32_t dim, int32_t heads, int32_t epeg_k,
                     int32_t compute, void *stream);
/* Round 6: rrt_rmsa_fused16 + rrt_linear16_f32(resid) of one R-MSA layer in ONE launch (modules/rmsa.py:100-132, :41-54,
 * :227-228; rrt.py:125 under autocast), what rrt_encoder_forward_f32 uses in BF16 / F16 for bags of at least two rounds of
 * (region pair, head) items -- regions of 65..96 or 113..128 tokens, a multiple of 16 regions, e.g. N = 30000 at
 * region_num = 16: block b runs item b and then the 64-column projection slab b - 256 of a pair whose items finished a
 * round earlier; out [L, dim] = resid + unpartition(o proj_w^T + proj_b), bit-identical to the two calls; o [Np, dim] is
 * scratch (the 16-bit attention output); cnt: Np / (2 P) ints of scratch (zeroed by the call).  The in-launch hand-over is
 * bounded like rrt_rmsa_fused_proj_f32's (RRT_E_HANDOVER, rrt_device_error). */
int rrt_rmsa_pair16_proj(const uint16_t *u, const uint16_t *qkv_w, const float *qkv_b, const float *pe_w,
                         const uint16_t *proj_w, const float *proj_b, const float *resid, float *out, uint16_t *o,
                         int32_t *cnt, int32_t dim, int32_t heads, int32_t epeg_k, const rrt_grid *g, int32_t compute,
                         void *stream);

/* RRT_COMPUTE_F32X3 stages.  A "split image" of a row-major fp32 array holds, per 32 consecutive elements, 32 bf16 hi
 * values then 32 bf16 lo values (128 bytes; hi = bf16(x), lo = bf16(x - hi)) -- 4 bytes per element like the original.
 *  cast_split        : dst = split image of src, n % 32 == 0 (the weights, once per forward);
 *  ln_partition_split: rrt_ln_partition_f32 with the rows written as a split image;
 *  linear_split      : C fp32 [M, N] = A . B^T + bias from split images of A [M, K] and B [N, K]; with resid != NULL
 *                      the un-partition + residual epilogue (M = H*H of g); K % 32 == 0;
 *  rmsa_fused_x3     : rrt_rmsa_fused_f32 on split images of u / qkv_w, o written as a split image; 48 < P <= 144. */
int rrt_cast_split(const float *src, void *dst, int64_t n, void *stream);
int rrt_ln_partition_split(const float *x, const float *gamma, const float *beta, void *u, int64_t L,
                           int32_t dim, const rrt_grid *g, void *stream);
int rrt_linear_split_f32(const void *A, const void *B, const float *bias, const float *resid, float *C,
                         int64_t M, int32_t N, int32_t K, const rrt_grid *g, void *stream);
int rrt_rmsa_fused_x3(const void *u, const void *qkv_w, const float *qkv_b, const float *pe_w, void *o,
                      int32_t n_regions, int32_t P, int32_t dim, int32_t heads, int32_t epeg_k, void *stream);

/* CR-MSA (rmsa.py:303-335), three kernels around the inner MSA (g8 = the 8x8 grid):
 *  logits  : LayerNorm statistics mean_rstd [L,2] and logits [Np8, k] in region-major order
 *            (zero rows for pad tokens);
 *  combine : per region the combine softmax over its P tokens -> rep [k, R8, dim], and the
 *            per-token dispatch weights wdisp [Np8, k] = minmax_p(logit) * softmax_k(logit);
 *            with mean_rstd == NULL, x1 must already hold LN(x1) in region-major order
 *            [Np8, dim] (the crmsa_mlp path) and gamma/beta are ignored;
 *  dispatch: y = LN(x1 + sum_n wdisp[.,n] * rep2[n, region] (+ x0)), the final norm fused. */
int rrt_crmsa_logits_f32(const float *x1, const float *gamma, const float *beta, const float *phi,
                         float *mean_rstd, float *logits, int64_t L, int32_t dim, int32_t k,
                         const rrt_grid *g8, void *stream);
int rrt_crmsa_combine_f32(const float *x1, const float *gamma, const float *beta,
                          const float *mean_rstd, const float *logits, float *wdisp, float *rep,
                          int64_t L, int32_t dim, int32_t k, const rrt_grid *g8, void *stream);
/* logits + combine in one pass over x1 (dim = 512, k <= 3, CR-MSA regions of <= 144 tokens: one block per region, the rows
 * are read once and stay in registers; rrt_encoder_forward_f32 uses it under RRT_CRMSA_REGION=1 -- better with several
 * bags in flight, worse with one); same outputs as the two calls above
 * (mean_rstd and logits may be NULL); RRT_E_UNSUPPORTED outside that range. */
int rrt_crmsa_region_f32(const float *x1, const float *gamma, const float *beta, const float *phi,
                         float *mean_rstd, float *logits, float *wdisp, float *rep,
                         int64_t L, int32_t dim, int32_t k, const rrt_grid *g8, void *stream);
/* The same at full chip width, what rrt_encoder_forward_f32 uses (dim = 512, k <= 8, regions of 4..576 tokens -- bags up
 * to ~36 k patches --, at least one R-MSA layer): 4 / 8 / 16 blocks per region (regions of <= 144 / 288 / 576 tokens), each
 * with its share of the rows; the block that arrives last merges the partial records like an online softmax (nobody
 * waits for anybody).  scratch: 256 + 64 * nb * km * 520 * 4 bytes with nb = 16 for regions of more than 288 tokens, else
 * 8, and km = 3 for k <= 3, else 8; logits must be given (the merging block reads them); mean_rstd may be NULL. */
int rrt_crmsa_region4_f32(const float *x1, const float *gamma, const float *beta, const float *phi,
                          float *mean_rstd, float *logits, float *wdisp, float *rep,
                          int64_t L, int32_t dim, int32_t k, const rrt_grid *g8,
                          void *scratch, size_t scratch_bytes, void *stream);
/* Round 6: the one-pass form for LARGER regions (what rrt_encoder_forward_f32 uses above 144 tokens per CR-MSA region, i.e.
 * bags of more than ~9.2 k patches: replaces modules/rmsa.py:303-316 like the two calls at the top of this group, reading x1
 * once instead of twice): always 4 blocks per region, each STREAMING its quarter of the rows in trips with a per-wave online
 * softmax; records, merge, scratch layout and arguments exactly as rrt_crmsa_region4_f32 (dim = 512, k <= 8, regions of
 * 16..576 tokens). */
int rrt_crmsa_stream4_f32(const float *x1, const float *gamma, const float *beta, const float *phi,
                          float *mean_rstd, float *logits, float *wdisp, float *rep,
                          int64_t L, int32_t dim, int32_t k, const rrt_grid *g8,
                          void *scratch, size_t scratch_bytes, void *stream);
int rrt_crmsa_dispatch_ln_f32(const float *x1, const float *x0, const float *wdisp,
                              const float *rep2, const float *gamma,
                              const float *beta, float *y, int64_t L, int32_t dim, int32_t k,
                              const rrt_grid *g8, void *stream);
/* crmsa_mlp logits (rmsa.py:248-252, :305): logits[r, n] = sum_j tanh(hid[r, j]) * w2[n, j] */
int rrt_crmsa_mlp_logits_f32(const float *hid, const float *w2, float *logits, int64_t rows,
                             int32_t hdim, int32_t k, void *stream);
/* final LayerNorm only (cr_msa=False path): y = LN(x1 (+ x0)) */
int rrt_layernorm_f32(const float *x1, const float *x0, const float *gamma, const float *beta,
                      float *y, int64_t L, int32_t dim, void *stream);

/* ---- row f1: the RRTMIL slide classifier around the encoder (modules/rrt.py:204-246) ----
 * patch_to_emb Linear(input_dim, dim)+act (rrt.py:208-217; Dropout is identity in eval) ->
 * RRTEncoder -> DAttention pooling (modules/datten.py) -> predictor Linear(dim, n_classes). */
typedef struct rrt_mil_desc {
  rrt_encoder_desc enc;
  int32_t input_dim;       /* patch feature width (multiple of 32) */
  int32_t emb_act;         /* RRT_ACT_RELU / RRT_ACT_GELU / RRT_ACT_NONE (RRTMIL act=) */
  int32_t n_classes;
  int32_t pool_hidden;     /* DAttention D = 128 (multiple of 4) */
  int32_t pool_act;        /* RRT_ACT_RELU / GELU / TANH / NONE (da_act) */
  int32_t pool_gated;      /* 1: AttentionGated (datten.py:40-83) */
  int32_t input16;         /* (ABI 25) 0: x is fp32.  RRT_COMPUTE_BF16 / RRT_COMPUTE_F16: x points to 16-bit features of that
                            * type [n_tokens, input_dim] (half the PCIe / HBM bytes of the bag, rrt_mil_amd.BagFeeder(dtype=...));
                            * needs enc.compute == input16 and input_dim % 64 == 0, otherwise RRT_E_UNSUPPORTED.  Under autocast
                            * the reference's first op (patch_to_emb Linear, rrt.py:208-229) rounds the fp32 features to exactly
                            * these 16-bit values, so the logits are bit-identical to the fp32-fed call. */
} rrt_mil_desc;

/* Row-major fp32, state_dict layout.  Non-gated: pool_a = pool_fn.attention.attention.0,
 * pool_c = its last Linear; gated: pool_a/pool_b/pool_c = attention_a.0 / attention_b.0 / attention_c.
 * Biases may be NULL (da_bias=False). */
typedef struct rrt_mil_weights {
  rrt_encoder_weights enc;
  const float *emb_w, *emb_b;         /* patch_to_emb.0  [dim, input_dim], [dim] */
  const float *pool_a_w, *pool_a_b;   /* [pool_hidden, dim], [pool_hidden] */
  const float *pool_b_w, *pool_b_b;   /* gated only */
  const float *pool_c_w, *pool_c_b;   /* [1, pool_hidden], [1] */
  const float *pred_w, *pred_b;       /* predictor [n_classes, dim], [n_classes] */
} rrt_mil_weights;

/* The pooling alone (datten.py:28-38 / :69-83 AFTER the first Linear + activation; training keeps those as autograd-visible
 * layers): s_n = c_w . h_n + c_b with h = hid_a (or hid_a * hid_b, gated), attn = softmax_n(s), pooled = sum_n attn_n y_n.
 * Outputs pooled [dim], attn [N] (normalised), a_raw [N] (the scores).  ... and its adjoint: given d_pooled [dim] (and
 * optionally d_attn [N] with c_ext = sum_n attn_n d_attn_n as a device scalar, d_raw [N]) ->
 * dy [N, dim] = attn_n d_pooled, dhid_a / dhid_b [N, hidden], dwcb [hidden + 4] = d c_w | d c_b (at [hidden]).
 * workspace: rrt_attn_pool_workspace_size bytes (either call). */
int rrt_attn_pool_workspace_size(int64_t n_tokens, int32_t dim, int32_t hidden, size_t *bytes);
int rrt_attn_pool_f32(const float *y, const float *hid_a, const float *hid_b, const float *c_w, const float *c_b,
                      float *pooled, float *attn, float *a_raw, int64_t n_tokens, int32_t dim, int32_t hidden,
                      void *workspace, size_t workspace_bytes, void *stream);
int rrt_attn_pool_backward_f32(const float *y, const float *hid_a, const float *hid_b, const float *c_w,
                               const float *attn, const float *pooled, const float *d_pooled,
                               const float *d_attn, const float *d_raw, const float *c_ext, float *dy,
                               float *dhid_a, float *dhid_b, float *dwcb, int64_t n_tokens, int32_t dim,
                               int32_t hidden, void *workspace, size_t workspace_bytes, void *stream);

int rrt_mil_workspace_size(const rrt_mil_desc *desc, int64_t n_tokens, size_t *bytes);

/* RRTMIL.forward (eval, one bag): x [n_tokens, input_dim] -> logits [n_classes].
 * attn (optional, [n_tokens]): the attention row the reference returns with return_attn=True
 * (softmax over the bag, or the raw scores when no_norm != 0).  feat (optional, [n_tokens, dim]):
 * the encoder output. */
int rrt_mil_forward_f32(const rrt_mil_desc *desc, const rrt_mil_weights *w, const float *x,
                        float *logits, float *attn, int32_t no_norm, float *feat, int64_t n_tokens,
                        void *workspace, size_t workspace_bytes, void *stream);

/* DAttention + predictor alone on encoder features y [n_tokens, dim] (datten.py:28-38,69-83; rrt.py:241).
 * pooled (optional, [dim]) receives the bag embedding.  workspace: rrt_pool_workspace_size bytes. */
int rrt_pool_workspace_size(int64_t n_tokens, int32_t dim, int32_t hidden, int32_t gated, size_t *bytes);
int rrt_pool_predict_f32(const float *y, const float *a_w, const float *a_b, const float *b_w,
                         const float *b_b, const float *c_w, const float *c_b, const float *pred_w,
                         const float *pred_b, float *pooled, float *logits, float *attn, int32_t no_norm,
                         int64_t n_tokens, int32_t dim, int32_t hidden, int32_t act, int32_t n_classes,
                         int32_t compute, void *workspace, size_t workspace_bytes, void *stream);

/* nn.Linear with an activation epilogue: C = act(A . B^T + bias)   (patch_to_emb, rrt.py:208-217) */
int rrt_linear_act_f32(const float *A, const float *B, const float *bias, float *C, int64_t M, int32_t N,
                       int32_t K, int32_t act, int32_t compute, void *stream);

/* ---- batch-of-bags executor (BASELINE configs[4]: mixed-size bags, every bag an independent B=1 forward;
 * the reference loops `for bag in loader: model(bag)`, main.py:466-467 / :558-560) ----
 * Bags are independent units, and one bag's forward is a dependent chain of ~10 kernels that leaves
 * launch ramps, tails and store phases idle; the executor keeps `n_streams` bags in flight on its own HIP
 * streams (one workspace each, owned by the executor) so those gaps are filled by another bag's kernels.
 * rrt_executor_forward is ordered on the caller's `stream` like any other call here; the caller's stream carries the
 * first share of the bags itself and the executor's streams 1 .. n_streams-1 the others (fork: they wait for `stream`;
 * join: `stream` waits for them BEHIND its own bags -- a caller stream that only sits on the join's barrier packets is one
 * more active hardware queue, and four bag queues plus that one lose 13 % on MI355X's four pipes); a one-stream executor is
 * plain launches on `stream`, and so is a call of fewer than 8 bags on any executor (a call uses one stream per four
 * bags, at most n_streams: forking for 2-4 bags costs more than it hides -- the kernels and bits are those of the
 * executor's configured width either way); it never synchronises the host unless a
 * workspace has to grow.  One executor per host thread and device.  Set GPU_MAX_HW_QUEUES >= 8 in the
 * process environment so every stream gets its own hardware queue. */
typedef struct rrt_executor rrt_executor;
typedef struct rrt_bag {
  const float *x;      /* [n_tokens, dim] device */
  float *y;            /* [n_tokens, dim] device, != x */
  int64_t n_tokens;
} rrt_bag;
int rrt_executor_create(const rrt_encoder_desc *desc, int32_t n_streams, int64_t max_tokens,
                        rrt_executor **out);
/* The same on streams the CALLER owns (n_streams distinct hipStream_t; they are used, never destroyed): a host framework
 * hands over streams of its own pool -- rrt_mil_amd passes torch.cuda.Stream handles -- so that the process keeps ONE set of
 * streams however many executors it makes (HIP maps streams to hardware queues in creation order; streams created and
 * destroyed by executors of different widths end up sharing queues). */
int rrt_executor_create_on_streams(const rrt_encoder_desc *desc, int32_t n_streams, void *const *streams,
                                   int64_t max_tokens, rrt_executor **out);
int rrt_executor_forward(rrt_executor *ex, const rrt_encoder_weights *w, const rrt_bag *bags,
                         int32_t n_bags, void *stream);
int rrt_executor_destroy(rrt_executor *ex);

/* ---- row f2 building blocks: backward stages (assembled by rrt_encoder_backward_f32 below; parity-tested
 * on their own as well) ----
 * nn.Linear backward for Y[M,N] = X[M,K] . W[N,K]^T + b:  dX[M,K] = dY . W,  dW[N,K] = dY^T . X,
 * db[N] = column sums of dY.  Any of dX / dW / db may be NULL.  N must be a multiple of 32 when dX is
 * requested (it is the reduction length of that product). */
/* Region attention backward (rmsa.py:103-122): qkv [n_regions*P, 3*dim] as the forward stage wrote it (q scaled),
 * o = the forward output, d_o its gradient  ->  d_qkv (gradient w.r.t. the qkv linear's raw output, same layout)
 * and d_pe_w [heads, epeg_k] (NULL allowed; the conv bias gradient is exactly zero).  Head dim 64: any P (<= 208: one resident kernel; larger: a streaming four-kernel variant);
 * other head dims: no EPEG and P <= 128.  workspace: rrt_region_attention_backward_workspace_size bytes. */
int rrt_region_attention_backward_workspace_size(int32_t n_regions, int32_t P, int32_t dim, int32_t heads,
                                                 int32_t epeg_k, size_t *bytes);
int rrt_region_attention_backward_f32(const float *qkv, const float *pe_w, const float *o, const float *d_o,
                                      float *d_qkv, float *d_pe_w, int32_t n_regions, int32_t P, int32_t dim,
                                      int32_t heads, int32_t epeg_k, void *workspace, size_t workspace_bytes,
                                      void *stream);
/* LayerNorm backward (eps 1e-5): dx [L, dim] = d/dx of LN(x) . dy (+ add, the residual branch's gradient,
 * optional); dgamma_dbeta [2, dim].  g != NULL: dy is region-major padded [H*H, dim] (the qkv-linear backward's
 * output) and token t reads its slot -- the adjoint of zero-pad + region_partition.  workspace: 512*2*dim floats. */
int rrt_layernorm_backward_f32(const float *dy, const float *x, const float *gamma, const float *add,
                               float *dx, float *dgamma_dbeta, int64_t L, int32_t dim, const rrt_grid *g,
                               void *workspace, size_t workspace_bytes, void *stream);
/* Fixed-order sum of S partial vectors: out[i] = sum_s part[s * n + i] (the parameter-gradient reductions of the backward:
 * LayerNorm's d gamma | d beta, the EPEG taps, CR-MSA's norm / phi rows, the weight gradients' split-K chunks).
 * out_tr != NULL: out takes only [0, split) and the tail [tr_k, tr_dim] (n - split = tr_k * tr_dim, split % 4 == 0) leaves
 * transposed in out_tr [tr_dim, tr_k].  deferred = 0: the stage's own launch (256 elements per block); 1: through the
 * backward's job list and its one launch at the end of the pass (a block per 128-byte line of every partial), the job queued
 * `copies` (1 .. 16) times -- copy c reads the same partials and writes out + c * n (out_tr + c * tr_dim * tr_k).
 * Bit-reproducible in either form; the two forms sum in different orders. */
int rrt_reduce_partials_f32(const float *part, float *out, float *out_tr, int32_t S, int64_t n, int64_t split,
                            int32_t tr_dim, int32_t tr_k, int32_t deferred, int32_t copies, void *stream);
int rrt_linear_backward_workspace_size(int64_t M, int32_t N, int32_t K, size_t *bytes);
int rrt_linear_backward_f32(const float *dY, const float *X, const float *W, float *dX, float *dW,
                            float *db, int64_t M, int32_t N, int32_t K, int32_t compute,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ---- row f2: training.  Forward that stashes what the backward needs, and the backward itself ----
 * Supported: the default path (1-D 'attn' EPEG R-MSA layers, CR-MSA with the phi matrix or the MLP phi,
 * all_shortcut), head dim 64 in R-MSA (any multiple of 4 in CR-MSA's inner attention), bags of any size, dim <= 1024,
 * F32 compute.  Anything else: RRT_E_UNSUPPORTED.
 * drop_p / drop_seed: the train-mode proj_drop of every InnerAttention (rmsa.py:70,132; p = drop_out): a stateless
 * mask, element kept iff hash(seed, layer, index) >= p * 2^32, kept values scaled by 1/(1-p); the backward call
 * must receive the same (drop_p, drop_seed) as its forward.  drop_p = 0: no dropout.
 * branch_scale (HOST pointer, may be NULL = all ones): stochastic depth, TransLayer.drop_path (rrt.py:102,125,129;
 * timm's DropPath at batch size 1 keeps or drops a whole residual branch): 2 * (RRT_MAX_RMSA_LAYERS + 1) multipliers,
 * first the attention branches (index n_rmsa_layers = CR-MSA's), then the FFN branches in the same order; each is
 * 1 (drop_path = 0), 1 / keep_prob (kept) or 0 (dropped).  The caller draws them; forward and backward get the same.
 * Gradients mirror the parameters: norm = [2, dim] (d gamma then d beta); pe bias gradients are exactly zero and
 * are not written.  NULL gradient pointers are not allowed for parameters the model has. */
typedef struct rrt_attn_grads {
  float *norm;                 /* [2, dim] */
  float *qkv_w, *qkv_b;        /* qkv_b NULL when qkv_bias = False */
  float *proj_w, *proj_b;
  float *pe_w;                 /* [heads, epeg_k] or NULL (epeg_2d: [heads, k, k]; epeg_type value_*: [dim, k] / [dim, k, k]) */
  float *pe_b;                 /* epeg_type value_* with a conv bias: [dim]; NULL otherwise (the 'attn' conv bias has an
                                  exactly zero gradient: it cancels in the softmax) */
  float *norm2;                /* ffn = 1: [2, dim] */
  float *fc1_w, *fc1_b;        /* ffn = 1: [ffn_hidden, dim], [ffn_hidden] */
  float *fc2_w, *fc2_b;        /* ffn = 1: [dim, ffn_hidden], [dim] */
} rrt_attn_grads;
typedef struct rrt_encoder_grads {
  rrt_attn_grads rmsa[RRT_MAX_RMSA_LAYERS];
  rrt_attn_grads crmsa;
  float *phi;                  /* [dim, crmsa_k]                              (crmsa_mlp = 0) */
  float *phi0_w, *phi2_w;      /* [dim/4, dim], [crmsa_k, dim/4]              (crmsa_mlp = 1) */
  float *norm;                 /* [2, dim] final LayerNorm */
  float *pos_w[3], *pos_b[3];  /* pos_embedding.proj / proj1 / proj2 (pos != none) */
} rrt_encoder_grads;

int rrt_encoder_train_sizes(const rrt_encoder_desc *desc, int64_t n_tokens, size_t *stash_bytes,
                            size_t *backward_workspace_bytes);
/* y = RRTEncoder(x) (eval-equivalent arithmetic), intermediates kept in the caller-owned stash */
int rrt_encoder_forward_train_f32(const rrt_encoder_desc *desc, const rrt_encoder_weights *w, const float *x,
                                  float *y, int64_t n_tokens, void *stash, size_t stash_bytes,
                                  float drop_p, uint64_t drop_seed, const float *branch_scale, void *stream);
/* given dy = dL/dy: every parameter gradient and (optional) dx = dL/dx.  x and the stash are those of the forward. */
int rrt_encoder_backward_f32(const rrt_encoder_desc *desc, const rrt_encoder_weights *w, const float *x,
                             const float *dy, const void *stash, size_t stash_bytes,
                             const rrt_encoder_grads *grads, float *dx, int64_t n_tokens, void *workspace,
                             size_t workspace_bytes, float drop_p, uint64_t drop_seed,
                             const float *branch_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RRT_HIP_H */
