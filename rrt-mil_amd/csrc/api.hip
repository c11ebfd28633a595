// api.hip -- the C ABI of librrt_hip.so (include/rrt_hip.h): geometry, workspace
// carving and the launch sequence of one RRTEncoder forward (modules/rrt.py:165-202).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "internal.h"

namespace {

thread_local char g_detail[256] = "";

int unsupported(const char* why) {
  snprintf(g_detail, sizeof(g_detail), "%s", why);
  return RRT_E_UNSUPPORTED;
}

int64_t ceil_sqrt(int64_t n) {
  int64_t r = (int64_t)floor(sqrt((double)n));
  while (r * r > n) --r;
  while ((r + 1) * (r + 1) <= n) ++r;
  return r * r == n ? r : r + 1;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Workspace {
  float *uo, *qkv, *xa, *xb, *mean_rstd, *logits, *wdisp, *rep, *rep_qkv, *rep_o, *rep2, *v8, *hid;
  float *ffn_ln, *ffn_hid, *xp;
  float* pe_out;     // value-EPEG ablation: the conv's output [Np, D]
  float* smap;       // 2-D 'attn' EPEG on regions whose [P, P] score map does not fit the LDS: the maps [R * heads][P][P]
  uint16_t* w16;     // reduced-precision modes: 16-bit copies of the R-MSA layers' qkv / proj weights (4 D^2 per layer)
  uint16_t* wcr16;   // ... and of CR-MSA's inner qkv / proj weights (4 D^2), bf16 / fp16 modes
  uint16_t *rep16, *repo16;   // 16-bit representatives [k * 64, D] and their attention output (the fused 16-bit inner MSA)
  float* cr_part;    // crmsa_region4_kernel: partial records of the region quarters
  int* cr_cnt;       // ... and the 64 arrival counters (zeroed by an R-MSA GEMM of the same forward)
  int* proj_cnt;     // rmsa_fused_kernel<.., PROJ>: one arrival counter per region (zeroed by the layer's LayerNorm + partition)
  float* cr_pstat;   // row records of the last R-MSA layer's projection slabs [N][D / 64][2 + k] (crmsa_combine_parts_kernel)
  size_t bytes;
};

// One carve function used for both the size query (base = null) and the real carve.
Workspace carve(const rrt_encoder_desc& d, int64_t N, const rrt_grid& g, const rrt_grid& g8, char* base) {
  Workspace w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? (float*)(base + off) : nullptr;
    off = align_up(off + nfloat * sizeof(float), 256);
    return p;
  };
  const size_t D = d.dim;
  const size_t Np = (size_t)g.H * g.H, Np8 = (size_t)g8.H * g8.H;
  const size_t R8 = (size_t)g8.regions_side * g8.regions_side;
  // the weight images first, so that their place does not depend on n_tokens (desc.weights16_valid)
  if (d.n_rmsa_layers > 0) w.w16 = (uint16_t*)take((size_t)d.n_rmsa_layers * 4 * D * D);     // (F32X3: (hi, lo) pairs = 4 bytes per weight)
  if (d.cr_msa) w.wcr16 = (uint16_t*)take(2 * D * D);                                         // 4 D^2 16-bit weights
  if (d.n_rmsa_layers > 0) {
    // w16: its place does not depend on n_tokens (desc.weights16_valid); carved in every mode (4 D^2 floats
    // = 4 MiB per layer at D = 512): the size must not depend on desc.compute, which callers flip between calls on one
    // workspace
    w.uo = take(Np * D);
    w.qkv = take(Np * 3 * D);
    w.xa = take((size_t)N * D);
    if (d.n_rmsa_layers > 1 || d.ffn) w.xb = take((size_t)N * D);
    w.proj_cnt = (int*)take((size_t)g.regions_side * g.regions_side);
    if (d.epeg && d.epeg_type != RRT_EPEG_ATTN) w.pe_out = take(Np * D);
    if (d.epeg && d.epeg_2d && d.epeg_type == RRT_EPEG_ATTN) {
      const size_t sm = attn_scoremap_scratch_floats(g.regions_side * g.regions_side, g.s * g.s, d.n_heads, d.epeg_k, 1);
      if (sm) w.smap = take(sm);
    }
  }
  if (d.ffn) {
    if (!w.xa) w.xa = take((size_t)N * D);
    if (!w.xb) w.xb = take((size_t)N * D);
    w.ffn_ln = take((size_t)N * D);
    w.ffn_hid = take((size_t)N * d.ffn_hidden);
  }
  if (d.cr_msa) {
    const size_t k = d.crmsa_k;
    w.mean_rstd = take((size_t)N * 2);
    w.logits = take(Np8 * k);
    w.wdisp = take(Np8 * k);
    w.rep = take(k * R8 * D);
    w.rep_qkv = take(k * R8 * 3 * D);
    w.rep_o = take(k * R8 * D);
    w.rep2 = take(k * R8 * D);
    w.rep16 = (uint16_t*)take(k * R8 * D / 2);
    w.repo16 = (uint16_t*)take(k * R8 * D / 2);
    if (d.n_rmsa_layers > 0) {
      w.cr_part = take(crmsa_region4_scratch_floats(to_dev(g8), d.crmsa_k));
      w.cr_cnt = (int*)take(64);
      if (D % 64 == 0) w.cr_pstat = take(crmsa_parts_floats(N, (int)D, d.crmsa_k));
    }
    if (d.crmsa_mlp) {
      w.v8 = take(Np8 * D);
      w.hid = take(Np8 * (D / 4));
    }
  }
  if (d.pos) w.xp = take((size_t)N * D);
  w.bytes = off ? off : 256;
  return w;
}

int check_desc(const rrt_encoder_desc* d, int64_t N) {
  if (!d || N <= 0) return RRT_E_INVALID;
  if (d->dim <= 0 || d->dim % 32 != 0) return unsupported("dim must be a positive multiple of 32");
  if (d->dim > 2048) return unsupported("dim > 2048");
  if (d->n_rmsa_layers < 0 || d->n_rmsa_layers > RRT_MAX_RMSA_LAYERS) return unsupported("n_layers-1 out of range");
  if (d->n_rmsa_layers > 0) {
    if (d->n_heads <= 0 || d->dim % d->n_heads != 0) return unsupported("n_heads must divide dim");
    if (d->epeg && (d->epeg_k <= 0 || d->epeg_k % 2 == 0)) return unsupported("epeg_k must be odd");
    if (d->epeg_type < RRT_EPEG_ATTN || d->epeg_type > RRT_EPEG_VALUE_AF) return unsupported("epeg_type must be attn / value_bf / value_af");
    if (d->epeg && (d->epeg_2d || d->epeg_type != RRT_EPEG_ATTN) && d->epeg_k > 63) return unsupported("epeg ablations: epeg_k <= 63");
    if (d->region_size <= 0 && d->region_num <= 0) return unsupported("region_num must be positive");
  }
  if (d->cr_msa) {
    if (d->crmsa_k <= 0 || d->crmsa_k > RRT_MAX_CRMSA_K) return unsupported("crmsa_k must be in [1,8]");
    if (d->crmsa_heads <= 0 || d->dim % d->crmsa_heads != 0) return unsupported("crmsa_heads must divide dim");
  }
  if (d->ffn) {
    if (d->ffn_hidden <= 0 || d->ffn_hidden % 32 != 0)
      return unsupported("ffn: hidden width int(dim * mlp_ratio) must be a positive multiple of 32");
    if (d->ffn_act != RRT_ACT_GELU && d->ffn_act != RRT_ACT_RELU) return unsupported("ffn_act must be gelu or relu");
  }
  if (d->pos) {
    if (d->pos != RRT_POS_PEG && d->pos != RRT_POS_PPEG) return unsupported("pos must be none / peg / ppeg");
    if (d->peg_k <= 0 || d->peg_k % 2 == 0 || d->peg_k > 11) return unsupported("peg_k must be odd and <= 11");
    if (d->pos_pos != -1 && d->pos_pos != 0) return unsupported("pos_pos must be -1 or 0");
  }
  if (N > (int64_t)4000000) return unsupported("bag larger than 4e6 tokens");
  if (d->compute < 0 || d->compute > RRT_COMPUTE_F32X3) return unsupported("compute must be RRT_COMPUTE_F32/BF16/F16/F32X3");
  return RRT_OK;
}

hipError_t inner_attention(const float* u, int n_regions, int P, const rrt_attn_weights& w, int dim,
                           int heads, int epeg_k, float* qkv, float* o, int prec, bool solo, hipStream_t st) {
  const int M = n_regions * P;
  LinearEpilogue ep{};
  ep.prec = prec;
  ep.solo = solo;
  ep.bias = w.qkv_b;
  ep.q_cols = dim;
  ep.q_scale = 1.0f / sqrtf((float)(dim / heads));   // head_dim ** -0.5, modules/rmsa.py:65,103
  hipError_t e = launch_linear(u, w.qkv_w, qkv, M, 3 * dim, dim, ep, st);
  if (e != hipSuccess) return e;
  return launch_region_attention(qkv, epeg_k > 0 ? w.pe_w : nullptr, o, n_regions, P, dim, heads,
                                 epeg_k, st);
}

}  // namespace

GridDev to_dev(const rrt_grid& g) {
  GridDev d;
  d.L = (int)g.L;
  d.H = g.H;
  d.s = g.s;
  d.rs = g.regions_side;
  d.P = g.s * g.s;
  d.Np = g.H * g.H;
  d.inv_H = 1.0f / (float)d.H;
  d.inv_s = 1.0f / (float)d.s;
  d.inv_rs = 1.0f / (float)d.rs;
  d.inv_P = 1.0f / (float)d.P;
  d.Rt = d.rs * d.rs;
  return d;
}

extern "C" {

int rrt_abi_version(void) { return RRT_ABI_VERSION; }

const char* rrt_strerror(int code) {
  switch (code) {
    case RRT_OK: return "ok";
    case RRT_E_INVALID: return "invalid argument (null pointer or non-positive size)";
    case RRT_E_UNSUPPORTED: return g_detail[0] ? g_detail : "unsupported configuration";
    case RRT_E_WORKSPACE: return "workspace too small (see rrt_encoder_workspace_size)";
    case RRT_E_HANDOVER: return "an earlier merged R-MSA launch gave up its in-launch hand-over wait; outputs in flight are invalid "
                                "(synchronise, then rrt_device_error(1) to clear)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

int rrt_region_grid(int64_t L, int32_t region_num, int32_t region_size, int32_t min_region_num,
                    float min_region_ratio, rrt_grid* out) {
  if (!out || L <= 0) return RRT_E_INVALID;
  if (region_size <= 0 && region_num <= 0) return RRT_E_INVALID;
  int64_t H = ceil_sqrt(L), s;
  if (region_size > 0) {
    H += ((-H) % region_size + region_size) % region_size;
    s = region_size;
  } else {
    H += ((-H) % region_num + region_num) % region_num;
    s = H / region_num;
  }
  int64_t add = H * H - L;
  // "if padding much, give up region attention" (ablation escape hatch, rmsa.py:191-196);
  // evaluated in double like the reference's Python floats
  if ((double)add > (double)L / ((double)min_region_ratio + 1e-8) || L < min_region_num) {
    H = ceil_sqrt(L);
    H += H % 2;
    add = H * H - L;
    s = H;
  }
  if (H > 46340) return RRT_E_INVALID;
  out->L = L;
  out->H = (int32_t)H;
  out->s = (int32_t)s;
  out->regions_side = (int32_t)(H / s);
  out->add = add;
  return RRT_OK;
}

int rrt_encoder_workspace_size(const rrt_encoder_desc* desc, int64_t n_tokens, size_t* bytes) {
  if (!bytes) return RRT_E_INVALID;
  int rc = check_desc(desc, n_tokens);
  if (rc) return rc;
  rrt_grid g{}, g8{};
  if (desc->n_rmsa_layers > 0) {
    rc = rrt_region_grid(n_tokens, desc->region_num, desc->region_size, desc->min_region_num,
                         desc->min_region_ratio, &g);
    if (rc) return rc;
  }
  rc = rrt_region_grid(n_tokens, 8, 0, 0, 0.f, &g8);
  if (rc) return rc;
  *bytes = carve(*desc, n_tokens, g, g8, nullptr).bytes;
  return RRT_OK;
}

// Mirrors the kernel choice of encoder_forward's R-MSA layer loop (same predicates, same order); measurement only.
int rrt_encoder_plan(const rrt_encoder_desc* desc, int64_t n_tokens, int32_t* flags) {
  if (!flags) return RRT_E_INVALID;
  *flags = 0;
  int rc = check_desc(desc, n_tokens);
  if (rc) return rc;
  if (desc->n_rmsa_layers <= 0) return RRT_OK;
  rrt_grid g{};
  rc = rrt_region_grid(n_tokens, desc->region_num, desc->region_size, desc->min_region_num, desc->min_region_ratio, &g);
  if (rc) return rc;
  const GridDev gd = to_dev(g);
  const int D = desc->dim, ek = desc->epeg ? desc->epeg_k : 0;
  const bool want_x3 = desc->compute == RRT_COMPUTE_F32X3;
  const int compute = want_x3 ? RRT_COMPUTE_F32 : desc->compute;
  const bool epeg_variant = desc->epeg && (desc->epeg_2d || desc->epeg_type != RRT_EPEG_ATTN);
  if (epeg_variant) return RRT_OK;
  const bool rows_ok = rmsa_fused_supported_rows(gd.Np, D);
  if (compute != RRT_COMPUTE_F32 && D % 64 == 0 && rmsa_fused16_supported(gd.P, D, desc->n_heads, ek) && rows_ok) {
    *flags = RRT_PLAN_FUSED16;
    return RRT_OK;
  }
  if (want_x3 && D % 256 == 0 && rmsa_fused_x3_supported(gd.P, D, desc->n_heads, ek) && rows_ok) {
    *flags = RRT_PLAN_FUSED_X3;
    return RRT_OK;
  }
  if (rmsa_fused_supported(gd.P, D, desc->n_heads, ek) && rows_ok) {
    *flags = RRT_PLAN_FUSED;
    if (compute == RRT_COMPUTE_F32 && rmsa_fused_proj_supported(gd.rs * gd.rs, gd.P, D, desc->n_heads, ek, compute)) {
      *flags |= RRT_PLAN_FUSED_PROJ;
      rrt_grid g8{};
      if (rrt_region_grid(n_tokens, 8, 0, 0, 0.f, &g8) == RRT_OK && desc->solo != 0 && desc->crmsa_k <= 4 && desc->cr_msa &&
          !desc->crmsa_mlp && !desc->ffn && D % 64 == 0 &&
          crmsa_combine_parts_supported(D, desc->crmsa_k, to_dev(g8)))
        *flags |= RRT_PLAN_CRMSA_PARTS;
    }
  }
  return RRT_OK;
}

// Serialises the MFMA-bound R-MSA core (fused kernel, or qkv linear + attention) of forwards that run
// concurrently on different streams: two of them co-running just time-slice the matrix pipes (each takes
// twice as long), while one of them next to another bag's memory- and latency-bound kernels overlaps well.
struct rrt_phase_gate {
  hipEvent_t done;     // completion of the most recent gated phase (any stream)
  bool armed;
};

// TransLayer's optional FFN (rrt.py:127-129): xo = xi + fc2(act(fc1(LN2(xi)))).  LN2 -> GEMM with the activation in its
// epilogue -> GEMM with the residual in its epilogue (identity slot -> token map).
static int ffn_apply(const rrt_encoder_desc* desc, const rrt_attn_weights& lw, const float* xi, float* xo,
                     const Workspace& ws, int64_t N, hipStream_t st) {
  if (!lw.norm2_w || !lw.norm2_b || !lw.fc1_w || !lw.fc1_b || !lw.fc2_w || !lw.fc2_b) return RRT_E_INVALID;
  const int D = desc->dim;
  GridDev gid{};
  const int Hs = (int)ceil_sqrt(N);
  gid.L = (int)N;
  gid.H = gid.s = Hs;
  gid.rs = 1;
  gid.P = gid.Np = Hs * Hs;
  gid.inv_H = gid.inv_s = 1.0f / (float)Hs;
  gid.inv_rs = 1.0f;
  gid.inv_P = 1.0f / (float)gid.P;
  gid.Rt = 1;
  hipError_t fe = launch_layernorm(xi, nullptr, lw.norm2_w, lw.norm2_b, ws.ffn_ln, (int)N, D, st);
  if (fe != hipSuccess) return (int)fe;
  LinearEpilogue e1{};
  e1.prec = desc->compute;
  e1.bias = lw.fc1_b;
  e1.act = desc->ffn_act;
  fe = launch_linear(ws.ffn_ln, lw.fc1_w, ws.ffn_hid, (int)N, desc->ffn_hidden, D, e1, st);
  if (fe != hipSuccess) return (int)fe;
  LinearEpilogue e2{};
  e2.prec = desc->compute;
  e2.bias = lw.fc2_b;
  e2.resid = xi;
  e2.g = gid;
  return (int)launch_linear(ws.ffn_hid, lw.fc2_w, xo, (int)N, D, desc->ffn_hidden, e2, st);
}

// CR-MSA's first pass as a by-product of the last R-MSA layer's projection slabs: that layer's output must BE CR-MSA's
// input (no FFN behind the attention, not the batch entry point's per-bag stop), phi a plain parameter, 64-column slabs
// ... and the forward has the GPU to itself (rrt_encoder_desc.solo): the statistics cost the merged launch ~3 us of matrix-pipe
// time and save ~6 us of the latency-bound CR-MSA front -- one bag in flight 0.2238 -> 0.2218 ms, but with four bags in
// flight the tail hides behind other bags' tails anyway and the matrix pipe is the scarce thing: 5.26 k against 5.31 k
// slides/s (round 5, same box); likewise k > 4 representatives (twice the record) stay with the region kernels
// smallest CR-MSA region (tokens) that takes crmsa_stream4_kernel instead of crmsa_region4_kernel (A/B builds: -DRRT_STREAM4_MIN_P=16)
#ifndef RRT_STREAM4_MIN_P
#define RRT_STREAM4_MIN_P 145
#endif
static bool crmsa_parts_wanted(const rrt_encoder_desc& d, const Workspace& ws, const GridDev& g8, bool stops_before_crmsa) {
  return d.solo != 0 && d.cr_msa && !d.crmsa_mlp && !d.ffn && d.crmsa_k <= 4 && !stops_before_crmsa && ws.cr_pstat != nullptr &&
         crmsa_combine_parts_supported(d.dim, d.crmsa_k, g8);
}

static int encoder_forward(const rrt_encoder_desc* desc_in, const rrt_encoder_weights* w, const float* x,
                           float* y, int64_t n_tokens, void* workspace, size_t workspace_bytes,
                           void* stream, void** events, rrt_phase_gate* gate = nullptr, float* rmsa_out = nullptr,
                           uint16_t* y16 = nullptr, bool* y16_done = nullptr) {
  // y16 / y16_done (the slide classifier, 16-bit modes): where the forward's last kernel is CR-MSA's dispatch + LayerNorm it
  // also leaves the output rows as 16-bit values there and sets the flag
  if (y16_done) *y16_done = false;
  if (!desc_in || !w || !x || !y || x == y) return RRT_E_INVALID;
  int rc = check_desc(desc_in, n_tokens);
  if (rc) return rc;
  if (handover_err_peek(false)) return RRT_E_HANDOVER;    // (a host read of pinned memory; rrt_hip.h, rrt_device_error)
  // RRT_COMPUTE_F32X3 concerns the R-MSA layers' two big projections (below); every other GEMM of the call is exact fp32
  rrt_encoder_desc dloc = *desc_in;
  const bool want_x3 = dloc.compute == RRT_COMPUTE_F32X3;
  if (want_x3) dloc.compute = RRT_COMPUTE_F32;
  const rrt_encoder_desc* const desc = &dloc;
  hipStream_t st = (hipStream_t)stream;
  const int D = desc->dim;
  const int64_t N = n_tokens;
  rrt_grid g{}, g8{};
  if (desc->n_rmsa_layers > 0) {
    rc = rrt_region_grid(N, desc->region_num, desc->region_size, desc->min_region_num,
                         desc->min_region_ratio, &g);
    if (rc) return rc;
  }
  // CR-MSA never receives region_num / region_size / min_region_* (modules/rrt.py:148): grid 8x8
  rc = rrt_region_grid(N, 8, 0, 0, 0.f, &g8);
  if (rc) return rc;
  Workspace ws = carve(*desc, N, g, g8, nullptr);
  if (!workspace || workspace_bytes < ws.bytes) return RRT_E_WORKSPACE;
  ws = carve(*desc, N, g, g8, (char*)workspace);

  hipError_t e = hipSuccess;
#define RRT_TRY(call)            \
  do {                           \
    e = (call);                  \
    if (e != hipSuccess) return (int)e; \
  } while (0)
  // optional stage-boundary events (bench.py / profiling): events[i] recorded after stage i-1
#define RRT_MARK(i)                                                          \
  do {                                                                       \
    if (events && events[i]) RRT_TRY(hipEventRecord((hipEvent_t)events[i], st)); \
  } while (0)
  RRT_MARK(RRT_EV_START);

  auto ffn_block = [&](const rrt_attn_weights& lw, const float* xi, float* xo) -> int {
    return ffn_apply(desc, lw, xi, xo, ws, N, st);
  };

  const float* xin = x;   // current activations [N, D]
  // PEG / PPEG (ablation): before the first layer (pos_pos = -1) or before layer index 1 (pos_pos = 0), rrt.py:181-187
  auto pos_embed = [&]() -> int {
    if (!w->pos_w[0] || (desc->pos == RRT_POS_PPEG && (!w->pos_w[1] || !w->pos_w[2]))) return RRT_E_INVALID;
    hipError_t pe = launch_peg(xin, w->pos_w, w->pos_b, ws.xp, (int)N, D, desc->peg_k, desc->peg_1d,
                               desc->pos == RRT_POS_PPEG, st);
    if (pe != hipSuccess) return (int)pe;
    xin = ws.xp;
    return RRT_OK;
  };
  if (desc->pos && desc->pos_pos == -1) {
    rc = pos_embed();
    if (rc) return rc;
  }
  // Reduced-precision modes on regions the 16-bit fused kernel covers: every tensor that is only a matrix-core
  // operand (LayerNorm output u, the weights, the attention output O) lives in HBM in 16 bits.  The weights are
  // cast once per call (one launch for all layers) unless the caller vouches for the images already in the
  // workspace (desc.weights16_valid; the ABI itself keeps no state).
  // EPEG ablations (epeg_2d, epeg_type = value_*): unfused path with their own kernels (epeg_variants.hip)
  const bool epeg_variant = desc->epeg && (desc->epeg_2d || desc->epeg_type != RRT_EPEG_ATTN);
  // The images are written whenever a reduced / emulated mode is asked for and the caller does not vouch for them --
  // whether or not THIS bag's regions take the 16-bit kernels: the workspace (and the caller's validity key) outlives
  // the bag, and the next, smaller bag on it may take them (a 20 k-token bag followed by a 9 k-token one).
  bool lowp16 = false;
  // CR-MSA's inner MSA over the 64 k representatives on the same 16-bit kernels (one fused launch + one GEMM instead
  // of GEMM + attention + GEMM on fp32 data): head dim 64 only (crmsa_heads = dim / 64)
  const bool inner16 = desc->cr_msa && desc->compute != RRT_COMPUTE_F32 && D % 64 == 0 &&
                       rmsa_fused16_supported(64, D, desc->crmsa_heads, 0);
  if (desc->compute != RRT_COMPUTE_F32 && D % 64 == 0) {
    Cast16Jobs jobs{};
    if (desc->n_rmsa_layers > 0 && !epeg_variant) {
      const GridDev gd = to_dev(g);
      lowp16 = rmsa_fused16_supported(gd.P, D, desc->n_heads, desc->epeg ? desc->epeg_k : 0) &&
               rmsa_fused_supported_rows(gd.Np, D);
      for (int li = 0; li < desc->n_rmsa_layers; ++li) {
        const rrt_attn_weights& lw = w->rmsa[li];
        if (!lw.qkv_w || !lw.proj_w) return RRT_E_INVALID;
        uint16_t* base = ws.w16 + (size_t)li * 4 * D * D;
        jobs.src[jobs.count] = lw.qkv_w; jobs.dst[jobs.count] = base; jobs.n4[jobs.count++] = (size_t)3 * D * D / 4;
        jobs.src[jobs.count] = lw.proj_w; jobs.dst[jobs.count] = base + (size_t)3 * D * D; jobs.n4[jobs.count++] = (size_t)D * D / 4;
      }
    }
    if (inner16) {
      if (!w->crmsa.qkv_w || !w->crmsa.proj_w) return RRT_E_INVALID;
      jobs.src[jobs.count] = w->crmsa.qkv_w; jobs.dst[jobs.count] = ws.wcr16; jobs.n4[jobs.count++] = (size_t)3 * D * D / 4;
      jobs.src[jobs.count] = w->crmsa.proj_w; jobs.dst[jobs.count] = ws.wcr16 + (size_t)3 * D * D; jobs.n4[jobs.count++] = (size_t)D * D / 4;
    }
    if (jobs.count && !desc->weights16_valid) RRT_TRY(launch_cast16(jobs, desc->compute, st));
  }
  // RRT_COMPUTE_F32X3: the qkv and proj GEMMs of the R-MSA layers emulated in fp32 on the bf16 matrix cores (operands
  // as (hi, lo) bf16 pairs, three MFMAs per product; rmsa_fused_x3.hip, cast16.hip); attention and everything else as F32
  bool x3 = false;
  if (want_x3 && desc->n_rmsa_layers > 0 && !epeg_variant) {
    const GridDev gd = to_dev(g);
    // dim % 256 == 0 only: at other widths the LayerNorm-type kernels keep their lane-predicated column guards, and those
    // mis-summed now and then with split-kernel waves co-resident (root cause not established) -- such
    // widths get the exact fp32 kernels instead (more accurate than what was asked for)
    x3 = D % 256 == 0 && rmsa_fused_x3_supported(gd.P, D, desc->n_heads, desc->epeg ? desc->epeg_k : 0) &&
         rmsa_fused_supported_rows(gd.Np, D);
    Cast16Jobs jobs{};
    for (int li = 0; li < desc->n_rmsa_layers; ++li) {
      const rrt_attn_weights& lw = w->rmsa[li];
      if (!lw.qkv_w || !lw.proj_w) return RRT_E_INVALID;
      uint16_t* base = ws.w16 + (size_t)li * 8 * D * D;          // 4 D^2 weights x 2 bf16
      jobs.src[jobs.count] = lw.qkv_w; jobs.dst[jobs.count] = base; jobs.n4[jobs.count++] = (size_t)3 * D * D / 4;
      jobs.src[jobs.count] = lw.proj_w; jobs.dst[jobs.count] = base + (size_t)6 * D * D; jobs.n4[jobs.count++] = (size_t)D * D / 4;
    }
    if (!desc->weights16_valid) RRT_TRY(launch_cast_split(jobs, st));
  }
  bool parts_done = false;      // the last R-MSA layer left CR-MSA's row records in ws.cr_pstat
  // ---- R-MSA TransLayers: x = x + unpart(InnerAttention(part(pad(LN(x)))))  (rrt.py:117-125)
  for (int li = 0; li < desc->n_rmsa_layers; ++li) {
    if (li == 1 && desc->pos && desc->pos_pos == 0) {
      rc = pos_embed();
      if (rc) return rc;
    }
    const rrt_attn_weights& lw = w->rmsa[li];
    if (!lw.norm_w || !lw.norm_b || !lw.qkv_w || !lw.proj_w || !lw.proj_b) return RRT_E_INVALID;
    if (desc->epeg && !lw.pe_w) return RRT_E_INVALID;
    const GridDev gd = to_dev(g);
    // without FFN the layers ping-pong xa / xb; with it attention writes xa and the FFN writes xb
    // (rmsa_out: the batch entry point takes the R-MSA layers' result of each bag in its own buffer)
    const bool to_out = rmsa_out != nullptr && li == desc->n_rmsa_layers - 1;
    float* xout = desc->ffn ? ws.xa : (to_out ? rmsa_out : ((li & 1) ? ws.xb : ws.xa));
    float* const fout = to_out ? rmsa_out : ws.xb;      // the layer's FFN output
    const int ek = desc->epeg ? desc->epeg_k : 0;
    if (lowp16) {
      // u (16-bit) in the uo buffer, O (16-bit) in the qkv buffer; qkv, scores and probabilities never leave the CU
      uint16_t* u16 = (uint16_t*)ws.uo;
      uint16_t* o16 = (uint16_t*)ws.qkv;
      const uint16_t* wq16 = ws.w16 + (size_t)li * 4 * D * D;
      // round 6: bags of at least two rounds of (pair, head) items (N = 30000 at region_num = 16: four) CAN take the
      // out-projection as a phase of the pair launch's blocks (rmsa_pair16.hip, PROJ; bit-identical, tested) -- built on the
      // round-5 review's request and MEASURED SLOWER than the two launches: 109-116 us against 64.5 + 35.6 at N = 30000
      // (profiles/r06_trace_pair16_proj.txt: a slab costs a block 25 K cycles -- 13 K of DMA-issue-bound K loop, the same
      // bound the separate projection runs at with two blocks per CU hiding each other's prologue and epilogue -- plus 4 K
      // waiting for the write-through O stores; the block owns its CU, so nothing overlaps the slab).  Off unless a tuning
      // build asks for it (RRT_PAIR16_PROJ=1).
      static const bool want_merged16 = rrt_tune_env("RRT_PAIR16_PROJ") != nullptr;
      const bool merged16 = want_merged16 && ws.proj_cnt != nullptr &&
                            rmsa_pair16_proj_supported(gd.rs * gd.rs, gd.P, D, desc->n_heads, ek);
      RRT_TRY(launch_ln_partition16(xin, lw.norm_w, lw.norm_b, u16, D, gd, desc->compute, st, merged16 ? ws.proj_cnt : nullptr,
                                    merged16 ? gd.rs * gd.rs / 2 : 0));
      rrt_phase_gate* const gt16 = (gate && gd.P > 112) ? gate : nullptr;
      if (gt16 && gt16->armed) RRT_TRY(hipStreamWaitEvent(st, gt16->done, 0));
      if (li == 0) RRT_MARK(RRT_EV_LN_PARTITION);
      if (merged16) {
        PairProj pj{};
        pj.Wp = wq16 + (size_t)3 * D * D;
        pj.bias = lw.proj_b;
        pj.resid = xin;
        pj.out = xout;
        pj.cnt = ws.proj_cnt;
        pj.zero64 = ws.cr_cnt;
        pj.g = gd;
        RRT_TRY(launch_rmsa_pair16_proj(u16, wq16, lw.qkv_b, desc->epeg ? lw.pe_w : nullptr, o16, gd.rs * gd.rs, gd.P, D,
                                        desc->n_heads, ek, desc->compute, pj, st));
        if (gt16) { RRT_TRY(hipEventRecord(gt16->done, st)); gt16->armed = true; }
        if (li == 0) { RRT_MARK(RRT_EV_QKV); RRT_MARK(RRT_EV_ATTN); RRT_MARK(RRT_EV_PROJ); }
        xin = xout;
        if (desc->ffn) {
          rc = ffn_block(lw, xout, fout);
          if (rc) return rc;
          xin = fout;
        }
        continue;
      }
      RRT_TRY(launch_rmsa_fused16(u16, wq16, lw.qkv_b, desc->epeg ? lw.pe_w : nullptr, o16, gd.rs * gd.rs, gd.P, D,
                                  desc->n_heads, ek, desc->compute, st));
      if (gt16) { RRT_TRY(hipEventRecord(gt16->done, st)); gt16->armed = true; }
      if (li == 0) { RRT_MARK(RRT_EV_QKV); RRT_MARK(RRT_EV_ATTN); }
      LinearEpilogue ep{};
      ep.prec = desc->compute;
      ep.bias = lw.proj_b;
      ep.resid = xin;
      ep.g = gd;
      ep.zero64 = ws.cr_cnt;
      ep.solo = desc->solo != 0;          // (picks the tile shape: one bag in flight / several, launch_linear16)
      RRT_TRY(launch_linear16(o16, wq16 + (size_t)3 * D * D, xout, gd.Np, D, D, ep, st));
      if (li == 0) RRT_MARK(RRT_EV_PROJ);
      xin = xout;
      if (desc->ffn) {
        rc = ffn_block(lw, xout, fout);
        if (rc) return rc;
        xin = fout;
      }
      continue;
    }
    if (x3) {
      // u and O as split images (4 bytes per element: the uo / qkv buffers as they are)
      const uint16_t* wq = ws.w16 + (size_t)li * 8 * D * D;
      RRT_TRY(launch_ln_partition_split(xin, lw.norm_w, lw.norm_b, ws.uo, D, gd, st));
      rrt_phase_gate* const gt3 = (gate && gd.P > 112) ? gate : nullptr;
      if (gt3 && gt3->armed) RRT_TRY(hipStreamWaitEvent(st, gt3->done, 0));
      if (li == 0) RRT_MARK(RRT_EV_LN_PARTITION);
      RRT_TRY(launch_rmsa_fused_x3(ws.uo, wq, lw.qkv_b, desc->epeg ? lw.pe_w : nullptr, ws.qkv, gd.rs * gd.rs, gd.P, D,
                                   desc->n_heads, ek, st));
      if (gt3) { RRT_TRY(hipEventRecord(gt3->done, st)); gt3->armed = true; }
      if (li == 0) { RRT_MARK(RRT_EV_QKV); RRT_MARK(RRT_EV_ATTN); }
      LinearEpilogue ep{};
      ep.bias = lw.proj_b;
      ep.resid = xin;
      ep.g = gd;
      ep.zero64 = ws.cr_cnt;
      RRT_TRY(launch_linear_split(ws.qkv, wq + (size_t)6 * D * D, xout, gd.Np, D, D, ep, st));
      if (li == 0) RRT_MARK(RRT_EV_PROJ);
      xin = xout;
      if (desc->ffn) {
        rc = ffn_block(lw, xout, fout);
        if (rc) return rc;
        xin = fout;
      }
      continue;
    }
    // the exact fp32 path of bags that fill the chip twice over: fused R-MSA kernel with the out-projection as a
    // later phase of the same launch (rmsa_fused.hip, PROJ); its arrival counters are zeroed by LayerNorm + partition
    const bool merged = !epeg_variant && desc->compute == RRT_COMPUTE_F32 && ws.proj_cnt != nullptr &&
                        rmsa_fused_supported_rows(gd.Np, D) &&
                        rmsa_fused_proj_supported(gd.rs * gd.rs, gd.P, D, desc->n_heads, ek, desc->compute);
    const bool parts = merged && li == desc->n_rmsa_layers - 1 && !epeg_variant &&
                       rmsa_fused_supported(gd.P, D, desc->n_heads, ek) &&
                       crmsa_parts_wanted(*desc, ws, to_dev(g8), rmsa_out != nullptr);
    if (parts && (!w->crmsa.norm_w || !w->crmsa.norm_b || !w->phi)) return RRT_E_INVALID;
    RRT_TRY(launch_ln_partition(xin, lw.norm_w, lw.norm_b, ws.uo, D, gd, st, merged ? ws.proj_cnt : nullptr,
                                merged ? gd.rs * gd.rs : 0));
    if (epeg_variant) {
      if (!lw.pe_w) return RRT_E_INVALID;
      const int nreg = gd.rs * gd.rs;
      {
        LinearEpilogue ep{};
        ep.prec = desc->compute;
        ep.bias = lw.qkv_b;
        ep.q_cols = D;
        ep.q_scale = 1.0f / sqrtf((float)(D / desc->n_heads));
        RRT_TRY(launch_linear(ws.uo, lw.qkv_w, ws.qkv, gd.Np, 3 * D, D, ep, st));
      }
      if (desc->epeg_type == RRT_EPEG_ATTN) {           // epeg_2d: k x k stencil over the score map
        // (regions of more than ~180 tokens: the score maps live in the workspace instead of the LDS)
        RRT_TRY(launch_attn_scoremap(ws.qkv, lw.pe_w, ws.uo, ws.smap, nreg, gd.P, D, desc->n_heads, desc->epeg_k, st));
      } else {
        RRT_TRY(launch_value_pe(ws.qkv, lw.pe_w, lw.pe_b, ws.pe_out, nreg, gd.P, gd.s, D, desc->n_heads, desc->epeg_k,
                                desc->epeg_2d, st));
        if (desc->epeg_type == RRT_EPEG_VALUE_BF)       // v += pe before attn @ v (rmsa.py:114-118)
          RRT_TRY(launch_add_cols(ws.qkv + 2 * D, ws.pe_out, (size_t)gd.Np, D, 3 * D, st));
        RRT_TRY(launch_region_attention(ws.qkv, nullptr, ws.uo, nreg, gd.P, D, desc->n_heads, 0, st));
        if (desc->epeg_type == RRT_EPEG_VALUE_AF)       // x += pe after it (rmsa.py:124-129)
          RRT_TRY(launch_add_cols(ws.uo, ws.pe_out, (size_t)gd.Np, D, D, st));
      }
      LinearEpilogue ep{};
      ep.prec = desc->compute;
      ep.bias = lw.proj_b;
      ep.resid = xin;
      ep.g = gd;
      ep.zero64 = ws.cr_cnt;
      RRT_TRY(launch_linear(ws.uo, lw.proj_w, xout, gd.Np, D, D, ep, st));
      xin = xout;
      if (desc->ffn) {
        rc = ffn_block(lw, xout, fout);
        if (rc) return rc;
        xin = fout;
      }
      continue;
    }
    const bool fused = rmsa_fused_supported(gd.P, D, desc->n_heads, ek) && rmsa_fused_supported_rows(gd.Np, D);
    // the gate only pays for launches that fill the matrix pipes of the whole chip on their own: the fused
    // kernel on regions of >= 113 tokens (measured on the configs[4] mix: gating small or unfused bags costs 5 %)
    rrt_phase_gate* const gt = (gate && fused && gd.P > 112) ? gate : nullptr;
    if (gt && gt->armed) RRT_TRY(hipStreamWaitEvent(st, gt->done, 0));
    if (li == 0) RRT_MARK(RRT_EV_LN_PARTITION);    // after the gate: the mark brackets the kernel, not the wait
    if (fused && merged) {
      // ... and the out-projection + un-partition + residual as a later phase of the same launch's blocks (fp32,
      // bags of >= two rounds of (region, head) items): one launch per R-MSA layer
      FusedProj pj{};
      pj.Wp = lw.proj_w;
      pj.bias = lw.proj_b;
      pj.resid = xin;
      pj.out = xout;
      pj.cnt = ws.proj_cnt;
      pj.zero64 = ws.cr_cnt;
      pj.g = gd;
      // the LAST R-MSA layer feeding CR-MSA directly: its slabs also leave LayerNorm 2's statistics and the logits' dot
      // products of every row (x1 is in their registers), and CR-MSA's first pass shrinks to the combine
      if (parts) {
        pj.part = ws.cr_pstat;
        pj.ln_g = w->crmsa.norm_w;
        pj.phi = w->phi;
        pj.k = desc->crmsa_k;
        parts_done = true;
      }
      RRT_TRY(launch_rmsa_fused(ws.uo, lw.qkv_w, lw.qkv_b, desc->epeg ? lw.pe_w : nullptr, ws.qkv,
                                gd.rs * gd.rs, gd.P, D, desc->n_heads, ek, desc->compute, st, nullptr, &pj));
      if (gt) { RRT_TRY(hipEventRecord(gt->done, st)); gt->armed = true; }
      if (li == 0) { RRT_MARK(RRT_EV_QKV); RRT_MARK(RRT_EV_ATTN); RRT_MARK(RRT_EV_PROJ); }
      xin = xout;
      if (desc->ffn) {
        rc = ffn_block(lw, xout, fout);
        if (rc) return rc;
        xin = fout;
      }
      continue;
    }
    if (fused) {
      // qkv projection + EPEG + attention in one kernel per (region, head): qkv never reaches HBM.
      // O goes to the qkv workspace (first Np*D floats), u stays in uo.
      RRT_TRY(launch_rmsa_fused(ws.uo, lw.qkv_w, lw.qkv_b, desc->epeg ? lw.pe_w : nullptr, ws.qkv,
                                gd.rs * gd.rs, gd.P, D, desc->n_heads, ek, desc->compute, st));
      static const bool gate_proj = rrt_tune_env("RRT_GATE_PROJ") != nullptr;
      if (gt && !gate_proj) { RRT_TRY(hipEventRecord(gt->done, st)); gt->armed = true; }
      if (li == 0) { RRT_MARK(RRT_EV_QKV); RRT_MARK(RRT_EV_ATTN); }
      LinearEpilogue ep{};
      ep.prec = desc->compute;
      ep.bias = lw.proj_b;
      ep.resid = xin;
      ep.g = gd;
      ep.zero64 = ws.cr_cnt;
      RRT_TRY(launch_linear(ws.qkv, lw.proj_w, xout, gd.Np, D, D, ep, st));
      if (gt && gate_proj) { RRT_TRY(hipEventRecord(gt->done, st)); gt->armed = true; }
      if (li == 0) RRT_MARK(RRT_EV_PROJ);
      xin = xout;
      if (desc->ffn) {
        rc = ffn_block(lw, xout, fout);
        if (rc) return rc;
        xin = fout;
      }
      continue;
    }
    {
      LinearEpilogue ep{};
      ep.prec = desc->compute;
      ep.bias = lw.qkv_b;
      ep.q_cols = D;
      ep.q_scale = 1.0f / sqrtf((float)(D / desc->n_heads));   // head_dim ** -0.5, rmsa.py:65,103
      RRT_TRY(launch_linear(ws.uo, lw.qkv_w, ws.qkv, gd.Np, 3 * D, D, ep, st));
    }
    if (li == 0) RRT_MARK(RRT_EV_QKV);
    RRT_TRY(launch_region_attention(ws.qkv, desc->epeg ? lw.pe_w : nullptr, ws.uo, gd.rs * gd.rs, gd.P, D,
                                    desc->n_heads, desc->epeg ? desc->epeg_k : 0, st));
    if (li == 0) RRT_MARK(RRT_EV_ATTN);
    LinearEpilogue ep{};
    ep.prec = desc->compute;
    ep.bias = lw.proj_b;
    ep.resid = xin;
    ep.g = gd;
    ep.zero64 = ws.cr_cnt;
    RRT_TRY(launch_linear(ws.uo, lw.proj_w, xout, gd.Np, D, D, ep, st));
    if (li == 0) RRT_MARK(RRT_EV_PROJ);
    xin = xout;
    if (desc->ffn) {
      rc = ffn_block(lw, xout, fout);
      if (rc) return rc;
      xin = fout;
    }
  }
  if (rmsa_out != nullptr) {           // batch entry point: stop here, the R-MSA result of this bag in rmsa_out
    if (xin != rmsa_out) RRT_TRY(hipMemcpyAsync(rmsa_out, xin, (size_t)N * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    return RRT_OK;
  }
  const float* x0 = desc->all_shortcut ? x : nullptr;
  if (!w->norm_w || !w->norm_b) return RRT_E_INVALID;
  if (!desc->cr_msa) {
    RRT_TRY(launch_layernorm(xin, x0, w->norm_w, w->norm_b, y, (int)N, D, st));
    RRT_MARK(RRT_EV_END);
    return RRT_OK;
  }
  // ---- CR-MSA TransLayer (rmsa.py:290-337) + all_shortcut + final LayerNorm (rrt.py:190-195)
  const rrt_attn_weights& cw = w->crmsa;
  if (!cw.norm_w || !cw.norm_b || !cw.qkv_w || !cw.proj_w || !cw.proj_b) return RRT_E_INVALID;
  if (desc->crmsa_mlp ? (!w->phi0_w || !w->phi2_w) : !w->phi) return RRT_E_INVALID;
  const GridDev gd8 = to_dev(g8);
  const int k = desc->crmsa_k, R8 = gd8.rs * gd8.rs;
  bool rep16_done = false;
  static const bool no_region_inflight = rrt_tune_env("RRT_NO_REGION_INFLIGHT") != nullptr;
  static const bool region_lowp = rrt_tune_env("RRT_REGION_INFLIGHT_LOWP") != nullptr;      // (A/B: also in the 16-bit modes)
  const bool region_inflight = !no_region_inflight && !desc->solo && !desc->crmsa_mlp && !parts_done && !x3 &&
                               (desc->compute == RRT_COMPUTE_F32 || region_lowp) && crmsa_region_supported(D, k, gd8);
  if (desc->crmsa_mlp) {
    // MLP phi (rmsa.py:248-252,305): v = LN(x1) materialised in region-major order, hidden = v W1^T on
    // the matrix cores, logits = tanh(hidden) W2^T; the combine then runs on the normalised rows
    RRT_TRY(launch_ln_partition(xin, cw.norm_w, cw.norm_b, ws.v8, D, gd8, st));
    LinearEpilogue ep{};
    ep.prec = desc->compute;
    RRT_TRY(launch_linear(ws.v8, w->phi0_w, ws.hid, gd8.Np, D / 4, D, ep, st));
    RRT_TRY(launch_crmsa_mlp_logits(ws.hid, w->phi2_w, ws.logits, gd8.Np, D / 4, k, st));
    RRT_TRY(launch_crmsa_combine(ws.v8, nullptr, nullptr, nullptr, ws.logits, ws.wdisp, ws.rep, inner16 ? ws.rep16 : nullptr,
                                 desc->compute, D, k, gd8, st));
    rep16_done = inner16;
  } else if (parts_done) {
    // LayerNorm 2's statistics and the logits' dot products came out of the projection slabs: one pass over x1, 64 x D / 64
    // independent blocks (crmsa_combine_parts_kernel)
    RRT_TRY(launch_crmsa_combine_parts(xin, ws.cr_pstat, cw.norm_w, cw.norm_b, w->phi, ws.wdisp, ws.rep,
                                       inner16 ? ws.rep16 : nullptr, desc->compute, D, k, gd8, st));
    rep16_done = inner16;
  } else if (region_inflight) {
    // round 6: exact fp32 with SEVERAL bags in flight -- logits + combine as ONE sixteen-wave block per region
    // (crmsa_region_kernel, its k = 1 .. 3 forms with gamma . phi in registers): 64 blocks, so three quarters of the chip stay
    // with the other bags' fused R-MSA launches, which is what the line is made of (the fused launches' union is 187 of the
    // 189 us a bag takes).  Same box, four bags in flight: 5.30-5.31 k -> 5.36-5.38 k slides/s with the round-1 kernel
    // (profiles/r06_region1_in_flight_ab.txt); one bag in flight it loses (20 us on a quarter of the chip): the hint decides.
    // The 16-bit modes keep crmsa_region4 (their chip-wide kernels are short: a 64-block front becomes the critical path).
    RRT_TRY(launch_crmsa_region(xin, cw.norm_w, cw.norm_b, w->phi, nullptr, nullptr, ws.wdisp, ws.rep, k, gd8, st,
                                inner16 ? ws.rep16 : nullptr, desc->compute));
    rep16_done = inner16;
  } else if (ws.cr_cnt && desc->n_rmsa_layers > 0 && crmsa_region4_supported(D, k, gd8) && gd8.P < RRT_STREAM4_MIN_P) {
    // logits + combine in one pass over x1: four blocks per region, the last to arrive merges (crmsa_region4_kernel).
    // Its counters were zeroed by this forward's last R-MSA out-projection.  Regions of more than 144 tokens (8 / 16
    // blocks per region: the kernel covers them, tests) stay with the two chip-wide kernels: measured on MI355X
    // (tools/bench_crmsa.py) the merge of 8-16 partial records per region costs more than the second pass over x1
    // -- N = 15000: 36 vs 24 us, N = 30000: 46-74 vs 39 us.
    RRT_TRY(launch_crmsa_region4(xin, cw.norm_w, cw.norm_b, w->phi, nullptr, ws.logits, ws.wdisp, ws.rep,
                                 inner16 ? ws.rep16 : nullptr, desc->compute, ws.cr_part, ws.cr_cnt, k, gd8, st));
    rep16_done = inner16;
  } else if (crmsa_region_enabled() && crmsa_region_supported(D, k, gd8)) {
    // logits + combine in one pass over x1 (one block of 16 waves per region, the rows stay in registers)
    RRT_TRY(launch_crmsa_region(xin, cw.norm_w, cw.norm_b, w->phi, nullptr, nullptr, ws.wdisp, ws.rep, k, gd8, st));
  } else if (ws.cr_cnt && desc->n_rmsa_layers > 0 && gd8.P >= RRT_STREAM4_MIN_P && crmsa_stream4_supported(D, k, gd8)) {
    // round 6: regions of more than 144 tokens in ONE pass over x1 as well -- four blocks per region that stream their rows
    // with an online softmax per wave, crmsa_region4's records and merge (crmsa_stream4_kernel); counters as above
    RRT_TRY(launch_crmsa_stream4(xin, cw.norm_w, cw.norm_b, w->phi, nullptr, ws.logits, ws.wdisp, ws.rep,
                                 inner16 ? ws.rep16 : nullptr, desc->compute, ws.cr_part, ws.cr_cnt, k, gd8, st));
    rep16_done = inner16;
  } else {
    RRT_TRY(launch_crmsa_logits(xin, cw.norm_w, cw.norm_b, w->phi, ws.mean_rstd, ws.logits, D, k, gd8, st));
    RRT_TRY(launch_crmsa_combine(xin, cw.norm_w, cw.norm_b, ws.mean_rstd, ws.logits, ws.wdisp, ws.rep,
                                 inner16 ? ws.rep16 : nullptr, desc->compute, D, k, gd8, st));
    rep16_done = inner16;
  }
  RRT_MARK(RRT_EV_CR_COMBINE);
  // inner MSA over the representatives: batch = k, sequence = R8 regions, no EPEG (rmsa.py:322)
  if (inner16) {
    // reduced-precision modes: qkv projection + attention of the (n, head) pairs as ONE launch of the 16-bit fused R-MSA
    // kernel (k "regions" of 64 tokens, no EPEG), then the out-projection on 16-bit operands
    if (!rep16_done) {
      Cast16Jobs cj{};
      cj.src[0] = ws.rep; cj.dst[0] = ws.rep16; cj.n4[0] = (size_t)k * R8 * D / 4; cj.count = 1;
      RRT_TRY(launch_cast16(cj, desc->compute, st));
    }
    RRT_TRY(launch_rmsa_fused16(ws.rep16, ws.wcr16, cw.qkv_b, nullptr, ws.repo16, k, R8, D, desc->crmsa_heads, 0,
                                desc->compute, st));
    LinearEpilogue ep{};
    ep.prec = desc->compute;
    ep.bias = cw.proj_b;
    RRT_TRY(launch_linear16(ws.repo16, ws.wcr16 + (size_t)3 * D * D, ws.rep2, k * R8, D, D, ep, st));
  } else {
    static const bool inner_solo = rrt_tune_env("RRT_INNER_SOLO") != nullptr;     // (A/B: the K-split 16-wave GEMMs also with bags in flight)
    RRT_TRY(inner_attention(ws.rep, k, R8, cw, D, desc->crmsa_heads, 0, ws.rep_qkv, ws.rep_o, desc->compute, desc->solo != 0 || inner_solo, st));
    LinearEpilogue ep{};
    ep.prec = desc->compute;
    ep.bias = cw.proj_b;
    ep.solo = desc->solo != 0 || inner_solo;
    RRT_TRY(launch_linear(ws.rep_o, cw.proj_w, ws.rep2, k * R8, D, D, ep, st));
  }
  RRT_MARK(RRT_EV_CR_INNER);
  if (desc->ffn) {
    // x2 = x1 + dispatch (no LayerNorm yet) -> FFN -> (+ shortcut) -> final LayerNorm.  xin is xb or the
    // caller's x: xa is free for x2, and xin is dead once the dispatch has read it.
    RRT_TRY(launch_crmsa_dispatch_ln(xin, nullptr, ws.wdisp, ws.rep2, nullptr, nullptr, ws.xa, D, k, gd8, st));
    rc = ffn_block(cw, ws.xa, ws.xb);
    if (rc) return rc;
    RRT_TRY(launch_layernorm(ws.xb, x0, w->norm_w, w->norm_b, y, (int)N, D, st));
    RRT_MARK(RRT_EV_END);
    return RRT_OK;
  }
  const bool want16 = y16 != nullptr && (desc->compute == RRT_COMPUTE_BF16 || desc->compute == RRT_COMPUTE_F16) && D % 4 == 0;
  RRT_TRY(launch_crmsa_dispatch_ln(xin, x0, ws.wdisp, ws.rep2, w->norm_w, w->norm_b, y, D, k,
                                   gd8, st, want16 ? y16 : nullptr, want16 ? desc->compute : 0));
  if (want16 && y16_done) *y16_done = true;
  RRT_MARK(RRT_EV_END);
#undef RRT_TRY
#undef RRT_MARK
  return RRT_OK;
}

int rrt_encoder_forward_f32(const rrt_encoder_desc* desc, const rrt_encoder_weights* w, const float* x,
                            float* y, int64_t n_tokens, void* workspace, size_t workspace_bytes,
                            void* stream) {
  return encoder_forward(desc, w, x, y, n_tokens, workspace, workspace_bytes, stream, nullptr);
}

int rrt_phase_gate_create(rrt_phase_gate** out) {
  if (!out) return RRT_E_INVALID;
  rrt_phase_gate* g = new rrt_phase_gate();
  g->armed = false;
  hipError_t e = hipEventCreateWithFlags(&g->done, hipEventDisableTiming);
  if (e != hipSuccess) {
    delete g;
    return (int)e;
  }
  *out = g;
  return RRT_OK;
}

int rrt_phase_gate_destroy(rrt_phase_gate* g) {
  if (!g) return RRT_OK;
  (void)hipEventDestroy(g->done);
  delete g;
  return RRT_OK;
}

int rrt_encoder_forward_gated_f32(const rrt_encoder_desc* desc, const rrt_encoder_weights* w, const float* x,
                                  float* y, int64_t n_tokens, void* workspace, size_t workspace_bytes,
                                  void* stream, rrt_phase_gate* gate, void** events) {
  return encoder_forward(desc, w, x, y, n_tokens, workspace, workspace_bytes, stream, events, gate);
}

int rrt_encoder_forward_events_f32(const rrt_encoder_desc* desc, const rrt_encoder_weights* w,
                                   const float* x, float* y, int64_t n_tokens, void* workspace,
                                   size_t workspace_bytes, void* stream, void** events) {
  return encoder_forward(desc, w, x, y, n_tokens, workspace, workspace_bytes, stream, events);
}

// ---- batch > 1 (modules/rrt.py:165-202 on (B, N, D) input).  The reference's region_partition puts the regions of all
// bags of the batch on one leading axis (rmsa.py:28-39), so the R-MSA layers treat the bags independently, but CR-MSA's
// inner attention runs over the 64 B representatives of ALL bags (rmsa.py:316-322: batch = k, sequence = B * 64): the bags
// are coupled there and only there.  Here: the R-MSA layers bag by bag (the single-bag path, results in x1 [B, N, D]),
// statistics + combine per bag into rep [k, 64 B, D] (region b * 64 + w), ONE inner MSA over sequences of 64 B tokens, the
// dispatch + final LayerNorm per bag.  Correctness path for the literal drop-in contract (every reference trainer uses
// B = 1): the chip-wide two-kernel statistics / combine and the generic attention kernel, fp32 data; inference only.
namespace {
struct BatchWs {
  Workspace one;          // the single-bag workspace (R-MSA layers, FFN / MLP-phi temporaries), reused bag after bag
  float *x1, *mean_rstd, *logits, *wdisp, *rep, *rep_qkv, *rep_o, *rep2;
  size_t one_bytes, bytes;
};
BatchWs carve_batch(const rrt_encoder_desc& d, int64_t B, int64_t N, const rrt_grid& g, const rrt_grid& g8, char* base) {
  BatchWs w{};
  // (the single-bag carve is sized by its QUERY form: with ffn = 1 the query counts xa / xb twice -- a null pointer does not
  //  tell it they are taken -- so the queried size, which encoder_forward checks against, exceeds what the real carve uses)
  w.one_bytes = carve(d, N, g, g8, nullptr).bytes;
  w.one = carve(d, N, g, g8, base);
  size_t off = align_up(w.one_bytes, 256);
  auto take = [&](size_t nfloat) {
    float* p = base ? (float*)(base + off) : nullptr;
    off = align_up(off + nfloat * sizeof(float), 256);
    return p;
  };
  const size_t D = d.dim, Np8 = (size_t)g8.H * g8.H, R8 = (size_t)g8.regions_side * g8.regions_side, k = d.crmsa_k;
  w.x1 = take((size_t)B * N * D);
  if (d.cr_msa) {
    w.mean_rstd = take((size_t)B * N * 2);
    w.logits = take((size_t)B * Np8 * k);
    w.wdisp = take((size_t)B * Np8 * k);
    w.rep = take(k * R8 * B * D);
    w.rep_qkv = take(k * R8 * B * 3 * D);
    w.rep_o = take(k * R8 * B * D);
    w.rep2 = take(k * R8 * B * D);
  }
  w.bytes = off;
  return w;
}
}  // namespace

int rrt_encoder_batch_workspace_size(const rrt_encoder_desc* desc, int32_t batch, int64_t n_tokens, size_t* bytes) {
  if (!bytes || batch <= 0) return RRT_E_INVALID;
  int rc = check_desc(desc, n_tokens);
  if (rc) return rc;
  rrt_grid g{}, g8{};
  if (desc->n_rmsa_layers > 0) {
    rc = rrt_region_grid(n_tokens, desc->region_num, desc->region_size, desc->min_region_num, desc->min_region_ratio, &g);
    if (rc) return rc;
  }
  rc = rrt_region_grid(n_tokens, 8, 0, 0, 0.f, &g8);
  if (rc) return rc;
  *bytes = carve_batch(*desc, batch, n_tokens, g, g8, nullptr).bytes;
  return RRT_OK;
}

int rrt_encoder_forward_batch_f32(const rrt_encoder_desc* desc_in, const rrt_encoder_weights* w, const float* x, float* y,
                                  int32_t batch, int64_t n_tokens, void* workspace, size_t workspace_bytes, void* stream) {
  if (!desc_in || !w || !x || !y || x == y || batch <= 0) return RRT_E_INVALID;
  int rc = check_desc(desc_in, n_tokens);
  if (rc) return rc;
  if (batch > 1024) return unsupported("batch > 1024");
  rrt_encoder_desc dloc = *desc_in;
  dloc.weights16_valid = 0;     // (the first bag writes the images; the later bags of THIS call may reuse them, below)
  dloc.solo = 0;
  const rrt_encoder_desc* const desc = &dloc;
  const int64_t N = n_tokens, B = batch;
  const int D = desc->dim;
  hipStream_t st = (hipStream_t)stream;
  rrt_grid g{}, g8{};
  if (desc->n_rmsa_layers > 0) {
    rc = rrt_region_grid(N, desc->region_num, desc->region_size, desc->min_region_num, desc->min_region_ratio, &g);
    if (rc) return rc;
  }
  rc = rrt_region_grid(N, 8, 0, 0, 0.f, &g8);
  if (rc) return rc;
  BatchWs bw = carve_batch(*desc, B, N, g, g8, nullptr);
  if (!workspace || workspace_bytes < bw.bytes) return RRT_E_WORKSPACE;
  bw = carve_batch(*desc, B, N, g, g8, (char*)workspace);
  hipError_t e = hipSuccess;
#define RRT_TRY(call)                   \
  do {                                  \
    e = (call);                         \
    if (e != hipSuccess) return (int)e; \
  } while (0)
  // ---- positional encoder + R-MSA layers, bag by bag -> x1 [B, N, D]
  for (int64_t b = 0; b < B; ++b) {
    rc = encoder_forward(&dloc, w, x + (size_t)b * N * D, y + (size_t)b * N * D, N, workspace, bw.one_bytes, stream, nullptr,
                         nullptr, bw.x1 + (size_t)b * N * D);
    if (rc) return rc;
    dloc.weights16_valid = 1;   // same workspace, same weights, same mode: the images of bag 0 stand
  }
  dloc.weights16_valid = 0;
  if (!w->norm_w || !w->norm_b) return RRT_E_INVALID;
  if (!desc->cr_msa) {
    for (int64_t b = 0; b < B; ++b)
      RRT_TRY(launch_layernorm(bw.x1 + (size_t)b * N * D, desc->all_shortcut ? x + (size_t)b * N * D : nullptr, w->norm_w,
                               w->norm_b, y + (size_t)b * N * D, (int)N, D, st));
    return RRT_OK;
  }
  // ---- CR-MSA over the batch (rmsa.py:290-337)
  const rrt_attn_weights& cw = w->crmsa;
  if (!cw.norm_w || !cw.norm_b || !cw.qkv_w || !cw.proj_w || !cw.proj_b) return RRT_E_INVALID;
  if (desc->crmsa_mlp ? (!w->phi0_w || !w->phi2_w) : !w->phi) return RRT_E_INVALID;
  GridDev gd8 = to_dev(g8);
  const int k = desc->crmsa_k, R8 = gd8.rs * gd8.rs;
  gd8.Rt = R8 * (int)B;                                  // rep / rep2 rows: [k][B * R8]
  const size_t Np8 = (size_t)gd8.Np;
  // (F32X3 concerns the R-MSA projections only; CR-MSA's GEMMs round their operands in the bf16 / fp16 modes)
  const int prec = desc->compute == RRT_COMPUTE_F32X3 ? RRT_COMPUTE_F32 : desc->compute;
  for (int64_t b = 0; b < B; ++b) {
    const float* xb1 = bw.x1 + (size_t)b * N * D;
    float* lg = bw.logits + (size_t)b * Np8 * k;
    float* wd = bw.wdisp + (size_t)b * Np8 * k;
    float* repb = bw.rep + (size_t)b * R8 * D;          // region 0 of bag b in every representative's row
    if (desc->crmsa_mlp) {
      RRT_TRY(launch_ln_partition(xb1, cw.norm_w, cw.norm_b, bw.one.v8, D, gd8, st));
      LinearEpilogue ep{};
      ep.prec = prec;
      RRT_TRY(launch_linear(bw.one.v8, w->phi0_w, bw.one.hid, gd8.Np, D / 4, D, ep, st));
      RRT_TRY(launch_crmsa_mlp_logits(bw.one.hid, w->phi2_w, lg, gd8.Np, D / 4, k, st));
      RRT_TRY(launch_crmsa_combine(bw.one.v8, nullptr, nullptr, nullptr, lg, wd, repb, nullptr, 0, D, k, gd8, st));
    } else {
      float* mr = bw.mean_rstd + (size_t)b * N * 2;
      RRT_TRY(launch_crmsa_logits(xb1, cw.norm_w, cw.norm_b, w->phi, mr, lg, D, k, gd8, st));
      RRT_TRY(launch_crmsa_combine(xb1, cw.norm_w, cw.norm_b, mr, lg, wd, repb, nullptr, 0, D, k, gd8, st));
    }
  }
  // inner MSA: batch = k, sequence = the B * R8 representatives of all bags, no EPEG (rmsa.py:322)
  RRT_TRY(inner_attention(bw.rep, k, R8 * (int)B, cw, D, desc->crmsa_heads, 0, bw.rep_qkv, bw.rep_o, prec, false, st));
  {
    LinearEpilogue ep{};
    ep.prec = prec;
    ep.bias = cw.proj_b;
    RRT_TRY(launch_linear(bw.rep_o, cw.proj_w, bw.rep2, k * R8 * (int)B, D, D, ep, st));
  }
  for (int64_t b = 0; b < B; ++b) {
    const float* xb1 = bw.x1 + (size_t)b * N * D;
    const float* x0 = desc->all_shortcut ? x + (size_t)b * N * D : nullptr;
    const float* wd = bw.wdisp + (size_t)b * Np8 * k;
    const float* rep2b = bw.rep2 + (size_t)b * R8 * D;
    float* yb = y + (size_t)b * N * D;
    if (desc->ffn) {
      // x2 = x1 + dispatch (no LayerNorm yet) -> FFN -> (+ shortcut) -> final LayerNorm
      RRT_TRY(launch_crmsa_dispatch_ln(xb1, nullptr, wd, rep2b, nullptr, nullptr, bw.one.xa, D, k, gd8, st));
      rc = ffn_apply(desc, cw, bw.one.xa, bw.one.xb, bw.one, N, st);
      if (rc) return rc;
      RRT_TRY(launch_layernorm(bw.one.xb, x0, w->norm_w, w->norm_b, yb, (int)N, D, st));
    } else {
      RRT_TRY(launch_crmsa_dispatch_ln(xb1, x0, wd, rep2b, w->norm_w, w->norm_b, yb, D, k, gd8, st));
    }
  }
#undef RRT_TRY
  return RRT_OK;
}

// ------------------------------------------------------------------ stage entry points
int rrt_ln_partition_f32(const float* x, const float* gamma, const float* beta, float* u, int64_t L,
                         int32_t dim, const rrt_grid* g, void* stream) {
  if (!x || !gamma || !beta || !u || !g || L != g->L || dim <= 0 || dim % 4) return RRT_E_INVALID;
  if (dim > 2048) return unsupported("dim > 2048");
  return (int)launch_ln_partition(x, gamma, beta, u, dim, to_dev(*g), (hipStream_t)stream);
}

int rrt_linear_f32(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N,
                   int32_t K, int32_t q_cols, float q_scale, int32_t compute, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return RRT_E_INVALID;
  if (K % 32) return unsupported("linear: K must be a multiple of 32");
  if (compute < 0 || compute > 2) return unsupported("compute must be RRT_COMPUTE_F32/BF16/F16");
  LinearEpilogue ep{};
  ep.prec = compute;
  ep.bias = bias;
  ep.q_cols = q_cols;
  ep.q_scale = q_scale;
  return (int)launch_linear(A, B, C, (int)M, N, K, ep, (hipStream_t)stream);
}

int rrt_linear_unpartition_residual_f32(const float* A, const float* B, const float* bias,
                                        const float* resid, float* out, int32_t N, int32_t K,
                                        const rrt_grid* g, int32_t compute, void* stream) {
  if (!A || !B || !resid || !out || !g || N <= 0 || K <= 0) return RRT_E_INVALID;
  if (K % 32) return unsupported("linear: K must be a multiple of 32");
  if (compute < 0 || compute > 2) return unsupported("compute must be RRT_COMPUTE_F32/BF16/F16");
  LinearEpilogue ep{};
  ep.prec = compute;
  ep.bias = bias;
  ep.resid = resid;
  ep.g = to_dev(*g);
  return (int)launch_linear(A, B, out, ep.g.Np, N, K, ep, (hipStream_t)stream);
}

int rrt_region_attention_f32(const float* qkv, const float* pe_w, float* o, int32_t n_regions, int32_t P,
                             int32_t dim, int32_t heads, int32_t epeg_k, void* stream) {
  if (!qkv || !o || n_regions <= 0 || P <= 0 || dim <= 0 || heads <= 0 || dim % heads) return RRT_E_INVALID;
  if (pe_w && epeg_k > 0 && epeg_k % 2 == 0) return unsupported("epeg_k must be odd");
  return (int)launch_region_attention(qkv, pe_w, o, n_regions, P, dim, heads, epeg_k, (hipStream_t)stream);
}

int rrt_rmsa_fused_f32(const float* u, const float* qkv_w, const float* qkv_b, const float* pe_w, float* o,
                       int32_t n_regions, int32_t P, int32_t dim, int32_t heads, int32_t epeg_k,
                       int32_t compute, void* stream) {
  if (!u || !qkv_w || !o || n_regions <= 0 || P <= 0 || dim <= 0 || heads <= 0) return RRT_E_INVALID;
  if (compute < 0 || compute > 2) return unsupported("compute must be RRT_COMPUTE_F32/BF16/F16");
  const int ek = pe_w ? epeg_k : 0;
  if (!rmsa_fused_supported(P, dim, heads, ek) || !rmsa_fused_supported_rows((long)n_regions * P, dim))
    return unsupported("rmsa_fused: needs head dim 64, 48 < P <= 208, epeg_k <= 63 (use linear + region_attention)");
  return (int)launch_rmsa_fused(u, qkv_w, qkv_b, pe_w, o, n_regions, P, dim, heads, ek, compute,
                                (hipStream_t)stream);
}

int rrt_device_error(int32_t clear) { return handover_err_peek(clear != 0); }

int rrt_rmsa_fused_proj_f32(const float* u, const float* qkv_w, const float* qkv_b, const float* pe_w,
                            const float* proj_w, const float* proj_b, const float* resid, float* out, float* o_scratch,
                            int32_t* counters, int32_t dim, int32_t heads, int32_t epeg_k, const rrt_grid* g,
                            void* stream) {
  return rrt_debug_rmsa_fused_proj_f32(u, qkv_w, qkv_b, pe_w, proj_w, proj_b, resid, out, o_scratch, counters, dim, heads,
                                       epeg_k, g, 0, 0, 0, stream);
}

}  // extern "C"
// the merged launch of one layer through the stage entry points (tests): optional test knobs, optional CR-MSA row records
static int fused_proj_stage(const float* u, const float* qkv_w, const float* qkv_b, const float* pe_w,
                            const float* proj_w, const float* proj_b, const float* resid, float* out, float* o_scratch,
                            int32_t* counters, int32_t dim, int32_t heads, int32_t epeg_k, const rrt_grid* g,
                            int32_t lag, int32_t spin_limit, int32_t wait_extra, const float* ln2_gamma, const float* phi,
                            int32_t crmsa_k, float* part, void* stream);
extern "C" {
int rrt_debug_rmsa_fused_proj_f32(const float* u, const float* qkv_w, const float* qkv_b, const float* pe_w,
                                  const float* proj_w, const float* proj_b, const float* resid, float* out, float* o_scratch,
                                  int32_t* counters, int32_t dim, int32_t heads, int32_t epeg_k, const rrt_grid* g,
                                  int32_t lag, int32_t spin_limit, int32_t wait_extra, void* stream) {
  return fused_proj_stage(u, qkv_w, qkv_b, pe_w, proj_w, proj_b, resid, out, o_scratch, counters, dim, heads, epeg_k, g, lag,
                          spin_limit, wait_extra, nullptr, nullptr, 0, nullptr, stream);
}

int rrt_rmsa_fused_proj_stats_f32(const float* u, const float* qkv_w, const float* qkv_b, const float* pe_w,
                                  const float* proj_w, const float* proj_b, const float* resid, float* out, float* o_scratch,
                                  int32_t* counters, const float* ln2_gamma, const float* phi, int32_t crmsa_k, float* part,
                                  int32_t dim, int32_t heads, int32_t epeg_k, const rrt_grid* g, void* stream) {
  if (!ln2_gamma || !phi || !part || crmsa_k < 1 || crmsa_k > RRT_MAX_CRMSA_K) return RRT_E_INVALID;
  return fused_proj_stage(u, qkv_w, qkv_b, pe_w, proj_w, proj_b, resid, out, o_scratch, counters, dim, heads, epeg_k, g, 0, 0, 0,
                          ln2_gamma, phi, crmsa_k, part, stream);
}

int rrt_crmsa_combine_parts_f32(const float* x1, const float* part, const float* gamma, const float* beta, const float* phi,
                                float* wdisp, float* rep, int64_t n_tokens, int32_t dim, int32_t k, const rrt_grid* g8,
                                void* stream) {
  if (!x1 || !part || !gamma || !beta || !phi || !wdisp || !rep || !g8 || n_tokens <= 0 || g8->L != n_tokens) return RRT_E_INVALID;
  const GridDev gd = to_dev(*g8);
  if (!crmsa_combine_parts_supported(dim, k, gd)) return unsupported("crmsa_combine_parts: dim % 64 == 0, dim <= 512, 1 <= k <= 8");
  return (int)launch_crmsa_combine_parts(x1, part, gamma, beta, phi, wdisp, rep, nullptr, 0, dim, k, gd, (hipStream_t)stream);
}
}  // extern "C"

static int fused_proj_stage(const float* u, const float* qkv_w, const float* qkv_b, const float* pe_w,
                            const float* proj_w, const float* proj_b, const float* resid, float* out, float* o_scratch,
                            int32_t* counters, int32_t dim, int32_t heads, int32_t epeg_k, const rrt_grid* g,
                            int32_t lag, int32_t spin_limit, int32_t wait_extra, const float* ln2_gamma, const float* phi,
                            int32_t crmsa_k, float* part, void* stream) {
  if (!u || !qkv_w || !proj_w || !resid || !out || !o_scratch || !counters || !g || dim <= 0 || heads <= 0 || out == resid ||
      lag < 0 || spin_limit < 0 || wait_extra < 0)
    return RRT_E_INVALID;
  if (handover_err_peek(false)) return RRT_E_HANDOVER;
  const GridDev gd = to_dev(*g);
  const int ek = pe_w ? epeg_k : 0, R = gd.rs * gd.rs;
  if (!rmsa_fused_proj_supported(R, gd.P, dim, heads, ek, RRT_COMPUTE_F32) || !rmsa_fused_supported_rows(gd.Np, dim))
    return unsupported("rmsa_fused_proj: needs what rmsa_fused needs (head dim 64, 64 < P <= 208, epeg_k <= 63) and at "
                       "least two rounds of (region, head) items (heads * regions >= 2 x the CU count)");
  hipError_t e = hipMemsetAsync(counters, 0, (size_t)R * sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  FusedProj pj{};
  pj.Wp = proj_w;
  pj.bias = proj_b;
  pj.resid = resid;
  pj.out = out;
  pj.cnt = counters;
  pj.g = gd;
  pj.lag = lag;
  pj.spin_limit = spin_limit;
  pj.wait_for = wait_extra > 0 ? heads + wait_extra : 0;
  pj.part = part;
  pj.ln_g = ln2_gamma;
  pj.phi = phi;
  pj.k = crmsa_k;
  if (lag > 0 && (lag < 8 * heads || lag > heads * R)) return unsupported("rmsa_fused_proj: lag must be in [8 * heads, heads * regions]");
  return (int)launch_rmsa_fused(u, qkv_w, qkv_b, pe_w, o_scratch, R, gd.P, dim, heads, ek, RRT_COMPUTE_F32,
                                (hipStream_t)stream, nullptr, &pj);
}
extern "C" {

// ---- 16-bit operand stages of the reduced-precision modes (cast16.hip, linear_f32.hip IN16, rmsa_fused16.hip)
int rrt_cast16(const float* src, uint16_t* dst, int64_t n, int32_t compute, void* stream) {
  if (!src || !dst || n <= 0 || n % 4) return RRT_E_INVALID;
  if (compute != RRT_COMPUTE_BF16 && compute != RRT_COMPUTE_F16) return unsupported("cast16: compute must be BF16 or F16");
  Cast16Jobs jobs{};
  jobs.src[0] = src; jobs.dst[0] = dst; jobs.n4[0] = (size_t)n / 4; jobs.count = 1;
  return (int)launch_cast16(jobs, compute, (hipStream_t)stream);
}

int rrt_ln_partition16(const float* x, const float* gamma, const float* beta, uint16_t* u, int64_t L, int32_t dim,
                       const rrt_grid* g, int32_t compute, void* stream) {
  if (!x || !gamma || !beta || !u || !g || L != g->L || dim <= 0 || dim % 4) return RRT_E_INVALID;
  if (dim > 2048) return unsupported("dim > 2048");
  if (compute != RRT_COMPUTE_BF16 && compute != RRT_COMPUTE_F16) return unsupported("compute must be BF16 or F16");
  return (int)launch_ln_partition16(x, gamma, beta, u, dim, to_dev(*g), compute, (hipStream_t)stream);
}

int rrt_linear16_f32(const uint16_t* A, const uint16_t* B, const float* bias, const float* resid, float* C, int64_t M,
                     int32_t N, int32_t K, const rrt_grid* g, int32_t compute, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || (resid && !g)) return RRT_E_INVALID;
  if (K % 64) return unsupported("linear16: K must be a multiple of 64");
  if (compute != RRT_COMPUTE_BF16 && compute != RRT_COMPUTE_F16) return unsupported("compute must be BF16 or F16");
  LinearEpilogue ep{};
  ep.prec = compute;
  ep.bias = bias;
  if (resid) {
    ep.resid = resid;
    ep.g = to_dev(*g);
    if (M != ep.g.Np) return RRT_E_INVALID;
  }
  return (int)launch_linear16(A, B, C, (int)M, N, K, ep, (hipStream_t)stream);
}

int rrt_rmsa_fused16(const uint16_t* u, const uint16_t* qkv_w, const float* qkv_b, const float* pe_w, uint16_t* o,
                     int32_t n_regions, int32_t P, int32_t dim, int32_t heads, int32_t epeg_k, int32_t compute,
                     void* stream) {
  if (!u || !qkv_w || !o || n_regions <= 0 || P <= 0 || dim <= 0 || heads <= 0) return RRT_E_INVALID;
  if (compute != RRT_COMPUTE_BF16 && compute != RRT_COMPUTE_F16) return unsupported("compute must be BF16 or F16");
  const int ek = pe_w ? epeg_k : 0;
  if (!rmsa_fused16_supported(P, dim, heads, ek) || !rmsa_fused_supported_rows((long)n_regions * P, dim))
    return unsupported("rmsa_fused16: needs head dim 64, 16 < P <= 256, epeg_k <= 63");
  return (int)launch_rmsa_fused16(u, qkv_w, qkv_b, pe_w, o, n_regions, P, dim, heads, ek, compute, (hipStream_t)stream);
}

int rrt_rmsa_pair16_proj(const uint16_t* u, const uint16_t* qkv_w, const float* qkv_b, const float* pe_w, const uint16_t* proj_w,
                         const float* proj_b, const float* resid, float* out, uint16_t* o, int32_t* cnt, int32_t dim,
                         int32_t heads, int32_t epeg_k, const rrt_grid* g, int32_t compute, void* stream) {
  if (!u || !qkv_w || !proj_w || !resid || !out || !o || !cnt || !g || dim <= 0 || heads <= 0) return RRT_E_INVALID;
  if (compute != RRT_COMPUTE_BF16 && compute != RRT_COMPUTE_F16) return unsupported("compute must be BF16 or F16");
  if (handover_err_peek(false)) return RRT_E_HANDOVER;
  const GridDev gd = to_dev(*g);
  const int ek = pe_w ? epeg_k : 0, R = gd.rs * gd.rs;
  if (!rmsa_pair16_proj_supported(R, gd.P, dim, heads, ek))
    return unsupported("rmsa_pair16_proj: head dim 64, regions of 65..96 or 113..128 tokens, a multiple of 16 regions, "
                       "at least two rounds of (pair, head) items");
  hipError_t e = hipMemsetAsync(cnt, 0, (size_t)(R / 2) * sizeof(int), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  PairProj pj{};
  pj.Wp = proj_w;
  pj.bias = proj_b;
  pj.resid = resid;
  pj.out = out;
  pj.cnt = cnt;
  pj.g = gd;
  return (int)launch_rmsa_pair16_proj(u, qkv_w, qkv_b, pe_w, o, R, gd.P, dim, heads, ek, compute, pj, (hipStream_t)stream);
}

// ---- RRT_COMPUTE_F32X3 stages: fp32 as (hi, lo) bf16 pairs (cast16.hip), three bf16 MFMAs per product
int rrt_cast_split(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0 || n % 32) return RRT_E_INVALID;
  Cast16Jobs jobs{};
  jobs.src[0] = src; jobs.dst[0] = (uint16_t*)dst; jobs.n4[0] = (size_t)n / 4; jobs.count = 1;
  return (int)launch_cast_split(jobs, (hipStream_t)stream);
}

int rrt_ln_partition_split(const float* x, const float* gamma, const float* beta, void* u, int64_t L, int32_t dim,
                           const rrt_grid* g, void* stream) {
  if (!x || !gamma || !beta || !u || !g || L != g->L || dim <= 0 || dim % 32) return RRT_E_INVALID;
  if (dim > 2048) return unsupported("dim > 2048");
  return (int)launch_ln_partition_split(x, gamma, beta, u, dim, to_dev(*g), (hipStream_t)stream);
}

int rrt_linear_split_f32(const void* A, const void* B, const float* bias, const float* resid, float* C, int64_t M,
                         int32_t N, int32_t K, const rrt_grid* g, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || (resid && !g)) return RRT_E_INVALID;
  if (K % 32) return unsupported("linear_split: K must be a multiple of 32");
  LinearEpilogue ep{};
  ep.bias = bias;
  if (resid) {
    ep.resid = resid;
    ep.g = to_dev(*g);
    if (M != ep.g.Np) return RRT_E_INVALID;
  }
  return (int)launch_linear_split(A, B, C, (int)M, N, K, ep, (hipStream_t)stream);
}

int rrt_rmsa_fused_x3(const void* u, const void* qkv_w, const float* qkv_b, const float* pe_w, void* o,
                      int32_t n_regions, int32_t P, int32_t dim, int32_t heads, int32_t epeg_k, void* stream) {
  if (!u || !qkv_w || !o || n_regions <= 0 || P <= 0 || dim <= 0 || heads <= 0) return RRT_E_INVALID;
  const int ek = pe_w ? epeg_k : 0;
  if (!rmsa_fused_x3_supported(P, dim, heads, ek) || !rmsa_fused_supported_rows((long)n_regions * P, dim))
    return unsupported("rmsa_fused_x3: needs head dim 64, 48 < P <= 144, epeg_k <= 63");
  return (int)launch_rmsa_fused_x3(u, qkv_w, qkv_b, pe_w, o, n_regions, P, dim, heads, ek, (hipStream_t)stream);
}

int rrt_crmsa_logits_f32(const float* x1, const float* gamma, const float* beta, const float* phi,
                         float* mean_rstd, float* logits, int64_t L, int32_t dim, int32_t k,
                         const rrt_grid* g8, void* stream) {
  if (!x1 || !gamma || !beta || !phi || !mean_rstd || !logits || !g8 || L != g8->L) return RRT_E_INVALID;
  if (k <= 0 || k > RRT_MAX_CRMSA_K || dim % 4 || dim > 2048) return unsupported("crmsa: k in [1,8], dim%4==0, dim<=2048");
  return (int)launch_crmsa_logits(x1, gamma, beta, phi, mean_rstd, logits, dim, k, to_dev(*g8),
                                  (hipStream_t)stream);
}

int rrt_crmsa_combine_f32(const float* x1, const float* gamma, const float* beta, const float* mean_rstd,
                          const float* logits, float* wdisp, float* rep, int64_t L, int32_t dim, int32_t k,
                          const rrt_grid* g8, void* stream) {
  if (!x1 || !logits || !wdisp || !rep || !g8 || L != g8->L) return RRT_E_INVALID;
  if (mean_rstd && (!gamma || !beta)) return RRT_E_INVALID;
  if (k <= 0 || k > RRT_MAX_CRMSA_K || dim % 4) return unsupported("crmsa: k in [1,8], dim%4==0");
  return (int)launch_crmsa_combine(x1, gamma, beta, mean_rstd, logits, wdisp, rep, nullptr, 0, dim, k, to_dev(*g8),
                                   (hipStream_t)stream);
}

int rrt_crmsa_region_f32(const float* x1, const float* gamma, const float* beta, const float* phi, float* mean_rstd,
                         float* logits, float* wdisp, float* rep, int64_t L, int32_t dim, int32_t k, const rrt_grid* g8,
                         void* stream) {
  if (!x1 || !gamma || !beta || !phi || !wdisp || !rep || !g8 || L != g8->L) return RRT_E_INVALID;
  const GridDev gd = to_dev(*g8);
  if (!crmsa_region_supported(dim, k, gd)) return unsupported("crmsa_region: dim = 512, k <= 3, regions of <= 144 tokens");
  return (int)launch_crmsa_region(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, k, gd, (hipStream_t)stream);
}

int rrt_crmsa_region4_f32(const float* x1, const float* gamma, const float* beta, const float* phi, float* mean_rstd,
                          float* logits, float* wdisp, float* rep, int64_t L, int32_t dim, int32_t k, const rrt_grid* g8,
                          void* scratch, size_t scratch_bytes, void* stream) {
  if (!x1 || !gamma || !beta || !phi || !logits || !wdisp || !rep || !g8 || !scratch || L != g8->L) return RRT_E_INVALID;
  const GridDev gd = to_dev(*g8);
  if (!crmsa_region4_supported(dim, k, gd)) return unsupported("crmsa_region4: dim = 512, k <= 8, regions of 4..576 tokens");
  if (scratch_bytes < 256 + crmsa_region4_scratch_floats(gd, k) * sizeof(float)) return RRT_E_WORKSPACE;
  hipError_t e = hipMemsetAsync(scratch, 0, 256, (hipStream_t)stream);      // the arrival counters
  if (e != hipSuccess) return (int)e;
  return (int)launch_crmsa_region4(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, nullptr, 0,
                                   (float*)((char*)scratch + 256), (int*)scratch, k, gd, (hipStream_t)stream);
}

int rrt_crmsa_stream4_f32(const float* x1, const float* gamma, const float* beta, const float* phi, float* mean_rstd,
                          float* logits, float* wdisp, float* rep, int64_t L, int32_t dim, int32_t k, const rrt_grid* g8,
                          void* scratch, size_t scratch_bytes, void* stream) {
  if (!x1 || !gamma || !beta || !phi || !logits || !wdisp || !rep || !g8 || !scratch || L != g8->L) return RRT_E_INVALID;
  const GridDev gd = to_dev(*g8);
  if (!crmsa_stream4_supported(dim, k, gd) || !crmsa_region4_supported(dim, k, gd))
    return unsupported("crmsa_stream4: dim = 512, k <= 8, regions of 16..576 tokens");
  if (scratch_bytes < 256 + crmsa_region4_scratch_floats(gd, k) * sizeof(float)) return RRT_E_WORKSPACE;
  hipError_t e = hipMemsetAsync(scratch, 0, 256, (hipStream_t)stream);      // the arrival counters
  if (e != hipSuccess) return (int)e;
  return (int)launch_crmsa_stream4(x1, gamma, beta, phi, mean_rstd, logits, wdisp, rep, nullptr, 0,
                                   (float*)((char*)scratch + 256), (int*)scratch, k, gd, (hipStream_t)stream);
}

int rrt_crmsa_dispatch_ln_f32(const float* x1, const float* x0, const float* wdisp,
                              const float* rep2, const float* gamma, const float* beta, float* y, int64_t L,
                              int32_t dim, int32_t k, const rrt_grid* g8, void* stream) {
  if (!x1 || !wdisp || !rep2 || !gamma || !beta || !y || !g8 || L != g8->L) return RRT_E_INVALID;
  if (k <= 0 || k > RRT_MAX_CRMSA_K || dim % 4 || dim > 2048) return unsupported("crmsa: k in [1,8], dim%4==0, dim<=2048");
  return (int)launch_crmsa_dispatch_ln(x1, x0, wdisp, rep2, gamma, beta, y, dim, k, to_dev(*g8),
                                       (hipStream_t)stream);
}

int rrt_crmsa_mlp_logits_f32(const float* hid, const float* w2, float* logits, int64_t rows, int32_t hdim,
                             int32_t k, void* stream) {
  if (!hid || !w2 || !logits || rows <= 0 || hdim <= 0) return RRT_E_INVALID;
  if (k <= 0 || k > RRT_MAX_CRMSA_K) return unsupported("crmsa: k in [1,8]");
  return (int)launch_crmsa_mlp_logits(hid, w2, logits, (int)rows, hdim, k, (hipStream_t)stream);
}

int rrt_layernorm_f32(const float* x1, const float* x0, const float* gamma, const float* beta, float* y,
                      int64_t L, int32_t dim, void* stream) {
  if (!x1 || !gamma || !beta || !y || L <= 0 || dim <= 0 || dim % 4) return RRT_E_INVALID;
  if (dim > 2048) return unsupported("dim > 2048");
  return (int)launch_layernorm(x1, x0, gamma, beta, y, (int)L, dim, (hipStream_t)stream);
}

}  // extern "C"

// ------------------------------------------------------------------ row f1: RRTMIL around the encoder
namespace {

struct PoolWs {
  float *hid_a, *hid_b, *a_raw, *part;
  size_t bytes;
};
PoolWs carve_pool(int64_t N, int dim, int hidden, int gated, char* base) {
  PoolWs w{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? (float*)(base + off) : nullptr;
    off = align_up(off + nfloat * sizeof(float), 256);
    return p;
  };
  const size_t nb = (size_t)(N + POOL_CHUNK - 1) / POOL_CHUNK;
  w.hid_a = take((size_t)N * hidden);
  if (gated) w.hid_b = take((size_t)N * hidden);
  w.a_raw = take((size_t)N);
  w.part = take(nb * ((size_t)dim + 4));
  w.bytes = off;
  return w;
}

int check_pool(int64_t N, int dim, int hidden, int act, int n_classes) {
  if (N <= 0 || dim <= 0 || hidden <= 0 || n_classes <= 0) return RRT_E_INVALID;
  if (dim % 32) return unsupported("pool: dim must be a multiple of 32");
  if (hidden % 4) return unsupported("pool: hidden width must be a multiple of 4");
  if (act < RRT_ACT_NONE || act > RRT_ACT_TANH) return unsupported("pool: act must be none/relu/gelu/tanh");
  if (N > (int64_t)1000000) return unsupported("pool: bag larger than 1e6 tokens");
  return RRT_OK;
}

int pool_predict(const float* y, const float* a_w, const float* a_b, const float* b_w, const float* b_b,
                 const float* c_w, const float* c_b, const float* pred_w, const float* pred_b, float* pooled,
                 float* logits, float* attn, int no_norm, int64_t N, int dim, int hidden, int act,
                 int n_classes, int compute, const PoolWs& ws, hipStream_t st, const uint16_t* y16 = nullptr,
                 const uint16_t* a_w16 = nullptr, const uint16_t* b_w16 = nullptr) {
  hipError_t e;
  LinearEpilogue ep{};
  ep.prec = compute;
  ep.bias = a_b;
  ep.act = act;
  // (round 5: the 16-bit-operand route as patch_to_emb's -- rows cast once by a launch of their own, GEMM on 16-bit images --
  // was measured here and lost: configs[2] 10.55-10.60 k against 10.71-10.80 k slides/s; the cast's 28 MB cost more than the
  // narrow GEMM saved.  Round 6: the encoder's last kernel leaves the rows in 16 bits as a by-product (y16: 9 MB more written,
  // no launch, no second read of y) and the weights ride in the classifier's one cast launch: the fp32-operand product --
  // 23 us for 1.2 GFLOP: 142 blocks on 256 CUs, twice the bytes through the LDS-DMA path -- becomes a 16-bit-operand one)
  const bool p16 = y16 != nullptr && a_w16 != nullptr && (!b_w || b_w16 != nullptr) && dim % 64 == 0 &&
                   (compute == RRT_COMPUTE_BF16 || compute == RRT_COMPUTE_F16);
  e = p16 ? launch_linear16(y16, a_w16, ws.hid_a, (int)N, hidden, dim, ep, st)
          : launch_linear(y, a_w, ws.hid_a, (int)N, hidden, dim, ep, st);
  if (e != hipSuccess) return (int)e;
  if (b_w) {
    ep.bias = b_b;
    ep.act = RRT_ACT_SIGMOID;
    e = p16 ? launch_linear16(y16, b_w16, ws.hid_b, (int)N, hidden, dim, ep, st)
            : launch_linear(y, b_w, ws.hid_b, (int)N, hidden, dim, ep, st);
    if (e != hipSuccess) return (int)e;
  }
  e = launch_pool_partial(y, ws.hid_a, b_w ? ws.hid_b : nullptr, c_w, c_b, ws.a_raw, ws.part, (int)N, dim,
                          hidden, st);
  if (e != hipSuccess) return (int)e;
  e = launch_pool_merge(ws.part, ws.a_raw, pred_w, pred_b, pooled, logits, attn, no_norm, (int)N, dim,
                        n_classes, st);
  return (int)e;
}

int check_mil(const rrt_mil_desc* d, int64_t N) {
  if (!d) return RRT_E_INVALID;
  int rc = check_desc(&d->enc, N);
  if (rc) return rc;
  if (d->input_dim <= 0 || d->input_dim % 32) return unsupported("input_dim must be a positive multiple of 32");
  if (d->emb_act != RRT_ACT_NONE && d->emb_act != RRT_ACT_RELU && d->emb_act != RRT_ACT_GELU)
    return unsupported("emb_act must be none/relu/gelu");
  return check_pool(N, d->enc.dim, d->pool_hidden, d->pool_act, d->n_classes);
}

}  // namespace

extern "C" {

int rrt_linear_act_f32(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N,
                       int32_t K, int32_t act, int32_t compute, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return RRT_E_INVALID;
  if (K % 32) return unsupported("linear: K must be a multiple of 32");
  if (compute < 0 || compute > 2) return unsupported("compute must be RRT_COMPUTE_F32/BF16/F16");
  if (act < RRT_ACT_NONE || act > RRT_ACT_SIGMOID) return unsupported("unknown activation");
  LinearEpilogue ep{};
  ep.prec = compute;
  ep.bias = bias;
  ep.act = act;
  return (int)launch_linear(A, B, C, (int)M, N, K, ep, (hipStream_t)stream);
}

int rrt_pool_workspace_size(int64_t n_tokens, int32_t dim, int32_t hidden, int32_t gated, size_t* bytes) {
  if (!bytes) return RRT_E_INVALID;
  int rc = check_pool(n_tokens, dim, hidden, RRT_ACT_NONE, 1);
  if (rc) return rc;
  *bytes = carve_pool(n_tokens, dim, hidden, gated, nullptr).bytes;
  return RRT_OK;
}

int rrt_pool_predict_f32(const float* y, const float* a_w, const float* a_b, const float* b_w, const float* b_b,
                         const float* c_w, const float* c_b, const float* pred_w, const float* pred_b,
                         float* pooled, float* logits, float* attn, int32_t no_norm, int64_t n_tokens,
                         int32_t dim, int32_t hidden, int32_t act, int32_t n_classes, int32_t compute,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (!y || !a_w || !c_w || !pred_w || !logits) return RRT_E_INVALID;
  int rc = check_pool(n_tokens, dim, hidden, act, n_classes);
  if (rc) return rc;
  if (compute < 0 || compute > 2) return unsupported("compute must be RRT_COMPUTE_F32/BF16/F16");
  const int gated = b_w != nullptr;
  if (!workspace || workspace_bytes < carve_pool(n_tokens, dim, hidden, gated, nullptr).bytes) return RRT_E_WORKSPACE;
  PoolWs ws = carve_pool(n_tokens, dim, hidden, gated, (char*)workspace);
  return pool_predict(y, a_w, a_b, b_w, b_b, c_w, c_b, pred_w, pred_b, pooled, logits, attn, no_norm, n_tokens,
                      dim, hidden, act, n_classes, compute, ws, (hipStream_t)stream);
}

// ---- the pooling alone, forward and backward (training: the first Linear + activation stay autograd-visible layers)
int rrt_attn_pool_workspace_size(int64_t n_tokens, int32_t dim, int32_t hidden, size_t* bytes) {
  if (!bytes) return RRT_E_INVALID;
  int rc = check_pool(n_tokens, dim, hidden, RRT_ACT_NONE, 1);
  if (rc) return rc;
  const size_t nb = (size_t)(n_tokens + POOL_CHUNK - 1) / POOL_CHUNK;
  const size_t fwd = align_up(nb * ((size_t)dim + 4) * sizeof(float), 256);
  const size_t bwd = align_up(pool_backward_part_floats((int)n_tokens, hidden) * sizeof(float), 256);
  *bytes = fwd > bwd ? fwd : bwd;
  return RRT_OK;
}

int rrt_attn_pool_f32(const float* y, const float* hid_a, const float* hid_b, const float* c_w, const float* c_b,
                      float* pooled, float* attn, float* a_raw, int64_t n_tokens, int32_t dim, int32_t hidden,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!y || !hid_a || !c_w || !pooled || !attn || !a_raw) return RRT_E_INVALID;
  int rc = check_pool(n_tokens, dim, hidden, RRT_ACT_NONE, 1);
  if (rc) return rc;
  size_t need = 0;
  (void)rrt_attn_pool_workspace_size(n_tokens, dim, hidden, &need);
  if (!workspace || workspace_bytes < need) return RRT_E_WORKSPACE;
  hipError_t e = launch_pool_partial(y, hid_a, hid_b, c_w, c_b, a_raw, (float*)workspace, (int)n_tokens, dim, hidden,
                                     (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  return (int)launch_pool_merge((const float*)workspace, a_raw, nullptr, nullptr, pooled, nullptr, attn, 0, (int)n_tokens, dim, 0,
                                (hipStream_t)stream);
}

int rrt_attn_pool_backward_f32(const float* y, const float* hid_a, const float* hid_b, const float* c_w, const float* attn,
                               const float* pooled, const float* d_pooled, const float* d_attn, const float* d_raw,
                               const float* c_ext, float* dy, float* dhid_a, float* dhid_b, float* dwcb, int64_t n_tokens,
                               int32_t dim, int32_t hidden, void* workspace, size_t workspace_bytes, void* stream) {
  if (!y || !hid_a || !c_w || !attn || !pooled || !d_pooled || !dy || !dhid_a || !dwcb || (hid_b && !dhid_b)) return RRT_E_INVALID;
  int rc = check_pool(n_tokens, dim, hidden, RRT_ACT_NONE, 1);
  if (rc) return rc;
  if (d_attn && !c_ext) return RRT_E_INVALID;     // c_ext = sum_n attn_n d_attn_n (device scalar) goes with d_attn
  size_t need = 0;
  (void)rrt_attn_pool_workspace_size(n_tokens, dim, hidden, &need);
  if (!workspace || workspace_bytes < need) return RRT_E_WORKSPACE;
  return (int)launch_pool_backward(y, hid_a, hid_b, c_w, attn, pooled, d_pooled, d_attn, d_raw, c_ext, dy, dhid_a, dhid_b, dwcb,
                                   (float*)workspace, (int)n_tokens, dim, hidden, (hipStream_t)stream);
}

int rrt_mil_workspace_size(const rrt_mil_desc* desc, int64_t n_tokens, size_t* bytes) {
  if (!bytes) return RRT_E_INVALID;
  int rc = check_mil(desc, n_tokens);
  if (rc) return rc;
  size_t enc = 0;
  rc = rrt_encoder_workspace_size(&desc->enc, n_tokens, &enc);
  if (rc) return rc;
  const size_t act = align_up((size_t)n_tokens * desc->enc.dim * sizeof(float), 256);
  // (+ the 16-bit images of the feature matrix and of patch_to_emb's weight for the reduced-precision modes: the same size
  // in every mode, so that a workspace sized once serves a module that switches modes)
  *bytes = 2 * act + align_up(enc, 256) +
           align_up(carve_pool(n_tokens, desc->enc.dim, desc->pool_hidden, desc->pool_gated, nullptr).bytes, 256) +
           align_up((size_t)n_tokens * desc->input_dim * 2, 256) + align_up((size_t)desc->enc.dim * desc->input_dim * 2, 256) +
           // (round 6) the encoder's output rows in 16 bits + the pooling Linears' 16-bit weight images
           align_up((size_t)n_tokens * desc->enc.dim * 2, 256) + 2 * align_up((size_t)desc->pool_hidden * desc->enc.dim * 2, 256);
  return RRT_OK;
}

int rrt_mil_forward_f32(const rrt_mil_desc* desc, const rrt_mil_weights* w, const float* x, float* logits,
                        float* attn, int32_t no_norm, float* feat, int64_t n_tokens, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (!desc || !w || !x || !logits) return RRT_E_INVALID;
  int rc = check_mil(desc, n_tokens);
  if (rc) return rc;
  if (!w->emb_w || !w->pool_a_w || !w->pool_c_w || !w->pred_w || (desc->pool_gated && !w->pool_b_w))
    return RRT_E_INVALID;
  size_t need = 0, enc_bytes = 0;
  rc = rrt_mil_workspace_size(desc, n_tokens, &need);
  if (rc) return rc;
  if (!workspace || workspace_bytes < need) return RRT_E_WORKSPACE;
  rc = rrt_encoder_workspace_size(&desc->enc, n_tokens, &enc_bytes);
  if (rc) return rc;
  const int D = desc->enc.dim;
  const size_t act = align_up((size_t)n_tokens * D * sizeof(float), 256);
  char* base = (char*)workspace;
  float* emb = (float*)base;
  float* y = feat ? feat : (float*)(base + act);
  char* enc_ws = base + 2 * act;
  PoolWs pws = carve_pool(n_tokens, D, desc->pool_hidden, desc->pool_gated, enc_ws + align_up(enc_bytes, 256));
  hipStream_t st = (hipStream_t)stream;

  // F32X3 concerns the R-MSA layers' big products only (encoder_forward): the GEMMs around the encoder are exact fp32
  const int gemm_prec = desc->enc.compute == RRT_COMPUTE_F32X3 ? RRT_COMPUTE_F32 : desc->enc.compute;
  LinearEpilogue ep{};
  ep.prec = gemm_prec;
  ep.bias = w->emb_b;
  ep.act = desc->emb_act;
  // Reduced-precision modes (round 5): the features and the weight are cast to 16 bits by one streaming launch and the GEMM
  // runs on 16-bit operands through LDS-DMA.  The fp32-operand kernel rounds the same values to the same 16-bit numbers,
  // but after staging them in LDS as fp32: twice the DMA bytes through a path that is issue-bound (41 us at
  // N = 9000 x 1024 against ~10 + ~13 here).
  static const bool no_fc16 = rrt_tune_env("RRT_NO_FC16") != nullptr;
  static const bool no_pool16 = rrt_tune_env("RRT_NO_POOL16") != nullptr;
  uint16_t *y16 = nullptr, *pa16 = nullptr, *pb16 = nullptr;
  bool pool16 = false, y16_done = false;
  hipError_t e;
  if (desc->input16 != 0 && (desc->input16 != gemm_prec || (gemm_prec != RRT_COMPUTE_BF16 && gemm_prec != RRT_COMPUTE_F16) ||
                             desc->input_dim % 64 != 0 || no_fc16))
    return unsupported("rrt_mil_forward: 16-bit features (input16) need enc.compute == input16 (bf16 / f16) and input_dim % 64 == 0");
  if (!no_fc16 && (gemm_prec == RRT_COMPUTE_BF16 || gemm_prec == RRT_COMPUTE_F16) && desc->input_dim % 64 == 0) {
    char* tail = enc_ws + align_up(enc_bytes, 256) + align_up(pws.bytes, 256);
    uint16_t* x16 = (uint16_t*)tail;
    uint16_t* w16 = (uint16_t*)(tail + align_up((size_t)n_tokens * desc->input_dim * 2, 256));
    y16 = (uint16_t*)((char*)w16 + align_up((size_t)D * desc->input_dim * 2, 256));
    pa16 = (uint16_t*)((char*)y16 + align_up((size_t)n_tokens * D * 2, 256));
    pb16 = (uint16_t*)((char*)pa16 + align_up((size_t)desc->pool_hidden * D * 2, 256));
    Cast16Jobs cj{};
    if (desc->input16) {                 // the caller's features ARE the 16-bit operand: only the weight is cast
      x16 = (uint16_t*)x;
      cj.src[0] = w->emb_w; cj.dst[0] = w16; cj.n4[0] = (size_t)D * desc->input_dim / 4;
      cj.count = 1;
    } else {
      cj.src[0] = x; cj.dst[0] = x16; cj.n4[0] = (size_t)n_tokens * desc->input_dim / 4;
      cj.src[1] = w->emb_w; cj.dst[1] = w16; cj.n4[1] = (size_t)D * desc->input_dim / 4;
      cj.count = 2;
    }
    // the pooling Linears' weights in the same launch (their 16-bit-operand product reads the encoder's y16, below)
    pool16 = !no_pool16 && D % 64 == 0 && ((size_t)desc->pool_hidden * D) % 4 == 0 && cj.count + 2 <= CAST16_MAX_JOBS;
    if (pool16) {
      cj.src[cj.count] = w->pool_a_w; cj.dst[cj.count] = pa16; cj.n4[cj.count++] = (size_t)desc->pool_hidden * D / 4;
      if (desc->pool_gated) {
        cj.src[cj.count] = w->pool_b_w; cj.dst[cj.count] = pb16; cj.n4[cj.count++] = (size_t)desc->pool_hidden * D / 4;
      }
    }
    e = launch_cast16(cj, gemm_prec, st);
    if (e != hipSuccess) return (int)e;
    e = launch_linear16(x16, w16, emb, (int)n_tokens, D, desc->input_dim, ep, st);
  } else {
    e = launch_linear(x, w->emb_w, emb, (int)n_tokens, D, desc->input_dim, ep, st);
  }
  if (e != hipSuccess) return (int)e;
  rc = encoder_forward(&desc->enc, &w->enc, emb, y, n_tokens, enc_ws, enc_bytes, stream, nullptr, nullptr, nullptr,
                       pool16 ? y16 : nullptr, &y16_done);
  if (rc) return rc;
  const bool p16 = pool16 && y16_done;
  return pool_predict(y, w->pool_a_w, w->pool_a_b, desc->pool_gated ? w->pool_b_w : nullptr, w->pool_b_b,
                      w->pool_c_w, w->pool_c_b, w->pred_w, w->pred_b, nullptr, logits, attn, no_norm, n_tokens,
                      D, desc->pool_hidden, desc->pool_act, desc->n_classes, gemm_prec, pws, st, p16 ? y16 : nullptr,
                      p16 ? pa16 : nullptr, (p16 && desc->pool_gated) ? pb16 : nullptr);
}

}  // extern "C"

// ------------------------------------------------------------------ batch-of-bags executor
#define RRT_EXEC_MAX_STREAMS 8
struct rrt_executor {
  rrt_encoder_desc desc;
  int n_streams;
  int device;
  hipStream_t streams[RRT_EXEC_MAX_STREAMS];
  void* ws[RRT_EXEC_MAX_STREAMS];
  size_t ws_bytes[RRT_EXEC_MAX_STREAMS];
  hipEvent_t fork, join[RRT_EXEC_MAX_STREAMS];
  rrt_phase_gate gate;
  // per workspace: (weights.version, compute) of the 16-bit weight images it holds (0 = none)
  uint64_t w16_version[RRT_EXEC_MAX_STREAMS];
  int w16_compute[RRT_EXEC_MAX_STREAMS];
  bool own_streams;      // false: the streams are the caller's (rrt_executor_create_on_streams), never destroyed here
  // slot 0 (its workspace) runs on whatever stream the CALLER passes: two calls from different streams must not overlap there
  hipEvent_t done0;
  bool has_done0;
};

extern "C" {

int rrt_executor_destroy(rrt_executor* ex) {
  if (!ex) return RRT_OK;
  (void)hipDeviceSynchronize();       // slot 0's workspace is used on the callers' streams
  for (int s = 0; s < ex->n_streams; ++s) {
    if (ex->streams[s]) (void)hipStreamSynchronize(ex->streams[s]);
    if (ex->ws[s]) (void)hipFree(ex->ws[s]);
    if (ex->join[s]) (void)hipEventDestroy(ex->join[s]);
    if (ex->streams[s] && ex->own_streams) (void)hipStreamDestroy(ex->streams[s]);
  }
  if (ex->fork) (void)hipEventDestroy(ex->fork);
  if (ex->done0) (void)hipEventDestroy(ex->done0);
  if (ex->gate.done) (void)hipEventDestroy(ex->gate.done);
  delete ex;
  return RRT_OK;
}

static int executor_create(const rrt_encoder_desc* desc, int32_t n_streams, int64_t max_tokens, void* const* user_streams,
                           rrt_executor** out) {
  if (!desc || !out || max_tokens <= 0) return RRT_E_INVALID;
  if (n_streams < 1 || n_streams > RRT_EXEC_MAX_STREAMS) return unsupported("executor: n_streams must be in [1,8]");
  size_t need = 0;
  int rc = rrt_encoder_workspace_size(desc, max_tokens, &need);
  if (rc) return rc;
  rrt_executor* ex = new rrt_executor();
  memset(ex, 0, sizeof(*ex));
  ex->desc = *desc;
  ex->n_streams = n_streams;
  ex->own_streams = user_streams == nullptr;
  hipError_t e = hipGetDevice(&ex->device);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ex->fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ex->gate.done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ex->done0, hipEventDisableTiming);
  for (int s = 0; s < n_streams && e == hipSuccess; ++s) {
    if (user_streams) {
      ex->streams[s] = (hipStream_t)user_streams[s];
    } else {
      static const char* smode = rrt_tune_env("RRT_EXEC_STREAMS");     // tuning build: how the executor's streams are made
      if (smode && smode[0] == 'p') {                                   // "prio": priorities spread over the device's range
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const int span = lo - hi + 1;
        e = hipStreamCreateWithPriority(&ex->streams[s], hipStreamNonBlocking, hi + (span > 0 ? s % span : 0));
      } else {
        e = hipStreamCreateWithFlags(&ex->streams[s], hipStreamNonBlocking);
      }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ex->join[s], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&ex->ws[s], need);
    if (e == hipSuccess) ex->ws_bytes[s] = need;
  }
  if (e != hipSuccess) {
    rrt_executor_destroy(ex);
    return (int)e;
  }
  *out = ex;
  return RRT_OK;
}

int rrt_executor_create(const rrt_encoder_desc* desc, int32_t n_streams, int64_t max_tokens, rrt_executor** out) {
  return executor_create(desc, n_streams, max_tokens, nullptr, out);
}

int rrt_executor_create_on_streams(const rrt_encoder_desc* desc, int32_t n_streams, void* const* streams, int64_t max_tokens,
                                   rrt_executor** out) {
  if (!streams) return RRT_E_INVALID;
  if (n_streams < 1 || n_streams > RRT_EXEC_MAX_STREAMS) return unsupported("executor: n_streams must be in [1,8]");
  for (int s = 0; s < n_streams; ++s)
    for (int t = 0; t < s; ++t)
      if (streams[s] == streams[t]) return RRT_E_INVALID;      // the bags in flight need distinct streams
  return executor_create(desc, n_streams, max_tokens, streams, out);
}

int rrt_executor_forward(rrt_executor* ex, const rrt_encoder_weights* w, const rrt_bag* bags, int32_t n_bags,
                         void* stream) {
  if (!ex || !w || (n_bags > 0 && !bags) || n_bags < 0) return RRT_E_INVALID;
  if (n_bags == 0) return RRT_OK;
  if (n_bags > 65536) return unsupported("executor: more than 65536 bags in one call");
  // validate everything before the first launch: nothing runs if any bag is unsupported
  for (int i = 0; i < n_bags; ++i) {
    if (!bags[i].x || !bags[i].y || bags[i].x == bags[i].y) return RRT_E_INVALID;
    int rc = check_desc(&ex->desc, bags[i].n_tokens);
    if (rc) return rc;
  }
  // longest-processing-time-first onto the least loaded stream (cost ~ tokens); submission order within
  // a stream follows that order, so the big bags start first and the small ones fill the tail
  // Slots used by THIS call: one per four bags.  A fork / join costs every extra stream ~10 us of event waits and releases
  // the streams in lockstep, which pays off only when each stream then carries a few bags: with 2-4 bags per call four
  // streams were SLOWER than one (round 4, profiles/r04_final_bench_bags.txt: 4026-4368 vs 4480 slides/s), from 16 bags per
  // call on they win (4936-5062).  A call of < 8 bags is therefore plain launches on the caller's stream.  The kernels (and
  // bits) stay those of the executor's configured width: solo = (n_streams == 1) below does not depend on the call.
  int S = n_bags / 4;
  if (S < 1) S = 1;
  if (S > ex->n_streams) S = ex->n_streams;
  int order_buf[256];
  int* order = n_bags <= 256 ? order_buf : new int[n_bags];
  for (int i = 0; i < n_bags; ++i) order[i] = i;
  for (int i = 1; i < n_bags; ++i) {        // insertion sort, stable, descending n_tokens (n_bags is small)
    int v = order[i], j = i - 1;
    while (j >= 0 && bags[order[j]].n_tokens < bags[v].n_tokens) { order[j + 1] = order[j]; --j; }
    order[j + 1] = v;
  }
  hipStream_t caller = (hipStream_t)stream;
  // Slot 0 runs on the CALLER's stream, slots 1 .. S-1 on the executor's.  Round 4: with all S slots on own streams the
  // caller's stream sat through the whole call with S barrier packets (the join) at its head -- a (S+1)-th active hardware
  // queue next to the S that carry bags.  At S = 4 that is five queues on four pipes: 4.3-4.45 k slides/s where the same
  // launches on four queues give 5.07-5.10 k (tools/bench_bags.py with the join compiled out; the five-stream bench line
  // shows the same loss).  Now the join's waits sit BEHIND the caller stream's own share of the bags, the fork only
  // concerns the other slots, and a one-stream executor is plain launches on the caller's stream.
  hipStream_t sts[RRT_EXEC_MAX_STREAMS];
  sts[0] = caller;
  for (int s = 1; s < S; ++s) sts[s] = ex->streams[s];
  hipError_t e = hipSuccess;
  // (the previous call may have come from ANOTHER stream: its slot-0 work, on that stream, owns workspace 0 until it is done)
  if (ex->has_done0) {
    e = hipStreamWaitEvent(caller, ex->done0, 0);
    if (e != hipSuccess) return (int)e;
  }
  static const bool no_fork = rrt_tune_env("RRT_EXEC_NOFORK") != nullptr;      // (bisecting, tuning build only)
  if (S > 1 && !no_fork) {
    e = hipEventRecord(ex->fork, caller);
    for (int s = 1; s < S && e == hipSuccess; ++s) e = hipStreamWaitEvent(sts[s], ex->fork, 0);
  }
  int rc = (int)e;
  int64_t load[RRT_EXEC_MAX_STREAMS] = {0};
  // phase gate (the R-MSA cores of the bags in flight take turns): opt-in, RRT_GATE=1.  It paid +1.5 % at two bags in
  // flight with the round-1 kernels; with the round-2 kernels free-running streams are faster at every S (configs[4]
  // mix, bf16: 7.72 k vs 7.49 k slides/s; fp32: 3.74 k vs 3.68 k)
  static const bool gate_on = rrt_tune_env("RRT_GATE") != nullptr;
  const bool gated = S == 2 && gate_on;
  // tuning build: RRT_EXEC_STAGGER=1 -> only the FIRST bag of every stream goes through the gate, so that the streams of a
  // call start one R-MSA core apart instead of in lockstep (the fork releases them together)
  static const bool stagger = rrt_tune_env("RRT_EXEC_STAGGER") != nullptr;
  if (stagger) ex->gate.armed = false;
  for (int k = 0; k < n_bags && rc == RRT_OK; ++k) {
    const rrt_bag& b = bags[order[k]];
    int s = 0;
    for (int t = 1; t < S; ++t)
      if (load[t] < load[s]) s = t;
    load[s] += b.n_tokens;
    size_t need = 0;
    rc = rrt_encoder_workspace_size(&ex->desc, b.n_tokens, &need);
    if (rc) break;
    if (need > ex->ws_bytes[s]) {            // grow: the only host synchronisation in this call
      e = hipStreamSynchronize(sts[s]);
      if (e == hipSuccess) e = hipFree(ex->ws[s]);
      ex->ws[s] = nullptr;
      ex->ws_bytes[s] = 0;
      if (e == hipSuccess) e = hipMalloc(&ex->ws[s], need);
      if (e != hipSuccess) { rc = (int)e; break; }
      ex->ws_bytes[s] = need;
      ex->w16_version[s] = 0;
    }
    rrt_encoder_desc d = ex->desc;
    d.solo = ex->n_streams == 1;
    d.weights16_valid = w->version != 0 && ex->w16_version[s] == w->version && ex->w16_compute[s] == d.compute;
    rc = encoder_forward(&d, w, b.x, b.y, b.n_tokens, ex->ws[s], ex->ws_bytes[s], sts[s], nullptr,
                         (gated || (stagger && k < S)) ? &ex->gate : nullptr);
    ex->w16_version[s] = rc == RRT_OK ? w->version : 0;
    ex->w16_compute[s] = d.compute;
  }
  // join even after an error so the caller's stream stays ordered after whatever was enqueued
  static const bool no_join = rrt_tune_env("RRT_EXEC_NOJOIN") != nullptr;      // (bisecting, tuning build only)
  for (int s = 1; s < S && !no_join; ++s) {
    hipError_t j = hipEventRecord(ex->join[s], sts[s]);
    if (j == hipSuccess) j = hipStreamWaitEvent(caller, ex->join[s], 0);
    if (rc == RRT_OK && j != hipSuccess) rc = (int)j;
  }
  if (hipEventRecord(ex->done0, caller) == hipSuccess) ex->has_done0 = true;
  if (order != order_buf) delete[] order;
  return rc;
}

}  // extern "C"

// ------------------------------------------------------------------ row f2 building blocks (backward stages)
extern "C" {

int rrt_region_attention_backward_workspace_size(int32_t n_regions, int32_t P, int32_t dim, int32_t heads,
                                                 int32_t epeg_k, size_t* bytes) {
  if (!bytes || n_regions <= 0 || P <= 0 || dim <= 0 || heads <= 0) return RRT_E_INVALID;
  *bytes = attn_bwd_workspace(n_regions, P, dim, heads, epeg_k);
  return RRT_OK;
}

int rrt_region_attention_backward_f32(const float* qkv, const float* pe_w, const float* o, const float* d_o,
                                      float* d_qkv, float* d_pe_w, int32_t n_regions, int32_t P, int32_t dim,
                                      int32_t heads, int32_t epeg_k, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (!qkv || !o || !d_o || !d_qkv || n_regions <= 0 || P <= 0 || dim <= 0 || heads <= 0) return RRT_E_INVALID;
  const int ek = pe_w ? epeg_k : 0;
  if (ek > 0 && ek % 2 == 0) return unsupported("epeg_k must be odd");
  if (!attn_bwd_supported(P, dim, heads, ek))
    return unsupported("attention backward: needs head dim 64 with P <= 208, or (no EPEG, P <= 128, head dim % 4 == 0)");
  if (!workspace || workspace_bytes < attn_bwd_workspace(n_regions, P, dim, heads, ek)) return RRT_E_WORKSPACE;
  return (int)launch_attention_backward(qkv, pe_w, o, d_o, d_qkv, d_pe_w, (float*)workspace, n_regions, P, dim,
                                        heads, ek, (hipStream_t)stream);
}

int rrt_layernorm_backward_f32(const float* dy, const float* x, const float* gamma, const float* add, float* dx,
                               float* dgamma_dbeta, int64_t L, int32_t dim, const rrt_grid* g, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!dy || !x || !gamma || !dx || !dgamma_dbeta || L <= 0 || dim <= 0 || dim % 4) return RRT_E_INVALID;
  if (dim > 2048) return unsupported("dim > 2048");
  if (g && g->L != L) return RRT_E_INVALID;
  if (!workspace || workspace_bytes < ln_bwd_workspace(dim)) return RRT_E_WORKSPACE;
  GridDev gd{};
  if (g) gd = to_dev(*g);
  return (int)launch_ln_backward(dy, x, gamma, add, dx, dgamma_dbeta, (float*)workspace, (int)L, dim,
                                 g ? &gd : nullptr, (hipStream_t)stream);
}

int rrt_reduce_partials_f32(const float* part, float* out, float* out_tr, int32_t S, int64_t n, int64_t split,
                            int32_t tr_dim, int32_t tr_k, int32_t deferred, int32_t copies, void* stream) {
  if (!part || !out || S <= 0 || n <= 0 || S > (1 << 20)) return RRT_E_INVALID;
  if (out_tr && (split < 0 || split % 4 || split > n || tr_dim <= 0 || tr_k <= 0 || n - split != (int64_t)tr_dim * tr_k))
    return RRT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (!deferred) {
    if (out_tr) return (int)launch_reduce_partials_scatter(part, out, S, (size_t)n, (size_t)split, out_tr, tr_dim, tr_k, st);
    return (int)launch_reduce_partials(part, out, S, (size_t)n, st);
  }
  if (copies < 1 || copies > REDUCE_MAX_JOBS) return RRT_E_INVALID;
  ReduceJobs rj{};
  for (int c = 0; c < copies; ++c) {
    hipError_t e = reduce_or_defer(&rj, part, out + (size_t)c * n, S, (size_t)n, st, (size_t)(out_tr ? split : 0),
                                   out_tr ? out_tr + (size_t)c * tr_dim * tr_k : nullptr, tr_dim, tr_k);
    if (e != hipSuccess) return (int)e;
  }
  return (int)launch_reduce_jobs(rj, st);
}

int rrt_linear_backward_workspace_size(int64_t M, int32_t N, int32_t K, size_t* bytes) {
  if (!bytes || M <= 0 || N <= 0 || K <= 0 || M > (int64_t)16000000) return RRT_E_INVALID;
  *bytes = linear_bwd_workspace((int)M, N, K);
  return RRT_OK;
}

int rrt_linear_backward_f32(const float* dY, const float* X, const float* W, float* dX, float* dW, float* db,
                            int64_t M, int32_t N, int32_t K, int32_t compute, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!dY || M <= 0 || N <= 0 || K <= 0 || M > (int64_t)16000000) return RRT_E_INVALID;
  if ((dX && !W) || (dW && !X)) return RRT_E_INVALID;
  if (dX && N % 32) return unsupported("linear backward: N must be a multiple of 32 for dX");
  if (compute < 0 || compute > 2) return unsupported("compute must be RRT_COMPUTE_F32/BF16/F16");
  if (!workspace || workspace_bytes < linear_bwd_workspace((int)M, N, K)) return RRT_E_WORKSPACE;
  return (int)launch_linear_backward(dY, X, W, dX, dW, db, (int)M, N, K, compute, workspace, (hipStream_t)stream);
}

}  // extern "C"

// ------------------------------------------------------------------ row f2: training forward + backward
namespace {

struct Stash {
  float *u[RRT_MAX_RMSA_LAYERS], *qkv[RRT_MAX_RMSA_LAYERS], *o[RRT_MAX_RMSA_LAYERS], *xout[RRT_MAX_RMSA_LAYERS];
  float *mean_rstd, *logits, *wdisp, *rep, *rep_qkv, *rep_o, *rep2, *x2, *v8, *hid;
  // ffn = 1: per TransLayer (index n_rmsa_layers = CR-MSA's) LN2 output, fc1 pre-activation, layer output;
  // xcr = CR-MSA's output before its FFN; hscr = one scratch for act(hpre)
  float *ffn_u[RRT_MAX_RMSA_LAYERS + 1], *ffn_hpre[RRT_MAX_RMSA_LAYERS + 1], *xf[RRT_MAX_RMSA_LAYERS + 1];
  float *xcr, *hscr, *xp;     // xp: output of the PEG / PPEG stage (pos != none)
  float *pe[RRT_MAX_RMSA_LAYERS];   // value-EPEG ablations: the conv's output per layer
  float *smap;                // 2-D 'attn' EPEG: score-map scratch of the forward (large regions only)
  size_t bytes;
};

Stash carve_stash(const rrt_encoder_desc& d, int64_t N, const rrt_grid& g, const rrt_grid& g8, char* base) {
  Stash s{};
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? (float*)(base + off) : nullptr;
    off = align_up(off + nfloat * sizeof(float), 256);
    return p;
  };
  const size_t D = d.dim, Np = (size_t)g.H * g.H, Np8 = (size_t)g8.H * g8.H;
  const size_t R8 = (size_t)g8.regions_side * g8.regions_side, k = d.crmsa_k;
  for (int l = 0; l < d.n_rmsa_layers; ++l) {
    s.u[l] = take(Np * D);
    s.qkv[l] = take(Np * 3 * D);
    s.o[l] = take(Np * D);
    s.xout[l] = take((size_t)N * D);
    if (d.epeg && d.epeg_type != RRT_EPEG_ATTN) s.pe[l] = take(Np * D);
  }
  if (d.n_rmsa_layers > 0 && d.epeg && d.epeg_2d && d.epeg_type == RRT_EPEG_ATTN) {
    const size_t sm = attn_scoremap_scratch_floats(g.regions_side * g.regions_side, g.s * g.s, d.n_heads, d.epeg_k, 1);
    if (sm) s.smap = take(sm);
  }
  if (d.cr_msa) {
    s.mean_rstd = take((size_t)N * 2);
    s.logits = take(Np8 * k);
    s.wdisp = take(Np8 * k);
    s.rep = take(k * R8 * D);
    s.rep_qkv = take(k * R8 * 3 * D);
    s.rep_o = take(k * R8 * D);
    s.rep2 = take(k * R8 * D);
    if (d.crmsa_mlp) {
      s.v8 = take(Np8 * D);
      s.hid = take(Np8 * (D / 4));
    }
  }
  s.x2 = take((size_t)N * D);
  if (d.ffn) {
    const int nl = d.n_rmsa_layers + (d.cr_msa ? 1 : 0);
    for (int l = 0; l < nl; ++l) {
      const int idx = l < d.n_rmsa_layers ? l : RRT_MAX_RMSA_LAYERS;
      s.ffn_u[idx] = take((size_t)N * D);
      s.ffn_hpre[idx] = take((size_t)N * d.ffn_hidden);
      s.xf[idx] = take((size_t)N * D);
    }
    if (d.cr_msa) s.xcr = take((size_t)N * D);
    s.hscr = take((size_t)N * d.ffn_hidden);
  }
  if (d.pos) s.xp = take((size_t)N * D);
  s.bytes = off;
  return s;
}

struct BwdWs {
  float *dx2, *dxa, *dxb, *dz, *dO, *dqkv, *lnpart, *attnpart, *dWd, *dC, *dlg, *Cw, *d_rep2, *d_rep_o,
      *d_rep_qkv, *d_rep, *rows, *dxpart, *th, *dhid, *dvphi, *w1t, *tnscratch, *fh, *fdh, *fdu, *attnpart_cr, *dpe, *smap;
  char *lin, *pegws;
  float* tappart[RRT_MAX_RMSA_LAYERS];     // the EPEG taps' partials of each layer (summed by the deferred reduce launch)
  size_t lnpart_stride;                    // lnpart holds one partial buffer per LayerNorm backward of the pass
  float *wt_qkv[RRT_MAX_RMSA_LAYERS], *wt_proj[RRT_MAX_RMSA_LAYERS], *wt_cr_qkv, *wt_cr_proj;   // W^T images, made in one launch
  size_t bytes;
};

BwdWs carve_bwd(const rrt_encoder_desc& d, int64_t N, const rrt_grid& g, const rrt_grid& g8, char* base) {
  BwdWs w{};
  size_t off = 0;
  auto takeb = [&](size_t nbytes) {
    char* p = base ? base + off : nullptr;
    off = align_up(off + nbytes, 256);
    return p;
  };
  auto take = [&](size_t nfloat) { return (float*)takeb(nfloat * sizeof(float)); };
  const size_t D = d.dim, Np = (size_t)g.H * g.H, Np8 = (size_t)g8.H * g8.H;
  const size_t R8 = (size_t)g8.regions_side * g8.regions_side, k = d.crmsa_k;
  w.dx2 = take((size_t)N * D);
  w.dxa = take((size_t)N * D);
  w.dxb = take((size_t)N * D);
  w.lnpart_stride = align_up(ln_bwd_workspace((int)D), 256) / sizeof(float);
  w.lnpart = take(w.lnpart_stride * (size_t)(2 * (d.n_rmsa_layers + 1) + 1));      // final + (norm, norm2) per layer + CR-MSA's norm2
  size_t lin = 0;
  if (d.n_rmsa_layers > 0) {
    w.dz = take(Np * D);
    w.dO = take(Np * D);
    w.dqkv = take(Np * 3 * D);
    const int R = g.regions_side * g.regions_side;
    const bool e2d = d.epeg && d.epeg_2d && d.epeg_type == RRT_EPEG_ATTN, evalue = d.epeg && d.epeg_type != RRT_EPEG_ATTN;
    if (e2d) {       // 2-D 'attn' EPEG: its own backward kernel (three [P, P] maps per (region, head) + the tap partials)
      w.smap = take(attn_scoremap_scratch_floats(R, g.s * g.s, d.n_heads, d.epeg_k, 3) + (size_t)R * d.n_heads * d.epeg_k * d.epeg_k);
      w.attnpart = take(64);
    } else {
      w.attnpart = take(attn_bwd_workspace(R, g.s * g.s, (int)D, d.n_heads, (d.epeg && !evalue) ? d.epeg_k : 0) / sizeof(float));
      if (d.epeg && !evalue)
        for (int li = 0; li < d.n_rmsa_layers; ++li) w.tappart[li] = take((size_t)R * d.n_heads * d.epeg_k);
    }
    if (evalue) w.dpe = take(Np * D);
    lin = linear_bwd_workspace((int)Np, 3 * (int)D, (int)D);
    const size_t l2 = linear_bwd_workspace((int)Np, (int)D, (int)D);
    if (l2 > lin) lin = l2;
  }
  if (d.cr_msa) {
    w.dWd = take(Np8 * k);
    w.dC = take(Np8 * k);
    w.dlg = take(Np8 * k);
    w.Cw = take(Np8 * k);
    w.d_rep2 = take(k * R8 * D);
    w.d_rep_o = take(k * R8 * D);
    w.d_rep_qkv = take(k * R8 * 3 * D);
    w.d_rep = take(k * R8 * D);
    w.rows = take((2 + k) * D);
    w.dxpart = take(crmsa_bwd_dx_workspace((int)D, (int)k) / sizeof(float));
    if (d.crmsa_mlp) {
      w.th = take(Np8 * (D / 4));
      w.dhid = take(Np8 * (D / 4));
      w.dvphi = take(Np8 * D);
      w.w1t = take(D * (D / 4));
      const size_t a1 = linear_bwd_workspace((int)Np8, (int)(D / 4), (int)D);     // dW1 partials dominate
      const size_t a2 = linear_bwd_workspace((int)Np8, (int)k, (int)(D / 4));
      w.tnscratch = take((a1 > a2 ? a1 : a2) / sizeof(float));
    }
    // attnpart is shared by the R-MSA layers' and CR-MSA's attention backward: the larger of the two geometries
    // (a small bag's R-MSA partials can be smaller than what CR-MSA's 64-token regions need)
    w.attnpart_cr = take(attn_bwd_workspace((int)k, (int)R8, (int)D, d.crmsa_heads, 0) / sizeof(float));
    const size_t l3 = linear_bwd_workspace((int)(k * R8), 3 * (int)D, (int)D);
    if (l3 > lin) lin = l3;
  }
  if (d.ffn) {
    w.fh = take((size_t)N * d.ffn_hidden);
    w.fdh = take((size_t)N * d.ffn_hidden);
    w.fdu = take((size_t)N * D);
    const size_t f1 = linear_bwd_workspace((int)N, (int)D, d.ffn_hidden), f2 = linear_bwd_workspace((int)N, d.ffn_hidden, (int)D);
    if (f1 > lin) lin = f1;
    if (f2 > lin) lin = f2;
  }
  w.lin = takeb(lin ? lin : 256);
  if (d.pos) w.pegws = takeb(peg_bwd_workspace((int)N, (int)D, d.peg_k, d.pos == RRT_POS_PPEG));
  for (int li = 0; li < d.n_rmsa_layers; ++li) {
    w.wt_qkv[li] = take(3 * D * D);
    w.wt_proj[li] = take(D * D);
  }
  if (d.cr_msa) {
    w.wt_cr_qkv = take(3 * D * D);
    w.wt_cr_proj = take(D * D);
  }
  w.bytes = off;
  return w;
}

int check_train(const rrt_encoder_desc* d, int64_t N, rrt_grid* g, rrt_grid* g8) {
  int rc = check_desc(d, N);
  if (rc) return rc;
  if (d->compute == RRT_COMPUTE_F32X3) return unsupported("training: RRT_COMPUTE_F32X3 is an inference mode (train in F32 or under autocast)");
  if (d->dim > 1024) return unsupported("training: dim > 1024");
  memset(g, 0, sizeof(*g));
  if (d->n_rmsa_layers > 0) {
    rc = rrt_region_grid(N, d->region_num, d->region_size, d->min_region_num, d->min_region_ratio, g);
    if (rc) return rc;
    const bool e2d = d->epeg && d->epeg_2d && d->epeg_type == RRT_EPEG_ATTN, evalue = d->epeg && d->epeg_type != RRT_EPEG_ATTN;
    // (the 2-D 'attn' EPEG has its own backward kernel, any head dim; the value variants run the plain attention backward)
    if (!e2d && !attn_bwd_supported(g->s * g->s, d->dim, d->n_heads, (d->epeg && !evalue) ? d->epeg_k : 0))
      return unsupported("training: the R-MSA attention backward needs head dim 64 (any region size, epeg_k <= 63); other head dims only without EPEG on regions of <= 128 tokens, or with epeg_2d");
  }
  rc = rrt_region_grid(N, 8, 0, 0, 0.f, g8);
  if (rc) return rc;
  if (d->cr_msa) {
    const int R8 = g8->regions_side * g8->regions_side;
    if (!attn_bwd_supported(R8, d->dim, d->crmsa_heads, 0))
      return unsupported("training: CR-MSA's inner attention needs a head dim that is a multiple of 4");
  }
  return RRT_OK;
}

}  // namespace

extern "C" {

int rrt_encoder_train_sizes(const rrt_encoder_desc* desc, int64_t n_tokens, size_t* stash_bytes,
                            size_t* backward_workspace_bytes) {
  if (!desc || !stash_bytes || !backward_workspace_bytes) return RRT_E_INVALID;
  rrt_grid g{}, g8{};
  int rc = check_train(desc, n_tokens, &g, &g8);
  if (rc) return rc;
  *stash_bytes = carve_stash(*desc, n_tokens, g, g8, nullptr).bytes;
  *backward_workspace_bytes = carve_bwd(*desc, n_tokens, g, g8, nullptr).bytes;
  return RRT_OK;
}

namespace {
struct DropCfg {
  unsigned thresh;
  float scale;
  unsigned seed(uint64_t base, int layer) const { return (unsigned)(base ^ (base >> 32)) + 0x9E3779B9u * (unsigned)(layer + 1); }
};
int make_drop(float p, DropCfg* d) {
  if (!(p >= 0.f) || p >= 1.f) return RRT_E_INVALID;
  d->thresh = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
  if (p > 0.f && d->thresh == 0) d->thresh = 1;
  d->scale = 1.0f / (1.0f - p);
  return RRT_OK;
}
constexpr int DROP_LAYER_CRMSA = 100;
// stochastic depth (TransLayer.drop_path, rrt.py:102,125,129): per-call multipliers of the residual branches,
// host array [2][RRT_MAX_RMSA_LAYERS + 1] = {attention branches, FFN branches}, index n_rmsa_layers = CR-MSA's
struct Branch {
  float attn[RRT_MAX_RMSA_LAYERS + 1], ffn[RRT_MAX_RMSA_LAYERS + 1];
};
int make_branch(const float* bs, Branch* b) {
  for (int i = 0; i <= RRT_MAX_RMSA_LAYERS; ++i) {
    b->attn[i] = bs ? bs[i] : 1.0f;
    b->ffn[i] = bs ? bs[RRT_MAX_RMSA_LAYERS + 1 + i] : 1.0f;
    if (!(b->attn[i] >= 0.f) || !(b->ffn[i] >= 0.f)) return RRT_E_INVALID;
  }
  return RRT_OK;
}
}  // namespace

int rrt_encoder_forward_train_f32(const rrt_encoder_desc* desc, const rrt_encoder_weights* w, const float* x,
                                  float* y, int64_t n_tokens, void* stash, size_t stash_bytes, float drop_p,
                                  uint64_t drop_seed, const float* branch_scale, void* stream) {
  if (!desc || !w || !x || !y || x == y) return RRT_E_INVALID;
  DropCfg dc{};
  if (make_drop(drop_p, &dc)) return RRT_E_INVALID;
  Branch br{};
  if (make_branch(branch_scale, &br)) return RRT_E_INVALID;
  rrt_grid g{}, g8{};
  int rc = check_train(desc, n_tokens, &g, &g8);
  if (rc) return rc;
  Stash s = carve_stash(*desc, n_tokens, g, g8, nullptr);
  if (!stash || stash_bytes < s.bytes) return RRT_E_WORKSPACE;
  s = carve_stash(*desc, n_tokens, g, g8, (char*)stash);
  hipStream_t st = (hipStream_t)stream;
  const int D = desc->dim;
  const int64_t N = n_tokens;
  hipError_t e = hipSuccess;
#define RRT_TRY(call)                   \
  do {                                  \
    e = (call);                         \
    if (e != hipSuccess) return (int)e; \
  } while (0)
  // TransLayer's FFN (ffn = 1): xf = xi + fc2(act(fc1(LN2(xi)))), stashing LN2's output and the pre-activation
  GridDev gid{};
  {
    const int Hs = (int)ceil_sqrt(N);
    gid.L = (int)N;
    gid.H = gid.s = Hs;
    gid.rs = 1;
    gid.P = gid.Np = Hs * Hs;
    gid.inv_H = gid.inv_s = 1.0f / (float)Hs;
    gid.inv_rs = 1.0f;
    gid.inv_P = 1.0f / (float)gid.P;
  }
  auto train_ffn = [&](const rrt_attn_weights& lw, const float* xi, int idx) -> int {
    if (!lw.norm2_w || !lw.norm2_b || !lw.fc1_w || !lw.fc1_b || !lw.fc2_w || !lw.fc2_b) return RRT_E_INVALID;
    hipError_t fe = launch_layernorm(xi, nullptr, lw.norm2_w, lw.norm2_b, s.ffn_u[idx], (int)N, D, st);
    if (fe != hipSuccess) return (int)fe;
    LinearEpilogue e1{};
    e1.prec = desc->compute;
    e1.bias = lw.fc1_b;
    fe = launch_linear(s.ffn_u[idx], lw.fc1_w, s.ffn_hpre[idx], (int)N, desc->ffn_hidden, D, e1, st);
    if (fe != hipSuccess) return (int)fe;
    // the Mlp's two dropouts (rrt.py:38-40): after the activation and after fc2 (before the residual)
    fe = launch_act_forward(s.ffn_hpre[idx], s.hscr, (size_t)N * desc->ffn_hidden, desc->ffn_act, dc.thresh,
                            dc.seed(drop_seed, 200 + 2 * idx), dc.scale, st);
    if (fe != hipSuccess) return (int)fe;
    LinearEpilogue e2{};
    e2.bias = lw.fc2_b;
    e2.resid = xi;
    e2.g = gid;
    const float bsf = br.ffn[idx < RRT_MAX_RMSA_LAYERS ? idx : desc->n_rmsa_layers];
    e2.drop_thresh = dc.thresh;
    e2.drop_scale = dc.scale * bsf;
    e2.drop_on = dc.thresh || bsf != 1.0f;
    e2.drop_seed = dc.seed(drop_seed, 201 + 2 * idx);
    return (int)launch_linear(s.hscr, lw.fc2_w, s.xf[idx], (int)N, D, desc->ffn_hidden, e2, st);
  };
  const float* xin = x;
  const bool pos_first = desc->pos && desc->pos_pos == -1, pos_mid = desc->pos && desc->pos_pos == 0;
  if (desc->pos && (!w->pos_w[0] || (desc->pos == RRT_POS_PPEG && (!w->pos_w[1] || !w->pos_w[2])))) return RRT_E_INVALID;
  if (pos_first) {
    RRT_TRY(launch_peg(xin, w->pos_w, w->pos_b, s.xp, (int)N, D, desc->peg_k, desc->peg_1d, desc->pos == RRT_POS_PPEG, st));
    xin = s.xp;
  }
  for (int li = 0; li < desc->n_rmsa_layers; ++li) {
    if (pos_mid && li == 1) {
      RRT_TRY(launch_peg(xin, w->pos_w, w->pos_b, s.xp, (int)N, D, desc->peg_k, desc->peg_1d, desc->pos == RRT_POS_PPEG, st));
      xin = s.xp;
    }
    const rrt_attn_weights& lw = w->rmsa[li];
    if (!lw.norm_w || !lw.norm_b || !lw.qkv_w || !lw.proj_w || !lw.proj_b) return RRT_E_INVALID;
    if (desc->epeg && !lw.pe_w) return RRT_E_INVALID;
    const GridDev gd = to_dev(g);
    RRT_TRY(launch_ln_partition(xin, lw.norm_w, lw.norm_b, s.u[li], D, gd, st));
    LinearEpilogue ep{};
    ep.prec = desc->compute;
    ep.bias = lw.qkv_b;
    ep.q_cols = D;
    ep.q_scale = 1.0f / sqrtf((float)(D / desc->n_heads));
    const bool e2d = desc->epeg && desc->epeg_2d && desc->epeg_type == RRT_EPEG_ATTN;
    const bool evalue = desc->epeg && desc->epeg_type != RRT_EPEG_ATTN;
    // the inference path's fused kernel (projection + EPEG + attention per (region, head)) with the q | k | v tiles
    // also written to the stash, where the bag's regions fit it; else projection to the stash + attention from it
    const int ekf = desc->epeg ? desc->epeg_k : 0;
    const bool fused = !e2d && !evalue && desc->compute == RRT_COMPUTE_F32 &&
                       rmsa_fused_supported(gd.P, D, desc->n_heads, ekf) && rmsa_fused_supported_rows(gd.Np, D);
    if (fused) {
      RRT_TRY(launch_rmsa_fused(s.u[li], lw.qkv_w, lw.qkv_b, desc->epeg ? lw.pe_w : nullptr, s.o[li], gd.rs * gd.rs, gd.P, D,
                                desc->n_heads, ekf, RRT_COMPUTE_F32, st, s.qkv[li]));
    } else {
    RRT_TRY(launch_linear(s.u[li], lw.qkv_w, s.qkv[li], gd.Np, 3 * D, D, ep, st));
    if (e2d) {                                            // rmsa.py:78-79,106-108
      RRT_TRY(launch_attn_scoremap(s.qkv[li], lw.pe_w, s.o[li], s.smap, gd.rs * gd.rs, gd.P, D, desc->n_heads, desc->epeg_k, st));
    } else if (evalue) {                                  // rmsa.py:80-85,114-129; the stash keeps pe, v' = v + pe ('bf') and o + pe ('af')
      RRT_TRY(launch_value_pe(s.qkv[li], lw.pe_w, lw.pe_b, s.pe[li], gd.rs * gd.rs, gd.P, gd.s, D, desc->n_heads, desc->epeg_k,
                              desc->epeg_2d, st));
      if (desc->epeg_type == RRT_EPEG_VALUE_BF) RRT_TRY(launch_add_cols(s.qkv[li] + 2 * D, s.pe[li], (size_t)gd.Np, D, 3 * D, st));
      RRT_TRY(launch_region_attention(s.qkv[li], nullptr, s.o[li], gd.rs * gd.rs, gd.P, D, desc->n_heads, 0, st));
      if (desc->epeg_type == RRT_EPEG_VALUE_AF) RRT_TRY(launch_add_cols(s.o[li], s.pe[li], (size_t)gd.Np, D, D, st));
    } else
    RRT_TRY(launch_region_attention(s.qkv[li], desc->epeg ? lw.pe_w : nullptr, s.o[li], gd.rs * gd.rs, gd.P, D,
                                    desc->n_heads, desc->epeg ? desc->epeg_k : 0, st));
    }
    LinearEpilogue ep2{};
    ep2.bias = lw.proj_b;
    ep2.resid = xin;
    ep2.g = gd;
    ep2.drop_thresh = dc.thresh;
    ep2.drop_scale = dc.scale * br.attn[li];
    ep2.drop_on = dc.thresh || br.attn[li] != 1.0f;
    ep2.drop_seed = dc.seed(drop_seed, li);
    ep2.prec = ep2.drop_on ? RRT_COMPUTE_F32 : desc->compute;      // the dropout epilogue exists in fp32 only
    RRT_TRY(launch_linear(s.o[li], lw.proj_w, s.xout[li], gd.Np, D, D, ep2, st));
    xin = s.xout[li];
    if (desc->ffn) {
      rc = train_ffn(lw, xin, li);
      if (rc) return rc;
      xin = s.xf[li];
    }
  }
  const float* x0 = desc->all_shortcut ? x : nullptr;
  if (!w->norm_w || !w->norm_b) return RRT_E_INVALID;
  if (desc->cr_msa) {
    const rrt_attn_weights& cw = w->crmsa;
    if (!cw.norm_w || !cw.norm_b || !cw.qkv_w || !cw.proj_w || !cw.proj_b) return RRT_E_INVALID;
    if (desc->crmsa_mlp ? (!w->phi0_w || !w->phi2_w) : !w->phi) return RRT_E_INVALID;
    const GridDev gd8 = to_dev(g8);
    const int k = desc->crmsa_k, R8 = gd8.rs * gd8.rs;
    if (desc->crmsa_mlp) {
      // the LayerNorm statistics the backward needs come out of the logits kernel (run on the first D*k floats of
      // W1 as a stand-in phi; its logits are overwritten by the MLP's two lines below)
      RRT_TRY(launch_crmsa_logits(xin, cw.norm_w, cw.norm_b, w->phi0_w, s.mean_rstd, s.logits, D, k, gd8, st));
      RRT_TRY(launch_ln_partition(xin, cw.norm_w, cw.norm_b, s.v8, D, gd8, st));
      LinearEpilogue e0{};
      e0.prec = desc->compute;
      RRT_TRY(launch_linear(s.v8, w->phi0_w, s.hid, gd8.Np, D / 4, D, e0, st));
      RRT_TRY(launch_crmsa_mlp_logits(s.hid, w->phi2_w, s.logits, gd8.Np, D / 4, k, st));
      RRT_TRY(launch_crmsa_combine(s.v8, nullptr, nullptr, nullptr, s.logits, s.wdisp, s.rep, nullptr, 0, D, k, gd8, st));
    } else {
      RRT_TRY(launch_crmsa_logits(xin, cw.norm_w, cw.norm_b, w->phi, s.mean_rstd, s.logits, D, k, gd8, st));
      RRT_TRY(launch_crmsa_combine(xin, cw.norm_w, cw.norm_b, s.mean_rstd, s.logits, s.wdisp, s.rep, nullptr, 0, D, k, gd8, st));
    }
    LinearEpilogue ep{};
    ep.prec = desc->compute;
    ep.bias = cw.qkv_b;
    ep.q_cols = D;
    ep.q_scale = 1.0f / sqrtf((float)(D / desc->crmsa_heads));
    ep.solo = true;
    RRT_TRY(launch_linear(s.rep, cw.qkv_w, s.rep_qkv, k * R8, 3 * D, D, ep, st));
    RRT_TRY(launch_region_attention(s.rep_qkv, nullptr, s.rep_o, k, R8, D, desc->crmsa_heads, 0, st));
    LinearEpilogue ep2{};
    ep2.bias = cw.proj_b;
    // CR-MSA's branch is linear in rep2 (dispatch = sum_n wdisp * rep2): its stochastic-depth multiplier rides here
    const float bsc = br.attn[desc->n_rmsa_layers];
    ep2.drop_thresh = dc.thresh;
    ep2.drop_scale = dc.scale * bsc;
    ep2.drop_on = dc.thresh || bsc != 1.0f;
    ep2.drop_seed = dc.seed(drop_seed, DROP_LAYER_CRMSA);
    ep2.prec = ep2.drop_on ? RRT_COMPUTE_F32 : desc->compute;
    ep2.solo = true;                            // (as the qkv product above: K split inside the block, dropout epilogue included)
    RRT_TRY(launch_linear(s.rep_o, cw.proj_w, s.rep2, k * R8, D, D, ep2, st));
    if (desc->ffn) {
      // CR-MSA's TransLayer: x1 + dispatch -> xcr, FFN -> xf, then the shortcut, then the final LayerNorm
      RRT_TRY(launch_crmsa_dispatch_ln(xin, nullptr, s.wdisp, s.rep2, nullptr, nullptr, s.xcr, D, k, gd8, st));
      rc = train_ffn(cw, s.xcr, RRT_MAX_RMSA_LAYERS);
      if (rc) return rc;
      RRT_TRY(launch_layernorm(s.xf[RRT_MAX_RMSA_LAYERS], x0, nullptr, nullptr, s.x2, (int)N, D, st));
    } else {
      RRT_TRY(launch_crmsa_dispatch_ln(xin, x0, s.wdisp, s.rep2, nullptr, nullptr, s.x2, D, k, gd8, st));  // x2, no LN
    }
  } else {
    RRT_TRY(launch_layernorm(xin, x0, nullptr, nullptr, s.x2, (int)N, D, st));                            // x2 = x1 (+ x)
  }
  RRT_TRY(launch_layernorm(s.x2, nullptr, w->norm_w, w->norm_b, y, (int)N, D, st));
#undef RRT_TRY
  return RRT_OK;
}

int rrt_encoder_backward_f32(const rrt_encoder_desc* desc, const rrt_encoder_weights* w, const float* x,
                             const float* dy, const void* stash, size_t stash_bytes, const rrt_encoder_grads* gr,
                             float* dx, int64_t n_tokens, void* workspace, size_t workspace_bytes, float drop_p,
                             uint64_t drop_seed, const float* branch_scale, void* stream) {
  if (!desc || !w || !x || !dy || !gr) return RRT_E_INVALID;
  DropCfg dc{};
  if (make_drop(drop_p, &dc)) return RRT_E_INVALID;
  Branch br{};
  if (make_branch(branch_scale, &br)) return RRT_E_INVALID;
  rrt_grid g{}, g8{};
  int rc = check_train(desc, n_tokens, &g, &g8);
  if (rc) return rc;
  Stash s = carve_stash(*desc, n_tokens, g, g8, nullptr);
  if (!stash || stash_bytes < s.bytes) return RRT_E_WORKSPACE;
  s = carve_stash(*desc, n_tokens, g, g8, (char*)stash);
  BwdWs b = carve_bwd(*desc, n_tokens, g, g8, nullptr);
  if (!workspace || workspace_bytes < b.bytes) return RRT_E_WORKSPACE;
  b = carve_bwd(*desc, n_tokens, g, g8, (char*)workspace);
  hipStream_t st = (hipStream_t)stream;
  const int D = desc->dim, L = desc->n_rmsa_layers;
  const int N = (int)n_tokens;
  hipError_t e = hipSuccess;
#define RRT_TRY(call)                   \
  do {                                  \
    e = (call);                         \
    if (e != hipSuccess) return (int)e; \
  } while (0)
  if (!gr->norm) return RRT_E_INVALID;
  {
    // W^T of every attention Linear, for the dX products (dX = dY . W as the forward GEMM on W^T): one launch up front
    TransposeJobs tj{};
    // (a NULL weight or a full table used to drop the job silently: the dX product then read an unwritten W^T image)
    bool tj_bad = false;
    auto add = [&](const float* wsrc, float* wt, int n_out, int k_in) {
      if (!wsrc || !wt || tj.n >= TRANSPOSE_MAX_JOBS) { tj_bad = true; return; }
      tj.in[tj.n] = wsrc; tj.out[tj.n] = wt; tj.R[tj.n] = n_out; tj.C[tj.n] = k_in;
      ++tj.n;
    };
    for (int li = 0; li < desc->n_rmsa_layers; ++li) {
      add(w->rmsa[li].qkv_w, b.wt_qkv[li], 3 * D, D);
      add(w->rmsa[li].proj_w, b.wt_proj[li], D, D);
    }
    if (desc->cr_msa) {
      add(w->crmsa.qkv_w, b.wt_cr_qkv, 3 * D, D);
      add(w->crmsa.proj_w, b.wt_cr_proj, D, D);
    }
    if (tj_bad) return RRT_E_INVALID;
    RRT_TRY(launch_transpose_batch(tj, st));
  }
  // Parameter-gradient sums that nothing downstream reads leave the dependent chain: the stages below append their reduce
  // jobs here and one launch at the end of the pass runs them all (round 6: four 5-9 us launches -> one).  Each stage
  // therefore gets partial buffers of its own (ln_part(), tappart[li], dxpart).
  ReduceJobs rj{};
  static const bool no_defer = rrt_tune_env("RRT_NO_DEFER_REDUCE") != nullptr;      // (A/B, tuning build only)
  ReduceJobs* const defer = no_defer ? nullptr : &rj;
  int ln_calls = 0;
  auto ln_part = [&]() { return b.lnpart + (size_t)(ln_calls++) * b.lnpart_stride; };
  // final LayerNorm
  RRT_TRY(launch_ln_backward(dy, s.x2, w->norm_w, nullptr, b.dx2, gr->norm, ln_part(), N, D, nullptr, st, defer));
  const float* cur = b.dx2;   // gradient w.r.t. the activations entering the stage being undone
  // FFN backward (ffn = 1): given d xf in `cur`, the gradient w.r.t. the FFN's input xi goes to the other
  // ping-pong buffer; fc2 / fc1 through the linear backward, the activation through its own pass, LN2 last
  auto ffn_backward = [&](const rrt_attn_weights& lw, const rrt_attn_grads& lg, const float* xi, int idx) -> int {
    if (!lg.norm2 || !lg.fc1_w || !lg.fc1_b || !lg.fc2_w || !lg.fc2_b) return RRT_E_INVALID;
    const int Hd = desc->ffn_hidden;
    hipError_t fe = launch_act_forward(s.ffn_hpre[idx], b.fh, (size_t)N * Hd, desc->ffn_act, dc.thresh,
                                       dc.seed(drop_seed, 200 + 2 * idx), dc.scale, st);
    if (fe != hipSuccess) return (int)fe;
    const float* dyf = cur;                        // d(fc2 output): the residual's gradient through the second mask
    const float bsf = br.ffn[idx < RRT_MAX_RMSA_LAYERS ? idx : desc->n_rmsa_layers];
    if (dc.thresh || bsf != 1.0f) {
      fe = launch_copy_drop_mask(cur, b.fdu, (size_t)N * D, dc.thresh, dc.seed(drop_seed, 201 + 2 * idx), dc.scale * bsf, st);
      if (fe != hipSuccess) return (int)fe;
      dyf = b.fdu;
    }
    fe = launch_linear_backward(dyf, b.fh, lw.fc2_w, b.fdh, lg.fc2_w, lg.fc2_b, N, D, Hd, desc->compute, b.lin, st);
    if (fe != hipSuccess) return (int)fe;
    fe = launch_act_backward(b.fdh, s.ffn_hpre[idx], (size_t)N * Hd, desc->ffn_act, dc.thresh,
                             dc.seed(drop_seed, 200 + 2 * idx), dc.scale, st);
    if (fe != hipSuccess) return (int)fe;
    fe = launch_linear_backward(b.fdh, s.ffn_u[idx], lw.fc1_w, b.fdu, lg.fc1_w, lg.fc1_b, N, Hd, D, desc->compute, b.lin, st);
    if (fe != hipSuccess) return (int)fe;
    float* nxt = (cur == b.dxa) ? b.dxb : b.dxa;
    fe = launch_ln_backward(b.fdu, xi, lw.norm2_w, cur, nxt, lg.norm2, ln_part(), N, D, nullptr, st, defer);
    if (fe != hipSuccess) return (int)fe;
    cur = nxt;
    return RRT_OK;
  };
  // PEG / PPEG backward: d(stage output) in `cur` -> d(stage input) in the other buffer, plus the convs' gradients
  auto peg_backward = [&](const float* xstage_in) -> int {
    const bool ppeg = desc->pos == RRT_POS_PPEG;
    for (int i = 0; i < (ppeg ? 3 : 1); ++i)
      if (!gr->pos_w[i] || (w->pos_b[i] && !gr->pos_b[i])) return RRT_E_INVALID;
    float* nxt = (cur == b.dxa) ? b.dxb : b.dxa;
    float* dbp[3] = {w->pos_b[0] ? gr->pos_b[0] : nullptr, w->pos_b[1] ? gr->pos_b[1] : nullptr,
                     w->pos_b[2] ? gr->pos_b[2] : nullptr};
    hipError_t pe = launch_peg_backward(xstage_in, cur, w->pos_w, nxt, gr->pos_w, dbp, N, D, desc->peg_k,
                                        desc->peg_1d, ppeg, b.pegws, st);
    if (pe != hipSuccess) return (int)pe;
    cur = nxt;
    return RRT_OK;
  };
  if (desc->cr_msa && desc->ffn) {
    rc = ffn_backward(w->crmsa, gr->crmsa, s.xcr, RRT_MAX_RMSA_LAYERS);
    if (rc) return rc;
  }
  if (desc->cr_msa) {
    const rrt_attn_weights& cw = w->crmsa;
    const rrt_attn_grads& cg = gr->crmsa;
    if (!cg.norm || !cg.qkv_w || !cg.proj_w || !cg.proj_b || (cw.qkv_b && !cg.qkv_b)) return RRT_E_INVALID;
    if (desc->crmsa_mlp ? (!gr->phi0_w || !gr->phi2_w) : !gr->phi) return RRT_E_INVALID;
    const GridDev gd8 = to_dev(g8);
    const int k = desc->crmsa_k, R8 = gd8.rs * gd8.rs;
    const float* x1 = L > 0 ? (desc->ffn ? s.xf[L - 1] : s.xout[L - 1]) : (desc->pos && desc->pos_pos == -1 ? s.xp : x);
    const float* up = cur;                                   // d x2 (after the FFN backward when ffn = 1)
    float* dx1 = (cur == b.dxa) ? b.dxb : b.dxa;
    RRT_TRY(launch_crmsa_tokdot(up, nullptr, nullptr, nullptr, s.rep2, b.dWd, D, k, gd8, st));
    RRT_TRY(launch_crmsa_wsum(up, s.wdisp, b.d_rep2, D, k, gd8, st));
    RRT_TRY(launch_apply_drop_mask(b.d_rep2, k * R8, D, dc.thresh, dc.seed(drop_seed, DROP_LAYER_CRMSA),
                                   dc.scale * br.attn[L], st));
    RRT_TRY(launch_linear_backward(b.d_rep2, s.rep_o, cw.proj_w, b.d_rep_o, cg.proj_w, cg.proj_b, k * R8, D, D, desc->compute,
                                   b.lin, st, b.wt_cr_proj));
    RRT_TRY(launch_attention_backward(s.rep_qkv, nullptr, s.rep_o, b.d_rep_o, b.d_rep_qkv, nullptr, b.attnpart_cr, k,
                                      R8, D, desc->crmsa_heads, 0, st));
    RRT_TRY(launch_linear_backward(b.d_rep_qkv, s.rep, cw.qkv_w, b.d_rep, cg.qkv_w, cg.qkv_b, k * R8, 3 * D, D, desc->compute,
                                   b.lin, st, b.wt_cr_qkv));
    RRT_TRY(launch_crmsa_tokdot(x1, s.mean_rstd, cw.norm_w, cw.norm_b, b.d_rep, b.dC, D, k, gd8, st));
    RRT_TRY(launch_crmsa_bwd_region(s.logits, b.dC, b.dWd, b.dlg, b.Cw, k, gd8, st));
    if (desc->crmsa_mlp) {
      // phi = Linear(D, D/4) -> Tanh -> Linear(D/4, k) (rmsa.py:248-252): d hid, then the two weight gradients as
      // TN products over the Np8 slots and the logits' path into v as a GEMM row block (pad slots carry v = 0 and
      // tanh(0) = 0, so they add nothing to dW1 / dW2)
      const int hdim = D / 4;
      RRT_TRY(launch_crmsa_mlp_bwd_hidden(s.hid, b.dlg, w->phi2_w, b.th, b.dhid, (size_t)gd8.Np, hdim, k, st));
      RRT_TRY(launch_gemm_tn(b.dlg, b.th, gr->phi2_w, b.tnscratch, gd8.Np, k, hdim, st));          // dW2 [k, D/4]
      RRT_TRY(launch_gemm_tn(b.dhid, s.v8, gr->phi0_w, b.tnscratch, gd8.Np, hdim, D, st));         // dW1 [D/4, D]
      if (hdim % 32 == 0) {
        RRT_TRY(launch_transpose(w->phi0_w, b.w1t, hdim, D, st));                                  // W1^T [D, D/4]
        LinearEpilogue ev{};
        ev.prec = desc->compute;
        RRT_TRY(launch_linear(b.dhid, b.w1t, b.dvphi, gd8.Np, D, hdim, ev, st));                   // d v_phi = d hid . W1
      } else {                                             // hidden width not a GEMM K tile (dim % 128 != 0)
        RRT_TRY(launch_small_k_matmul(b.dhid, w->phi0_w, b.dvphi, gd8.Np, D, hdim, st));
      }
      // (the reduce launch writes LayerNorm's gradients where they belong when the caller's buffer takes 16-byte stores: the
      //  4 KB copy -- and, below, the transpose of d phi -- were two 5 us launches of the dependent chain)
      const bool direct = ((uintptr_t)cg.norm & 15) == 0;
      RRT_TRY(launch_crmsa_bwd_dx(x1, up, s.mean_rstd, cw.norm_w, cw.norm_b, b.dvphi, b.Cw, b.dlg, b.d_rep, dx1,
                                  direct ? cg.norm : b.rows, b.dxpart, D, k, gd8, true, st, nullptr, direct ? defer : nullptr));
      if (!direct) RRT_TRY(hipMemcpyAsync(cg.norm, b.rows, (size_t)2 * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (((uintptr_t)cg.norm & 15) == 0) {
      RRT_TRY(launch_crmsa_bwd_dx(x1, up, s.mean_rstd, cw.norm_w, cw.norm_b, w->phi, b.Cw, b.dlg, b.d_rep, dx1,
                                  cg.norm, b.dxpart, D, k, gd8, false, st, gr->phi, defer));
    } else {
      RRT_TRY(launch_crmsa_bwd_dx(x1, up, s.mean_rstd, cw.norm_w, cw.norm_b, w->phi, b.Cw, b.dlg, b.d_rep, dx1,
                                  b.rows, b.dxpart, D, k, gd8, false, st));
      RRT_TRY(hipMemcpyAsync(cg.norm, b.rows, (size_t)2 * D * sizeof(float), hipMemcpyDeviceToDevice, st));
      RRT_TRY(launch_transpose(b.rows + 2 * (size_t)D, gr->phi, k, D, st));      // [k, D] -> phi's [D, k]
    }
    cur = dx1;
  }
  for (int li = L - 1; li >= 0; --li) {
    const rrt_attn_weights& lw = w->rmsa[li];
    const rrt_attn_grads& lg = gr->rmsa[li];
    if (!lg.norm || !lg.qkv_w || !lg.proj_w || !lg.proj_b || (lw.qkv_b && !lg.qkv_b) || (desc->epeg && !lg.pe_w))
      return RRT_E_INVALID;
    const GridDev gd = to_dev(g);
    const bool pos_in = desc->pos && ((desc->pos_pos == -1 && li == 0) || (desc->pos_pos == 0 && li == 1));
    const float* xprev = li > 0 ? (desc->ffn ? s.xf[li - 1] : s.xout[li - 1]) : x;   // before a PEG stage, if any
    const float* xin = pos_in ? s.xp : xprev;
    if (desc->ffn) {
      rc = ffn_backward(lw, lg, s.xout[li], li);
      if (rc) return rc;
    }
    RRT_TRY(launch_partition_rows(cur, b.dz, D, gd, dc.thresh, dc.seed(drop_seed, li), dc.scale * br.attn[li], st));
    RRT_TRY(launch_linear_backward(b.dz, s.o[li], lw.proj_w, b.dO, lg.proj_w, lg.proj_b, gd.Np, D, D, desc->compute, b.lin, st,
                                   b.wt_proj[li]));
    const bool e2d = desc->epeg && desc->epeg_2d && desc->epeg_type == RRT_EPEG_ATTN;
    const bool evalue = desc->epeg && desc->epeg_type != RRT_EPEG_ATTN;
    if (e2d) {
      RRT_TRY(launch_attn_scoremap_backward(s.qkv[li], lw.pe_w, b.dO, b.dqkv, lg.pe_w, b.smap, gd.rs * gd.rs, gd.P, D,
                                            desc->n_heads, desc->epeg_k, st));
    } else if (evalue) {
      if (lw.pe_b && !lg.pe_b) return RRT_E_INVALID;
      const int nreg = gd.rs * gd.rs;
      if (desc->epeg_type == RRT_EPEG_VALUE_BF) {
        // attention ran on v' = v + pe: its backward gives dv' (v columns of dqkv) = d pe; the conv read v = v' - pe
        RRT_TRY(launch_attention_backward(s.qkv[li], nullptr, s.o[li], b.dO, b.dqkv, nullptr, b.attnpart, nreg, gd.P, D,
                                          desc->n_heads, 0, st));
        RRT_TRY(launch_copy_cols(b.dpe, b.dqkv, (size_t)gd.Np, D, 3 * D, 2 * D, st));
        RRT_TRY(launch_value_pe_backward(b.dpe, s.qkv[li], s.pe[li], lw.pe_w, b.dqkv, lg.pe_w, lw.pe_b ? lg.pe_b : nullptr, nreg,
                                         gd.P, gd.s, D, desc->n_heads, desc->epeg_k, desc->epeg_2d, st));
      } else {
        // proj read o + pe: d pe = dO; the attention's own output is (o + pe) - pe (into dz, dead since the proj backward)
        RRT_TRY(launch_sub(b.dz, s.o[li], s.pe[li], (size_t)gd.Np * D, st));
        RRT_TRY(launch_attention_backward(s.qkv[li], nullptr, b.dz, b.dO, b.dqkv, nullptr, b.attnpart, nreg, gd.P, D,
                                          desc->n_heads, 0, st));
        RRT_TRY(launch_value_pe_backward(b.dO, s.qkv[li], nullptr, lw.pe_w, b.dqkv, lg.pe_w, lw.pe_b ? lg.pe_b : nullptr, nreg,
                                         gd.P, gd.s, D, desc->n_heads, desc->epeg_k, desc->epeg_2d, st));
      }
    } else
    RRT_TRY(launch_attention_backward(s.qkv[li], desc->epeg ? lw.pe_w : nullptr, s.o[li], b.dO, b.dqkv,
                                      desc->epeg ? lg.pe_w : nullptr, b.attnpart, gd.rs * gd.rs, gd.P, D,
                                      desc->n_heads, desc->epeg ? desc->epeg_k : 0, st, defer, b.tappart[li]));
    RRT_TRY(launch_linear_backward(b.dqkv, s.u[li], lw.qkv_w, b.dz, lg.qkv_w, lg.qkv_b, gd.Np, 3 * D, D, desc->compute, b.lin,
                                   st, b.wt_qkv[li]));                         // dU -> dz (dead)
    float* nxt = (cur == b.dxa) ? b.dxb : b.dxa;
    RRT_TRY(launch_ln_backward(b.dz, xin, lw.norm_w, cur, nxt, lg.norm, ln_part(), N, D, &gd, st, defer));
    cur = nxt;
    if (pos_in) {
      rc = peg_backward(xprev);
      if (rc) return rc;
    }
  }
  if (desc->pos && desc->pos_pos == -1 && L == 0) {       // no R-MSA layer: the PEG stage feeds CR-MSA directly
    rc = peg_backward(x);
    if (rc) return rc;
  }
  if (dx) RRT_TRY(launch_layernorm(cur, desc->all_shortcut ? b.dx2 : nullptr, nullptr, nullptr, dx, N, D, st));
  RRT_TRY(launch_reduce_jobs(rj, st));
#undef RRT_TRY
  return RRT_OK;
}

}  // extern "C"

