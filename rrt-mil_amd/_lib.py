"""ctypes binding of librrt_hip.so (the C ABI declared in include/rrt_hip.h).

The library is the product: if it is missing this module raises -- there is no
CPU or PyTorch fallback behind the HIP path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librrt_hip.so")

RRT_MAX_RMSA_LAYERS = 8
RRT_MAX_CRMSA_K = 8
ABI_VERSION = 27

_f32p = C.POINTER(C.c_float)


class Grid(C.Structure):
    _fields_ = [("L", C.c_int64), ("H", C.c_int32), ("s", C.c_int32),
                ("regions_side", C.c_int32), ("add", C.c_int64)]


class EncoderDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_heads", C.c_int32), ("n_rmsa_layers", C.c_int32),
                ("region_num", C.c_int32), ("region_size", C.c_int32), ("min_region_num", C.c_int32),
                ("min_region_ratio", C.c_float), ("epeg", C.c_int32), ("epeg_k", C.c_int32),
                ("cr_msa", C.c_int32), ("crmsa_k", C.c_int32), ("crmsa_heads", C.c_int32),
                ("crmsa_mlp", C.c_int32), ("all_shortcut", C.c_int32), ("compute", C.c_int32),
                ("ffn", C.c_int32), ("ffn_act", C.c_int32), ("ffn_hidden", C.c_int32),
                ("pos", C.c_int32), ("pos_pos", C.c_int32), ("peg_k", C.c_int32), ("peg_1d", C.c_int32),
                ("epeg_2d", C.c_int32), ("epeg_type", C.c_int32), ("weights16_valid", C.c_int32),
                ("solo", C.c_int32)]


class AttnWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm_w", "norm_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
                                          "pe_w", "pe_b", "norm2_w", "norm2_b", "fc1_w", "fc1_b",
                                          "fc2_w", "fc2_b")]


class EncoderWeights(C.Structure):
    _fields_ = [("rmsa", AttnWeights * RRT_MAX_RMSA_LAYERS), ("crmsa", AttnWeights),
                ("phi", C.c_void_p), ("phi0_w", C.c_void_p), ("phi2_w", C.c_void_p),
                ("norm_w", C.c_void_p), ("norm_b", C.c_void_p),
                ("pos_w", C.c_void_p * 3), ("pos_b", C.c_void_p * 3), ("version", C.c_uint64)]


class AttnGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm", "qkv_w", "qkv_b", "proj_w", "proj_b", "pe_w", "pe_b", "norm2",
                                          "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class EncoderGrads(C.Structure):
    _fields_ = [("rmsa", AttnGrads * RRT_MAX_RMSA_LAYERS), ("crmsa", AttnGrads), ("phi", C.c_void_p),
                ("phi0_w", C.c_void_p), ("phi2_w", C.c_void_p), ("norm", C.c_void_p),
                ("pos_w", C.c_void_p * 3), ("pos_b", C.c_void_p * 3)]


class MilDesc(C.Structure):
    _fields_ = [("enc", EncoderDesc), ("input_dim", C.c_int32), ("emb_act", C.c_int32),
                ("n_classes", C.c_int32), ("pool_hidden", C.c_int32), ("pool_act", C.c_int32),
                ("pool_gated", C.c_int32), ("input16", C.c_int32)]


class MilWeights(C.Structure):
    _fields_ = [("enc", EncoderWeights)] + [(n, C.c_void_p) for n in (
        "emb_w", "emb_b", "pool_a_w", "pool_a_b", "pool_b_w", "pool_b_b", "pool_c_w", "pool_c_b",
        "pred_w", "pred_b")]


class Bag(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("n_tokens", C.c_int64)]


ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
ACT_BY_NAME = {"relu": ACT_RELU, "gelu": ACT_GELU, "tanh": ACT_TANH}   # anything else: no activation

# name -> (restype, argtypes); every symbol include/rrt_hip.h declares
SIGNATURES = {
    "rrt_abi_version": (C.c_int, []),
    "rrt_strerror": (C.c_char_p, [C.c_int]),
    "rrt_region_grid": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.POINTER(Grid)]),
    "rrt_encoder_workspace_size": (C.c_int, [C.POINTER(EncoderDesc), C.c_int64, C.POINTER(C.c_size_t)]),
    "rrt_encoder_plan": (C.c_int, [C.POINTER(EncoderDesc), C.c_int64, C.POINTER(C.c_int32)]),
    "rrt_encoder_forward_f32": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rrt_encoder_forward_events_f32": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p,
                                                 C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p,
                                                 C.POINTER(C.c_void_p)]),
    "rrt_encoder_batch_workspace_size": (C.c_int, [C.POINTER(EncoderDesc), C.c_int32, C.c_int64, C.POINTER(C.c_size_t)]),
    "rrt_encoder_forward_batch_f32": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p, C.c_void_p,
                                                C.c_int32, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rrt_phase_gate_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "rrt_phase_gate_destroy": (C.c_int, [C.c_void_p]),
    "rrt_encoder_forward_gated_f32": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p,
                                                C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p,
                                                C.c_void_p, C.POINTER(C.c_void_p)]),
    "rrt_ln_partition_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int32, C.POINTER(Grid), C.c_void_p]),
    "rrt_linear_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                 C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "rrt_linear_unpartition_residual_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Grid),
                                                      C.c_int32, C.c_void_p]),
    "rrt_region_attention_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "rrt_rmsa_fused_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p]),
    "rrt_rmsa_fused_proj_f32": (C.c_int, [C.c_void_p] * 10 + [C.c_int32] * 3 + [C.c_void_p] * 2),
    "rrt_device_error": (C.c_int, [C.c_int32]),
    "rrt_rmsa_fused_proj_stats_f32": (C.c_int, [C.c_void_p] * 12 + [C.c_int32, C.c_void_p] + [C.c_int32] * 3 + [C.c_void_p] * 2),
    "rrt_crmsa_combine_parts_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32, C.POINTER(Grid), C.c_void_p]),
    "rrt_debug_rmsa_fused_proj_f32": (C.c_int, [C.c_void_p] * 10 + [C.c_int32] * 3 + [C.c_void_p] + [C.c_int32] * 3 + [C.c_void_p]),
    "rrt_cast16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "rrt_ln_partition16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                     C.POINTER(Grid), C.c_int32, C.c_void_p]),
    "rrt_linear16_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int32, C.c_int32, C.POINTER(Grid), C.c_int32,
                                                      C.c_void_p]),
    "rrt_rmsa_fused16": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p]),
    "rrt_rmsa_pair16_proj": (C.c_int, [C.c_void_p] * 10 + [C.c_int32] * 3 + [C.POINTER(Grid), C.c_int32, C.c_void_p]),
    "rrt_cast_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rrt_ln_partition_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                         C.POINTER(Grid), C.c_void_p]),
    "rrt_linear_split_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int32, C.c_int32, C.POINTER(Grid), C.c_void_p]),
    "rrt_rmsa_fused_x3": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p]),
    "rrt_crmsa_logits_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_int32,
                                                         C.POINTER(Grid), C.c_void_p]),
    "rrt_crmsa_combine_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32,
                                                          C.POINTER(Grid), C.c_void_p]),
    "rrt_crmsa_region_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_int32, C.POINTER(Grid), C.c_void_p]),
    "rrt_crmsa_region4_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_int32, C.POINTER(Grid), C.c_void_p,
                                                         C.c_size_t, C.c_void_p]),
    "rrt_crmsa_stream4_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_int32, C.POINTER(Grid), C.c_void_p,
                                                         C.c_size_t, C.c_void_p]),
    "rrt_crmsa_dispatch_ln_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32,
                                                              C.POINTER(Grid), C.c_void_p]),
    "rrt_crmsa_mlp_logits_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "rrt_layernorm_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int32, C.c_void_p]),
    "rrt_mil_workspace_size": (C.c_int, [C.POINTER(MilDesc), C.c_int64, C.POINTER(C.c_size_t)]),
    "rrt_mil_forward_f32": (C.c_int, [C.POINTER(MilDesc), C.POINTER(MilWeights), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t,
                                      C.c_void_p]),
    "rrt_pool_workspace_size": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "rrt_pool_predict_f32": (C.c_int, [C.c_void_p] * 12 + [C.c_int32, C.c_int64] + [C.c_int32] * 5 +
                             [C.c_void_p, C.c_size_t, C.c_void_p]),
    "rrt_attn_pool_workspace_size": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "rrt_attn_pool_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rrt_attn_pool_backward_f32": (C.c_int, [C.c_void_p] * 14 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                                                 C.c_void_p]),
    "rrt_executor_create": (C.c_int, [C.POINTER(EncoderDesc), C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "rrt_executor_create_on_streams": (C.c_int, [C.POINTER(EncoderDesc), C.c_int32, C.POINTER(C.c_void_p), C.c_int64,
                                                 C.POINTER(C.c_void_p)]),
    "rrt_executor_forward": (C.c_int, [C.c_void_p, C.POINTER(EncoderWeights), C.POINTER(Bag), C.c_int32,
                                       C.c_void_p]),
    "rrt_executor_destroy": (C.c_int, [C.c_void_p]),
    "rrt_region_attention_backward_workspace_size": (C.c_int, [C.c_int32] * 5 + [C.POINTER(C.c_size_t)]),
    "rrt_region_attention_backward_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 5 + [C.c_void_p, C.c_size_t,
                                                                                       C.c_void_p]),
    "rrt_layernorm_backward_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.POINTER(Grid), C.c_void_p,
                                                                C.c_size_t, C.c_void_p]),
    "rrt_reduce_partials_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int32, C.c_int64, C.c_int64] + [C.c_int32] * 4 + [C.c_void_p]),
    "rrt_linear_backward_workspace_size": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "rrt_linear_backward_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                             C.c_void_p, C.c_size_t, C.c_void_p]),
    "rrt_encoder_train_sizes": (C.c_int, [C.POINTER(EncoderDesc), C.c_int64, C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_size_t)]),
    "rrt_encoder_forward_train_f32": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p,
                                                C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_float,
                                                C.c_uint64, C.POINTER(C.c_float), C.c_void_p]),
    "rrt_encoder_backward_f32": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.POINTER(EncoderGrads), C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_size_t, C.c_float, C.c_uint64, C.POINTER(C.c_float),
                                           C.c_void_p]),
    "rrt_linear_act_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                        C.c_void_p]),
}

COMPUTE_F32, COMPUTE_BF16, COMPUTE_F16, COMPUTE_F32X3 = 0, 1, 2, 3
POS_NONE, POS_PEG, POS_PPEG = 0, 1, 2
EPEG_ATTN, EPEG_VALUE_BF, EPEG_VALUE_AF = 0, 1, 2

PLAN_FUSED, PLAN_FUSED_PROJ, PLAN_FUSED16, PLAN_FUSED_X3, PLAN_CRMSA_PARTS = 1, 2, 4, 8, 16     # rrt_encoder_plan flags
# stage-boundary event slots of rrt_encoder_forward_events_f32 (enum in include/rrt_hip.h)
EV_START, EV_LN_PARTITION, EV_QKV, EV_ATTN, EV_PROJ, EV_CR_COMBINE, EV_CR_INNER, EV_END, EV_COUNT = range(9)

_lib = None


class RRTHipError(RuntimeError):
    pass


def load(path=None):
    """dlopen librrt_hip.so and bind every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RRT_HIP_LIB") or LIB_PATH   # RRT_HIP_LIB: ablation builds (tools/)
    if not os.path.exists(p):
        raise RRTHipError(
            f"{p} is missing: the HIP extension is the only compute path. Build it with "
            "`python rrt-mil_amd/build.py` (needs /opt/rocm/bin/hipcc).")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.rrt_abi_version() != ABI_VERSION:
        raise RRTHipError(f"librrt_hip.so ABI {lib.rrt_abi_version()} != binding {ABI_VERSION}; rebuild")
    if path is None:
        _lib = lib
    return lib


E_HANDOVER = -4      # rrt_hip.h RRT_E_HANDOVER: a merged launch gave up its bounded in-launch wait (sticky, process-wide)


def device_error(clear=False):
    """The process's sticky hand-over error word (include/rrt_hip.h rrt_device_error): non-zero after a merged R-MSA launch
    gave up its bounded wait; every later forward then returns RRT_E_HANDOVER until the word is cleared.  Recovery:
    ``torch.cuda.synchronize()``, discard the outputs of the forwards that were in flight, ``device_error(clear=True)``, run
    them again.  No device synchronisation happens here (a host read of pinned memory)."""
    return int(load().rrt_device_error(1 if clear else 0))


def check(rc, what=""):
    if rc != 0:
        msg = load().rrt_strerror(rc).decode()
        if rc == -2:
            raise NotImplementedError(f"rrt_hip {what}: {msg}")
        if rc == E_HANDOVER:
            msg += (" -- sticky until cleared: torch.cuda.synchronize(), discard the outputs of the forwards in flight, "
                    "rrt_mil_amd.device_error(clear=True), run them again")
        raise RRTHipError(f"rrt_hip {what} failed ({rc}): {msg}")


def region_grid(L, region_num=8, region_size=0, min_region_num=0, min_region_ratio=0.0):
    g = Grid()
    check(load().rrt_region_grid(L, region_num, region_size, min_region_num, min_region_ratio, C.byref(g)),
          "rrt_region_grid")
    return g
