"""Helpers for the -m gpu parity tests: call the C-ABI stage entry points on torch
device buffers (torch is only the allocator / stream provider here)."""
import ctypes as C

import numpy as np
import torch

from rrt_mil_amd import RRTEncoder, _lib

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def stream():
    return torch.cuda.current_stream().cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


def encoder_from_state(state, cfg):
    enc = RRTEncoder(**cfg).eval()
    enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()}, strict=True)
    return enc.to(DEV)


def run_encoder(x, state, cfg):
    enc = encoder_from_state(state, cfg)
    y = enc(dev(x).unsqueeze(0)).squeeze(0)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def linear(A, B, bias=None, q_cols=0, q_scale=1.0, compute=0):
    lib = _lib.load()
    M, K = A.shape
    N = B.shape[0]
    Cout = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(lib.rrt_linear_f32(p(A), p(B), p(bias), p(Cout), M, N, K, q_cols, q_scale, compute, stream()), "linear")
    torch.cuda.synchronize()
    return Cout


def region_attention(qkv, pe_w, n_regions, P, dim, heads, epeg_k):
    lib = _lib.load()
    o = torch.full((n_regions * P, dim), float("nan"), device=DEV)
    _lib.check(lib.rrt_region_attention_f32(p(qkv), p(pe_w), p(o), n_regions, P, dim, heads, epeg_k, stream()),
               "region_attention")
    torch.cuda.synchronize()
    return o


def dropout_keep(seed, layer, rows, cols, p):
    """numpy replica of csrc/common.h::rrt_drop_keep for one layer's proj output [rows, cols]: the boolean keep
    mask that rrt_encoder_forward_train_f32 applies for (drop_p = p, drop_seed = seed); layer = index of the R-MSA
    layer, 100 for CR-MSA's inner attention (api.hip DropCfg::seed)."""
    M32 = np.uint64(0xFFFFFFFF)
    base = np.uint64(seed)
    lseed = ((base ^ (base >> np.uint64(32))) + np.uint64(0x9E3779B9) * np.uint64(layer + 1)) & M32
    idx = np.arange(rows * cols, dtype=np.uint64)
    h = ((idx & M32) * np.uint64(0x9E3779B1) + (idx >> np.uint64(32)) * np.uint64(0x85EBCA77) + lseed) & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    thresh = np.uint64(max(1, int(p * 4294967296.0)))
    return (h >= thresh).reshape(rows, cols)
