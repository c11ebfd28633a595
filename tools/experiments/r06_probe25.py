#!/usr/bin/env python3
"""round 6, probe 25: the split-K TN product (weight gradients) with the XCD-aware block order against the plain 3-D grid
(RRT_TN_PLAIN_GRID=1, tuning build): dW-only calls of rrt_linear_backward_f32 for the encoder's two big shapes, HIP-event timed."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rrt_mil_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for M, N, K in ((9216, 1536, 512), (9216, 512, 512), (9216, 512, 2048), (30976, 1536, 512)):
    dY = torch.randn(M, N, device=dev); X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    dW = torch.empty(N, K, device=dev)
    need = C.c_size_t()
    _lib.check(lib.rrt_linear_backward_workspace_size(M, N, K, C.byref(need)), "ws")
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    def call():
        _lib.check(lib.rrt_linear_backward_f32(dY.data_ptr(), X.data_ptr(), W.data_ptr(), None, dW.data_ptr(), None, M, N, K, 0,
                                               ws.data_ptr(), ws.numel(), st), "bwd")
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for rep in range(5):
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * M * N * K
    print(f"dW {M}x{N}x{K}: {min(best):.1f} us (gemm_tn + reduce)  {fl / min(best) / 1e6:.1f} TFLOP/s  reps {[round(b, 1) for b in best]}", flush=True)
