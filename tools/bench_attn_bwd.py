#!/usr/bin/env python3
"""us per launch of rrt_region_attention_backward_f32:  python tools/bench_attn_bwd.py R P [epeg_k]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
R, P = int(sys.argv[1]), int(sys.argv[2]); ek = int(sys.argv[3]) if len(sys.argv) > 3 else 15
D, h = 512, 8
qkv = torch.randn(R * P, 3 * D, device="cuda") * 0.5; pe = torch.randn(h, ek, device="cuda") * 0.2
o = torch.randn(R * P, D, device="cuda"); do = torch.randn(R * P, D, device="cuda")
dqkv = torch.empty_like(qkv); dpe = torch.empty(h, ek, device="cuda")
need = C.c_size_t(); lib.rrt_region_attention_backward_workspace_size(R, P, D, h, ek, C.byref(need))
ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
f = lambda: _lib.check(lib.rrt_region_attention_backward_f32(qkv.data_ptr(), pe.data_ptr(), o.data_ptr(), do.data_ptr(), dqkv.data_ptr(), dpe.data_ptr(), R, P, D, h, ek, ws.data_ptr(), ws.numel(), st))
for _ in range(5): f()
torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30): f()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 30 * 1e3
print(f"attention backward R={R} P={P}: {us:.1f} us  ({8 * 2.0 * R * P * P * D / us / 1e6:.1f} TFLOP/s at 8 P^2 D products)")
