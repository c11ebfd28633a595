#!/usr/bin/env python3
"""us per call of the CR-MSA logits + combine stage: the two chip-wide kernels against crmsa_region4 (one pass over x1,
last-arrival merge), over bag sizes and k.  With a -DRRT_TUNING library (RRT_HIP_LIB=tools/_abl/librrt_tune.so)
RRT_REGION4_CFG selects the block shape."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
D = 512
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for L, k in ((9000, 3), (9000, 5), (15000, 5), (15000, 3), (30000, 3), (30000, 5)):
    g8 = _lib.region_grid(L, 8)
    Np8 = g8.H * g8.H
    x1 = torch.randn(L, D, device=dev)
    gm, bt = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    phi = torch.randn(D, k, device=dev) * 0.1
    mr = torch.empty(L, 2, device=dev)
    lg, wd = torch.empty(Np8, k, device=dev), torch.empty(Np8, k, device=dev)
    rep = torch.empty(k, 64, D, device=dev)
    scratch = torch.zeros(256 + 64 * 16 * 8 * 520 * 4, dtype=torch.uint8, device=dev)
    flush = torch.empty(64 << 20, device=dev)

    def two():
        lib.rrt_crmsa_logits_f32(p(x1), p(gm), p(bt), p(phi), p(mr), p(lg), L, D, k, C.byref(g8), st())
        lib.rrt_crmsa_combine_f32(p(x1), p(gm), p(bt), p(mr), p(lg), p(wd), p(rep), L, D, k, C.byref(g8), st())

    def four():
        rc = lib.rrt_crmsa_region4_f32(p(x1), p(gm), p(bt), p(phi), p(mr), p(lg), p(wd), p(rep), L, D, k, C.byref(g8),
                                       p(scratch), scratch.numel(), st())
        assert rc == 0, rc

    res = {}
    for name, fn in (("two", two), ("region4", four)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            flush.zero_()                      # x1 out of the caches, as after the big R-MSA kernels of a forward
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        res[name] = ts[len(ts) // 2]
    print(f"L={L} P8={g8.s * g8.s} k={k}: two kernels {res['two']:.1f} us, region4 {res['region4']:.1f} us "
          f"(cfg {os.environ.get('RRT_REGION4_CFG', 'default')})", flush=True)
