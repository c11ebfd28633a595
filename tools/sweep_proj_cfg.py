"""Out-projection (+ un-partition + residual) of one bag, fp32: every tile shape x resident-block cap over bag sizes.
Needs the tuning build (the product library does not read the environment):
    tools/build_ablation.sh tune -DRRT_TUNING
    RRT_HIP_LIB=tools/_abl/librrt_tune.so python tools/sweep_proj_cfg.py [L ...]
    RRT_HIP_LIB=tools/_abl/librrt_tune.so SWEEP_BF16=1 [SWEEP_RN=16] python tools/sweep_proj_cfg.py [L ...]     (16-bit operands)
"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
libc = C.CDLL(None)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


D = 512
BF, RN = bool(os.environ.get("SWEEP_BF16")), int(os.environ.get("SWEEP_RN", "8"))
ENV = b"RRT_LINEAR16_CFG" if BF else b"RRT_LINEAR_CFG_BIG"
for L in [int(a) for a in sys.argv[1:]] or [3000, 4096, 5000, 6000, 7000, 9000, 10500, 12000, 13000, 15000]:
    g = _lib.region_grid(L, RN)
    Np = g.H * g.H
    A = torch.randn(Np, D, device=dev); B = torch.randn(D, D, device=dev) / D ** 0.5
    bias = torch.randn(D, device=dev); resid = torch.randn(L, D, device=dev); out = torch.empty(L, D, device=dev)
    f = lambda: _lib.check(lib.rrt_linear_unpartition_residual_f32(A.data_ptr(), B.data_ptr(), bias.data_ptr(), resid.data_ptr(),
                                                                   out.data_ptr(), D, D, C.byref(g), 0, st))
    if BF:
        A16, B16 = A.bfloat16().view(torch.int16), B.bfloat16().view(torch.int16)
        f = lambda: _lib.check(lib.rrt_linear16_f32(A16.data_ptr(), B16.data_ptr(), bias.data_ptr(), resid.data_ptr(), out.data_ptr(), Np, D, D,
                                                    C.byref(g), 1, st))
    libc.unsetenv(ENV)
    res = [("default", timeit(f))]
    shapes = ((9, 1), (8, 1), (6, 1), (4, 1), (9, 2), (8, 2))
    if os.environ.get("SWEEP_WIDE"):          # round 4: 256-column tiles (tuning build instantiations)
        shapes = ((6, 1), (9, 2), (8, 2), (6, 2), (5, 2), (4, 2), (8, 4), (6, 4), (5, 4), (4, 4), (3, 4))
    for mt, nt in shapes:
        for cap in (256, 512, 768, 1024):
            if (nt >= 2 and cap > 512) or (nt == 1 and cap == 256):
                continue
            libc.setenv(ENV, f"{mt},{nt},{cap}".encode(), 1)
            try:
                res.append((f"{mt},{nt},{cap}", timeit(f)))
            except Exception as e:
                res.append((f"{mt},{nt},{cap}", float("inf")))
    best = min(res, key=lambda r: r[1])
    fl = 2.0 * Np * D * D
    print(f"L={L:6d} M={Np:6d}: default {res[0][1]:6.1f} us ({fl / res[0][1] / 1e6:5.1f} TF)   best {best[0]:>9} {best[1]:6.1f} us ({fl / best[1] / 1e6:5.1f} TF)   "
          + "  ".join(f"{n}:{t:.1f}" for n, t in res[1:]), flush=True)
