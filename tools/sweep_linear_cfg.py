"""Plain fp32 GEMM C = A . B^T + bias (rrt_linear_f32): every tile shape x resident-block cap for given (M, N, K) triples.
    RRT_HIP_LIB=tools/_abl/librrt_tune.so python tools/sweep_linear_cfg.py M N K [M N K ...]      (tuning build)
"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
libc = C.CDLL(None)
prec = int(os.environ.get("RRT_COMPUTE", "0"))


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


args = list(map(int, sys.argv[1:]))
for i in range(0, len(args), 3):
    M, N, K = args[i:i + 3]
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev); Cm = torch.empty(M, N, device=dev)
    f = lambda: _lib.check(lib.rrt_linear_f32(A.data_ptr(), B.data_ptr(), bias.data_ptr(), Cm.data_ptr(), M, N, K, 0, 1.0, prec, st))
    libc.unsetenv(b"RRT_LINEAR_CFG_BIG")
    res = [("default", timeit(f))]
    for mt, nt in ((9, 1), (8, 1), (6, 1), (4, 1), (9, 2), (8, 2)):
        for cap in (256, 512, 768, 1024):
            if (nt == 2 and cap > 512) or (nt == 1 and cap == 256):
                continue
            libc.setenv(b"RRT_LINEAR_CFG_BIG", f"{mt},{nt},{cap}".encode(), 1)
            try:
                res.append((f"{mt},{nt},{cap}", timeit(f)))
            except Exception:
                res.append((f"{mt},{nt},{cap}", float("inf")))
    best = min(res, key=lambda r: r[1])
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} prec={prec}: default {res[0][1]:6.1f} us ({fl / res[0][1] / 1e6:5.1f} TF)   best {best[0]:>9} {best[1]:6.1f} us "
          f"({fl / best[1] / 1e6:5.1f} TF)   " + "  ".join(f"{n}:{t:.1f}" for n, t in res[1:]), flush=True)
