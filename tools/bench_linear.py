"""GPU micro-benchmark of the stage entry points (not the judged bench).
    python tools/bench_linear.py linear M N K [M N K ...]
    python tools/bench_linear.py attn R P D heads epeg_k
    python tools/bench_linear.py proj16 L region_num        (the 16-bit out-projection + un-partition + residual of a bag)
"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib

lib = _lib.load()
dev = "cuda:0"


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


def main():
    kind = sys.argv[1]
    st = torch.cuda.current_stream().cuda_stream
    if kind == "linear":
        args = list(map(int, sys.argv[2:]))
        for i in range(0, len(args), 3):
            M, N, K = args[i:i + 3]
            A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) / K ** 0.5
            bias = torch.randn(N, device=dev); C = torch.empty(M, N, device=dev)
            f = lambda: _lib.check(lib.rrt_linear_f32(A.data_ptr(), B.data_ptr(), bias.data_ptr(), C.data_ptr(), M, N, K, 0, 1.0, int(os.environ.get("RRT_COMPUTE", "0")), st))
            med, mn = timeit(f)
            print(f"linear M={M} N={N} K={K}: median {med:.1f} us  min {mn:.1f} us  {2.0 * M * N * K / med / 1e6:.1f} TFLOP/s (median)")
    elif kind == "proj16":
        import ctypes as C
        L, rn = int(sys.argv[2]), int(sys.argv[3])
        g = _lib.region_grid(L, rn)
        Np, D = g.H * g.H, 512
        A = torch.randn(Np, D, device=dev).bfloat16().view(torch.int16)
        B = (torch.randn(D, D, device=dev) / D ** 0.5).bfloat16().view(torch.int16)
        bias = torch.randn(D, device=dev); resid = torch.randn(L, D, device=dev); out = torch.empty(L, D, device=dev)
        flush = torch.empty(96 << 20, device=dev)
        def f():
            _lib.check(lib.rrt_linear16_f32(A.data_ptr(), B.data_ptr(), bias.data_ptr(), resid.data_ptr(), out.data_ptr(), Np, D, D,
                                            C.byref(g), 1, st))
        med, mn = timeit(f)
        print(f"proj16 L={L} rn={rn} (M={Np}) cfg={os.environ.get('RRT_LINEAR16_CFG', 'default')}: median {med:.1f} us  min {mn:.1f} us")
    elif kind == "proj32":
        import ctypes as C
        L, rn = int(sys.argv[2]), int(sys.argv[3])
        g = _lib.region_grid(L, rn)
        Np, D = g.H * g.H, 512
        A = torch.randn(Np, D, device=dev); B = torch.randn(D, D, device=dev) / D ** 0.5
        bias = torch.randn(D, device=dev); resid = torch.randn(L, D, device=dev); out = torch.empty(L, D, device=dev)
        f = lambda: _lib.check(lib.rrt_linear_unpartition_residual_f32(A.data_ptr(), B.data_ptr(), bias.data_ptr(), resid.data_ptr(),
                                                                       out.data_ptr(), D, D, C.byref(g), 0, st))
        med, mn = timeit(f)
        print(f"proj32 L={L} rn={rn} (M={Np}) cfg={os.environ.get('RRT_LINEAR_CFG_BIG', 'default')}: median {med:.1f} us  min {mn:.1f} us  "
              f"{2.0 * Np * D * D / med / 1e6:.1f} TFLOP/s")
    elif kind == "attn":
        R, P, D, H, ek = map(int, sys.argv[2:7])
        qkv = torch.randn(R * P, 3 * D, device=dev) * 0.5
        pe = torch.randn(H, max(ek, 1), device=dev) * 0.2
        o = torch.empty(R * P, D, device=dev)
        f = lambda: _lib.check(lib.rrt_region_attention_f32(qkv.data_ptr(), pe.data_ptr() if ek else None, o.data_ptr(), R, P, D, H, ek, st))
        med, mn = timeit(f)
        fl = 4.0 * R * P * P * D
        print(f"attn R={R} P={P} D={D} h={H} ek={ek}: median {med:.1f} us  min {mn:.1f} us  {fl / med / 1e6:.1f} TFLOP/s")


main()
