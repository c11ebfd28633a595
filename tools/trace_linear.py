"""Timeline of linear_ws_kernel waves from the RRT_TRACE build.
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_linear.py [M N K]
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_linear.py proj [L region_num]     (the fp32 R-MSA out-projection
        with its un-partition + residual epilogue: M = H * H slots, N = K = 512)
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
raw = C.CDLL(os.environ["RRT_HIP_LIB"])
raw.rrt_debug_trace_linear.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
EV, WAVES = 32, 8192
st = torch.cuda.current_stream().cuda_stream
if len(sys.argv) > 1 and sys.argv[1] == "proj":
    L, rn = (list(map(int, sys.argv[2:4])) + [9000, 8][len(sys.argv) - 2:])[:2]
    g = _lib.region_grid(L, rn)
    M, N, K = g.H * g.H, 512, 512
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") / K ** 0.5
    bias = torch.randn(N, device="cuda"); resid = torch.randn(L, N, device="cuda"); out = torch.empty(L, N, device="cuda")
    call = lambda: _lib.check(lib.rrt_linear_unpartition_residual_f32(A.data_ptr(), B.data_ptr(), bias.data_ptr(), resid.data_ptr(),
                                                                      out.data_ptr(), N, K, C.byref(g), 0, st))
    print(f"out-projection of a bag of L={L} tokens (region_num={rn}): M={M} slots, N=K=512, un-partition + residual epilogue")
else:
    M, N, K = (list(map(int, sys.argv[1:4])) + [9216, 1536, 512][len(sys.argv) - 1:])[:3]
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") / K ** 0.5
    bias = torch.randn(N, device="cuda"); Cc = torch.empty(M, N, device="cuda")
    call = lambda: _lib.check(lib.rrt_linear_f32(A.data_ptr(), B.data_ptr(), bias.data_ptr(), Cc.data_ptr(), M, N, K, 0, 1.0, 0, st))
for _ in range(3):
    call()
buf = np.zeros(WAVES * EV, dtype=np.uint64)
raw.rrt_debug_trace_linear(None, 0, 1)
call()
raw.rrt_debug_trace_linear(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(WAVES, EV)
idx = np.arange(WAVES)
live = t[:, 1] > 0
t0 = t[live][:, 1].astype(np.int64)
tl = np.array([row[row > 0][-1] for row in t[live][:, 1:].astype(np.int64)])
print(f"{int(live.sum())} traced waves; first entry -> last event of any wave: {int(tl.max() - t0.min())} cycles; "
      f"block entries spread over {int(np.percentile(t0 - t0.min(), 90))} cycles (p90)")
wave = idx % 6
for role, sel, names in (("compute", live & (wave < 4), None), ("loader", live & (wave >= 4), None)):
    ts = t[sel][:, 1:].astype(np.int64)
    nev = int((ts > 0).sum(1).max())
    ok = (ts[:, :nev] > 0).all(1)
    ts = ts[ok][:, :nev]
    print(f"== {role}: {ok.sum()} waves, {nev} events")
    d = np.diff(ts, axis=1)
    for i in range(nev - 1):
        x = d[:, i]
        print(f"   ev{i + 1:02d}->ev{i + 2:02d}  median {np.median(x):8.0f}  p10 {np.percentile(x, 10):8.0f}  p90 {np.percentile(x, 90):8.0f}")
    life = ts[:, -1] - ts[:, 0]
    print(f"   lifetime median {np.median(life):.0f} p10 {np.percentile(life, 10):.0f} p90 {np.percentile(life, 90):.0f}")
print("compute events per tile: [barrier kt0, kt1, kt8, kt9, last MFMA issued, stores issued] x tiles, after ev01 = entry")
print("loader events: entry, first stage issued, then per tile [kt0 landed, barrier, kt8 landed, barrier]")
