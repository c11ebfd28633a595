"""Timeline of crmsa_combine_parts_kernel waves from the RRT_TRACE build.
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_combine_parts.py [L k]
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
raw = C.CDLL(os.environ["RRT_HIP_LIB"])
raw.rrt_debug_trace_crmsa.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
EV, WAVES = 32, 8192
L, k = (list(map(int, sys.argv[1:3])) + [9000, 3][len(sys.argv) - 1:])[:2]
D = 512
dev = "cuda"
g8 = _lib.region_grid(L, 8)
Np8 = g8.H * g8.H
x1 = torch.randn(L, D, device=dev); gm = torch.ones(D, device=dev); bt = torch.zeros(D, device=dev)
phi = torch.randn(D, k, device=dev) * 0.1
S = (2 + k + 3) // 4 * 4
part = torch.randn(L, D // 64, S, device=dev).abs() + 0.5
wd = torch.empty(Np8, k, device=dev); rep = torch.empty(k, 64, D, device=dev); gbs = torch.empty(16, device=dev)
flush = torch.empty(64 << 20, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: _lib.check(lib.rrt_crmsa_combine_parts_f32(p(x1), p(part), p(gm), p(bt), p(phi), p(wd), p(rep), L, D, k, C.byref(g8), st))
for _ in range(3):
    call()
if os.environ.get("TRACE_COLD") == "1":
    flush.zero_()
torch.cuda.synchronize()
buf = np.zeros(WAVES * EV, dtype=np.uint64)
raw.rrt_debug_trace_crmsa(None, 0, 1)
call()
raw.rrt_debug_trace_crmsa(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(WAVES, EV)
live = t[:, 1] > 0
ts_all = t[live][:, 1:].astype(np.int64)
nev_each = (ts_all > 0).sum(1)
print(f"L={L} k={k} P8={g8.s * g8.s}: {int(live.sum())} traced waves; events per wave: {np.bincount(nev_each)[1:].nonzero()[0] + 1}")
t0 = ts_all[:, 0]
print("wave entries (cycles after the first): p50 %d p90 %d max %d" % tuple(np.percentile(t0 - t0.min(), [50, 90, 100])))
nev = int(nev_each.max())
ts = ts_all[nev_each == nev][:, :nev]
d = np.diff(ts, axis=1)
for i in range(nev - 1):
    x = d[:, i]
    print(f"   ev{i + 1:02d}->ev{i + 2:02d}  median {np.median(x):8.0f}  p10 {np.percentile(x, 10):8.0f}  p90 {np.percentile(x, 90):8.0f}")
print(f"   lifetime median {np.median(ts[:, -1] - ts[:, 0]):.0f}; last event (cycles after the kernel's first entry) median "
      f"{np.median(ts[:, -1] - t0.min()):.0f} max {int((ts[:, -1] - t0.min()).max())}")
print("events: 1 entry | 2 every load requested | 3 logits in LDS (records landed; wave 3: G/B first) | 4 barrier | "
      "5 region statistics + coefficients | 6 barrier | 7 contraction (x1 rows landed) | 8 barrier | 9 representatives stored | "
      "10 (slab 0) dispatch weights written")
