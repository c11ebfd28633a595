"""Timeline of inner_msa_kernel's three roles from the RRT_TRACE build.
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_inner.py [k]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
raw = C.CDLL(os.environ["RRT_HIP_LIB"])
raw.rrt_debug_trace_inner.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
EV, WAVES = 32, 8192
k, D, heads = int(sys.argv[1]) if len(sys.argv) > 1 else 3, 512, 8
M = k * 64
dev = "cuda"
rep = torch.randn(M, D, device=dev); Wq = torch.randn(3 * D, D, device=dev) / D ** 0.5; bq = torch.randn(3 * D, device=dev) * 0.1
Wp = torch.randn(D, D, device=dev) / D ** 0.5; bp = torch.randn(D, device=dev) * 0.1; out = torch.empty(M, D, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
need = C.c_size_t(); lib.rrt_inner_msa_workspace_size(D, k, C.byref(need))
ws = torch.zeros(need.value, dtype=torch.uint8, device=dev)
call = lambda: _lib.check(lib.rrt_inner_msa_f32(p(rep), p(Wq), p(bq), p(Wp), p(bp), p(out), k, D, heads, p(ws), ws.numel(), st))
for _ in range(5):
    call()
torch.cuda.synchronize()
buf = np.zeros(WAVES * EV, dtype=np.uint64)
raw.rrt_debug_trace_inner(None, 0, 1)
call()
raw.rrt_debug_trace_inner(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(WAVES, EV)[:, 1:].astype(np.int64)
NA, NB, NC = k * heads * 24, k * heads, k * 4 * (D // 32)
nblk = min(NA + NB + NC, WAVES // 4)
tb = t[:nblk * 4].reshape(nblk, 4, EV - 1)
t0 = tb[:, :, 0][tb[:, :, 0] > 0].min()
print(f"k={k}: {NA} A + {NB} B + {NC} C blocks; traced {nblk}; launch span {tb.max() - t0} cycles")
for name, lo, hi in (("A", 0, NA), ("B", NA, NA + NB), ("C", NA + NB, min(NA + NB + NC, nblk))):
    if hi <= lo:
        continue
    ts = tb[lo:hi, 0, :]
    nev = int((ts > 0).sum(1).max())
    ts = ts[:, :nev]
    print(f"== role {name}: {hi - lo} blocks; entry since launch: p10 {np.percentile(ts[:, 0] - t0, 10):.0f} median {np.median(ts[:, 0] - t0):.0f} p90 {np.percentile(ts[:, 0] - t0, 90):.0f}; "
          f"end since launch: median {np.median(ts[:, -1] - t0):.0f} max {(ts[:, -1] - t0).max()}")
    d = np.diff(ts, axis=1)
    for i in range(nev - 1):
        print(f"   ev{i + 1:02d}->ev{i + 2:02d}  median {np.median(d[:, i]):8.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
print("A: 1 entry | 2 fragments requested | 3 MFMAs issued | 4 partial in memory | 5 counted")
print("B: 1 entry | 2 producers arrived | 3 K, V tiles in LDS | 4 scores + softmax | 5 O stores issued | 6 O in memory")
print("C: 1 entry | 2 row arrived | 3 partial tile | 4 stored")
