"""Timeline of rmsa_fused_kernel waves from the RRT_TRACE build.
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_fused.py [R P D heads epeg_k]
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
raw = C.CDLL(os.environ["RRT_HIP_LIB"])
raw.rrt_debug_trace_fused.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
EV, WAVES = 32, 8192
R, P, D, H, ek = (list(map(int, sys.argv[1:6])) + [64, 144, 512, 8, 15][len(sys.argv) - 1:])[:5]
u = torch.randn(R * P, D, device="cuda"); W = torch.randn(3 * D, D, device="cuda") / D ** 0.5
b = torch.randn(3 * D, device="cuda") * 0.1; pe = torch.randn(H, max(ek, 1), device="cuda") * 0.2
o = torch.empty(R * P, D, device="cuda")
st = torch.cuda.current_stream().cuda_stream
call = lambda: _lib.check(lib.rrt_rmsa_fused_f32(u.data_ptr(), W.data_ptr(), b.data_ptr(), pe.data_ptr() if ek else None, o.data_ptr(), R, P, D, H, ek, 0, st))
for _ in range(3):
    call()
buf = np.zeros(WAVES * EV, dtype=np.uint64)
raw.rrt_debug_trace_fused(None, 0, 1)
call()
raw.rrt_debug_trace_fused(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(WAVES, EV)
NW = 8                                            # waves per block (RRT_TRACE_INIT(blockIdx.x * 8 + wave))
nb = (t[:, 1] > 0).sum() // NW
tb = t[:nb * NW].reshape(nb, NW, EV)[:, :, 1:].astype(np.int64)
t0 = np.where(tb[:, :, 0] > 0, tb[:, :, 0], np.iinfo(np.int64).max).min(1)          # first wave entry of the block
first = tb[:, :, 0].min(1) <= np.percentile(tb[:, :, 0].min(1), 45)                     # first-round blocks (2 per CU)
for w in range(NW):
    ts = tb[:, w, :]
    nev = int((ts > 0).sum(1).max())
    rel = ts[:, :nev] - t0[:, None]
    ok = (ts[:, :nev] > 0).all(1)
    print(f"== wave {w}: {nev} events; time since the block's first wave entered (median | first-round blocks | step)")
    prev = None
    for i in range(nev):
        m = np.median(rel[ok, i]); m1 = np.median(rel[ok & first, i])
        print(f"   ev{i + 1:02d}  {m:9.0f}  {m1:9.0f}  {'' if prev is None else f'+{m - prev:.0f}'}")
        prev = m
print("marks: 1 entry | 2 K tile 0 published | 3,4 B_0, B_7 | 5 last proj MFMA | 6 QKV in LDS | 7 Q~ built | per tile: S^T issued, softmax done, PV issued, O stored | quarter done | partials published | merged")
