"""Timeline of rmsa_pair16_kernel waves from the RRT_TRACE build.
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_pair16.py [R P D heads epeg_k]
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
raw = C.CDLL(os.environ["RRT_HIP_LIB"])
raw.rrt_debug_trace_pair16.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
EV, WAVES = 32, 8192
R, P, D, H, ek = (list(map(int, sys.argv[1:6])) + [64, 144, 512, 8, 15][len(sys.argv) - 1:])[:5]
u = torch.randn(R * P, D, device="cuda").bfloat16().view(torch.int16)
W = (torch.randn(3 * D, D, device="cuda") / D ** 0.5).bfloat16().view(torch.int16)
b = torch.randn(3 * D, device="cuda") * 0.1; pe = torch.randn(H, max(ek, 1), device="cuda") * 0.2
o = torch.empty(R * P, D, device="cuda", dtype=torch.int16)
st = torch.cuda.current_stream().cuda_stream
call = lambda: _lib.check(lib.rrt_rmsa_fused16(u.data_ptr(), W.data_ptr(), b.data_ptr(), pe.data_ptr() if ek else None, o.data_ptr(), R, P, D, H, ek, 1, st))
for _ in range(3):
    call()
buf = np.zeros(WAVES * EV, dtype=np.uint64)
raw.rrt_debug_trace_pair16(None, 0, 1)
call()
raw.rrt_debug_trace_pair16(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(WAVES, EV)
idx = np.arange(WAVES)
live = t[:, 1] > 0
print(f"R={R} P={P} D={D} heads={H} epeg_k={ek}: {int(live.sum())} traced waves (first {WAVES // 8} blocks)")
for role, sel in (("waves cw=0", live & (idx % 4 == 0)), ("waves cw=1..3", live & (idx % 4 > 0))):
    ts = t[sel][:, 1:].astype(np.int64)
    nev = int(np.median((ts > 0).sum(1)))
    ok = (ts[:, :nev] > 0).all(1)
    ts = ts[ok][:, :nev]
    print(f"== {role}: {ok.sum()} waves, {nev} events")
    d = np.diff(ts, axis=1)
    for i in range(nev - 1):
        x = d[:, i]
        print(f"   ev{i + 1:02d}->ev{i + 2:02d}  median {np.median(x):8.0f}  p10 {np.percentile(x, 10):8.0f}  p90 {np.percentile(x, 90):8.0f}")
    life = ts[:, -1] - ts[:, 0]
    print(f"   lifetime median {np.median(life):.0f} p10 {np.percentile(life, 10):.0f} p90 {np.percentile(life, 90):.0f}")
# block start spread: when do blocks start relative to the first one (two rounds? one?)
t0 = t[live][:, 1].astype(np.int64)
print("block entry times (cycles after the first): p50 %d p90 %d max %d" % tuple(np.percentile(t0 - t0.min(), [50, 90, 100])))
print("events: 1 entry | 2 first stage issued | 3,5,7 barrier kt=0,1,4 passed | 4,6,8 next stage issued | 9 last projection MFMA | 10 ring dead |"
      " 11 Q^T,K in LDS | 12 stencil done | 13 V^T in LDS | per query tile: S^T issued, softmax done, PV issued, O stored")
