"""Timeline of the merged 16-bit launch (rmsa_pair16_kernel<.., PROJ>: item, then the projection slab of a pair that finished
a round earlier) from the RRT_TRACE build.
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_pair16_proj.py [L region_num epeg_k]
Traced: the first 512 blocks (8192 waves): blocks 0..255 run an item only, blocks 256..511 an item and a slab."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
raw = C.CDLL(os.environ["RRT_HIP_LIB"])
raw.rrt_debug_trace_pair16.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
EV, WAVES = 32, 8192
L, rn, ek = (list(map(int, sys.argv[1:4])) + [30000, 16, 15][len(sys.argv) - 1:])[:3]
D, H = 512, 8
g = _lib.region_grid(L, rn)
Np, P, R = g.H * g.H, g.s * g.s, rn * rn
u = torch.randn(Np, D, device="cuda").bfloat16().view(torch.int16)
W = (torch.randn(3 * D, D, device="cuda") / D ** 0.5).bfloat16().view(torch.int16)
Wp = (torch.randn(D, D, device="cuda") / D ** 0.5).bfloat16().view(torch.int16)
b = torch.randn(3 * D, device="cuda") * 0.1; bp = torch.randn(D, device="cuda") * 0.1
pe = torch.randn(H, max(ek, 1), device="cuda") * 0.2
x = torch.randn(L, D, device="cuda"); x1 = torch.empty_like(x)
o = torch.empty(Np, D, device="cuda", dtype=torch.int16)
cnt = torch.zeros(R, device="cuda", dtype=torch.int32)
st = torch.cuda.current_stream().cuda_stream
call = lambda: _lib.check(lib.rrt_rmsa_pair16_proj(u.data_ptr(), W.data_ptr(), b.data_ptr(), pe.data_ptr() if ek else None, Wp.data_ptr(),
                                                   bp.data_ptr(), x.data_ptr(), x1.data_ptr(), o.data_ptr(), cnt.data_ptr(), D, H, ek,
                                                   C.byref(g), 1, st))
for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    call()
e1.record(); torch.cuda.synchronize()
print(f"L={L} rn={rn} P={P}: merged launch (+ its counter memset) {e0.elapsed_time(e1) * 100:.1f} us per call")
buf = np.zeros(WAVES * EV, dtype=np.uint64)
raw.rrt_debug_trace_pair16(None, 0, 1)
call()
torch.cuda.synchronize()
raw.rrt_debug_trace_pair16(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(WAVES, EV)
blk = np.arange(WAVES) // 16
for role, sel in (("item-only blocks (0..255)", blk < 256), ("item + slab blocks (256..511)", blk >= 256)):
    ts = t[sel & (t[:, 1] > 0)][:, 1:].astype(np.int64)
    if not len(ts):
        continue
    nev = int(np.median((ts > 0).sum(1)))
    ok = (ts[:, :nev] > 0).all(1) & ((ts > 0).sum(1) == nev)
    ts = ts[ok][:, :nev]
    print(f"== {role}: {ok.sum()} waves, {nev} events")
    d = np.diff(ts, axis=1)
    for i in range(nev - 1):
        xx = d[:, i]
        print(f"   ev{i + 1:02d}->ev{i + 2:02d}  median {np.median(xx):8.0f}  p10 {np.percentile(xx, 10):8.0f}  p90 {np.percentile(xx, 90):8.0f}")
    life = ts[:, -1] - ts[:, 0]
    print(f"   lifetime median {np.median(life):.0f} p10 {np.percentile(life, 10):.0f} p90 {np.percentile(life, 90):.0f}")
print("item events: 1 entry | 2 first stage | 3,5,7 barrier kt=0,1,4 | 4,6,8 next stage issued | 9 last projection MFMA | 10 ring dead | 11 Q^T,K | "
      "12 stencil | 13 V^T | per query tile: S^T, softmax, PV, O stored | item: O in memory | slab: 1 entry, 2 items arrived, 3 first stages, "
      "4/5 K tile 0 landed / barrier, 6/7 K tile 4, 8 last MFMA, 9 stores issued")
