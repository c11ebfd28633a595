"""Timeline of rmsa_fused_kernel<.., PROJ> blocks (item, then projection slab) from the RRT_TRACE build.
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_fused_proj.py [N region_num]
Blocks are grouped by what they run: item only (b < lag), item + slab, slab only (b >= n_items)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
raw = C.CDLL(os.environ["RRT_HIP_LIB"])
raw.rrt_debug_trace_fused.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
EV, WAVES = 32, 8192
N, rn = (list(map(int, sys.argv[1:3])) + [9000, 8][len(sys.argv) - 1:])[:2]
D, H, ek = 512, 8, 15
g = _lib.region_grid(N, rn)
Np, R = g.H * g.H, g.regions_side ** 2
u = torch.randn(Np, D, device="cuda"); W = torch.randn(3 * D, D, device="cuda") / D ** 0.5
b = torch.randn(3 * D, device="cuda") * 0.1; pe = torch.randn(H, ek, device="cuda") * 0.2
Wp = torch.randn(D, D, device="cuda") / D ** 0.5; bp = torch.randn(D, device="cuda") * 0.1
res = torch.randn(N, D, device="cuda"); out = torch.empty(N, D, device="cuda")
o = torch.empty(Np, D, device="cuda"); cnt = torch.zeros(R, device="cuda", dtype=torch.int32)
st = torch.cuda.current_stream().cuda_stream
K = int(os.environ.get("TRACE_STATS_K", "0"))          # > 0: with CR-MSA's row records as a by-product (k representatives)
gm2 = torch.randn(D, device="cuda"); phi = torch.randn(D, max(K, 1), device="cuda") * 0.1
part = torch.empty(N * (D // 64) * (2 + max(K, 1)), device="cuda")
if K > 0:
    call = lambda: _lib.check(lib.rrt_rmsa_fused_proj_stats_f32(u.data_ptr(), W.data_ptr(), b.data_ptr(), pe.data_ptr(), Wp.data_ptr(),
                                                                bp.data_ptr(), res.data_ptr(), out.data_ptr(), o.data_ptr(), cnt.data_ptr(),
                                                                gm2.data_ptr(), phi.data_ptr(), K, part.data_ptr(), D, H, ek, C.byref(g), st))
else:
  call = lambda: _lib.check(lib.rrt_rmsa_fused_proj_f32(u.data_ptr(), W.data_ptr(), b.data_ptr(), pe.data_ptr(), Wp.data_ptr(), bp.data_ptr(),
                                                      res.data_ptr(), out.data_ptr(), o.data_ptr(), cnt.data_ptr(), D, H, ek, C.byref(g), st))
for _ in range(3):
    call()
torch.cuda.synchronize()
buf = np.zeros(WAVES * EV, dtype=np.uint64)
raw.rrt_debug_trace_fused(None, 0, 1)
call()
raw.rrt_debug_trace_fused(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(WAVES, EV)
NW = 8
n_items = R * H
props = torch.cuda.get_device_properties(0)
lag = min(props.multi_processor_count & ~7, n_items & ~7)
nb = min(n_items + lag, WAVES // NW)
tb = t[:nb * NW].reshape(nb, NW, EV)[:, :, 1:].astype(np.int64)
start = np.where(tb[:, :, 0] > 0, tb[:, :, 0], np.iinfo(np.int64).max).min(1)
launch0 = start.min()
last = tb.max(axis=(1, 2))
print(f"N={N} rn={rn} R={R} items={n_items} lag={lag} blocks traced={nb}; launch span {last.max() - launch0} cycles")
groups = {"item only": np.arange(0, lag), "item + slab": np.arange(lag, min(n_items, nb)), "slab only": np.arange(n_items, nb)}
for name, idx in groups.items():
    if len(idx) == 0:
        continue
    print(f"==== {name}: {len(idx)} blocks; block start since launch: median {np.median(start[idx] - launch0):.0f}, "
          f"block end: median {np.median(last[idx] - launch0):.0f}, lifetime median {np.median(last[idx] - start[idx]):.0f}")
    for w in (0, 4):
        ts = tb[idx, w, :]
        nev = int((ts > 0).sum(1).max())
        ok = (ts[:, :nev] > 0).all(1)
        rel = ts[:, :nev] - start[idx, None]
        print(f"  -- wave {w}: {nev} events (since the block's first wave entered: median, step)")
        prev = None
        for i in range(nev):
            m = np.median(rel[ok, i])
            print(f"     ev{i + 1:02d} {m:9.0f} {'' if prev is None else f'+{m - prev:.0f}'}")
            prev = m
print("item marks (wave 0): 1 entry | 2 K tile 0 | 3,4 B_0, B_7 | 5 last proj MFMA | 6 QKV in LDS | 7 Q~ | S^T, softmax, PV, O stored, partials | published | merged | O in memory")
print("slab marks compute: entry | region complete | requests issued | K tile 0 published | B_0, B_7, B_14 | last MFMA | stores issued")
print("slab marks loader: entry | region complete | stages issued | K tile 0 landed | K tile 1, 8, 15 landed")
