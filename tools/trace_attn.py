"""Timeline of region_attn_kernel waves from the RRT_TRACE build (tools/build_ablation.sh trace -DRRT_TRACE).
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_attn.py [R P D heads epeg_k]
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib

lib = _lib.load()
raw = C.CDLL(os.environ["RRT_HIP_LIB"])
raw.rrt_debug_trace_attn.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
EV, WAVES = 32, 8192
R, P, D, H, ek = (list(map(int, sys.argv[1:6])) + [64, 144, 512, 8, 15][len(sys.argv) - 1:])[:5]
qkv = torch.randn(R * P, 3 * D, device="cuda") * 0.5
pe = torch.randn(H, max(ek, 1), device="cuda") * 0.2
o = torch.empty(R * P, D, device="cuda")
st = torch.cuda.current_stream().cuda_stream
call = lambda: _lib.check(lib.rrt_region_attention_f32(qkv.data_ptr(), pe.data_ptr() if ek else None, o.data_ptr(), R, P, D, H, ek, st))
for _ in range(3):
    call()
buf = np.zeros(WAVES * EV, dtype=np.uint64)
raw.rrt_debug_trace_attn(None, 0, 1)
call()
raw.rrt_debug_trace_attn(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(WAVES, EV)
live = t[:, 1] > 0
t = t[live]
hw = (t[:, 0] & 0xFFFFFFFF).astype(np.int64)
xcc = (t[:, 0] >> 32).astype(np.int64) & 0xF
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
ts = t[:, 1:].astype(np.int64)
t0 = ts[:, 0].min()
nev = int((ts > 0).sum(1).max())
print(f"{len(t)} waves traced, {nev} events/wave; kernel span {(ts[ts > 0].max() - t0)} cycles (s_memtime ticks)")
names = ["entry", "Q landed", "Q~ built"]
c = 0
while len(names) < nev - 1:
    names += [f"c{c} barrier", f"c{c} S^T issued", f"c{c} max done", f"c{c} softmax done", f"c{c} PV issued"]
    c += 1
names = names[:nev - 1] + ["end"]
d = np.diff(ts[:, :nev], axis=1)
ok = (ts[:, :nev] > 0).all(1)
print("phase durations (cycles): median / p10 / p90 over waves")
for i in range(nev - 1):
    x = d[ok, i]
    print(f"  {names[i]:>16s} -> {names[i + 1]:<16s} {np.median(x):8.0f} {np.percentile(x, 10):8.0f} {np.percentile(x, 90):8.0f}")
life = ts[ok, nev - 1] - ts[ok, 0]
print(f"wave lifetime: median {np.median(life):.0f}  p10 {np.percentile(life, 10):.0f}  p90 {np.percentile(life, 90):.0f}")
start = ts[ok, 0] - t0
print("wave start time percentiles (cycles):", [int(np.percentile(start, q)) for q in (0, 25, 50, 75, 100)])
# residency: how many waves share a SIMD at once (by physical id)
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
print("distinct physical SIMDs used:", len(np.unique(key[ok])), " waves per SIMD (total):", np.round(ok.sum() / len(np.unique(key[ok])), 2))
