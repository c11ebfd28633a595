"""Timeline of attn_bwd_kernel waves (resident variant, P <= 208) from the RRT_TRACE build.
    tools/build_ablation.sh trace -DRRT_TRACE
    RRT_HIP_LIB=tools/_abl/librrt_trace.so python tools/trace_attn_bwd.py [R P epeg_k]
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
raw = C.CDLL(os.environ["RRT_HIP_LIB"])
raw.rrt_debug_trace_attn_bwd.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
EV, WAVES = 32, 8192
R, P, ek = (list(map(int, sys.argv[1:4])) + [64, 144, 15][len(sys.argv) - 1:])[:3]
D, heads = 512, 8
dev = "cuda"
qkv = torch.randn(R * P, 3 * D, device=dev) * 0.5
pe = torch.randn(heads, ek, device=dev) * 0.1
o = torch.randn(R * P, D, device=dev)
dO = torch.randn(R * P, D, device=dev)
dqkv = torch.empty(R * P, 3 * D, device=dev)
dpe = torch.empty(heads, ek, device=dev)
need = C.c_size_t()
_lib.check(lib.rrt_region_attention_backward_workspace_size(R, P, D, heads, ek, C.byref(need)), "ws")
ws = torch.zeros(need.value, dtype=torch.uint8, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: _lib.check(lib.rrt_region_attention_backward_f32(p(qkv), p(pe), p(o), p(dO), p(dqkv), p(dpe), R, P, D, heads, ek,
                                                                p(ws), ws.numel(), st), "attention_backward")
for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    call()
e1.record()
torch.cuda.synchronize()
print(f"R={R} P={P} epeg_k={ek}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call (trace build)")
buf = np.zeros(WAVES * EV, dtype=np.uint64)
raw.rrt_debug_trace_attn_bwd(None, 0, 1)
call()
torch.cuda.synchronize()
raw.rrt_debug_trace_attn_bwd(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(WAVES, EV)
live = t[:, 1] > 0
ts_all = t[live][:, 1:].astype(np.int64)
nev_each = (ts_all > 0).sum(1)
t0 = ts_all[:, 0]
print(f"{int(live.sum())} traced waves; events per wave: {dict(zip(*np.unique(nev_each, return_counts=True)))}")
print("block entries (cycles after the first): p50 %d p90 %d max %d" % tuple(np.percentile(t0 - t0.min(), [50, 90, 100])))
names = ["entry", "q,k,v in LDS", "Q~ built", "A0: dO/O rows, D", "A0: scores+softmax", "A0: dA, dS", "A0: dQ~ parked", "pass A done",
         "dO tile in LDS", "B0: S^T, dA^T", "B0: A, dS", "B0: dV, dK stored", "pass B done", "dQ~ tile in LDS", "dq written", "tap gradients"]
for nev in np.unique(nev_each):
    ts = ts_all[nev_each == nev][:, :nev]
    print(f"== waves with {nev} events: {len(ts)}")
    d = np.diff(ts, axis=1)
    if nev == 16:
        for i in range(nev - 1):
            x = d[:, i]
            print(f"   {names[i]:>22} -> {names[i + 1]:<22} median {np.median(x):8.0f}  p10 {np.percentile(x, 10):8.0f}  p90 {np.percentile(x, 90):8.0f}")
    print(f"   lifetime median {np.median(ts[:, -1] - ts[:, 0]):.0f}; last event (cycles after the kernel's first entry) median "
          f"{np.median(ts[:, -1] - t0.min()):.0f} max {int((ts[:, -1] - t0.min()).max())}")
