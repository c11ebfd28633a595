"""Does a small kernel run beside the fused kernel on the same CUs?  Stream A loops the fused kernel, stream B
loops one CR-MSA / proj kernel; report B's kernel time alone and while A runs.
    CORUN_REGIONS=32 python tools/corun_probe.py     # 256 fused blocks: all resident
Round-1 findings (MI355X): with every fused block resident (<= 256 blocks) crmsa_logits keeps its 11 us beside
the fused kernel -- co-residency works (108 KiB + 6 KiB LDS, 176 + 88 VGPRs); with 512 blocks (N = 9000) it
takes 85 us, the same as when co-residency is made impossible: workgroups of the fused kernel that are pending
in the dispatcher hold up other queues' workgroups until a resident block retires (~75 us)."""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib, synth
lib = _lib.load()
dev = "cuda:0"
N, D, k = 9000, 512, 3
g = _lib.region_grid(N, 8)
Np = g.H * g.H
u = torch.randn(Np, D, device=dev); Wq = torch.randn(3 * D, D, device=dev) / 22; bq = torch.zeros(3 * D, device=dev)
pe = torch.randn(8, 15, device=dev) * 0.1; o = torch.empty(Np, D, device=dev)
x1 = torch.randn(N, D, device=dev); gam = torch.ones(D, device=dev); bet = torch.zeros(D, device=dev)
phi = torch.randn(D, k, device=dev) / 22; mr = torch.empty(N, 2, device=dev); lg = torch.empty(Np, k, device=dev)
Wp = torch.randn(D, D, device=dev) / 22; bp = torch.zeros(D, device=dev); xo = torch.empty(N, D, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
NR = int(os.environ.get("CORUN_REGIONS", "64"))
def fused(st): _lib.check(lib.rrt_rmsa_fused_f32(u.data_ptr(), Wq.data_ptr(), bq.data_ptr(), pe.data_ptr(), o.data_ptr(), NR, 144, D, 8, 15, 0, st.cuda_stream))
def logits(st): _lib.check(lib.rrt_crmsa_logits_f32(x1.data_ptr(), gam.data_ptr(), bet.data_ptr(), phi.data_ptr(), mr.data_ptr(), lg.data_ptr(), N, D, k, C.byref(g), st.cuda_stream))
def proj(st): _lib.check(lib.rrt_linear_unpartition_residual_f32(o.data_ptr(), Wp.data_ptr(), bp.data_ptr(), x1.data_ptr(), xo.data_ptr(), D, D, C.byref(g), 0, st.cuda_stream))
def small_time(fn, with_fused, n=40):
    torch.cuda.synchronize()
    if with_fused:
        for _ in range(n // 2 + 6): fused(sa)
        time.sleep(0.0005)
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(sb); fn(sb); b.record(sb); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2], ts[0], ts[-1]
for name, fn in (("crmsa_logits", logits), ("proj", proj)):
    for _ in range(3): fn(sb); fused(sa)
    print(name, "alone  median/min/max us: %.1f %.1f %.1f" % small_time(fn, False))
    print(name, "co-run median/min/max us: %.1f %.1f %.1f" % small_time(fn, True))
t0 = time.perf_counter(); torch.cuda.synchronize()
for _ in range(20): fused(sa)
torch.cuda.synchronize(); print("fused alone us: %.1f" % ((time.perf_counter() - t0) / 20 * 1e6))
