"""What does each kernel of the CR-MSA tail cost the fused R-MSA kernel when it runs beside it?

Stream A launches the fused kernel K times back to back; stream B meanwhile loops ONE tail kernel (or the whole
tail chain) until A is done.  Reported per co-runner X:
    fused us/launch alone and beside X, how many X launches completed in the window, X's own solo time, and
    cost = (T_both - T_alone) / n_X  -- MFMA-kernel microseconds lost per launch of X --
against X's solo duration (cost / solo = 1 would mean no overlap at all, 0 perfect hiding).
    python tools/corun_matrix.py [K]
"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib, synth   # noqa: E402

lib = _lib.load()
dev = "cuda:0"
N, D, k = 9000, 512, 3
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = _lib.region_grid(N, 8)
Np = g.H * g.H
R = 64
t = lambda *s: torch.randn(*s, device=dev)
u = t(Np, D); Wq = t(3 * D, D) / 22; bq = torch.zeros(3 * D, device=dev); pe = t(8, 15) * 0.1
o = torch.empty(Np, D, device=dev)
x = t(N, D); x1 = t(N, D); gam = torch.ones(D, device=dev); bet = torch.zeros(D, device=dev)
phi = t(D, k) / 22
lg = torch.empty(Np, k, device=dev); wd = torch.empty(Np, k, device=dev); rep = torch.empty(k * R, D, device=dev)
scr = torch.zeros(256 + 64 * 8 * 3 * 520 * 4 + 4096, dtype=torch.uint8, device=dev)
Wp = t(D, D) / 22; bp = torch.zeros(D, device=dev); xo = torch.empty(N, D, device=dev)
rqkv = torch.empty(k * R, 3 * D, device=dev); ro = torch.empty(k * R, D, device=dev); rep2 = t(k * R, D)
ub = torch.empty(Np, D, device=dev); y = torch.empty(N, D, device=dev)
# CORUN_PRIO="a,b": stream priorities (lower = more urgent; e.g. "-1,0": the launch under test outranks its co-runner)
_pr = [int(v) for v in os.environ.get("CORUN_PRIO", "0,0").split(",")]
sa, sb = torch.cuda.Stream(priority=_pr[0]), torch.cuda.Stream(priority=_pr[1])
ck = _lib.check


def fused(st): ck(lib.rrt_rmsa_fused_f32(u.data_ptr(), Wq.data_ptr(), bq.data_ptr(), pe.data_ptr(), o.data_ptr(), R, 144, D, 8, 15, 0, st.cuda_stream))
def proj(st): ck(lib.rrt_linear_unpartition_residual_f32(o.data_ptr(), Wp.data_ptr(), bp.data_ptr(), x.data_ptr(), xo.data_ptr(), D, D, C.byref(g), 0, st.cuda_stream))
def lnp(st): ck(lib.rrt_ln_partition_f32(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), ub.data_ptr(), N, D, C.byref(g), st.cuda_stream))
def region4(st): ck(lib.rrt_crmsa_region4_f32(x1.data_ptr(), gam.data_ptr(), bet.data_ptr(), phi.data_ptr(), None, lg.data_ptr(), wd.data_ptr(), rep.data_ptr(), N, D, k, C.byref(g), scr.data_ptr(), scr.numel(), st.cuda_stream))
def rqkv_f(st): ck(lib.rrt_linear_f32(rep.data_ptr(), Wq.data_ptr(), bq.data_ptr(), rqkv.data_ptr(), k * R, 3 * D, D, D, 0.125, 0, st.cuda_stream))
def rattn(st): ck(lib.rrt_region_attention_f32(rqkv.data_ptr(), None, ro.data_ptr(), k, R, D, 8, 0, st.cuda_stream))
def rproj(st): ck(lib.rrt_linear_f32(ro.data_ptr(), Wp.data_ptr(), bp.data_ptr(), rep2.data_ptr(), k * R, D, D, 0, 1.0, 0, st.cuda_stream))
def disp(st): ck(lib.rrt_crmsa_dispatch_ln_f32(x1.data_ptr(), None, wd.data_ptr(), rep2.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), N, D, k, C.byref(g), st.cuda_stream))
def tail(st):
    region4(st); rqkv_f(st); rattn(st); rproj(st); disp(st); lnp(st)


def solo_us(fn, n=60):
    for _ in range(5): fn(sb)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(sb)
    for _ in range(n): fn(sb)
    b.record(sb)
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def window(fnA, X, nx_guess, KA=K):
    """KA launches of fnA on stream A; X looped on stream B (nx_guess launches enqueued: enough to cover).  Returns
    (A's us / launch, number of X launches that finished before A's last one did)."""
    torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    evs = []
    a0.record(sa)
    if X is not None:
        # B first gets a head start of a few launches so that it is in steady state when A starts
        for _ in range(3): X(sb)
    for i in range(KA): fnA(sa)
    a1.record(sa)
    if X is not None:
        for i in range(nx_guess):
            X(sb)
            e = torch.cuda.Event(enable_timing=True); e.record(sb); evs.append(e)
    torch.cuda.synchronize()
    ta = a0.elapsed_time(a1) * 1e3
    done = sum(1 for e in evs if a0.elapsed_time(e) * 1e3 <= ta) if evs else 0
    return ta / KA, done


# clocks up
for _ in range(60): fused(sa)
torch.cuda.synchronize()
for nameA, fnA in (("fused", fused), ("proj", proj)):
    alone = min(window(fnA, None, 0)[0] for _ in range(3))
    print(f"== {nameA} alone: {alone:.1f} us / launch (K = {K})")
    print(f"{'co-runner':12s} {'solo us':>8s} {'A us':>8s} {'nX':>5s} {'X us beside':>11s} {'cost us/X':>10s} {'cost/solo':>9s}")
    for name, X in (("ln_part", lnp), ("region4", region4), ("rep_qkv", rqkv_f), ("rep_attn", rattn), ("rep_proj", rproj),
                    ("dispatch", disp), ("tail(all)", tail), ("proj", proj), ("fused", fused)):
        if name == nameA:
            continue
        s_us = solo_us(X)
        guess = int(K * alone * 3 / max(s_us, 1.0)) + 20
        res = [window(fnA, X, guess) for _ in range(3)]
        a_us, nx = sorted(res)[1]
        cost = (a_us - alone) * K / max(nx, 1)
        print(f"{name:12s} {s_us:8.1f} {a_us:8.1f} {nx:5d} {K * a_us / max(nx, 1):11.1f} {cost:10.1f} {cost / s_us:9.2f}")
