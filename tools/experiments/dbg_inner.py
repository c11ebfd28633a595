"""debug: inner_msa_kernel vs the three-launch form, where do they differ"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import _lib
lib = _lib.load()
k, D, heads = int(sys.argv[1]) if len(sys.argv) > 1 else 3, 512, 8
M = k * 64
dev = "cuda"
torch.manual_seed(1)
rep = torch.randn(M, D, device=dev); Wq = torch.randn(3 * D, D, device=dev) / D ** 0.5; bq = torch.randn(3 * D, device=dev) * 0.1
Wp = torch.randn(D, D, device=dev) / D ** 0.5; bp = torch.randn(D, device=dev) * 0.1
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
need = C.c_size_t(); lib.rrt_inner_msa_workspace_size(D, k, C.byref(need))
ws = torch.zeros(need.value, dtype=torch.uint8, device=dev)
qkv_t = torch.empty(M, 3 * D, device=dev); o_t = torch.empty(M, D, device=dev); y_t = torch.empty(M, D, device=dev)
_lib.check(lib.rrt_linear_f32(p(rep), p(Wq), p(bq), p(qkv_t), M, 3 * D, D, D, 64 ** -0.5, 0, st))
_lib.check(lib.rrt_region_attention_f32(p(qkv_t), None, p(o_t), k, 64, D, heads, 0, st))
_lib.check(lib.rrt_linear_f32(p(o_t), p(Wp), p(bp), p(y_t), M, D, D, 0, 1.0, 0, st))
torch.cuda.synchronize()
import time
for it in range(6):
    out = torch.full((M, D), float("nan"), device=dev)
    t0 = time.perf_counter()
    _lib.check(lib.rrt_inner_msa_f32(p(rep), p(Wq), p(bq), p(Wp), p(bp), p(out), k, D, heads, p(ws), ws.numel(), st))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d = (out - y_t).abs()
    bad = d > 1e-3
    nanc = int(torch.isnan(out).sum())
    # scratch: partials and o
    scr = ws[256:].view(torch.float32)
    part = scr[:4 * M * 3 * D].view(4, M, 3 * D)
    qkv_sum = part.sum(0)
    qkv_sum[:, :D] = (qkv_sum[:, :D] + bq[:D]) * 64 ** -0.5
    qkv_sum[:, D:] += bq[D:]
    dq = (qkv_sum - qkv_t).abs().max().item()
    o = scr[4 * M * 3 * D:4 * M * 3 * D + M * D].view(M, D)
    do = (o - o_t).abs()
    print(f"it {it}: {dt*1e6:.0f} us  max diff {d[~torch.isnan(d)].max().item() if nanc < d.numel() else float('nan'):.3e} bad {int(bad.sum())} nan {nanc} | qkv partial-sum diff {dq:.2e} | o diff max {do.max().item():.2e} bad rows {sorted(set((do > 1e-3).nonzero()[:, 0].tolist()))[:10]} bad cols(head) {sorted(set(((do > 1e-3).nonzero()[:, 1] // 64).tolist()))}")
    if bad.any():
        idx = bad.nonzero()
        print("   out bad rows", sorted(set(idx[:, 0].tolist()))[:20], "col tiles(32)", sorted(set((idx[:, 1] // 32).tolist()))[:20])
print("device error", lib.rrt_device_error(0))
# per-split partials against the exact K-quarter products
scr = ws[256:].view(torch.float32)
part = scr[:4 * M * 3 * D].view(4, M, 3 * D)
for s_ in range(4):
    ref = rep[:, s_ * 128:(s_ + 1) * 128].double() @ Wq[:, s_ * 128:(s_ + 1) * 128].double().T
    dd = (part[s_].double() - ref).abs()
    badt = (dd > 1e-3)
    rows = sorted(set((badt.nonzero()[:, 0] // 16).tolist()))
    cols = sorted(set((badt.nonzero()[:, 1] // 16).tolist()))
    print(f"split {s_}: max diff {dd.max().item():.3e} bad {int(badt.sum())} / {dd.numel()}; bad row tiles(16) {rows[:12]} col tiles(16) {cols[:12]}{'...' if len(cols) > 12 else ''}")
    if s_ == 0 and badt.any():
        r, c = badt.nonzero()[0].tolist()
        print("   first bad", r, c, part[s_][r, c].item(), ref[r, c].item(), " row r other splits:", [part[x][r, c].item() for x in range(4)])
        # is the value some other element?
        v = part[s_][r, c].item()
        close = ((ref - v).abs() < 1e-4).nonzero()
        print("   value matches ref at", close[:5].tolist())
ref0 = rep[:, :128].double() @ Wq[:, :128].double().T
badt = ((part[0].double() - ref0).abs() > 1e-3).nonzero()
import collections
print("lr", sorted(collections.Counter((badt[:, 0] % 16).tolist()).items()))
print("lg", sorted(collections.Counter(((badt[:, 1] % 16) // 4).tolist()).items()))
print("r ", sorted(collections.Counter((badt[:, 1] % 4).tolist()).items()))
print("wave", sorted(collections.Counter(((badt[:, 1] // 16) % 4).tolist()).items()))
print("rowtile i", sorted(collections.Counter(((badt[:, 0] // 16) % 2).tolist()).items()))
print("block(col tile of 64)", sorted(collections.Counter((badt[:, 1] // 64).tolist()).items()))
print("block(row tile 32)", sorted(collections.Counter((badt[:, 0] // 32).tolist()).items()))
o = scr[4 * M * 3 * D:4 * M * 3 * D + M * D].view(M, D)
for r in (11, 12, 13, 15, 28):
    print("row", r, "got", [f"{v:.4f}" for v in o[r, :6].tolist()], "want", [f"{v:.4f}" for v in o_t[r, :6].tolist()], "ratio", [f"{(a / b):.3f}" for a, b in zip(o[r, :6].tolist(), o_t[r, :6].tolist())])
# does a bad row equal the attention computed with a shifted/other query or a missing key group?
q = qkv_t[:64, :64].double(); kk = qkv_t[:64, D:D + 64].double(); vv = qkv_t[:64, 2 * D:2 * D + 64].double()
S = q @ kk.T
for name, mask in (("all keys", torch.ones(64, dtype=torch.bool)), ("keys 0..47", torch.arange(64) < 48), ("keys%16<12", (torch.arange(64) % 16) < 12)):
    Sm = S.clone(); Sm[:, ~mask.to(S.device)] = -1e30
    Pm = torch.softmax(Sm, -1)
    print(name, "row 12 diff", (Pm @ vv)[12].sub(o[12, :64].double()).abs().max().item(), "row 3 diff", (Pm @ vv)[3].sub(o[3, :64].double()).abs().max().item())
do = (o - o_t).abs() > 1e-3
idx = do.nonzero()
print("bad count", len(idx), "lr", sorted(collections.Counter((idx[:, 0] % 16).tolist()).items()))
cols = idx[:, 1] % 64
print("dt", sorted(collections.Counter((cols // 16).tolist()).items()), "lg", sorted(collections.Counter(((cols % 16) // 4).tolist()).items()), "r", sorted(collections.Counter((cols % 4).tolist()).items()))
print("wave(q tile)", sorted(collections.Counter(((idx[:, 0] % 64) // 16).tolist()).items()), "head", sorted(collections.Counter((idx[:, 1] // 64).tolist()).items()))
