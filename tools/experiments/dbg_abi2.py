"""rrt_encoder_forward_f32 directly on two streams with separate workspaces, bf16, compared with a quiet run"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rrt_mil_amd import RRTEncoder, _lib, synth
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
valid = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
enc = RRTEncoder(**cfg).eval()
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.encoder_state(**cfg).items()}, strict=True)
enc = enc.to("cuda:0")
enc._desc.compute = _lib.COMPUTE_BF16 if (len(sys.argv) <= 3 or sys.argv[3] != "f32") else _lib.COMPUTE_F32
enc._desc.solo = 0
w = enc._weights()
x = torch.randn(N, 512, device="cuda:0")
need = C.c_size_t(); _lib.check(lib.rrt_encoder_workspace_size(C.byref(enc._desc), N, C.byref(need)))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
wss = [torch.zeros(need.value, dtype=torch.uint8, device="cuda:0") for _ in streams]
def fwd(si, y, v):
    enc._desc.weights16_valid = v
    _lib.check(lib.rrt_encoder_forward_f32(C.byref(enc._desc), C.byref(w), x.data_ptr(), y.data_ptr(), N, wss[si].data_ptr(), wss[si].numel(),
                                           streams[si].cuda_stream))
ref = torch.empty_like(x); fwd(0, ref, 0); torch.cuda.synchronize()
r2 = torch.empty_like(x); fwd(1, r2, 0); torch.cuda.synchronize()
print("quiet: two workspaces agree", torch.equal(ref, r2))
for trial in range(5):
    ys = [torch.empty_like(x) for _ in range(12)]
    for i, y in enumerate(ys):
        fwd(i % 2, y, valid)
    torch.cuda.synchronize()
    print(f"N={N} valid={valid} trial {trial}:", ["ok" if torch.equal(y, ref) else f"{(y - ref).abs().max().item():.1e}" for y in ys])
# ---- which workspace buffers differ between a quiet forward and forwards that overlap?
import numpy as np
from rrt_mil_amd.geometry import region_grid
D, k = 512, 3
g = region_grid(N, 8)
Np = g.Np
def al(v): return (v + 255) // 256 * 256
off = 0; table = []
def take(name, nfloat):
    global off
    table.append((name, off, nfloat * 4)); off = al(off + nfloat * 4)
take("w16", 1 * 4 * D * D); take("wcr16", 2 * D * D); take("uo", Np * D); take("qkv", Np * 3 * D); take("xa", N * D); take("proj_cnt", 64)
take("mean_rstd", N * 2); take("logits", Np * k); take("wdisp", Np * k); take("rep", k * 64 * D); take("rep_qkv", k * 64 * 3 * D)
take("rep_o", k * 64 * D); take("rep2", k * 64 * D); take("rep16", k * 64 * D // 2); take("repo16", k * 64 * D // 2)
print("carve total", off, "lib says", need.value)
y0 = torch.empty_like(x); fwd(0, y0, 1); torch.cuda.synchronize()
snap = wss[0].clone()
for trial in range(6):
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    for rep_ in range(3):
        fwd(0, ya, 1); fwd(1, yb, 1)
    torch.cuda.synchronize()
    for si, y in ((0, ya), (1, yb)):
        if torch.equal(y, ref):
            continue
        d = (wss[si] != snap).nonzero().flatten()
        names = []
        for name, o, sz in table:
            m = ((d >= o) & (d < o + sz)).sum().item()
            if m:
                names.append((name, m))
        tail = (d >= off).sum().item()
        print(f"trial {trial} stream {si}: y differs {(y - ref).abs().max().item():.1e}; differing workspace bytes by buffer: {names} beyond-table {tail}")
# ---- structure of the differing rep elements
rep_off, rep_sz = [(o, sz) for name, o, sz in table if name == "rep"][0]
import collections
for trial in range(8):
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    for rep_ in range(3):
        fwd(0, ya, 1); fwd(1, yb, 1)
    torch.cuda.synchronize()
    for si in (0, 1):
        a = wss[si][rep_off:rep_off + rep_sz].view(torch.float32).view(k, 64, D)
        b = snap[rep_off:rep_off + rep_sz].view(torch.float32).view(k, 64, D)
        d = (a != b).nonzero()
        if len(d) == 0:
            continue
        chunks = collections.Counter((int(n), int(r), int(c) // 64) for n, r, c in d.tolist())
        cls = collections.Counter((int(c) % 64) // 4 for _, _, c in d.tolist())
        comps = collections.Counter(int(c) % 4 for _, _, c in d.tolist())
        slabs = collections.Counter(int(c) // 64 for _, _, c in d.tolist())
        print("      components", sorted(comps.items()), "slabs", sorted(slabs.items()), "regions%8", sorted(collections.Counter(int(r) % 8 for _, r, _ in d.tolist()).items()))
        rel = ((a - b).abs() / (b.abs() + 1e-6))[a != b]
        print(f"trial {trial} stream {si}: {len(d)} rep elements differ in {len(chunks)} (n, region, slab) chunks: {sorted(chunks.items())[:6]} | col lane hist {sorted(cls.items())} | rel diff median {rel.median().item():.1e} max {rel.max().item():.1e}")
