#!/usr/bin/env python3
"""Soak of the merged R-MSA + out-projection launch: S streams x many forwards of mixed bag sizes, every output compared
bit for bit with the solo result of its (stream, size).   python tools/soak_merged.py [rounds] [streams]"""
import ctypes as C
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTEncoder, _lib, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
enc = RRTEncoder(**cfg).eval()
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.encoder_state(**cfg).items()}, strict=True)
enc = enc.to(dev)
lib = _lib.load()
# SOAK_DTYPE = f32 (default) | bf16 | f16 | f32x3; SOAK_SOLO = 0 (default) | 1 (the forward's one-bag-in-flight choices: K split
# inside 16-wave blocks for the representatives' GEMMs, CR-MSA's row records from the projection slabs)
enc._desc.compute = {"f32": _lib.COMPUTE_F32, "bf16": _lib.COMPUTE_BF16, "f16": _lib.COMPUTE_F16,
                     "f32x3": _lib.COMPUTE_F32X3}[os.environ.get("SOAK_DTYPE", "f32")]
enc._desc.solo = int(os.environ.get("SOAK_SOLO", "0"))
# SOAK_MIX=1 (round 6, the packed-fp32 hazard's setting): stream 0 runs exact-fp32 forwards with the one-bag-in-flight kernel
# choices (CR-MSA's first pass = crmsa_combine_parts_kernel from the projection slabs' records), the other streams run the
# SOAK_DTYPE (bf16 / f16) forwards beside it -- the streaming fp32 kernel of one bag next to another bag's 16-bit MFMA waves
MIX = os.environ.get("SOAK_MIX") == "1"
MODE_LOW, SOLO_LOW = enc._desc.compute, enc._desc.solo
w = enc._weights()
sizes = [9000, 6200, 7000, 12000, 5000, 10500, 8000, 9000][:max(S, 1)]
big = torch.from_numpy(synth.bag(12000, 512, tag="soak")).to(dev)
# two inputs per stream, alternating between rounds: a slab that read the PREVIOUS forward's attention output (a stale
# cache line) would not reproduce the solo result of the current one
xs2 = [[(big[:m] * (1.0 + 0.003 * i)).contiguous(), (big[:m].flip(0) * (0.9 - 0.002 * i)).contiguous()] for i, m in enumerate(sizes)]
xs = [p[0] for p in xs2]
streams = [torch.cuda.Stream() for _ in sizes]
wss, refs, ys = [], [], []
for i, m in enumerate(sizes):
    need = C.c_size_t()
    _lib.check(lib.rrt_encoder_workspace_size(C.byref(enc._desc), m, C.byref(need)), "ws")
    wss.append(torch.empty(need.value, dtype=torch.uint8, device=dev))
    ys.append([torch.empty_like(xs[i]) for _ in range(4)])


def run(i, j, v=0):
    if MIX:
        enc._desc.compute, enc._desc.solo = (_lib.COMPUTE_F32, 1) if i == 0 else (MODE_LOW, SOLO_LOW)
        enc._desc.weights16_valid = 0
    _lib.check(lib.rrt_encoder_forward_f32(C.byref(enc._desc), C.byref(w), xs2[i][v].data_ptr(), ys[i][j].data_ptr(), sizes[i],
                                           wss[i].data_ptr(), wss[i].numel(), streams[i].cuda_stream), "forward")


for i in range(len(sizes)):
    pair = []
    for v in range(2):
        run(i, 0, v)
        torch.cuda.synchronize()
        pair.append(ys[i][0].clone())
    assert not torch.equal(pair[0], pair[1])
    refs.append(pair)
bad = total = 0
for r in range(rounds):
    for j in range(4):
        for i in range(len(sizes)):
            run(i, j, (r + j) & 1)
    torch.cuda.synchronize()
    for i in range(len(sizes)):
        for j in range(4):
            total += 1
            bad += int(not torch.equal(ys[i][j], refs[i][(r + j) & 1]))
            ys[i][j].fill_(float("nan"))
print(f"soak [{os.environ.get('SOAK_DTYPE', 'f32')}, solo={SOLO_LOW}{', stream 0 = exact fp32 solo' if MIX else ''}]: {total} forwards on {len(sizes)} streams (sizes {sizes}), {bad} differ from the solo result")
sys.exit(1 if bad else 0)
