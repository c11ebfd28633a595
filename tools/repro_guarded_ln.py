"""Reproducer of the round-2 LayerNorm mis-sum (DESIGN.md sections 9 and 12): the column-GUARDED LayerNorm-type kernels
(crmsa_logits, dispatch + LayerNorm, LayerNorm + partition) next to split-bf16 (F32X3) waves of another bag.

    RRT_HIP_LIB=tools/_abl/librrt_guard.so   RRT_NO_CRMSA_REGION4=1 python tools/repro_guarded_ln.py [runs]
    RRT_HIP_LIB=tools/_abl/librrt_guardn.so  RRT_NO_CRMSA_REGION4=1 python tools/repro_guarded_ln.py [runs]
(librrt_guard: tools/build_ablation.sh guard -DRRT_TUNING -DRRT_FORCE_GUARDED; librrt_guardn: the same + -DRRT_PERMLANE_NOPS=4.)
Two forwards in flight on two streams (dim = 512, F32X3 arithmetic, the two-kernel CR-MSA statistics so that
crmsa_logits_kernel runs), compared bit for bit with a forward that had the chip to itself; prints how many of the
concurrent forwards differ and, for the first few, which output rows."""
import ctypes as C
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTEncoder, _lib, synth   # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = {"f32x3": _lib.COMPUTE_F32X3, "bf16": _lib.COMPUTE_BF16, "f32": _lib.COMPUTE_F32}[os.environ.get("REPRO_MODE", "f32x3")]
n = int(os.environ.get("REPRO_N", "3000"))
cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
enc = RRTEncoder(**cfg).eval()
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.encoder_state(**{k: v for k, v in cfg.items() if k != "region_num"}).items()})
enc = enc.to("cuda:0")
lib = _lib.load()
enc._desc.compute = mode
w = enc._weights()
ns = [n, int(os.environ.get("REPRO_N2", "9000"))]                 # stream 0: small bags, stream 1: big ones -> the kernels drift
xs = [torch.from_numpy(synth.bag(m, 512, tag=f"repro/x{m}")).to("cuda:0") for m in ns]
streams = [torch.cuda.Stream() for _ in range(2)]
wss = []
for m in ns:
    need = C.c_size_t()
    _lib.check(lib.rrt_encoder_workspace_size(C.byref(enc._desc), m, C.byref(need)), "ws")
    wss.append(torch.zeros(need.value, dtype=torch.uint8, device="cuda:0"))
reps = [8, 3]                                                       # forwards per stream and round (about equal time)
ys = [[torch.zeros_like(xs[i]) for _ in range(reps[i])] for i in range(2)]


def run(i, j):
    _lib.check(lib.rrt_encoder_forward_f32(C.byref(enc._desc), C.byref(w), xs[i].data_ptr(), ys[i][j].data_ptr(), ns[i],
                                           wss[i].data_ptr(), wss[i].numel(), streams[i].cuda_stream), "forward")


torch.cuda.synchronize()
refs = []
for i in range(2):
    run(i, 0)
    torch.cuda.synchronize()
    refs.append(ys[i][0].clone())
solo_bad = 0
for _ in range(10):
    for i in range(2):
        run(i, 0)
        torch.cuda.synchronize()
        solo_bad += int(not torch.equal(ys[i][0], refs[i]))
bad, total, shown = 0, 0, 0
for r in range(runs):
    for j in range(max(reps)):
        for i in range(2):
            if j < reps[i]:
                run(i, j)
    torch.cuda.synchronize()
    for i in range(2):
        for j in range(reps[i]):
            total += 1
            if not torch.equal(ys[i][j], refs[i]):
                bad += 1
                if shown < 6:
                    rows = torch.nonzero((ys[i][j] != refs[i]).any(dim=1)).flatten()
                    d = (ys[i][j] - refs[i]).abs().max().item()
                    print(f"  round {r} stream {i} forward {j}: {rows.numel()} rows differ (first {rows[:6].tolist()}), max |diff| {d:.3e}")
                    shown += 1
print(f"lib={os.path.basename(os.environ.get('RRT_HIP_LIB', 'librrt_hip.so'))} mode={os.environ.get('REPRO_MODE', 'f32x3')} "
      f"region4_off={bool(os.environ.get('RRT_NO_CRMSA_REGION4'))} N={ns}: solo {solo_bad}/20 differ, concurrent {bad}/{total} differ")
