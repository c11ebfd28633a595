#!/usr/bin/env python3
"""Row f1 timing: RRTMIL forward (BASELINE configs[2], C16-R50 shape: N=9000 x 1024 features -> fc 512 + ReLU ->
encoder(epeg_k=15, crmsa_k=1, all_shortcut) -> DAttention -> predictor) -- the one-call HIP path against the
composite path (torch ops for fc / pooling / predictor around the HIP encoder)."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTMIL, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
dev = torch.device("cuda:0")
cfg = dict(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1, all_shortcut=True)
mil = RRTMIL(**cfg).eval()
st = synth.mil_state(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1)
mil.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
mil = mil.to(dev)
x = torch.from_numpy(synth.bag(N, 1024, tag="mil", nonneg=True)).to(dev).unsqueeze(0)


def composite(x):
    with torch.no_grad():
        return mil.predictor(mil.pool_fn(mil.online_encoder(mil.dp(mil.patch_to_emb(x)))))


def timeit(fn, steps=200):
    for _ in range(20):
        fn(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


for mode in ("fp32", "bf16"):
    mil.online_encoder.compute_dtype = None if mode == "fp32" else torch.bfloat16
    a, b = timeit(mil), timeit(composite)
    print(f"{mode}: one-call {a:.3f} ms/slide ({1e3 / a:.0f} slides/s)   composite {b:.3f} ms/slide ({1e3 / b:.0f} slides/s)")
print("logits", mil(x).cpu().numpy(), composite(x).cpu().numpy())
