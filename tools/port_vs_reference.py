"""Container-only: time the oracle's eager port against the REAL reference on the same host cores (SURVEY §8(d):
"restatement / oracle time ratio ... must be within +-10 % and is reported").  Writes
profiles/port_vs_reference_container.json, which bench.py's cpu_baseline leg quotes (the reference itself never
travels to the GPU box)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rrt_mil_amd  # noqa: E402,F401
from rrt_mil_amd import synth  # noqa: E402
from oracle import rrt_oracle  # noqa: E402
from _ref import build_reference_encoder  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
torch.set_num_threads(threads)
cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
state = synth.encoder_state(**cfg)
enc = build_reference_encoder(state, **cfg)
st = {k: torch.from_numpy(v) for k, v in state.items()}
x = torch.from_numpy(synth.bag(9000, 512))


def once(fn):
    t = time.perf_counter()
    fn()
    return time.perf_counter() - t


# interleaved (the host's first seconds of a new op mix run slow: oneDNN primitive caches, thread pool, clocks)
ref_fn, port_fn = (lambda: enc(x.unsqueeze(0))), (lambda: rrt_oracle.forward_eager(x, st, cfg))
tr, tp = [], []
with torch.no_grad():
    for _ in range(3):
        ref_fn(), port_fn()
    for _ in range(21):
        tr.append(once(ref_fn))
        tp.append(once(port_fn))
t_ref, t_port = float(np.median(tr)), float(np.median(tp))
rec = {"reference_ms": round(t_ref * 1e3, 2), "port_ms": round(t_port * 1e3, 2), "ratio_port_over_reference": round(t_port / t_ref, 4),
       "threads": threads, "host": "build container", "torch": torch.__version__,
       "workload": "RRTEncoder(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8).eval(), N=9000, fp32, interleaved, median of 21"}
print(json.dumps(rec))
with open(os.path.join(ROOT, "profiles", "port_vs_reference_container.json"), "w") as fh:
    json.dump(rec, fh, indent=1)
