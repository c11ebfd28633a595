import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    if "cfg" in d:
        d["cfg"] = json.loads(bytes(d["cfg"]).decode())
    return d


def golden_names(prefix=""):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix))


STATE_KEYS = ("mlp_dim", "n_layers", "n_heads", "epeg", "epeg_k", "cr_msa", "crmsa_k",
              "crmsa_mlp", "qkv_bias", "epeg_bias", "ffn", "mlp_ratio")


def synth_case(g):
    """(x, state, cfg) regenerated from the closed-form recipe for a golden record."""
    from rrt_mil_amd import synth
    cfg = g["cfg"]
    state = synth.encoder_state(**{k: v for k, v in cfg.items() if k in STATE_KEYS})
    x = synth.bag(int(g["n"]), cfg.get("mlp_dim", 512))
    return x, state, cfg


@pytest.fixture(scope="session")
def golden():
    return load_golden
