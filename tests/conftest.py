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
              "crmsa_mlp", "qkv_bias", "epeg_bias", "ffn", "mlp_ratio", "pos", "peg_k", "peg_1d", "peg_bias",
              "epeg_2d", "epeg_type")


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


def mil_case(name):
    """(golden, cfg, state, feats) of an RRTMIL golden (G8 / G11 / G13), regenerated from the closed-form recipe."""
    from rrt_mil_amd import synth
    g = load_golden(name)
    cfg, N = g["cfg"], int(g["n"])
    enc_keys = {k: v for k, v in cfg.items() if k in ("epeg_k", "crmsa_k", "crmsa_mlp")}
    st = synth.mil_state(input_dim=cfg["input_dim"], n_classes=cfg["n_classes"], da_bias=cfg.get("da_bias", False),
                         da_gated=cfg.get("da_gated", False), da_act=cfg.get("da_act", "relu"), **enc_keys)
    if name.startswith("G8"):
        tag = "mil"
    elif name.startswith("G11"):
        tag = "mil/" + name[len("G11_rrtmil_"):]
    else:
        tag = "readme/" + name[len("G13_readme_"):]
    return g, cfg, st, synth.bag(N, cfg["input_dim"], tag=tag, nonneg=True)
