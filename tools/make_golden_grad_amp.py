"""Generate the round-2 fixtures from the REAL reference (build container only):

    python tools/make_golden_grad_amp.py [G15|G16|G17]

G15_grad_*   parameter / input gradients of the reference RRTEncoder in .train() mode with drop_out = 0
             (main.py:466-467 ``loss.backward()``): loss = <y, G>, G closed-form.  The reference module is cast
             to float64 first (its own code, torch autograd), so the values are exact before
             they are stored (float32 samples, float64 checksums) and the per-tensor bounds of the tests measure
             the implementation under test, not the fixture.  Small
             tensors are stored whole, large ones as sampled rows + checksums.
G16_amp_*    the reference forward under ``torch.autocast('cpu', dtype=torch.bfloat16 / float16)`` -- the
             reference's --amp path (main.py:101-102,439 uses torch.cuda.amp.autocast; there is no GPU next to
             the reference, so the CPU autocast policy is what can be pinned: Linear / matmul / conv2d / einsum
             run in the low-precision dtype and hand it on to softmax; LayerNorm and the residual stream stay
             fp32).  Stored next to the fp32 output of the same case.

G17_epeg_*   the reference forward with the EPEG ablations (epeg_2d, epeg_type = value_bf / value_af,
             modules/rmsa.py:76-85,106-129), fp32 eval.

Only arrays are written (inputs regenerate from rrt-mil_amd/synth.py).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rrt_mil_amd  # noqa: E402,F401  (the shim)
from rrt_mil_amd import synth  # noqa: E402
from _ref import build_reference_encoder, load_reference  # noqa: E402
from make_golden import STATE_KEYS, cfg_array, checksums  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)

# (tag -> (N, cfg)); the tags reuse tests/test_hip_parity.py::TRAIN_CASES inputs ("train/<tag>", "train/G/<tag>")
GRAD_CASES = {
    # D = 128, two heads of 64: small enough to store every gradient whole, and trainable on the HIP path
    "d128_n300": (300, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, epeg_k=15, crmsa_k=3)),
    "d128_n700_k21c5_l3": (700, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, epeg_k=21, crmsa_k=5, n_layers=3)),
    "d128_n500_heads1_sc": (500, dict(mlp_dim=128, n_heads=2, crmsa_heads=1, epeg_k=9, crmsa_k=3, all_shortcut=True)),
    "d128_n600_mlp": (600, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, epeg_k=13, crmsa_k=3, crmsa_mlp=True)),
    "d128_n400_ffn": (400, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, epeg_k=15, crmsa_k=3, ffn=True, mlp_ratio=2.0)),
    "d128_n260_ppeg": (260, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, crmsa_k=3, pos="ppeg", pos_pos=-1)),
    # the reference's default `--pos ppeg` invocation: pos_pos = 0 with n_layers = 2 never applies the stage
    # (rrt.py:185 needs i == 1): its parameters get NO gradient (None)
    "d128_n200_ppeg_unused": (200, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, crmsa_k=3, pos="ppeg", pos_pos=0)),
    # D = 512: sampled rows + checksums
    "default_n1500": (1500, dict(mlp_dim=512, epeg_k=15, crmsa_k=3)),
    "default_n9000": (9000, dict(mlp_dim=512, epeg_k=15, crmsa_k=3)),
    "c16_n2600": (2600, dict(mlp_dim=512, epeg_k=15, crmsa_k=1, all_shortcut=True)),
    "brca_r50_heads1_n2000": (2000, dict(mlp_dim=512, epeg_k=17, crmsa_k=3, crmsa_heads=1)),
    "nsclc_plip_mlp_n1800": (1800, dict(mlp_dim=512, epeg_k=13, crmsa_k=3, crmsa_heads=1, all_shortcut=True,
                                        crmsa_mlp=True)),
    "ffn_gelu_n1200": (1200, dict(mlp_dim=512, epeg_k=15, crmsa_k=3, ffn=True, mlp_ratio=2.0)),
    # round 3: the EPEG ablations train too (modules/rmsa.py:76-85,106-129 are differentiable in the reference)
    "d128_n300_attn2d": (300, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, epeg_k=7, crmsa_k=3, epeg_2d=True)),
    "d128_n500_valuebf": (500, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, epeg_k=9, crmsa_k=3, epeg_type="value_bf")),
    "d128_n400_valueaf2d_l3": (400, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, epeg_k=5, crmsa_k=3, epeg_type="value_af",
                                         epeg_2d=True, n_layers=3)),
    "attn2d_n1500_k9": (1500, dict(mlp_dim=512, epeg_k=9, crmsa_k=3, epeg_2d=True)),
    "valuebf2d_n2000_k5": (2000, dict(mlp_dim=512, epeg_k=5, crmsa_k=3, epeg_type="value_bf", epeg_2d=True)),
    "valueaf_n1500_nobias": (1500, dict(mlp_dim=512, epeg_k=15, crmsa_k=1, epeg_type="value_af", epeg_bias=False)),
    # regions of 144 tokens: the 2-D EPEG backward's three [P, P] maps per (region, head) no longer fit the LDS
    "d128_n8000_attn2d_k5": (8000, dict(mlp_dim=128, n_heads=2, crmsa_heads=2, epeg_k=5, crmsa_k=3, epeg_2d=True)),
}

FULL_LIMIT = 20000      # elements: tensors up to this size are stored whole


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def sums64(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), np.abs(a).max(), (a * a).sum()])


def pack(prefix, g, out):
    """gradient tensor -> fixture entries: whole or sampled rows (stored as float32: 6e-8 relative) + float64 checksums."""
    key = prefix.replace(".", "_")
    g = np.asarray(g, dtype=np.float64)
    out[key + "__sums"] = sums64(g)
    if g.size <= FULL_LIMIT:
        out[key + "__full"] = g.astype(np.float32)
    else:
        g2 = g.reshape(g.shape[0], -1)
        step = max(1, g2.shape[0] // 48)
        rows = np.arange(0, g2.shape[0], step)
        out[key + "__rows"] = rows
        out[key + "__vals"] = g2[rows].astype(np.float32)


def gen_grad(only=None):
    for tag, (N, cfg) in GRAD_CASES.items():
        if only and only not in tag:
            continue
        D = cfg["mlp_dim"]
        state = synth.encoder_state(**{k: v for k, v in cfg.items() if k in STATE_KEYS})
        enc = build_reference_encoder(state, drop_out=0., **cfg)
        enc = enc.double().train()                                   # reference code, float64 arithmetic
        x = torch.from_numpy(synth.bag(N, D, tag="train/" + tag)).double().requires_grad_(True)
        G = torch.from_numpy(synth.normal("train/G/" + tag, (N, D))).double()
        y = enc(x.unsqueeze(0)).squeeze(0)
        (y * G).sum().backward()
        out = {"cfg": cfg_array(cfg), "n": np.array(N), "y_sums": checksums(y.detach().numpy())}
        pack("dx", x.grad.numpy(), out)
        none = []
        for name, p in enc.named_parameters():
            if p.grad is None:
                none.append(name)
                continue
            pack("p." + name, p.grad.numpy(), out)
        out["none"] = np.frombuffer("\n".join(none).encode(), dtype=np.uint8) if none else np.zeros(0, np.uint8)
        save("G15_grad_" + tag, **out)
        del enc, y


AMP_CASES = {
    "bf16_d512_n1000": (1000, dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8), torch.bfloat16),
    "f16_d512_n1000": (1000, dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8), torch.float16),
    "bf16_d512_n9000": (9000, dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8), torch.bfloat16),
    "bf16_d512_n9000_c1_sc": (9000, dict(mlp_dim=512, epeg_k=15, crmsa_k=1, region_num=8, all_shortcut=True), torch.bfloat16),
    "bf16_d512_n3000_k21_c5": (3000, dict(mlp_dim=512, epeg_k=21, crmsa_k=5, region_num=8), torch.bfloat16),
    "bf16_d512_n15000_k21_c5": (15000, dict(mlp_dim=512, epeg_k=21, crmsa_k=5, region_num=8), torch.bfloat16),
    "bf16_d512_n30000_rn16": (30000, dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=16), torch.bfloat16),
}


def gen_amp(only=None):
    for tag, (N, cfg, dt) in AMP_CASES.items():
        if only and only not in tag:
            continue
        D = cfg["mlp_dim"]
        state = synth.encoder_state(**{k: v for k, v in cfg.items() if k in STATE_KEYS})
        enc = build_reference_encoder(state, **cfg)
        x = torch.from_numpy(synth.bag(N, D)).unsqueeze(0)
        with torch.no_grad():
            y32 = enc(x).squeeze(0).numpy()
            with torch.autocast("cpu", dtype=dt):
                ya = enc(x).squeeze(0)
        assert ya.dtype == torch.float32        # final LayerNorm runs in fp32 on the fp32 residual stream
        ya = ya.numpy()
        rows = np.unique(np.concatenate([np.arange(0, N, max(1, N // 120)), [N - 1]]))
        d = np.abs(ya.astype(np.float64) - y32)
        save("G16_amp_" + tag, cfg=cfg_array(cfg), n=np.array(N), rows=rows, y_rows=ya[rows], y32_rows=y32[rows],
             y_sums=checksums(ya), dist_fp32=np.array([d.max(), d.mean()]),
             dtype=np.frombuffer(str(dt).encode(), dtype=np.uint8))
        print(f"   autocast vs fp32: max {d.max():.3e} mean {d.mean():.3e}")


# G17: the EPEG ablations (row f4): epeg_2d and epeg_type = value_bf / value_af (modules/rmsa.py:76-85,106-129)
EPEG_CASES = {
    "attn2d_d64_n300": (300, dict(mlp_dim=64, epeg_k=15, crmsa_k=3, epeg_2d=True)),
    "attn2d_d512_n1000_k9": (1000, dict(mlp_dim=512, epeg_k=9, crmsa_k=3, epeg_2d=True)),
    "attn2d_d512_n9000": (9000, dict(mlp_dim=512, epeg_k=15, crmsa_k=3, epeg_2d=True)),
    "valuebf_d64_n300": (300, dict(mlp_dim=64, epeg_k=15, crmsa_k=3, epeg_type="value_bf")),
    "valuebf_d512_n9000": (9000, dict(mlp_dim=512, epeg_k=15, crmsa_k=3, epeg_type="value_bf")),
    "valuebf2d_d512_n3000_k5": (3000, dict(mlp_dim=512, epeg_k=5, crmsa_k=3, epeg_type="value_bf", epeg_2d=True)),
    "valueaf_d64_n700_l3": (700, dict(mlp_dim=64, epeg_k=9, crmsa_k=3, epeg_type="value_af", n_layers=3)),
    "valueaf_d512_n9000": (9000, dict(mlp_dim=512, epeg_k=15, crmsa_k=3, epeg_type="value_af")),
    "valueaf2d_d512_n1000_nobias": (1000, dict(mlp_dim=512, epeg_k=7, crmsa_k=1, epeg_type="value_af", epeg_2d=True,
                                               epeg_bias=False)),
    # round 3: regions of 225 tokens -- the forward's score map no longer fits the LDS (it lives in the workspace)
    "attn2d_d512_n13000_k7": (13000, dict(mlp_dim=512, epeg_k=7, crmsa_k=3, epeg_2d=True)),
}


def gen_epeg(only=None):
    for tag, (N, cfg) in EPEG_CASES.items():
        if only and only not in tag:
            continue
        D = cfg["mlp_dim"]
        state = synth.encoder_state(**{k: v for k, v in cfg.items() if k in STATE_KEYS})
        enc = build_reference_encoder(state, **cfg)
        x = synth.bag(N, D)
        with torch.no_grad():
            y = enc(torch.from_numpy(x).unsqueeze(0)).squeeze(0).numpy()
        if N <= 700:
            save("G17_epeg_" + tag, cfg=cfg_array(cfg), n=np.array(N), y=y)
        else:
            rows = np.unique(np.concatenate([np.arange(0, N, max(1, N // 120)), [N - 1]]))
            save("G17_epeg_" + tag, cfg=cfg_array(cfg), n=np.array(N), rows=rows, y_rows=y[rows], y_sums=checksums(y))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else ""
    sub = sys.argv[2] if len(sys.argv) > 2 else None
    load_reference()
    if which in ("", "G15"):
        gen_grad(sub)
    if which in ("", "G16"):
        gen_amp(sub)
    if which in ("", "G17"):
        gen_epeg(sub)
