"""CPU: the oracle (oracle/rrt_oracle.py) against the golden vectors that
tools/make_golden.py captured from the real reference (modules/rrt.py::RRTEncoder,
torch CPU fp32, eval).  Pins the oracle before anything is compared with it."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, synth_case
from oracle import rrt_oracle as O

SMALL = [n for n in golden_names("G") if not n.startswith(("G0", "G8", "G11", "G13", "G15", "G16", "G20"))   # G8/G11/G13 = RRTMIL goldens; G15 / G16 / G20: below
         and int(load_golden(n)["n"]) <= 4096]
LARGE = ["G3_d512_n9000", "G18_d512_n13000", "G19_d512_n5600_k21_c5", "G5_d512_n9000_c1_sc", "G17_epeg_attn2d_d512_n9000", "G17_epeg_valuebf_d512_n9000",
         "G17_epeg_valueaf_d512_n9000"]


def _ref_and_pick(g, y):
    if "y" in g:
        return g["y"], y
    return g["y_rows"], y[g["rows"]]


@pytest.mark.parametrize("name", SMALL + LARGE)
def test_eager_port_matches_reference(name):
    g = load_golden(name)
    x, st, cfg = synth_case(g)
    y = O.forward_eager(x, st, cfg).numpy()
    ref, got = _ref_and_pick(g, y)
    # same aten op sequence on the same torch build -> (near) bit-identical; 2e-6 leaves
    # room for a different BLAS thread split on another host
    assert np.abs(got - ref).max() <= 2e-6
    if "y_sums" in g:
        s = np.array([y.astype(np.float64).sum(), np.abs(y.astype(np.float64)).sum()])
        assert np.allclose(s, g["y_sums"][:2], rtol=1e-6, atol=1e-2)


@pytest.mark.parametrize("name", SMALL + ["G3_d512_n9000"])
def test_f64_truth_matches_reference(name):
    g = load_golden(name)
    x, st, cfg = synth_case(g)
    y = O.forward_f64(x, st, cfg)
    ref, got = _ref_and_pick(g, y)
    assert np.abs(got - ref).max() <= 5e-6     # fp32 reference vs fp64 truth: ~1e-6 observed


def test_stage_taps_d64():
    """Per-stage pins (forward hooks on the reference): R-MSA scores before EPEG,
    attention output before proj, x after the R-MSA layer, CR-MSA representatives."""
    g = load_golden("G1_d64_n300")
    x, st, cfg = synth_case(g)
    taps = {}
    O.forward_f64(x, st, cfg, taps)
    assert np.abs(taps["layers.0.attn.attn.scores_in"] - 0).max() > 0
    # pe:in is the raw score map S (hook input), pe:out the EPEG term E; oracle tap holds S+E
    SE = g["layers_0_attn_attn_pe__in"] + g["layers_0_attn_attn_pe__out"]
    assert np.abs(taps["layers.0.attn.attn.scores_in"] - SE).max() <= 2e-5
    assert np.abs(taps["layers.0.attn.attn.proj_in"] - g["layers_0_attn_attn_proj__in"]).max() <= 5e-6
    assert np.abs(taps["layers.0.out"] - g["layers_0__out"][0]).max() <= 5e-6
    assert np.abs(taps["cr_msa.rep"] - g["cr_msa_attn_attn__in"]).max() <= 5e-6
    assert np.abs(taps["cr_msa.out"] - g["cr_msa__out"][0]).max() <= 5e-6


def test_flops_formula():
    # BASELINE.md §3 table
    assert abs(O.flops_per_bag(9000) / 1e9 - 22.88) < 0.01
    assert abs(O.flops_per_bag(512) / 1e9 - 1.65) < 0.01
    assert abs(O.flops_per_bag(30000, region_num=16) / 1e9 - 74.25) < 0.01


MIL = ["G8_rrtmil_n1000"] + golden_names("G11") + golden_names("G13")


@pytest.mark.parametrize("name", MIL)
def test_mil_f64_truth_matches_reference(name):
    """Row f1: the oracle's float64 RRTMIL (patch_to_emb -> encoder -> DAttention -> predictor) against
    the real reference RRTMIL's logits / attention / raw scores (tools/make_golden.py G8, G11, and G13 =
    the six published configs of the reference README)."""
    from conftest import mil_case
    g, cfg, st, feats = mil_case(name)
    logits, attn, raw = O.mil_forward_f64(feats, st, cfg)
    assert np.abs(logits - g["logits"][0]).max() <= 2e-5
    assert np.abs(attn - g["attn"][0]).max() <= 1e-7          # softmax weights ~ 1/N
    if "attn_raw" in g:
        assert np.abs(raw - g["attn_raw"][0]).max() <= 2e-5


# ------------------------------------------------------------------ G15: the reference's own gradients (row f2)
def grad_fixture(name):
    """{'dx' | parameter name: (rows or None, values, float64 checksums)} + names the reference left without a gradient"""
    g = load_golden(name)
    out = {}
    for key in g:
        if not key.endswith("__sums"):
            continue
        base = key[:-len("__sums")]
        if base + "__full" in g:
            out[base] = (None, g[base + "__full"], g[key])
        else:
            out[base] = (g[base + "__rows"], g[base + "__vals"], g[key])
    none = bytes(g["none"]).decode().split("\n") if g["none"].size else []
    return g, out, none


def grad_compare(got, entry, tol, what, floor=0.0):
    """per-tensor bound: max error <= tol * that tensor's own largest entry (checksum[2]); floor only for tensors
    the caller knows to be zero up to rounding"""
    rows, vals, sums = entry
    got = np.asarray(got, dtype=np.float64)
    pick = got if rows is None else got.reshape(got.shape[0], -1)[rows]
    scale = max(float(sums[2]), floor, 1e-30)
    err = np.abs(pick.reshape(vals.shape) - vals).max() / scale
    assert np.isfinite(got).all(), what
    assert err <= tol, f"{what}: max error {err:.2e} of the tensor's own largest entry {sums[2]:.3e}"
    # whole-tensor checksums (sum and sum of squares) catch errors outside the sampled rows
    assert abs(got.sum() - sums[0]) <= 50 * tol * max(sums[1], floor), what
    assert abs((got * got).sum() - sums[3]) <= 50 * tol * max(sums[3], floor * floor), what


@pytest.mark.parametrize("name", golden_names("G15"))
def test_eager_port_gradients_match_reference(name):
    """The backward oracle (torch autograd over the eager port, float64 leaves) against the gradients the real
    reference produced (its own modules cast to float64, tools/make_golden_grad_amp.py): every parameter and dL/dx,
    each within 1e-6 of ITS OWN largest entry; parameters the reference leaves without a gradient get none here."""
    import torch
    from rrt_mil_amd import synth
    g, fx, none = grad_fixture(name)
    cfg, N = g["cfg"], int(g["n"])
    tag = name[len("G15_grad_"):]
    st = synth.encoder_state(**{k: v for k, v in cfg.items() if k in __import__("conftest").STATE_KEYS})
    x = synth.bag(N, cfg["mlp_dim"], tag="train/" + tag)
    G = synth.normal("train/G/" + tag, (N, cfg["mlp_dim"]))
    y, x_leaf, params = O.forward_eager(x, st, cfg, grad=True)
    (y * torch.from_numpy(G).double()).sum().backward()
    grad_compare(x_leaf.grad.numpy(), fx["dx"], 1e-6, "dx")
    for pname, p in params.items():
        if pname in none:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, pname
            continue
        key = "p_" + pname.replace(".", "_")
        zero = pname.endswith("pe.bias")                 # Identity 2: exactly zero up to rounding
        grad_compare(p.grad.numpy(), fx[key], 1e-6, pname, floor=1e-6 if zero else 0.0)
    assert set("p_" + n.replace(".", "_") for n in params if n not in none) | {"dx"} == set(fx)


# ------------------------------------------------------------------ G16: the reference under autocast (bf16 / fp16)
@pytest.mark.parametrize("name", golden_names("G16"))
def test_eager_port_autocast_matches_reference(name):
    """forward_eager(autocast=dtype) = the same aten op sequence under torch.autocast('cpu', dtype) against the real
    reference under the same context.  Same ops, same torch build: bit-identical here; the bound leaves room for one
    low-precision rounding flip per row on a host with a different BLAS split."""
    import torch
    g = load_golden(name)
    x, st, cfg = synth_case(g)
    dt = {"torch.bfloat16": torch.bfloat16, "torch.float16": torch.float16}[bytes(g["dtype"]).decode()]
    y = O.forward_eager(x, st, cfg, autocast=dt).numpy()
    d = np.abs(y[g["rows"]].astype(np.float64) - g["y_rows"])
    fp32_dist = float(g["dist_fp32"][0])
    assert d.max() <= 0.25 * fp32_dist and d.mean() <= 1e-5, (d.max(), d.mean(), fp32_dist)
    # and the autocast run really differs from the fp32 run by the recorded distance
    d32 = np.abs(g["y_rows"].astype(np.float64) - g["y32_rows"])
    assert d32.max() <= fp32_dist + 1e-7 and d32.max() >= 0.2 * fp32_dist


@pytest.mark.parametrize("name,attn", [("G16_amp_bf16_d512_n1000", False), ("G16_amp_bf16_d512_n1000", True),
                                       ("G16_amp_f16_d512_n1000", True), ("G16_amp_bf16_d512_n3000_k21_c5", True)])
def test_lowp_restatement_is_autocast_class(name, attn):
    """The float64 restatement of the HIP path's reduced modes (explicit bf16 / fp16 rounding of the GEMM operands,
    and of Q~ / K / P / V in the region attention) sits at least as close to the fp32 reference as the reference's
    own autocast run does, and within 1.25x that distance of the autocast run itself."""
    g = load_golden(name)
    x, st, cfg = synth_case(g)
    dtype = "bf16" if "bf16" in name else "f16"
    y = O.forward_f64(x, st, cfg, lowp=O.LowP(dtype, attn=attn))[g["rows"]]
    to_fp32 = np.abs(y - g["y32_rows"])
    to_amp = np.abs(y - g["y_rows"])
    amp_to_fp32 = np.abs(g["y_rows"].astype(np.float64) - g["y32_rows"])
    assert to_fp32.mean() <= amp_to_fp32.mean() and to_fp32.max() <= 1.1 * amp_to_fp32.max(), (to_fp32.max(), amp_to_fp32.max())
    assert to_amp.max() <= 1.25 * amp_to_fp32.max() and to_amp.mean() <= 1.25 * amp_to_fp32.mean()
    assert to_fp32.max() > 1e-6        # the rounding points really are on


def test_round_lowp():
    import torch
    a = np.concatenate([np.linspace(-3, 3, 10001), [0.0, 1e-30, -1e-30, 65504.0, 1e5, 2.0 ** -24, 1.00390625, 1.01171875]])
    t = torch.from_numpy(a.astype(np.float32))
    assert np.array_equal(O.round_lowp(a, "bf16"), t.to(torch.bfloat16).double().numpy())
    assert np.array_equal(O.round_lowp(a, "f16"), t.to(torch.float16).double().numpy())


def batch_case(g):
    """(x [B, N, D], state, cfg) of a G20 golden: bag b = synth.bag(N, D, tag=f"batch/b{b}")."""
    from rrt_mil_amd import synth
    from conftest import STATE_KEYS
    cfg, N, B = g["cfg"], int(g["n"]), int(g["b"])
    st = synth.encoder_state(**{k: v for k, v in cfg.items() if k in STATE_KEYS})
    x = np.stack([synth.bag(N, cfg.get("mlp_dim", 512), tag=f"batch/b{b}") for b in range(B)])
    return x, st, cfg


@pytest.mark.parametrize("name", golden_names("G20"))
def test_eager_port_matches_reference_at_batch_gt_1(name):
    """(B, N, D) input, B = 2 / 3: the reference's CR-MSA runs its inner attention over the regions of ALL bags
    (rmsa.py:296-322) -- bag 0 of a batch differs from bag 0 alone by `coupling` (recorded from the reference)."""
    g = load_golden(name)
    x, st, cfg = batch_case(g)
    y = O.forward_eager(x, st, cfg).numpy()
    assert y.shape == x.shape
    assert np.abs(y[:, g["rows"]] - g["y_rows"]).max() <= 2e-6
    for b in range(x.shape[0]):
        s = np.array([y[b].astype(np.float64).sum(), np.abs(y[b].astype(np.float64)).sum()])
        assert np.allclose(s, g["y_sums"][b][:2], rtol=1e-6, atol=1e-2)
    y0 = O.forward_eager(x[0], st, cfg).numpy()
    assert abs(float(np.abs(y[0] - y0).max()) - float(g["coupling"])) <= 1e-5 and float(g["coupling"]) > 1e-4
