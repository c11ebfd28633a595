"""CPU: the oracle (oracle/rrt_oracle.py) against the golden vectors that
tools/make_golden.py captured from the real reference (modules/rrt.py::RRTEncoder,
torch CPU fp32, eval).  Pins the oracle before anything is compared with it."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, synth_case
from oracle import rrt_oracle as O

SMALL = [n for n in golden_names("G") if not n.startswith(("G0", "G8", "G11", "G13"))   # G8/G11/G13 = RRTMIL goldens
         and int(load_golden(n)["n"]) <= 4096]
LARGE = ["G3_d512_n9000", "G5_d512_n9000_c1_sc"]


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
