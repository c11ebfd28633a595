"""CPU: the C-ABI library loads and exports every symbol include/rrt_hip.h declares;
host-only entry points (geometry, workspace size, error strings) behave; the Python
boundary mirrors the reference's constructor / state_dict surface.  No GPU compute."""
import ctypes as C
import inspect
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
import rrt_mil_amd
from rrt_mil_amd import RRTEncoder, _lib, synth
from rrt_mil_amd.build import build


@pytest.fixture(scope="module")
def lib():
    build()
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "rrt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rrt_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rrt_abi_version() == _lib.ABI_VERSION


def test_c_geometry_matches_reference_padding(lib):
    tab = load_golden("G0_geometry")["table"]
    for L, rn, rs, mrn, mrr, H, rsz, add in tab:
        g = _lib.region_grid(int(L), int(rn), int(rs), int(mrn), float(mrr))
        assert (g.H, g.s, g.add) == (int(H), int(rsz), int(add)), (L, rn, rs, mrn, mrr)
        assert g.regions_side * g.s == g.H


def test_workspace_and_errors(lib):
    enc = RRTEncoder()
    n = C.c_size_t()
    assert lib.rrt_encoder_workspace_size(C.byref(enc._desc), 9000, C.byref(n)) == 0
    # u/o + qkv + x1 dominate: (9216*512*4 + 9000*512) floats
    assert n.value >= (9216 * 512 * 4 + 9000 * 512) * 4
    assert lib.rrt_encoder_workspace_size(C.byref(enc._desc), 0, C.byref(n)) == -1
    ok = RRTEncoder(mlp_dim=64, crmsa_mlp=True)              # (rounds 1-2: unsupported; round 3 runs any width)
    assert lib.rrt_encoder_workspace_size(C.byref(ok._desc), 100, C.byref(n)) == 0
    bad = RRTEncoder(mlp_dim=512, crmsa_k=9)                  # more representatives than RRT_MAX_CRMSA_K
    rc = lib.rrt_encoder_workspace_size(C.byref(bad._desc), 100, C.byref(n))
    assert rc == -2 and b"crmsa_k" in lib.rrt_strerror(rc)
    with pytest.raises(NotImplementedError):
        _lib.check(rc, "x")
    assert lib.rrt_region_grid(0, 8, 0, 0, 0.0, C.byref(_lib.Grid())) == -1
    # null device pointers are rejected before anything is launched
    assert lib.rrt_encoder_forward_f32(C.byref(enc._desc), None, None, None, 10, None, 0, None) == -1


def test_encoder_plan_flags(lib):
    """rrt_encoder_plan mirrors the forward's kernel choice for the R-MSA layers (host-only; without a GPU the launcher
    assumes 256 CUs): the fused kernel on regions of 49-208 tokens, the out-projection as a phase of its launch from 81
    tokens on when the bag has two rounds of (region, head) items, the 16-bit / split kernels in their modes."""
    enc = RRTEncoder()
    fl = C.c_int32(-1)

    def plan(n, compute=_lib.COMPUTE_F32, e=enc, solo=1):
        e._desc.compute = compute
        e._desc.solo = solo
        assert lib.rrt_encoder_plan(C.byref(e._desc), n, C.byref(fl)) == 0
        e._desc.compute = _lib.COMPUTE_F32
        return fl.value
    both = _lib.PLAN_FUSED | _lib.PLAN_FUSED_PROJ
    parts = both | _lib.PLAN_CRMSA_PARTS      # round 5: the merged launch of the last layer also leaves CR-MSA's row records
    assert plan(9000) == parts and plan(5000) == parts and plan(12000) == parts
    assert plan(9000, e=RRTEncoder(crmsa_mlp=True)) == both and plan(9000, e=RRTEncoder(ffn=True)) == both
    assert plan(9000, e=RRTEncoder(cr_msa=False)) == both and plan(9000, solo=0) == both and plan(9000, e=RRTEncoder(crmsa_k=5)) == both
    assert plan(3000) == _lib.PLAN_FUSED and plan(4096) == _lib.PLAN_FUSED     # regions of <= 64 tokens: two launches
    assert plan(15000) == 0 and plan(600) == 0                                  # regions outside the fused kernel's range
    assert plan(9000, _lib.COMPUTE_BF16) == _lib.PLAN_FUSED16
    assert plan(9000, _lib.COMPUTE_F32X3) == _lib.PLAN_FUSED_X3
    assert plan(12000, _lib.COMPUTE_F32X3) == parts                            # x3 covers P <= 144: exact kernels beyond
    assert plan(2000, e=RRTEncoder(region_num=4)) == _lib.PLAN_FUSED           # 16 regions: one round of items
    assert plan(30000, e=RRTEncoder(region_num=16)) == parts
    assert lib.rrt_encoder_plan(C.byref(enc._desc), 9000, None) == -1


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.RRTHipError):
        _lib.load(str(tmp_path / "nope.so"))


def test_constructor_signature_matches_reference():
    # modules/rrt.py:134 -- names, order and defaults
    ref = ("mlp_dim=512,pos_pos=0,pos='none',peg_k=7,attn='rmsa',region_num=8,drop_out=0.1,n_layers=2,"
           "n_heads=8,drop_path=0.,ffn=False,ffn_act='gelu',mlp_ratio=4.,trans_dim=64,epeg=True,epeg_k=15,"
           "region_size=0,min_region_num=0,min_region_ratio=0,qkv_bias=True,peg_bias=True,peg_1d=False,"
           "cr_msa=True,crmsa_k=3,all_shortcut=False,crmsa_mlp=False,crmsa_heads=8,need_init=False")
    want = [(kv.split("=")[0], eval(kv.split("=")[1])) for kv in ref.split(",")]
    sig = inspect.signature(RRTEncoder.__init__)
    got = [(n, p.default) for n, p in sig.parameters.items() if n not in ("self", "kwargs")]
    assert got == want
    assert any(p.kind is p.VAR_KEYWORD for p in sig.parameters.values())


@pytest.mark.parametrize("cfg", [dict(), dict(mlp_dim=64), dict(n_layers=3), dict(cr_msa=False),
                                 dict(epeg=False), dict(qkv_bias=False), dict(crmsa_k=5, epeg_k=21),
                                 dict(crmsa_mlp=True)])
def test_state_dict_surface(cfg):
    """Reference checkpoints load with strict=True: same keys, same shapes."""
    enc = RRTEncoder(**cfg)
    shapes = synth.encoder_state_shapes(**cfg)
    sd = enc.state_dict()
    assert list(sd.keys()) == list(shapes.keys()) or set(sd.keys()) == set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert enc.final_dim == cfg.get("mlp_dim", 512)


def test_default_param_count():
    enc = RRTEncoder()
    assert sum(p.numel() for p in enc.parameters()) == 2105984
    assert len(enc.state_dict()) == 17


@pytest.mark.parametrize("n_layers", [1, 2, 3])
def test_weight_fingerprint_covers_every_parameter(n_layers):
    """The version that guards the cached 16-bit weight images (rrt_encoder_weights.version, weights16_valid) moves when
    ANY parameter of the encoder changes -- CR-MSA's included, and with no R-MSA layer at all (round-3 advisor finding) --
    and the cached ctypes pointer struct is rebuilt with it (host logic only: no GPU needed)."""
    enc = RRTEncoder(mlp_dim=64, n_layers=n_layers)
    w0 = enc._weights()
    v0 = w0.version
    assert enc._weights() is w0 and enc._weights_version() == v0          # nothing changed: the cached struct
    seen = {v0}
    for name, p in enc.named_parameters():
        with torch.no_grad():
            p.add_(1.0)                                                     # optimizer-style in-place update
        v = enc._weights_version()
        assert v not in seen, name
        seen.add(v)
    w1 = enc._weights()
    assert w1 is not w0 and w1.version == max(seen)
    # load_state_dict copies in place (version counters move); a new Parameter object is seen through its data_ptr
    enc.load_state_dict(enc.state_dict())
    assert enc._weights_version() not in seen
    seen.add(enc._weights_version())
    enc.cr_msa.attn.attn.qkv.weight = torch.nn.Parameter(enc.cr_msa.attn.attn.qkv.weight.detach().clone())
    w2 = enc._weights()
    assert w2.version not in seen and w2.crmsa.qkv_w == enc.cr_msa.attn.attn.qkv.weight.data_ptr()
    # .float() / .to() may replace parameter objects: the pointer struct follows
    enc = enc.double().float()
    assert enc._weights().crmsa.qkv_w == enc.cr_msa.attn.attn.qkv.weight.data_ptr()


@pytest.mark.parametrize("kw", [dict(), dict(ffn=True), dict(ffn=True, crmsa_mlp=True), dict(n_layers=1), dict(n_layers=3),
                                dict(pos="ppeg", pos_pos=-1), dict(cr_msa=False), dict(epeg_2d=True), dict(epeg_type="value_bf")])
def test_workspace_queries_cover_what_the_forwards_check(kw):
    """The size a workspace query returns is the size the forward checks its workspace against -- for the single-bag entry
    point and the batch one (round 4: with ffn = 1 the batch carve handed the inner forward the REAL size of the single-bag
    carve, which is smaller than the queried one it checks).  No GPU here: a call that passes validation fails at its first
    launch with a HIP error (> 0); a sizing bug shows up as RRT_E_WORKSPACE (-3) before anything is launched."""
    lib = _lib.load()
    enc = RRTEncoder(mlp_dim=512, **kw)
    d, w = enc._desc, enc._weights()
    n = 700
    x, y = torch.zeros(2, n, 512), torch.zeros(2, n, 512)
    for mode in (_lib.COMPUTE_F32, _lib.COMPUTE_BF16, _lib.COMPUTE_F32X3):
        d.compute = mode
        need = C.c_size_t()
        assert lib.rrt_encoder_workspace_size(C.byref(d), n, C.byref(need)) == 0
        buf = (C.c_char * need.value)()
        rc = lib.rrt_encoder_forward_f32(C.byref(d), C.byref(w), x.data_ptr(), y.data_ptr(), n, C.addressof(buf), need.value, None)
        assert rc > 0 or rc == 0, (mode, rc)
        assert lib.rrt_encoder_forward_f32(C.byref(d), C.byref(w), x.data_ptr(), y.data_ptr(), n, C.addressof(buf),
                                           need.value - 1, None) == -3
        assert lib.rrt_encoder_batch_workspace_size(C.byref(d), 2, n, C.byref(need)) == 0
        buf = (C.c_char * need.value)()
        rc = lib.rrt_encoder_forward_batch_f32(C.byref(d), C.byref(w), x.data_ptr(), y.data_ptr(), 2, n, C.addressof(buf),
                                               need.value, None)
        assert rc > 0 or rc == 0, (mode, rc)
        assert lib.rrt_encoder_forward_batch_f32(C.byref(d), C.byref(w), x.data_ptr(), y.data_ptr(), 2, n, C.addressof(buf),
                                                 need.value - 1, None) == -3


def test_executor_argument_checks_need_no_gpu():
    """rrt_executor_create_on_streams refuses duplicate / missing stream handles and widths outside [1, 8] before touching
    the device; the batch entry points refuse empty batches (host-side validation only)."""
    lib = _lib.load()
    enc = RRTEncoder(mlp_dim=512)
    h = C.c_void_p()
    dup = (C.c_void_p * 2)(0x1000, 0x1000)
    assert lib.rrt_executor_create_on_streams(C.byref(enc._desc), 2, dup, 9000, C.byref(h)) == -1
    assert lib.rrt_executor_create_on_streams(C.byref(enc._desc), 2, None, 9000, C.byref(h)) == -1
    ok = (C.c_void_p * 2)(0x1000, 0x2000)
    assert lib.rrt_executor_create_on_streams(C.byref(enc._desc), 9, ok, 9000, C.byref(h)) == -2
    assert lib.rrt_executor_create_on_streams(C.byref(enc._desc), 2, ok, 0, C.byref(h)) == -1
    need = C.c_size_t()
    assert lib.rrt_encoder_batch_workspace_size(C.byref(enc._desc), 0, 100, C.byref(need)) == -1
    assert lib.rrt_encoder_forward_batch_f32(C.byref(enc._desc), C.byref(enc._weights()), 1, 2, 0, 100, None, 0, None) == -1


def test_modules_deepcopy_and_pickle_with_warm_caches():
    """copy.deepcopy (EMA / teacher copies), pickle and torch.save of the whole module keep working after the C-ABI caches
    are warm (round 4 cached the ctypes pointer struct in the module: ctypes objects with pointers cannot be pickled) -- the
    copy owns its own parameters and builds its own struct."""
    import copy
    import io
    import pickle
    from rrt_mil_amd import RRTMIL
    enc = RRTEncoder(mlp_dim=64)
    w = enc._weights()
    e2 = copy.deepcopy(enc)
    w2 = e2._weights()
    assert w2 is not w and w2.norm_w == e2.norm.weight.data_ptr() != w.norm_w
    e3 = pickle.loads(pickle.dumps(enc))
    assert torch.equal(e3.norm.weight, enc.norm.weight) and e3._weights().norm_w == e3.norm.weight.data_ptr()
    buf = io.BytesIO()
    torch.save(enc, buf)
    mil = RRTMIL(input_dim=64, n_classes=2)
    mil.online_encoder._weights()
    m2 = copy.deepcopy(mil)
    assert m2.online_encoder._weights().norm_w == m2.online_encoder.norm.weight.data_ptr()
    sd = {k: v.clone() for k, v in mil.state_dict().items()}
    m2.load_state_dict(sd, strict=True)


def test_need_init_matches_reference_rule():
    enc = RRTEncoder(mlp_dim=64, need_init=True)
    assert float(enc.layers[0].attn.attn.qkv.bias.abs().max()) == 0.0
    assert float(enc.norm.weight.min()) == 1.0


def test_cpu_tensor_and_ablations_raise():
    enc = RRTEncoder(mlp_dim=64).eval()
    with pytest.raises(_lib.RRTHipError):
        enc(torch.zeros(1, 10, 64))
    for kw in (dict(pos='sincos'), dict(attn='ntrans'), dict(region_attn='ntrans'), dict(epeg_type='value_xx')):
        with pytest.raises(NotImplementedError):
            RRTEncoder(mlp_dim=64, **kw)


def test_epeg_ablation_state_dict_surface():
    """epeg_2d / epeg_type = value_bf / value_af (modules/rmsa.py:74-87): the conv's channels and kernel shape."""
    for kw, shape, desc in ((dict(epeg_2d=True), (8, 1, 15, 15), (1, _lib.EPEG_ATTN)),
                            (dict(epeg_type='value_bf'), (64, 1, 15, 1), (0, _lib.EPEG_VALUE_BF)),
                            (dict(epeg_type='value_af', epeg_2d=True, epeg_k=5), (64, 1, 5, 5), (1, _lib.EPEG_VALUE_AF))):
        enc = RRTEncoder(mlp_dim=64, **kw)
        assert enc.layers[0].attn.attn.pe.weight.shape == shape
        assert (enc._desc.epeg_2d, enc._desc.epeg_type) == desc
        st = synth.encoder_state(mlp_dim=64, **kw)
        enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
        assert enc.cr_msa.attn.attn.pe is None


def test_pos_state_dict_surface():
    """pos='ppeg' / 'peg' (modules/emb_position.py:24-82): pos_embedding.proj{,1,2} depth-wise convs."""
    for cfg in (dict(mlp_dim=64, pos='ppeg', pos_pos=-1), dict(mlp_dim=64, pos='peg', pos_pos=-1, peg_k=5, peg_1d=True, peg_bias=False)):
        enc = RRTEncoder(**cfg)
        st = synth.encoder_state(**cfg)
        enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
        assert enc._desc.pos == (_lib.POS_PPEG if cfg['pos'] == 'ppeg' else _lib.POS_PEG)
    assert RRTEncoder(mlp_dim=64, pos='peg', peg_k=5, peg_1d=True).pos_embedding.proj.weight.shape == (64, 1, 5, 1)
    # pos_pos = 0 applies the stage before layer index 1 (rrt.py:185): with the default n_layers = 2 there is no such
    # layer -- the reference's default `--pos ppeg` run never executes it; the parameters exist, the stage does not
    assert RRTEncoder(mlp_dim=64, pos='ppeg')._desc.pos == _lib.POS_NONE
    assert RRTEncoder(mlp_dim=64, pos='ppeg', n_layers=3)._desc.pos == _lib.POS_PPEG
    assert len(RRTEncoder(mlp_dim=64, pos='ppeg').pos_embedding.state_dict()) == 6


def test_ffn_state_dict_surface():
    """ffn=True (modules/rrt.py:25-41,48,105-106): norm2 + mlp.fc1/fc2 under every TransLayer, CR-MSA's too."""
    cfg = dict(mlp_dim=64, ffn=True, mlp_ratio=2.0, n_layers=3)
    enc = RRTEncoder(**cfg)
    st = synth.encoder_state(**cfg)
    enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    assert enc.cr_msa.mlp.fc1.weight.shape == (128, 64) and enc.layers[1].mlp.fc2.weight.shape == (64, 128)
    assert enc._desc.ffn == 1 and enc._desc.ffn_hidden == 128 and enc._desc.ffn_act == _lib.ACT_GELU
    assert RRTEncoder(mlp_dim=64, ffn=True, ffn_act='relu')._desc.ffn_act == _lib.ACT_RELU


def test_rrtmil_state_dict_surface():
    """RRTMIL (modules/rrt.py:204-225) keys/shapes: reference checkpoints load strictly."""
    from rrt_mil_amd import RRTMIL
    m = RRTMIL(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1, all_shortcut=True)
    st = synth.mil_state(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 2696450
    # RRTMIL.apply(initialize_weights) ran: zero biases, unit LayerNorm
    m2 = RRTMIL()
    assert float(m2.predictor.bias.detach().abs().max()) == 0.0
    assert float(m2.online_encoder.norm.weight.detach().min()) == 1.0


def test_library_has_no_packed_fp32(lib):
    """The linked library's gfx950 code (every bundle, every kernel) contains no v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
    outside the kernels rrt-mil_amd/build.py::PACKED_FP32_OK names (today: none).  Compiler-formed packed fp32 next to another
    bag's bf16-MFMA waves is what mis-summed CR-MSA's representatives in round 5; the build applies the
    same check to every object it compiles -- this one reads the .so that actually ships (no GPU needed: llvm-objdump)."""
    from rrt_mil_amd import build as B
    if not os.path.exists(B.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    seen = []
    census = B.packed_fp32_census(B.LIB, seen)
    assert len(seen) > 300, f"only {len(seen)} device functions found: the disassembly did not see the kernels"
    allowed = tuple(p for pats in B.PACKED_FP32_OK.values() for p in pats)
    bad = {k: n for k, n in census.items() if not any(p in k for p in allowed)}
    assert not bad, bad
