"""Closed-form synthetic parameters and bags (pure numpy, no RNG stream).

Every value is splitmix64(tensor-name hash, element index) mapped to a uniform,
so the golden-fixture generator (tools/make_golden.py, run next to the
reference), the tests and bench.py regenerate bit-identical arrays on any box
without sharing a torch RNG stream.

State-dict names / shapes are the reference's (modules/rrt.py:134-163,
modules/rmsa.py:57-89, :233-259); see SURVEY.md §3.3.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, lo=-1.0, hi=1.0, dtype=np.float32) -> np.ndarray:
    """Deterministic U[lo,hi) array keyed by ``name``."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF) << np.uint64(32)
    with np.errstate(over="ignore"):
        h = _splitmix64(key + np.arange(n, dtype=np.uint64))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).astype(dtype).reshape(shape)


def normal(name: str, shape, dtype=np.float32) -> np.ndarray:
    """Deterministic N(0,1) array (Box-Muller on two keyed uniforms)."""
    u1 = uniform(name + "/u1", shape, 0.0, 1.0, np.float64)
    u2 = uniform(name + "/u2", shape, 0.0, 1.0, np.float64)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(dtype)


def encoder_state_shapes(mlp_dim=512, n_layers=2, n_heads=8, epeg=True, epeg_k=15,
                         cr_msa=True, crmsa_k=3, crmsa_mlp=False, qkv_bias=True,
                         epeg_bias=True, ffn=False, mlp_ratio=4., pos='none', peg_k=7, peg_1d=False,
                         peg_bias=True, epeg_2d=False, epeg_type='attn', **_unused):
    """Ordered {state_dict key: shape} of the default-path RRTEncoder."""
    D = mlp_dim
    sh = {"norm.weight": (D,), "norm.bias": (D,)}

    def inner(prefix, with_pe):
        sh[prefix + "qkv.weight"] = (3 * D, D)
        if qkv_bias:
            sh[prefix + "qkv.bias"] = (3 * D,)
        sh[prefix + "proj.weight"] = (D, D)
        sh[prefix + "proj.bias"] = (D,)
        if with_pe:                    # modules/rmsa.py:74-87: channels = heads ('attn') or features ('value_*')
            ch = n_heads if epeg_type == 'attn' else D
            sh[prefix + "pe.weight"] = (ch, 1, epeg_k, epeg_k if epeg_2d else 1)
            if epeg_bias:
                sh[prefix + "pe.bias"] = (ch,)

    def mlp(prefix):                       # TransLayer(ffn=True): norm2 + Mlp, modules/rrt.py:25-41,48,106
        if ffn:
            hid = int(D * mlp_ratio)
            sh[prefix + "norm2.weight"] = (D,)
            sh[prefix + "norm2.bias"] = (D,)
            sh[prefix + "mlp.fc1.weight"] = (hid, D)
            sh[prefix + "mlp.fc1.bias"] = (hid,)
            sh[prefix + "mlp.fc2.weight"] = (D, hid)
            sh[prefix + "mlp.fc2.bias"] = (D,)

    for i in range(n_layers - 1):
        sh[f"layers.{i}.norm.weight"] = (D,)
        sh[f"layers.{i}.norm.bias"] = (D,)
        inner(f"layers.{i}.attn.attn.", epeg)
        mlp(f"layers.{i}.")
    if cr_msa:
        sh["cr_msa.norm.weight"] = (D,)
        sh["cr_msa.norm.bias"] = (D,)
        inner("cr_msa.attn.attn.", False)
        if crmsa_mlp:
            sh["cr_msa.attn.phi.0.weight"] = (D // 4, D)
            sh["cr_msa.attn.phi.2.weight"] = (crmsa_k, D // 4)
        else:
            sh["cr_msa.attn.phi"] = (D, crmsa_k)
        mlp("cr_msa.")
    if pos in ("peg", "ppeg"):            # modules/emb_position.py:24-82 under pos_embedding.*
        for name, kk in (("proj", peg_k), ("proj1", 5), ("proj2", 3)):
            if name != "proj" and pos == "peg":
                continue
            sh[f"pos_embedding.{name}.weight"] = (D, 1, kk, 1 if peg_1d else kk)
            if peg_bias:
                sh[f"pos_embedding.{name}.bias"] = (D,)
    return sh


def encoder_state(**cfg):
    """Closed-form fp32 parameters, scaled like the reference's default inits
    (Linear/Conv ~ U(+-1/sqrt(fan_in)); LayerNorm gamma ~ 1+-0.25, beta ~ +-0.1
    so that gamma/beta mistakes show up in parity tests)."""
    out = {}
    for k, shape in encoder_state_shapes(**cfg).items():
        if k.endswith(("norm.weight", "norm2.weight")):
            out[k] = 1.0 + uniform(k, shape, -0.25, 0.25)
        elif k.endswith(("norm.bias", "norm2.bias")):
            out[k] = uniform(k, shape, -0.1, 0.1)
        elif k.startswith("pos_embedding.") and k.endswith(".weight"):
            b = 1.0 / np.sqrt(shape[2] * shape[3])
            out[k] = uniform(k, shape, -b, b)
        elif k.endswith("pe.weight"):
            b = 1.0 / np.sqrt(shape[2] * shape[3])
            out[k] = uniform(k, shape, -b, b)
        elif k.endswith("pe.bias"):
            out[k] = uniform(k, shape, -0.25, 0.25)
        elif k.endswith(".bias"):
            out[k] = uniform(k, shape, -0.05, 0.05)
        elif k.endswith("phi"):
            b = 1.0 / np.sqrt(shape[0])
            out[k] = uniform(k, shape, -3 * b, 3 * b)
        else:  # Linear weights (out, in)
            b = 1.0 / np.sqrt(shape[-1])
            out[k] = uniform(k, shape, -b, b)
    return out


def bag(n_tokens: int, dim: int = 512, tag: str = "bag", nonneg: bool = False) -> np.ndarray:
    """Synthetic (N, D) patch-embedding bag: N(0,1), or relu(N(0,1)) like pooled
    ResNet-50 features."""
    x = normal(f"{tag}/{n_tokens}x{dim}", (n_tokens, dim))
    return np.maximum(x, 0.0) if nonneg else x


def mil_state(input_dim=1024, n_classes=2, da_bias=False, da_gated=False, da_act="relu", **enc_cfg):
    """Closed-form parameters of RRTMIL (modules/rrt.py:204-225): encoder state under
    ``online_encoder.`` plus patch_to_emb / pool_fn / predictor tensors (state_dict names of
    modules/datten.py: ``attention.{0,2}`` -- ``{0,1}`` without an activation -- or the gated
    ``attention_a.0 / attention_b.0 / attention_c``)."""
    out = {"patch_to_emb.0.weight": uniform("mil/fc.w", (512, input_dim), -1, 1) / np.sqrt(input_dim),
           "patch_to_emb.0.bias": uniform("mil/fc.b", (512,), -0.05, 0.05)}
    out = {k: v.astype(np.float32) for k, v in out.items()}
    for k, v in encoder_state(**enc_cfg).items():
        out["online_encoder." + k] = v
    D = enc_cfg.get("mlp_dim", 512)
    w0 = (uniform("mil/da0", (128, D), -1, 1) / np.sqrt(D)).astype(np.float32)
    w2 = (uniform("mil/da2", (1, 128), -1, 1) / np.sqrt(128) * 4).astype(np.float32)
    if da_gated:
        names = ("pool_fn.attention.attention_a.0", "pool_fn.attention.attention_c")
        out["pool_fn.attention.attention_b.0.weight"] = (uniform("mil/dab", (128, D), -1, 1) / np.sqrt(D)).astype(np.float32)
        if da_bias:
            out["pool_fn.attention.attention_b.0.bias"] = uniform("mil/dabb", (128,), -0.05, 0.05)
    else:
        last = 2 if da_act in ("relu", "gelu", "tanh") else 1
        names = ("pool_fn.attention.attention.0", f"pool_fn.attention.attention.{last}")
    out[names[0] + ".weight"], out[names[1] + ".weight"] = w0, w2
    if da_bias:
        out[names[0] + ".bias"] = uniform("mil/da0b", (128,), -0.05, 0.05)
        out[names[1] + ".bias"] = uniform("mil/da2b", (1,), -0.05, 0.05)
    out["predictor.weight"] = (uniform("mil/pred.w", (n_classes, D), -1, 1) / np.sqrt(D)).astype(np.float32)
    out["predictor.bias"] = uniform("mil/pred.b", (n_classes,), -0.05, 0.05)
    return out
