"""CPU oracle for the RRTEncoder forward path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
import this module; the product path (rrt-mil_amd/) never does.

Two independent restatements of the reference algorithm, written from the
specification in SURVEY.md §3.3 (not copied from the reference sources):

* ``forward_f64``   -- numpy float64, *algebraic* form (GEMM-shaped CR-MSA
                       combine/dispatch, EPEG as an explicit stencil).  This is
                       the ground truth the HIP kernels are compared with.
* ``forward_eager`` -- torch fp32 on the CPU with the *same aten op sequence* as
                       the reference (depth-wise conv2d EPEG, materialised 4-D
                       einsum combine/dispatch, .contiguous() region copies).
                       This is what bench.py times as the CPU baseline ("port").

Reduced precision (the reference's --amp path, main.py:101-102,439; BASELINE configs[2..4]):

* ``forward_eager(..., autocast=torch.bfloat16)`` runs the same aten op sequence under
  ``torch.autocast('cpu', dtype)`` -- torch's own autocast policy (Linear / matmul / conv2d / einsum in
  the low-precision dtype, softmax on what they hand over, LayerNorm and the residual stream in
  fp32).  Pinned by the G16 fixtures (the real reference under the same context).
* ``forward_f64(..., lowp=...)`` states WHERE the HIP path's reduced modes round, in float64 with
  explicit round-to-nearest-even to bf16 / fp16 of exactly the tensors the kernels store or feed to
  the matrix cores in 16 bits (the operands of every GEMM; with ``attn=True`` also Q~, K, P and V of
  the region attention).  Everything else -- accumulation, LayerNorm, softmax, residuals, CR-MSA's
  logits / combine / dispatch -- is exact.  The GPU tests hold the kernels to this restatement
  tightly; its distance to the autocast fixtures is what the autocast-class claim rests on.

Parity pin: both are checked against tests/golden/*.npz, which were produced
by importing the real reference (/root/reference/modules/rrt.py::RRTEncoder,
torch 2.10.0 CPU) with tools/make_golden.py in the build container.

Reference lines followed (file:line in /root/reference):
  RRTEncoder.forward            modules/rrt.py:165-202
  TransLayer.forward_trans      modules/rrt.py:117-131
  padding / partition / reverse modules/rmsa.py:175-202, :28-54
  InnerAttention.forward        modules/rmsa.py:91-134   (EPEG :106-108)
  CrossRegionAttntion.forward   modules/rmsa.py:290-337
"""
import math
import numpy as np


# --------------------------------------------------------------------------- geometry
def grid(L, region_num=8, region_size=0, min_region_num=0, min_region_ratio=0.0):
    """(H, s, add_length) -- modules/rmsa.py:175-202 (same body at :261-288)."""
    H = int(np.ceil(np.sqrt(L)))
    if region_size and region_size > 0:
        H += (-H) % region_size
        s = region_size
    else:
        H += (-H) % region_num
        s = H // region_num
    add = H * H - L
    if add > L / (min_region_ratio + 1e-8) or L < min_region_num:
        H = int(np.ceil(np.sqrt(L)))
        H += (-H) % 2
        add = H * H - L
        s = H
    return H, s, add


def partition_index(H, s):
    """perm[slot] = padded-grid token index held by region-major slot
    (modules/rmsa.py:28-39: view(B,H/s,s,W/s,s,C).permute(0,1,3,2,4,5))."""
    t = np.arange(H * H).reshape(H // s, s, H // s, s)
    return t.transpose(0, 2, 1, 3).reshape(-1)


_DEFAULTS = dict(mlp_dim=512, region_num=8, n_layers=2, n_heads=8, epeg=True, epeg_k=15,
                 region_size=0, min_region_num=0, min_region_ratio=0.0, qkv_bias=True,
                 cr_msa=True, crmsa_k=3, all_shortcut=False, crmsa_mlp=False, crmsa_heads=8,
                 epeg_bias=True, ffn=False, ffn_act="gelu", mlp_ratio=4.0, pos="none", pos_pos=0, peg_k=7,
                 peg_1d=False, peg_bias=True, epeg_2d=False, epeg_type="attn")


def _cfg(cfg):
    c = dict(_DEFAULTS)
    c.update(cfg or {})
    return c


# --------------------------------------------------------------------------- float64 truth
def _ln64(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)      # biased, like nn.LayerNorm
    return (x - mu) / np.sqrt(var + eps) * g + b


def _softmax64(a, axis):
    a = a - a.max(axis=axis, keepdims=True)
    e = np.exp(a)
    return e / e.sum(axis=axis, keepdims=True)


def round_lowp(a, dtype):
    """float64 array -> values representable in ``dtype`` ('bf16' | 'f16'), round-to-nearest-even
    (through float32 first, like a kernel that holds the value in a 32-bit register), as float64."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    if dtype in ("f16", "fp16", "float16"):
        with np.errstate(over="ignore"):
            return a32.astype(np.float16).astype(np.float64)
    if dtype not in ("bf16", "bfloat16"):
        raise ValueError(dtype)
    u = a32.view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    out = (r & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32).astype(np.float64)
    nan = np.isnan(a32)
    if nan.any():
        out[nan] = np.nan
    return out


class LowP:
    """Rounding points of the HIP path's reduced-precision modes (see the module docstring)."""

    def __init__(self, dtype="bf16", attn=True, stencil16=None):
        # stencil16: None = as the library chooses (the pair kernel where it applies), True / False = force one form
        self.dtype, self.attn, self.stencil16 = dtype, attn, stencil16

    def r(self, a):
        return round_lowp(a, self.dtype)


def split_hi_lo(a):
    """fp32 value -> (hi, lo) bf16 pair of RRT_COMPUTE_F32X3: hi = bf16(x), lo = bf16(x - hi) (as float64 arrays)."""
    a32 = np.ascontiguousarray(a, dtype=np.float32).astype(np.float64)
    hi = round_lowp(a32, "bf16")
    lo = round_lowp((a32 - hi).astype(np.float32), "bf16")
    return hi, lo


def split_matmul_t(A, B):
    """A . B^T as the F32X3 kernels form it: three products of the (hi, lo) parts, the lo.lo term dropped."""
    ah, al = split_hi_lo(A)
    bh, bl = split_hi_lo(B)
    return ah @ bh.T + ah @ bl.T + al @ bh.T


class SplitX3:
    """forward_f64(lowp=SplitX3()): the rounding points of RRT_COMPUTE_F32X3 -- the qkv / proj GEMMs and the two
    attention products of the R-MSA layers on (hi, lo) bf16 operand pairs; everything else exact."""
    dtype, attn = "split", False


def _conv_dw64(img, w, b, two_d):
    """depth-wise Conv2d, zero padded: img [B, C, H, W], w [C, 1, k, kw] (kw = k or 1), b [C] or None"""
    k, kw = w.shape[2], w.shape[3]
    ph, pw = k // 2, (kw // 2 if two_d else 0)
    Hh, Ww = img.shape[2], img.shape[3]
    pad = np.pad(img, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    out = np.zeros_like(img)
    for a in range(k):
        for bb in range(kw):
            out += w[None, :, 0, a, bb, None, None] * pad[:, :, a:a + Hh, bb:bb + Ww]
    if b is not None:
        out += b[None, :, None, None]
    return out


def _inner_attention64(x, st, pfx, heads, epeg_k, taps=None, lowp=None, attn_lowp=None, epeg_2d=False,
                       epeg_type="attn"):
    """x: [B_, P, D] -> [B_, P, D]   (modules/rmsa.py:91-134; the default 1-D 'attn' EPEG and, in exact arithmetic
    only, the epeg_2d / epeg_type = value_bf / value_af ablations of :76-85,:106-129)."""
    B_, P, D = x.shape
    variant = (pfx + "pe.weight" in st) and (epeg_2d or epeg_type != "attn")
    if variant:
        assert lowp is None or attn_lowp is None
        hd = D // heads
        W = st[pfx + "qkv.weight"].astype(np.float64)
        xx, WW = (x, W) if lowp is None else (lowp.r(x), lowp.r(W))
        qkv = xx @ WW.T
        if pfx + "qkv.bias" in st:
            qkv = qkv + st[pfx + "qkv.bias"].astype(np.float64)
        qkv = qkv.reshape(B_, P, 3, heads, hd).transpose(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]                    # [B_,h,P,hd]
        pw = st[pfx + "pe.weight"].astype(np.float64)
        pb = st[pfx + "pe.bias"].astype(np.float64) if pfx + "pe.bias" in st else None
        S = q @ k.transpose(0, 1, 3, 2)
        if epeg_type == "attn":                                             # rmsa.py:106-108, k x k kernel
            S = S + _conv_dw64(S, pw, pb, True)
        A = _softmax64(S, -1)
        s_ = int(np.ceil(np.sqrt(P)))
        if epeg_type != "attn":                                             # rmsa.py:114-118 / :124-129
            img = v.transpose(0, 3, 1, 2).reshape(B_, D, s_, s_)            # channel c = d * h + head
            pe = _conv_dw64(img, pw, pb, epeg_2d)
        if epeg_type == "value_bf":
            v = v + pe.reshape(B_, heads, hd, P).transpose(0, 1, 3, 2)
        O = (A @ v).transpose(0, 2, 1, 3).reshape(B_, P, D)
        if epeg_type == "value_af":
            O = O + pe.reshape(B_, D, P).transpose(0, 2, 1)
        Wp = st[pfx + "proj.weight"].astype(np.float64)
        if lowp is not None:
            O, Wp = lowp.r(O), lowp.r(Wp)
        return O @ Wp.T + st[pfx + "proj.bias"].astype(np.float64)
    hd = D // heads
    W = st[pfx + "qkv.weight"].astype(np.float64)
    x3 = isinstance(lowp, SplitX3)
    if x3:
        qkv = split_matmul_t(x.reshape(-1, D), W).reshape(B_, P, 3 * D)
        lowp = None
    else:
        if lowp is not None:                                                # GEMM operands in 16 bits
            x, W = lowp.r(x), lowp.r(W)
        qkv = x @ W.T
    if pfx + "qkv.bias" in st:
        qkv = qkv + st[pfx + "qkv.bias"].astype(np.float64)
    qkv = qkv.reshape(B_, P, 3, heads, hd).transpose(2, 0, 3, 1, 4)       # [3,B_,h,P,hd]
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    if x3:
        # RRT_COMPUTE_F32X3 attention: Q~ log2(e), K, V and exp2(S - max) as (hi, lo) pairs, three products each
        qt = q
        if pfx + "pe.weight" in st:
            w = st[pfx + "pe.weight"].astype(np.float64).reshape(heads, epeg_k)
            half = epeg_k // 2
            qp = np.pad(q, ((0, 0), (0, 0), (half, half), (0, 0)))
            qt = q.copy()
            for t in range(epeg_k):
                qt += w[None, :, t, None, None] * qp[:, :, t:t + P, :]
        qh, ql = split_hi_lo(qt * 1.4426950408889634)
        kh, kl = split_hi_lo(k)
        vh, vl = split_hi_lo(v)
        kT = lambda a: a.transpose(0, 1, 3, 2)
        S = qh @ kT(kh) + qh @ kT(kl) + ql @ kT(kh)
        e = np.exp2(S - S.max(-1, keepdims=True))
        eh, el = split_hi_lo(e)
        O = (eh @ vh + eh @ vl + el @ vh) / e.sum(-1, keepdims=True)
        O = O.transpose(0, 2, 1, 3).reshape(B_, P, D)
        Wp = st[pfx + "proj.weight"].astype(np.float64)
        if taps is not None:
            taps[pfx + "proj_in"] = O
        return split_matmul_t(O.reshape(-1, D), Wp).reshape(B_, P, D) + st[pfx + "proj.bias"].astype(np.float64)
    if attn_lowp is not None:
        # the kernels' form (DESIGN.md identities 1, 2): Q~ = Q + T Q, then Q~, K, V and the unnormalised probabilities
        # exp(S - max) go to the matrix cores in 16 bits; the row sum is taken over the unrounded probabilities (fp32
        # registers) and divides the fp32 accumulator at the end.  Two forms of the stencil:
        #   rmsa_fused16 (one region per block): built in fp32 from the unrounded Q, rounded once;
        #   rmsa_pair16  (>= 8 regions of 17..176 tokens): on the matrix cores -- Q log2(e) is rounded to 16 bits FIRST, the
        #   taps (identity included) enter as hi + lo 16-bit pairs, fp32 accumulation, then Q~ is rounded.
        pair = attn_lowp.stencil16 if attn_lowp.stencil16 is not None else (B_ >= 8 and 16 < P <= 176)
        L2E = 1.4426950408889634
        if pair:
            q16 = attn_lowp.r(q * L2E)
            qt = q16
            if pfx + "pe.weight" in st:
                w = st[pfx + "pe.weight"].astype(np.float64).reshape(heads, epeg_k).copy()
                half = epeg_k // 2
                w[:, half] += 1.0                                           # the identity tap rides in the band
                w = np.float32(w).astype(np.float64)                        # (the table holds fp32 values)
                whi = attn_lowp.r(w)
                w16 = whi + attn_lowp.r(w - whi)
                qp = np.pad(q16, ((0, 0), (0, 0), (half, half), (0, 0)))
                qt = np.zeros_like(q16)
                for t in range(epeg_k):
                    qt += w16[None, :, t, None, None] * qp[:, :, t:t + P, :]
            S = attn_lowp.r(qt) @ attn_lowp.r(k).transpose(0, 1, 3, 2)
        else:
            qt = q
            if pfx + "pe.weight" in st:
                w = st[pfx + "pe.weight"].astype(np.float64).reshape(heads, epeg_k)
                half = epeg_k // 2
                qp = np.pad(q, ((0, 0), (0, 0), (half, half), (0, 0)))
                qt = q.copy()
                for t in range(epeg_k):
                    qt += w[None, :, t, None, None] * qp[:, :, t:t + P, :]
            # (the kernels fold log2(e) into the stencil taps, round Q~ log2(e) and exponentiate with exp2)
            S = attn_lowp.r(qt * L2E) @ attn_lowp.r(k).transpose(0, 1, 3, 2)
        e = np.exp2(S - S.max(-1, keepdims=True))
        O = (attn_lowp.r(e) @ attn_lowp.r(v)) / e.sum(-1, keepdims=True)
        O = O.transpose(0, 2, 1, 3).reshape(B_, P, D)
        Wp = st[pfx + "proj.weight"].astype(np.float64)
        if lowp is not None:
            O, Wp = lowp.r(O), lowp.r(Wp)
        return O @ Wp.T + st[pfx + "proj.bias"].astype(np.float64)
    S = q @ k.transpose(0, 1, 3, 2)                                         # [B_,h,P,P]
    if pfx + "pe.weight" in st:
        w = st[pfx + "pe.weight"].astype(np.float64).reshape(heads, epeg_k)
        half = epeg_k // 2
        Sp = np.pad(S, ((0, 0), (0, 0), (half, half), (0, 0)))             # zero rows = region edge
        E = np.zeros_like(S)
        for t in range(epeg_k):                                             # stencil along the QUERY axis
            E += w[None, :, t, None, None] * Sp[:, :, t:t + P, :]
        if pfx + "pe.bias" in st:
            E += st[pfx + "pe.bias"].astype(np.float64)[None, :, None, None]
        S = S + E
    A = _softmax64(S, -1)
    O = (A @ v).transpose(0, 2, 1, 3).reshape(B_, P, D)
    Wp = st[pfx + "proj.weight"].astype(np.float64)
    if taps is not None:
        taps[pfx + "proj_in"] = O
    if x3:
        return split_matmul_t(O.reshape(-1, D), Wp).reshape(B_, P, D) + st[pfx + "proj.bias"].astype(np.float64)
    if lowp is not None:
        O, Wp = lowp.r(O), lowp.r(Wp)
    out = O @ Wp.T + st[pfx + "proj.bias"].astype(np.float64)
    if taps is not None:
        taps[pfx + "scores_in"] = S
    return out


def _ffn64(x, st, pfx, act, lowp=None):
    """TransLayer's optional FFN (ffn=True): x + fc2(act(fc1(LN2(x)))), modules/rrt.py:25-41,127-129."""
    g = lambda k: st[pfx + k].astype(np.float64)
    r = (lambda a: a) if lowp is None else lowp.r
    u = _ln64(x, g("norm2.weight"), g("norm2.bias"))
    h = _act64(r(u) @ r(g("mlp.fc1.weight")).T + g("mlp.fc1.bias"), "gelu" if act == "gelu" else "relu")
    return x + r(h) @ r(g("mlp.fc2.weight")).T + g("mlp.fc2.bias")


def _pos64(x, st, kind, conv_1d):
    """PEG / PPEG (modules/emb_position.py:24-82): wrap-pad the tokens to an H x H grid (PPEG: zero-pad to 7 x 7),
    depth-wise convs with zero borders + identity, keep the first N tokens."""
    N, D = x.shape
    H = int(np.ceil(np.sqrt(N)))
    add = H * H - N
    xh = np.concatenate([x, x[:add]], 0)
    if kind == "ppeg" and H < 7:
        xh = np.concatenate([xh, np.zeros((49 - H * H, D))], 0)
        H = 7
    grid = xh.reshape(H, H, D)
    out = grid.copy()
    for name in ("proj", "proj1", "proj2") if kind == "ppeg" else ("proj",):
        w = st[f"pos_embedding.{name}.weight"].astype(np.float64)           # [D, 1, kh, kw]
        kh, kw = w.shape[2], w.shape[3]
        gp = np.pad(grid, ((kh // 2, kh // 2), (kw // 2, kw // 2), (0, 0)))
        for di in range(kh):
            for dj in range(kw):
                out += gp[di:di + H, dj:dj + H, :] * w[:, 0, di, dj]
        if f"pos_embedding.{name}.bias" in st:
            out += st[f"pos_embedding.{name}.bias"].astype(np.float64)
    return out.reshape(H * H, D)[:N]


def forward_f64(x, state, cfg=None, taps=None, lowp=None):
    """x: (N, D) array -> (N, D) float64.  ``state``: {reference state_dict key: array}.
    lowp: None (exact) or a LowP -- the rounding points of the HIP path's bf16 / fp16 modes."""
    c = _cfg(cfg)
    attn_lowp = lowp if (lowp is not None and lowp.attn) else None
    lowp_small = None if isinstance(lowp, SplitX3) else lowp       # F32X3 touches the R-MSA layers' two projections only
    st = state
    x = np.asarray(x, dtype=np.float64)
    N, D = x.shape
    x0 = x
    use_pos = c["pos"] in ("peg", "ppeg")
    if use_pos and c["pos_pos"] == -1:                                      # modules/rrt.py:181-182
        x = _pos64(x, st, c["pos"], c["peg_1d"])
    for li in range(c["n_layers"] - 1):                                     # R-MSA TransLayers
        if use_pos and li == 1 and c["pos_pos"] == 0:                       # modules/rrt.py:185-187
            x = _pos64(x, st, c["pos"], c["peg_1d"])
        p = f"layers.{li}."
        u = _ln64(x, st[p + "norm.weight"].astype(np.float64), st[p + "norm.bias"].astype(np.float64))
        H, s, add = grid(N, c["region_num"], c["region_size"], c["min_region_num"], c["min_region_ratio"])
        up = np.concatenate([u, np.zeros((add, D))], 0)                     # pad rows are exact zeros (T5)
        perm = partition_index(H, s)
        U = up[perm].reshape(-1, s * s, D)
        Z = _inner_attention64(U, st, p + "attn.attn.", c["n_heads"], c["epeg_k"], taps, lowp, attn_lowp,
                               c["epeg_2d"], c["epeg_type"])
        z = np.empty((H * H, D))
        z[perm] = Z.reshape(-1, D)
        x = x + z[:N]
        if c["ffn"]:
            x = _ffn64(x, st, p, c["ffn_act"], lowp_small)
        if taps is not None:
            taps[p + "out"] = x
    if c["cr_msa"]:
        p = "cr_msa."
        k = c["crmsa_k"]
        v = _ln64(x, st[p + "norm.weight"].astype(np.float64), st[p + "norm.bias"].astype(np.float64))
        # CR-MSA never receives region_num/region_size from RRTEncoder (trap T3): always 8
        H, s, add = grid(N, 8, 0, 0, 0.0)
        vp = np.concatenate([v, np.zeros((add, D))], 0)
        perm = partition_index(H, s)
        V = vp[perm].reshape(-1, s * s, D)                                   # [R,P,D]
        if c["crmsa_mlp"]:
            W1 = st[p + "attn.phi.0.weight"].astype(np.float64)
            h1 = np.tanh(V @ W1.T) if lowp_small is None else np.tanh(lowp_small.r(V) @ lowp_small.r(W1).T)
            Lg = (h1 @ st[p + "attn.phi.2.weight"].astype(np.float64).T).transpose(0, 2, 1)
        else:
            Lg = (V @ st[p + "attn.phi"].astype(np.float64)).transpose(0, 2, 1)   # [R,k,P]
        Cw = _softmax64(Lg, -1)                                              # combine: over P
        Dw = _softmax64(Lg, 1)                                               # dispatch: over k
        mn, mx = Lg.min(-1, keepdims=True), Lg.max(-1, keepdims=True)
        Mm = (Lg - mn) / (mx - mn + 1e-8)
        rep = (Cw @ V).transpose(1, 0, 2)                                    # [k,R,D]
        # bf16 / fp16 modes, head dim 64: the representatives go through the SAME fused 16-bit kernel as the R-MSA regions
        # (k "regions" of 64 tokens, no EPEG) -- 16-bit operands of both projections and of the two attention products.
        # Other head dims (crmsa_heads = 1): fp32 attention, only the GEMM operands round.
        inner16 = lowp_small if (lowp_small is not None and D % 64 == 0 and D == 64 * c["crmsa_heads"]) else None
        rep2 = _inner_attention64(rep, st, p + "attn.attn.", c["crmsa_heads"], 0, taps, lowp_small, inner16)
        out = np.einsum("rnp,nrd->rpd", Mm * Dw, rep2)                       # [R,P,D]
        z = np.empty((H * H, D))
        z[perm] = out.reshape(-1, D)
        x = x + z[:N]
        if c["ffn"]:
            x = _ffn64(x, st, p, c["ffn_act"], lowp_small)
        if taps is not None:
            taps["cr_msa.rep"] = rep
            taps["cr_msa.out"] = x
    if c["all_shortcut"]:
        x = x + x0
    return _ln64(x, st["norm.weight"].astype(np.float64), st["norm.bias"].astype(np.float64))


# --------------------------------------------------------------------------- eager-equivalent torch/CPU port
def forward_eager(x, state, cfg=None, grad=False, drop=None, autocast=None, branch=None):
    """torch fp32 CPU port issuing the reference's aten op sequence; x: (N, D)
    torch tensor or array -> torch (N, D); or (B, N, D) -> (B, N, D): at B > 1 the reference couples the bags
    inside CR-MSA (its inner attention runs over the regions of all bags, rmsa.py:296-322).  Used as the timed CPU baseline.
    grad=True: float64 leaves with requires_grad (x and every parameter) and a recorded graph -- torch autograd
    then yields the reference's gradients (the oracle of the backward, row f2); returns (y, x_leaf, params).
    drop = (p, {layer: keep mask [rows, D]}): train-mode proj_drop (rmsa.py:132) with GIVEN masks (layer index, or
    "cr_msa"), i.e. nn.Dropout's arithmetic x * keep / (1 - p) without its random number generator.
    branch = {(layer index | "cr_msa", "attn" | "ffn"): multiplier}: stochastic depth with GIVEN draws -- timm's
    DropPath at batch 1 multiplies a whole residual branch by 0 or 1 / keep_prob (rrt.py:102,125,129).
    autocast = torch.bfloat16 / torch.float16: the whole sequence under torch.autocast('cpu', dtype), i.e. the
    reference's --amp forward (main.py:101-102,439) with torch's CPU autocast policy."""
    import contextlib
    import torch
    import torch.nn.functional as F
    c = _cfg(cfg)
    st = {k_: (v_ if isinstance(v_, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v_)))
          for k_, v_ in state.items()}
    x = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    if grad:
        st = {k_: v_.double().requires_grad_(True) for k_, v_ in st.items()}
        x = x.double().requires_grad_(True)
        x_leaf = x
    batched = x.dim() == 3                                                   # (B,N,D): returned with its batch axis
    if not batched:
        x = x.unsqueeze(0)                                                   # (1,N,D), modules/rrt.py:166-175
    B, N, D = x.shape
    x0 = x

    def part(t, H, s):                                                       # modules/rmsa.py:28-39
        t = t.view(B, H // s, s, H // s, s, D)
        return t.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, s * s, D)

    def unpart(t, H, s):                                                     # modules/rmsa.py:41-54
        t = t.view(B, H // s, H // s, s, s, D)
        return t.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H * H, D)

    def proj_drop(z, key):                                                   # modules/rmsa.py:132
        if drop is None:
            return z
        keep = torch.from_numpy(np.ascontiguousarray(drop[1][key])).to(z.dtype).reshape(z.shape)
        return z * keep / (1.0 - drop[0])

    def inner(t, pfx, heads, epeg_k, key=None):                              # modules/rmsa.py:91-134
        B_, P, _ = t.shape
        hd = D // heads
        qkv = F.linear(t, st[pfx + "qkv.weight"], st.get(pfx + "qkv.bias"))
        qkv = qkv.reshape(B_, P, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        q = q * (hd ** -0.5)
        attn = q @ k.transpose(-2, -1)
        has_pe = pfx + "pe.weight" in st
        et = c["epeg_type"] if has_pe else None
        pad = (epeg_k // 2, epeg_k // 2) if c["epeg_2d"] else (epeg_k // 2, 0)
        if et == "attn":                                                     # rmsa.py:106-108
            pe = F.conv2d(attn, st[pfx + "pe.weight"], st.get(pfx + "pe.bias"), padding=pad, groups=heads)
            attn = attn + pe
        attn = attn.softmax(dim=-1)
        s_ = int(np.ceil(np.sqrt(P)))
        if et == "value_bf":                                                 # rmsa.py:114-118
            pe = F.conv2d(v.permute(0, 3, 1, 2).reshape(B_, D, s_, s_), st[pfx + "pe.weight"], st.get(pfx + "pe.bias"),
                          padding=pad, groups=D)
            v = v + pe.reshape(B_, heads, hd, P).permute(0, 1, 3, 2)
        o = (attn @ v).transpose(1, 2).reshape(B_, P, D)
        if et == "value_af":                                                 # rmsa.py:124-129
            pe = F.conv2d(v.permute(0, 3, 1, 2).reshape(B_, D, s_, s_), st[pfx + "pe.weight"], st.get(pfx + "pe.bias"),
                          padding=pad, groups=D)
            o = o + pe.reshape(B_, D, P).transpose(-1, -2)
        return proj_drop(F.linear(o, st[pfx + "proj.weight"], st[pfx + "proj.bias"]), key)

    def bm(key, kind):
        return 1.0 if branch is None else float(branch.get((key, kind), 1.0))

    def ffn(t, pfx, key=None):                                               # modules/rrt.py:25-41,127-129
        u = F.layer_norm(t, (D,), st[pfx + "norm2.weight"], st[pfx + "norm2.bias"], 1e-5)
        h = F.linear(u, st[pfx + "mlp.fc1.weight"], st[pfx + "mlp.fc1.bias"])
        h = F.gelu(h) if c["ffn_act"] == "gelu" else F.relu(h)
        h = proj_drop(h, ("ffn1", key))                                      # Mlp.drop after the activation ...
        return t + bm(key, "ffn") * proj_drop(F.linear(h, st[pfx + "mlp.fc2.weight"], st[pfx + "mlp.fc2.bias"]), ("ffn2", key))  # ... and after fc2

    def pos_embed(t):                                                        # modules/emb_position.py:24-82
        Hh = int(np.ceil(np.sqrt(N)))
        add = Hh * Hh - N
        t = torch.cat([t, t[:, :add, :]], dim=1)
        if c["pos"] == "ppeg" and Hh < 7:
            t = torch.cat([t, torch.zeros((B, 49 - Hh * Hh, D), dtype=t.dtype)], dim=1)
            add += 49 - Hh * Hh
            Hh = 7
        feat = t.transpose(1, 2).reshape(B, D, Hh, Hh)
        out = feat
        for name in ("proj", "proj1", "proj2") if c["pos"] == "ppeg" else ("proj",):
            wt = st[f"pos_embedding.{name}.weight"]
            out = out + F.conv2d(feat, wt, st.get(f"pos_embedding.{name}.bias"),
                                 padding=(wt.shape[2] // 2, wt.shape[3] // 2), groups=D)
        out = out.flatten(2).transpose(1, 2)
        return out[:, :-add] if add > 0 else out

    use_pos = c["pos"] in ("peg", "ppeg")
    amp = contextlib.nullcontext() if autocast is None else torch.autocast("cpu", dtype=autocast)
    with (contextlib.nullcontext() if grad else torch.no_grad()), amp:
        if use_pos and c["pos_pos"] == -1:
            x = pos_embed(x)
        for li in range(c["n_layers"] - 1):
            if use_pos and li == 1 and c["pos_pos"] == 0:
                x = pos_embed(x)
            p = f"layers.{li}."
            u = F.layer_norm(x, (D,), st[p + "norm.weight"], st[p + "norm.bias"], 1e-5)
            H, s, add = grid(N, c["region_num"], c["region_size"], c["min_region_num"], c["min_region_ratio"])
            if add > 0:
                u = torch.cat([u, torch.zeros((B, add, D), dtype=u.dtype)], dim=1)
            z = unpart(inner(part(u, H, s), p + "attn.attn.", c["n_heads"], c["epeg_k"], li), H, s)
            if add > 0:
                z = z[:, :-add]
            x = x + bm(li, "attn") * z
            if c["ffn"]:
                x = ffn(x, p, li)
        if c["cr_msa"]:
            p = "cr_msa."
            v = F.layer_norm(x, (D,), st[p + "norm.weight"], st[p + "norm.bias"], 1e-5)
            H, s, add = grid(N, 8, 0, 0, 0.0)
            if add > 0:
                v = torch.cat([v, torch.zeros((B, add, D), dtype=v.dtype)], dim=1)
            xr = part(v, H, s)
            if c["crmsa_mlp"]:
                lg = F.linear(torch.tanh(F.linear(xr, st[p + "attn.phi.0.weight"])),
                              st[p + "attn.phi.2.weight"]).transpose(1, 2)
            else:
                lg = torch.einsum("wpc,cn->wpn", xr, st[p + "attn.phi"]).transpose(1, 2)
            cw = lg.softmax(dim=-1)
            dw = lg.softmax(dim=1)
            mn = lg.min(dim=-1)[0].unsqueeze(-1)
            mx = lg.max(dim=-1)[0].unsqueeze(-1)
            mm = (lg - mn) / (mx - mn + 1e-8)
            rep = torch.einsum("wpc,wnp->wnpc", xr, cw).sum(dim=-2).transpose(0, 1)
            rep = inner(rep, p + "attn.attn.", c["crmsa_heads"], 0, "cr_msa").transpose(0, 1)
            o = torch.einsum("wnc,wnp->wnpc", rep, mm)
            o = torch.einsum("wnpc,wnp->wnpc", o, dw).sum(dim=1)
            z = unpart(o, H, s)
            if add > 0:
                z = z[:, :-add]
            x = x + bm("cr_msa", "attn") * z
            if c["ffn"]:
                x = ffn(x, p, "cr_msa")
        if c["all_shortcut"]:
            x = x + x0
        x = F.layer_norm(x, (D,), st["norm.weight"], st["norm.bias"], 1e-5)
    if grad:
        return (x if batched else x.squeeze(0)), x_leaf, st
    return x if batched else x.squeeze(0)


# --------------------------------------------------------------------------- RRTMIL (row f1) in float64
def _act64(v, name):
    """nn.ReLU / nn.GELU (exact erf form) / nn.Tanh; any other name = no activation layer
    (modules/rrt.py:210-213, modules/datten.py:14-19,51-56)."""
    if name == "relu":
        return np.maximum(v, 0.0)
    if name == "gelu":
        from scipy.special import erf
        return 0.5 * v * (1.0 + erf(v / np.sqrt(2.0)))
    if name == "tanh":
        return np.tanh(v)
    return v


def pool_predict_f64(y, state, da_act="relu", gated=False):
    """DAttention pooling + predictor on encoder features y (N, D): modules/datten.py:28-38
    (Attention.forward) / :69-83 (AttentionGated.forward), modules/rrt.py:241.
    Returns (logits (n_classes,), attention (N,) softmax over the bag, raw scores (N,), pooled (D,))."""
    st = {k: np.asarray(v, dtype=np.float64) for k, v in state.items() if k.startswith(("pool_fn.", "predictor."))}
    y = np.asarray(y, dtype=np.float64)
    pf = "pool_fn.attention."

    def lin(v, name):
        o = v @ st[name + ".weight"].T
        return o + st[name + ".bias"] if name + ".bias" in st else o

    if gated:
        a = _act64(lin(y, pf + "attention_a.0"), da_act)
        b = 1.0 / (1.0 + np.exp(-lin(y, pf + "attention_b.0")))
        raw = lin(a * b, pf + "attention_c")[:, 0]
    else:
        last = 2 if da_act in ("relu", "gelu", "tanh") else 1
        raw = lin(_act64(lin(y, pf + "attention.0"), da_act), pf + f"attention.{last}")[:, 0]
    attn = _softmax64(raw, 0)                       # softmax over the N patches (datten.py:32)
    pooled = attn @ y
    logits = pooled @ st["predictor.weight"].T + st["predictor.bias"]
    return logits, attn, raw, pooled


def mil_forward_f64(feats, state, cfg=None):
    """RRTMIL.forward (eval), modules/rrt.py:227-246: patch_to_emb (+act) -> RRTEncoder -> DAttention ->
    predictor.  ``cfg``: the RRTMIL constructor kwargs; ``state``: its state_dict as arrays."""
    cfg = dict(cfg or {})
    f = np.asarray(feats, dtype=np.float64)
    emb = f @ np.asarray(state["patch_to_emb.0.weight"], np.float64).T + np.asarray(state["patch_to_emb.0.bias"], np.float64)
    emb = _act64(emb, str(cfg.get("act", "relu")).lower())
    enc_state = {k[len("online_encoder."):]: v for k, v in state.items() if k.startswith("online_encoder.")}
    enc_cfg = {k: v for k, v in cfg.items() if k in _DEFAULTS}
    y = forward_f64(emb, enc_state, enc_cfg)
    logits, attn, raw, _ = pool_predict_f64(y, state, cfg.get("da_act", "relu"), bool(cfg.get("da_gated", False)))
    return logits, attn, raw


def flops_per_bag(N, D=512, region_num=8, heads=8, epeg_k=15, crmsa_k=3):
    """Algorithmic FLOPs per bag, SURVEY.md §8(d) / BASELINE.md §3."""
    H, s, _ = grid(N, region_num)
    Np, P = H * H, s * s
    H8, _, _ = grid(N, 8)
    Np8 = H8 * H8
    k = crmsa_k
    return (8 * Np * D * D + 4 * Np * P * D + 2 * Np * P * heads * epeg_k
            + 6 * Np8 * D * k + 8 * k * 64 * D * D + 4 * k * 64 * 64 * D)
