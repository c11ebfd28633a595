"""RRTEncoder -- drop-in for the reference's modules/rrt.py:133-202 on MI355X.

Same constructor signature, parameter names / shapes (state_dict-compatible,
strict=True), ``final_dim`` attribute and (N,D) / (B,N,D) / (B,C,H,W) -> same-shape
forward.  The parameter holders are real nn.LayerNorm / nn.Linear / nn.Conv2d
sub-modules under the reference's names (so ``.apply(initialize_weights)``,
optimizers and checkpoints behave identically); the forward itself is ONE call
into librrt_hip.so (rrt_encoder_forward_f32, include/rrt_hip.h) per bag on the
current HIP stream.  With gradients enabled and something to differentiate the call goes
through a torch.autograd.Function over rrt_encoder_forward_train_f32 /
rrt_encoder_backward_f32 (train() adds proj dropout and stochastic depth, eval() does not).
There is no PyTorch/CPU fallback: CPU tensors and configurations outside the HIP path raise.
"""
import os
import warnings
import contextlib
import ctypes as C
import math

import numpy as np

import torch
from torch import nn

from . import _lib


def initialize_weights(module):
    """modules/rrt.py:9-23: xavier-normal Linear/Conv2d, zero bias, LayerNorm (1, 0)."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.xavier_normal_(m.weight)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


class InnerAttention(nn.Module):
    """Parameter holder mirroring modules/rmsa.py:56-89 (qkv, proj, pe)."""

    def __init__(self, dim, head_dim=None, num_heads=8, qkv_bias=True, proj_drop=0., epeg=True,
                 epeg_k=15, epeg_2d=False, epeg_bias=True, epeg_type='attn', **_ignored):
        super().__init__()
        if epeg_type not in ('attn', 'value_bf', 'value_af'):
            raise NotImplementedError(f"epeg_type={epeg_type!r}")
        head_dim = head_dim or dim // num_heads
        if head_dim * num_heads != dim:
            raise NotImplementedError("head_dim * num_heads must equal dim")
        self.dim, self.num_heads, self.head_dim = dim, num_heads, head_dim
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, head_dim * num_heads * 3, bias=qkv_bias)
        self.proj = nn.Linear(head_dim * num_heads, dim)
        self.proj_drop_p = proj_drop
        self.epeg_k, self.epeg_2d, self.epeg_type = epeg_k, bool(epeg_2d), epeg_type
        # modules/rmsa.py:74-87: the conv runs over the score map ('attn': one channel per head) or over v's token
        # image ('value_*': one channel per feature), with a (k, 1) or -- epeg_2d -- a k x k kernel
        ch = num_heads if epeg_type == 'attn' else head_dim * num_heads
        ks, pad = (epeg_k, epeg_k // 2) if epeg_2d else ((epeg_k, 1), (epeg_k // 2, 0))
        self.pe = nn.Conv2d(ch, ch, ks, padding=pad, groups=ch, bias=epeg_bias) if epeg else None

    def extra_repr(self):
        return f'dim={self.dim}, num_heads={self.num_heads}'


class RegionAttntion(nn.Module):
    """Holder mirroring modules/rmsa.py:152-173 (name kept as spelled in the reference)."""

    def __init__(self, dim, head_dim=None, num_heads=8, region_size=0, qkv_bias=True, drop=0.,
                 region_num=8, epeg=False, min_region_num=0, min_region_ratio=0., region_attn='native',
                 **kwargs):
        super().__init__()
        if region_attn != 'native':
            raise NotImplementedError("region_attn='ntrans' (Nystrom ablation) is out of scope")
        self.dim, self.num_heads = dim, num_heads
        self.region_size = region_size if region_size > 0 else None
        self.region_num = region_num
        self.min_region_num, self.min_region_ratio = min_region_num, min_region_ratio
        self.attn = InnerAttention(dim, head_dim=head_dim, num_heads=num_heads, qkv_bias=qkv_bias,
                                   proj_drop=drop, epeg=epeg, **kwargs)


class CrossRegionAttntion(nn.Module):
    """Holder mirroring modules/rmsa.py:232-259."""

    def __init__(self, dim, head_dim=None, num_heads=8, region_size=0, qkv_bias=True, drop=0.,
                 region_num=8, epeg=False, min_region_num=0, min_region_ratio=0., crmsa_k=3,
                 crmsa_mlp=False, region_attn='native', **kwargs):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.region_size = region_size if region_size > 0 else None
        self.region_num = region_num
        self.min_region_num, self.min_region_ratio = min_region_num, min_region_ratio
        self.attn = InnerAttention(dim, head_dim=head_dim, num_heads=num_heads, qkv_bias=qkv_bias,
                                   proj_drop=drop, epeg=epeg, **kwargs)
        self.crmsa_mlp = crmsa_mlp
        self.crmsa_k = crmsa_k
        if crmsa_mlp:
            self.phi = nn.Sequential(nn.Linear(dim, dim // 4, bias=False), nn.Tanh(),
                                     nn.Linear(dim // 4, crmsa_k, bias=False))
        else:
            self.phi = nn.Parameter(torch.empty((dim, crmsa_k)))
            nn.init.kaiming_uniform_(self.phi, a=math.sqrt(5))


class PEG(nn.Module):
    """Holder mirroring modules/emb_position.py:60-82 (one depth-wise conv + identity)."""

    def __init__(self, dim=512, k=7, bias=True, conv_1d=False):
        super().__init__()
        ks, pad = ((k, 1), (k // 2, 0)) if conv_1d else (k, k // 2)
        self.proj = nn.Conv2d(dim, dim, ks, 1, pad, groups=dim, bias=bias)


class PPEG(nn.Module):
    """Holder mirroring modules/emb_position.py:24-58 (k, 5 and 3 depth-wise convs + identity)."""

    def __init__(self, dim=512, k=7, conv_1d=False, bias=True):
        super().__init__()

        def conv(kk):
            ks, pad = ((kk, 1), (kk // 2, 0)) if conv_1d else (kk, kk // 2)
            return nn.Conv2d(dim, dim, ks, 1, pad, groups=dim, bias=bias)

        self.proj, self.proj1, self.proj2 = conv(k), conv(5), conv(3)


class Mlp(nn.Module):
    """Holder mirroring modules/rrt.py:25-41 (TransLayer's FFN when ffn=True)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.ReLU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class TransLayer(nn.Module):
    """Holder mirroring modules/rrt.py:43-110: pre-norm residual attention block (+ optional FFN)."""

    def __init__(self, norm_layer=nn.LayerNorm, dim=512, head=8, drop_out=0.1, drop_path=0., ffn=False,
                 ffn_act='gelu', mlp_ratio=4., trans_dim=64, attn='rmsa', n_region=8, epeg=False,
                 region_size=0, min_region_num=0, min_region_ratio=0, qkv_bias=True, crmsa_k=3,
                 epeg_k=15, **kwargs):
        super().__init__()
        self.norm = norm_layer(dim)
        self.norm2 = norm_layer(dim) if ffn else nn.Identity()
        common = dict(dim=dim, num_heads=head, drop=drop_out, region_num=n_region, head_dim=dim // head,
                      epeg=epeg, region_size=region_size, min_region_num=min_region_num,
                      min_region_ratio=min_region_ratio, qkv_bias=qkv_bias)
        if attn == 'rmsa':
            self.attn = RegionAttntion(epeg_k=epeg_k, **common, **kwargs)
        elif attn == 'crmsa':
            self.attn = CrossRegionAttntion(crmsa_k=crmsa_k, **common, **kwargs)
        elif attn == 'ntrans':
            raise NotImplementedError("attn='ntrans' (Nystrom ablation) is out of scope")
        else:
            raise NotImplementedError
        # timm's DropPath (rrt.py:102) holds no parameters: at batch size 1 it keeps or drops a whole residual
        # branch; the draw happens in RRTEncoder._branch_scales and rides into the kernels as a multiplier
        self.drop_path = nn.Identity()
        self.drop_path_p = float(drop_path)
        self.ffn = ffn
        act_layer = nn.GELU if ffn_act == 'gelu' else nn.ReLU
        self.mlp = (Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop_out)
                    if ffn else nn.Identity())


class _EncoderFunction(torch.autograd.Function):
    """One bag through rrt_encoder_forward_train_f32 / rrt_encoder_backward_f32 (row f2).  The parameters are
    passed as inputs so that autograd routes their gradients; the stash (intermediates the backward needs)
    lives in a tensor owned by the graph node."""

    @staticmethod
    def forward(ctx, enc, x2d, *params):
        lib = _lib.load()
        n = x2d.shape[0]
        # under autocast (the reference's --amp training) the GEMMs round their operands to bf16 / fp16 like the
        # inference path does; accumulation, every other kernel and the weight gradients stay fp32
        enc._desc.compute = enc._compute_mode()
        stash_b, ws_b = C.c_size_t(), C.c_size_t()
        _lib.check(lib.rrt_encoder_train_sizes(C.byref(enc._desc), n, C.byref(stash_b), C.byref(ws_b)),
                   "rrt_encoder_train_sizes")
        stash = torch.empty(stash_b.value, dtype=torch.uint8, device=x2d.device)
        y = torch.empty_like(x2d)
        w = enc._weights()
        # train-mode proj_drop: a stateless mask keyed by a seed drawn from torch's CPU generator (so
        # torch.manual_seed reproduces a run); the backward regenerates the same mask from (p, seed).
        # eval() with gradients enabled (fine-tuning with frozen dropout, saliency maps): same graph, no dropout
        drop_p = float(enc.drop_out) if enc.training else 0.0
        seed = enc._next_drop_seed() if drop_p > 0 else 0
        branch = enc._branch_scales()
        with torch.cuda.device(x2d.device):
            rc = lib.rrt_encoder_forward_train_f32(C.byref(enc._desc), C.byref(w), x2d.data_ptr(), y.data_ptr(), n,
                                                   stash.data_ptr(), stash.numel(), drop_p, seed, branch,
                                                   torch.cuda.current_stream(x2d.device).cuda_stream)
        _lib.check(rc, "rrt_encoder_forward_train_f32")
        ctx.enc, ctx.stash, ctx.ws_bytes, ctx.drop, ctx.compute = enc, stash, ws_b.value, (drop_p, seed, branch), enc._desc.compute
        # the backward reads the parameters through their live pointers: an in-place update between forward and
        # backward (optimizer.step() before a second backward through a retained graph) must not go unnoticed
        ctx.versions = [(p_.data_ptr(), p_._version) for p_ in params]
        ctx.save_for_backward(x2d)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        enc = ctx.enc
        (x2d,) = ctx.saved_tensors
        if ctx.versions != [(p_.data_ptr(), p_._version) for p_ in enc.parameters()]:
            raise RuntimeError("RRTEncoder: a parameter was modified in place (or replaced) between this forward and "
                               "its backward; the backward kernels read the live weights, so the gradients would be "
                               "wrong -- run the forward again")
        n, dev = x2d.shape[0], x2d.device
        dy = dy.contiguous().float()
        grads, gstruct = enc._grad_buffers(dev)
        dx = torch.empty_like(x2d) if ctx.needs_input_grad[1] else None
        ws = torch.empty(ctx.ws_bytes, dtype=torch.uint8, device=dev)
        w = enc._weights()
        enc._desc.compute = ctx.compute
        with torch.cuda.device(dev):
            rc = lib.rrt_encoder_backward_f32(C.byref(enc._desc), C.byref(w), x2d.data_ptr(), dy.data_ptr(),
                                              ctx.stash.data_ptr(), ctx.stash.numel(), C.byref(gstruct),
                                              dx.data_ptr() if dx is not None else None, n, ws.data_ptr(), ws.numel(),
                                              ctx.drop[0], ctx.drop[1], ctx.drop[2],
                                              torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "rrt_encoder_backward_f32")
        return (None, dx) + tuple(grads)      # (the stash lives as long as the graph node: retain_graph works)


_HWQ_WARNED = False


def _warn_hw_queues(streams):
    """HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; it is
    read once, when the HIP runtime initialises.  With torch's own streams and RCCL's in the process, two of the
    executor's streams can land on ONE queue and serialise (measured: the one-stream rate, -15 %).  The library cannot
    see the setting from inside HIP, so the host side checks the environment and says so -- once."""
    global _HWQ_WARNED
    if _HWQ_WARNED or streams < 2:
        return
    import os
    import warnings
    try:
        q = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        q = 4
    from . import HW_QUEUES
    if HW_QUEUES == "late":            # HIP was initialised before the package could set the variable: the default holds
        q = 4
    if q < 2 * streams + 4:
        _HWQ_WARNED = True
        warnings.warn(f"rrt_mil_amd: {streams} bags in flight but GPU_MAX_HW_QUEUES={q} (HIP default 4): the executor's "
                      "streams may share a hardware queue and serialise.  Export GPU_MAX_HW_QUEUES=16 before the "
                      "process initialises HIP (before `import torch`); see INTEGRATION.md section 5.", RuntimeWarning)


_NULL_CTX = contextlib.nullcontext()
_SIDE_STREAM = {}         # device -> the stream an executor call runs on when the caller sits on the default stream
_LEAD_STREAM = {}         # device -> handle of THE stream that carries the first share of every multi-bag call of this process
_BAG_STREAMS = {}          # device -> [torch.cuda.Stream]: the streams every executor of this process runs its bags on


class RRTEncoder(nn.Module):
    def __init__(self, mlp_dim=512, pos_pos=0, pos='none', peg_k=7, attn='rmsa', region_num=8,
                 drop_out=0.1, n_layers=2, n_heads=8, drop_path=0., ffn=False, ffn_act='gelu',
                 mlp_ratio=4., trans_dim=64, epeg=True, epeg_k=15, region_size=0, min_region_num=0,
                 min_region_ratio=0, qkv_bias=True, peg_bias=True, peg_1d=False, cr_msa=True,
                 crmsa_k=3, all_shortcut=False, crmsa_mlp=False, crmsa_heads=8, need_init=False,
                 **kwargs):
        super().__init__()
        if pos == 'sincos':
            raise NotImplementedError("pos='sincos' is out of scope (the reference's SINCOS needs numpy < 1.24 and a "
                                      "4-D input it never receives)")
        self.final_dim = mlp_dim
        self.norm = nn.LayerNorm(self.final_dim)
        self.all_shortcut = all_shortcut
        self.layers = nn.Sequential(*[
            TransLayer(dim=mlp_dim, head=n_heads, drop_out=drop_out, drop_path=drop_path, ffn=ffn,
                       ffn_act=ffn_act, mlp_ratio=mlp_ratio, trans_dim=trans_dim, attn=attn,
                       n_region=region_num, epeg=epeg, region_size=region_size,
                       min_region_num=min_region_num, min_region_ratio=min_region_ratio,
                       qkv_bias=qkv_bias, epeg_k=epeg_k, **kwargs)
            for _ in range(n_layers - 1)])
        # the reference does not forward region_num / epeg / region_size / min_region_* here
        # (modules/rrt.py:148): CR-MSA always runs an 8x8 grid without EPEG
        self.cr_msa = (TransLayer(dim=mlp_dim, head=crmsa_heads, drop_out=drop_out, drop_path=drop_path,
                                  ffn=ffn, ffn_act=ffn_act, mlp_ratio=mlp_ratio, trans_dim=trans_dim,
                                  attn='crmsa', qkv_bias=qkv_bias, crmsa_k=crmsa_k, crmsa_mlp=crmsa_mlp,
                                  **kwargs) if cr_msa else nn.Identity())
        if pos == 'ppeg':
            self.pos_embedding = PPEG(dim=mlp_dim, k=peg_k, bias=peg_bias, conv_1d=peg_1d)
        elif pos == 'peg':
            self.pos_embedding = PEG(mlp_dim, k=peg_k, bias=peg_bias, conv_1d=peg_1d)
        else:
            self.pos_embedding = nn.Identity()
        self.pos_pos = pos_pos
        self.drop_out = drop_out
        self._desc = _lib.EncoderDesc(
            dim=mlp_dim, n_heads=n_heads, n_rmsa_layers=n_layers - 1, region_num=region_num,
            region_size=region_size, min_region_num=min_region_num, min_region_ratio=min_region_ratio,
            epeg=int(bool(epeg)), epeg_k=epeg_k, cr_msa=int(bool(cr_msa)), crmsa_k=crmsa_k,
            crmsa_heads=crmsa_heads, crmsa_mlp=int(bool(crmsa_mlp)), all_shortcut=int(bool(all_shortcut)),
            compute=_lib.COMPUTE_F32, ffn=int(bool(ffn)),
            ffn_act=_lib.ACT_GELU if ffn_act == 'gelu' else _lib.ACT_RELU, ffn_hidden=int(mlp_dim * mlp_ratio),
            pos={'peg': _lib.POS_PEG, 'ppeg': _lib.POS_PPEG}.get(pos, _lib.POS_NONE), pos_pos=pos_pos, peg_k=peg_k,
            peg_1d=int(bool(peg_1d)), epeg_2d=int(bool(kwargs.get('epeg_2d', False))),
            epeg_type={'attn': _lib.EPEG_ATTN, 'value_bf': _lib.EPEG_VALUE_BF,
                       'value_af': _lib.EPEG_VALUE_AF}[kwargs.get('epeg_type', 'attn')])
        if self._desc.pos and (pos_pos not in (-1, 0) or (pos_pos == 0 and n_layers - 1 < 2)):
            # the reference only applies pos_embedding before the first layer (pos_pos = -1) or before layer index 1
            # (pos_pos = 0, which needs a second R-MSA layer: the default `--pos ppeg` run with n_layers = 2 never
            # reaches it, rrt.py:181-187): the stage is absent from the kernels and its parameters get no gradient
            self._desc.pos = _lib.POS_NONE
        # None: exact fp32 unless the call runs under torch autocast (then bf16 / fp16 arithmetic like the
        # reference's --amp path); or force torch.float32 / bfloat16 / float16, or "f32x3" (inference: the two big
        # projections emulated in fp32 on the bf16 matrix cores, ~1e-6 from the exact path, include/rrt_hip.h)
        self.compute_dtype = None
        self._ws = None           # cached workspace tensor (grown on demand, per device)
        if need_init:
            self.apply(initialize_weights)

    # per-process state that must not travel with copy.deepcopy / pickle / torch.save(module): raw pointers (the cached ctypes
    # weight struct, the executor handle), device workspaces and the caches keyed on them
    _TRANSIENT = ("_w_struct", "_plist", "_w_fp", "_w16_key", "_ws", "_ws_need", "_ex", "_ex_key", "_ex_desc")

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in self._TRANSIENT:
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.__dict__.setdefault("_ws", None)

    # ------------------------------------------------------------------ C-ABI plumbing
    @staticmethod
    def _ptr(t):
        if t is None:
            return None
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.RRTHipError("parameters must be contiguous fp32 tensors (bf16 / fp16 arithmetic is selected by "
                                   "torch.autocast or compute_dtype, not by casting the module)")
        return t.data_ptr()

    def _attn_weights(self, layer):
        ia = layer.attn.attn
        w = _lib.AttnWeights()
        w.norm_w, w.norm_b = self._ptr(layer.norm.weight), self._ptr(layer.norm.bias)
        w.qkv_w, w.qkv_b = self._ptr(ia.qkv.weight), self._ptr(ia.qkv.bias)
        w.proj_w, w.proj_b = self._ptr(ia.proj.weight), self._ptr(ia.proj.bias)
        if ia.pe is not None:
            w.pe_w, w.pe_b = self._ptr(ia.pe.weight), self._ptr(ia.pe.bias)   # [h,1,k,1] == [h,k]
        if layer.ffn:
            w.norm2_w, w.norm2_b = self._ptr(layer.norm2.weight), self._ptr(layer.norm2.bias)
            w.fc1_w, w.fc1_b = self._ptr(layer.mlp.fc1.weight), self._ptr(layer.mlp.fc1.bias)
            w.fc2_w, w.fc2_b = self._ptr(layer.mlp.fc2.weight), self._ptr(layer.mlp.fc2.bias)
        return w

    def _weights_version(self):
        """A number that changes whenever ANY parameter of the encoder does (storage pointer or autograd version counter:
        optimizer steps, load_state_dict, .to(), in-place ops all bump one of them; writes through ``p.data`` do not --
        call invalidate_weight_cache() after those).  Lets the library keep the 16-bit weight images of the
        reduced-precision modes across calls (rrt_encoder_desc.weights16_valid, rrt_encoder_weights.version) -- the
        R-MSA layers' qkv / proj images AND CR-MSA's inner qkv / proj images (round-3 advisor finding: the fingerprint
        used to cover the R-MSA layers only, so a CR-MSA-only update, or any update at n_layers = 1, kept stale images) --
        and lets this module keep its ctypes weight struct between calls."""
        slots = self.__dict__.get("_plist")
        if slots is None:
            # (the modules' own parameter dicts, not the Parameter objects: assigning a new nn.Parameter to an existing
            #  submodule is then seen too; replacing a whole submodule needs invalidate_weight_cache())
            slots = self.__dict__["_plist"] = [(m._parameters, k) for m in self.modules() for k in m._parameters
                                               if m._parameters[k] is not None]
        fp = tuple([(d[k].data_ptr(), d[k]._version) for d, k in slots])
        if fp != self.__dict__.get("_w_fp"):
            self.__dict__["_w_fp"], self.__dict__["_w_ver"] = fp, self.__dict__.get("_w_ver", 0) + 1
            self.__dict__["_w_struct"] = None
        return self._w_ver

    def invalidate_weight_cache(self):
        """Forget the cached 16-bit weight images and the cached pointer struct (needed only after writing weights
        through ``.data`` or replacing a parameter object)."""
        self.__dict__["_w_fp"], self.__dict__["_w16_key"] = None, None
        self.__dict__["_w_struct"], self.__dict__["_plist"] = None, None

    def _apply(self, fn, *a, **kw):
        # .to() / .cuda() / .float() may REPLACE parameter objects: drop the cached list with them
        r = super()._apply(fn, *a, **kw)
        self.__dict__["_plist"] = None
        self.__dict__["_w_struct"] = None
        return r

    def _weights(self):
        """The C struct of parameter pointers (rrt_encoder_weights).  Cached between calls while no parameter changed
        storage or version (the struct holds raw pointers: it is rebuilt whenever the fingerprint moves)."""
        ver = self._weights_version()
        w = self.__dict__.get("_w_struct")
        if w is not None:
            return w
        w = _lib.EncoderWeights()
        w.version = ver
        for i, layer in enumerate(self.layers.children()):
            w.rmsa[i] = self._attn_weights(layer)
        if self._desc.cr_msa:
            w.crmsa = self._attn_weights(self.cr_msa)
            if self._desc.crmsa_mlp:
                w.phi0_w = self._ptr(self.cr_msa.attn.phi[0].weight)
                w.phi2_w = self._ptr(self.cr_msa.attn.phi[2].weight)
            else:
                w.phi = self._ptr(self.cr_msa.attn.phi)
        w.norm_w, w.norm_b = self._ptr(self.norm.weight), self._ptr(self.norm.bias)
        if self._desc.pos:
            for i, name in enumerate(("proj", "proj1", "proj2")):
                conv = getattr(self.pos_embedding, name, None)
                if conv is not None:
                    w.pos_w[i], w.pos_b[i] = self._ptr(conv.weight), self._ptr(conv.bias)
        self.__dict__["_w_struct"] = w
        return w

    def _grad_buffers(self, device):
        """Fresh gradient tensors in the order of self.parameters() and the C struct pointing into them.
        LayerNorm pairs share one [2, dim] buffer (d gamma, d beta); pe.bias gets zeros (its gradient is
        exactly zero: a per-head constant added to every score cancels in the softmax)."""
        D = self.final_dim
        gs = _lib.EncoderGrads()
        by_name = {}

        def ln(prefix):
            t = torch.empty((2, D), dtype=torch.float32, device=device)
            by_name[prefix + ".weight"], by_name[prefix + ".bias"] = t[0], t[1]
            return t.data_ptr()

        def attn(prefix, layer, ag):
            ia = layer.attn.attn
            ag.norm = ln(prefix + "norm")
            for nm, mod in (("qkv", ia.qkv), ("proj", ia.proj)):
                gw = torch.empty_like(mod.weight)
                by_name[f"{prefix}attn.attn.{nm}.weight"] = gw
                setattr(ag, nm + "_w", gw.data_ptr())
                if mod.bias is not None:
                    gb = torch.empty_like(mod.bias)
                    by_name[f"{prefix}attn.attn.{nm}.bias"] = gb
                    setattr(ag, nm + "_b", gb.data_ptr())
            if ia.pe is not None:
                gw = torch.empty_like(ia.pe.weight)
                by_name[prefix + "attn.attn.pe.weight"] = gw
                ag.pe_w = gw.data_ptr()
                if ia.pe.bias is not None:
                    if ia.epeg_type == 'attn':          # a per-head constant on every score: cancels in the softmax
                        by_name[prefix + "attn.attn.pe.bias"] = torch.zeros_like(ia.pe.bias)
                    else:                               # value_bf / value_af: the conv output is added to v / x
                        gb = torch.empty_like(ia.pe.bias)
                        by_name[prefix + "attn.attn.pe.bias"] = gb
                        ag.pe_b = gb.data_ptr()
            if layer.ffn:
                ag.norm2 = ln(prefix + "norm2")
                for nm in ("fc1", "fc2"):
                    mod = getattr(layer.mlp, nm)
                    gw, gb = torch.empty_like(mod.weight), torch.empty_like(mod.bias)
                    by_name[f"{prefix}mlp.{nm}.weight"], by_name[f"{prefix}mlp.{nm}.bias"] = gw, gb
                    setattr(ag, nm + "_w", gw.data_ptr())
                    setattr(ag, nm + "_b", gb.data_ptr())

        gs.norm = ln("norm")
        for i, layer in enumerate(self.layers.children()):
            attn(f"layers.{i}.", layer, gs.rmsa[i])
        if self._desc.cr_msa:
            attn("cr_msa.", self.cr_msa, gs.crmsa)
            if self._desc.crmsa_mlp:
                g0 = torch.empty_like(self.cr_msa.attn.phi[0].weight)
                g2 = torch.empty_like(self.cr_msa.attn.phi[2].weight)
                by_name["cr_msa.attn.phi.0.weight"], by_name["cr_msa.attn.phi.2.weight"] = g0, g2
                gs.phi0_w, gs.phi2_w = g0.data_ptr(), g2.data_ptr()
            else:
                gphi = torch.empty_like(self.cr_msa.attn.phi)
                by_name["cr_msa.attn.phi"] = gphi
                gs.phi = gphi.data_ptr()
        if self._desc.pos:
            for i, name in enumerate(("proj", "proj1", "proj2")):
                conv = getattr(self.pos_embedding, name, None)
                if conv is None:
                    continue
                gw = torch.empty_like(conv.weight)
                by_name[f"pos_embedding.{name}.weight"] = gw
                gs.pos_w[i] = gw.data_ptr()
                if conv.bias is not None:
                    gb = torch.empty_like(conv.bias)
                    by_name[f"pos_embedding.{name}.bias"] = gb
                    gs.pos_b[i] = gb.data_ptr()
        elif not isinstance(self.pos_embedding, nn.Identity):
            # constructed but never applied: no gradient, exactly as the reference's autograd leaves None (an
            # optimizer then skips these parameters instead of decaying them)
            for name, _ in self.pos_embedding.named_parameters():
                by_name["pos_embedding." + name] = None
        # the struct holds raw pointers: the caller keeps `grads` alive through the C call and then hands the
        # tensors to autograd as their ONLY owner -- AccumulateGrad then adopts them instead of cloning (an extra
        # reference here cost ~20 device copies per step)
        grads = [by_name[name] for name, _ in self.named_parameters()]
        return grads, gs

    def _workspace(self, n_tokens, device):
        # the size query is a C call + a geometry computation: remembered per (bag size, arithmetic) -- every other field
        # of the descriptor is fixed at construction
        cache = self.__dict__.setdefault("_ws_need", {})
        key = (int(n_tokens), int(self._desc.compute))
        need = cache.get(key)
        if need is None:
            lib = _lib.load()
            c_need = C.c_size_t()
            _lib.check(lib.rrt_encoder_workspace_size(C.byref(self._desc), n_tokens, C.byref(c_need)),
                       "rrt_encoder_workspace_size")
            need = cache[key] = int(c_need.value)
            if len(cache) > 4096:
                cache.clear()
        ws = self._ws
        if ws is None or ws.device != device or ws.numel() < need:
            ws = self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return ws

    def _compute_mode(self):
        dt = self.compute_dtype
        if dt is None and torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
        return {None: _lib.COMPUTE_F32, torch.float32: _lib.COMPUTE_F32, torch.bfloat16: _lib.COMPUTE_BF16,
                torch.float16: _lib.COMPUTE_F16, "f32x3": _lib.COMPUTE_F32X3}[dt]

    def forward_bag(self, x2d, out=None):
        """One bag: x2d (N, D) fp32 device tensor -> (N, D).  Enqueued on the current stream."""
        lib = _lib.load()
        if not x2d.is_cuda:
            raise _lib.RRTHipError("rrt_mil_amd.RRTEncoder runs on MI355X only: move the bag to a "
                                   "'cuda' (HIP) device; there is no CPU fallback")
        if x2d.dtype in (torch.bfloat16, torch.float16):
            x2d = x2d.float()        # e.g. the bf16 output of an autocast patch_to_emb; HBM tensors stay fp32
        if x2d.dtype != torch.float32:
            raise NotImplementedError(f"unsupported bag dtype {x2d.dtype}")
        x2d = x2d.contiguous()
        n, d = x2d.shape
        if d != self.final_dim:
            raise ValueError(f"expected feature dim {self.final_dim}, got {d}")
        y = out if out is not None else torch.empty_like(x2d)
        if self._stochastic():
            # train() without a graph (torch.no_grad(): MC dropout, EMA / teacher forwards, frozen models): the
            # reference still applies proj dropout and stochastic depth; same kernels as the training forward, the
            # stash is scratch
            return self._forward_bag_stochastic(x2d, y)
        desc, d_ = self._desc, self.__dict__
        mode = desc.compute = self._compute_mode()
        ws = self._workspace(n, x2d.device)
        w = self._weights()
        dev = x2d.device
        # kernels launch on the bag's device, whatever the current one is (the context switch is skipped when it already is)
        ctx = torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else _NULL_CTX
        with ctx:
            stream = torch.cuda.current_stream(dev).cuda_stream
            # the 16-bit weight images at the head of this workspace are still those of these weights?
            lowp = mode != _lib.COMPUTE_F32
            key = (ws.data_ptr(), mode, w.version, stream) if lowp else None
            desc.weights16_valid = int(lowp and key == d_.get("_w16_key"))
            # scheduling hint (rrt_hip.h: deterministic for a given setting, low-order bits may differ between settings):
            # a plain forward is the reference's loop, one bag at a time; callers that keep several forwards in flight on
            # their own streams set enc.solo = False
            desc.solo = int(d_.get("solo", True))
            rc = lib.rrt_encoder_forward_f32(C.byref(desc), C.byref(w), x2d.data_ptr(), y.data_ptr(), n,
                                             ws.data_ptr(), ws.numel(), stream)
            desc.weights16_valid = 0
            d_["_w16_key"] = key if rc == 0 else None
        if rc:
            _lib.check(rc, "rrt_encoder_forward_f32")
        return y

    def forward_batch(self, x3d):
        """(B, N, D) -> (B, N, D) with the reference's batch semantics (CR-MSA's inner attention over the representatives
        of ALL bags of the batch, modules/rmsa.py:316-322).  Inference; one library call on the current stream."""
        lib = _lib.load()
        if not x3d.is_cuda:
            raise _lib.RRTHipError("rrt_mil_amd.RRTEncoder runs on MI355X only; there is no CPU fallback")
        x3d = x3d.float().contiguous()
        b, n, d = x3d.shape
        if d != self.final_dim:
            raise ValueError(f"expected feature dim {self.final_dim}, got {d}")
        desc = self._desc
        desc.compute = self._compute_mode()
        need = C.c_size_t()
        _lib.check(lib.rrt_encoder_batch_workspace_size(C.byref(desc), b, n, C.byref(need)), "rrt_encoder_batch_workspace_size")
        ws = torch.empty(need.value, dtype=torch.uint8, device=x3d.device)
        y = torch.empty_like(x3d)
        w = self._weights()
        with torch.cuda.device(x3d.device):
            desc.weights16_valid = 0
            rc = lib.rrt_encoder_forward_batch_f32(C.byref(desc), C.byref(w), x3d.data_ptr(), y.data_ptr(), b, n,
                                                   ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x3d.device).cuda_stream)
        _lib.check(rc, "rrt_encoder_forward_batch_f32")
        return y

    def _stochastic(self):
        return self.training and (self.drop_out > 0 or any(l.drop_path_p > 0 for l in self._trans_layers()))

    def _trans_layers(self):
        layers = list(self.layers.children())
        if self._desc.cr_msa:
            layers.append(self.cr_msa)
        return layers

    def _branch_scales(self):
        """Stochastic depth for this call (timm DropPath at batch 1, rrt.py:102,125,129): one multiplier per residual
        branch -- 1 / keep_prob if the branch is kept, 0 if dropped, 1 outside train() -- as the host float array
        rrt_encoder_forward_train_f32 takes (NULL when there is nothing to draw).  ``drop_path_draws`` (a list of
        multipliers, attention branches then FFN branches) pins them (tests)."""
        if not self.training or not any(l.drop_path_p > 0 for l in self._trans_layers()):
            return None
        n = _lib.RRT_MAX_RMSA_LAYERS + 1
        arr = (C.c_float * (2 * n))(*([1.0] * (2 * n)))
        layers = list(self.layers.children())
        slots = [(i, l) for i, l in enumerate(layers)] + ([(len(layers), self.cr_msa)] if self._desc.cr_msa else [])
        fixed = getattr(self, "drop_path_draws", None)
        k = 0
        for part in (0, 1):                      # 0: attention branches, 1: FFN branches (ffn=True only)
            for idx, layer in slots:
                if part == 1 and not layer.ffn:
                    continue
                keep = 1.0 - layer.drop_path_p
                if fixed is not None:
                    m = float(fixed[k])
                else:
                    m = (1.0 / keep if keep > 0 else 1.0) if float(torch.rand(())) < keep else 0.0
                arr[part * n + idx] = m
                k += 1
        return arr

    def _forward_bag_stochastic(self, x2d, y):
        lib = _lib.load()
        self._desc.compute = self._compute_mode()
        stash_b, ws_b = C.c_size_t(), C.c_size_t()
        _lib.check(lib.rrt_encoder_train_sizes(C.byref(self._desc), x2d.shape[0], C.byref(stash_b), C.byref(ws_b)),
                   "rrt_encoder_train_sizes")
        stash = torch.empty(stash_b.value, dtype=torch.uint8, device=x2d.device)
        w = self._weights()
        drop_p = float(self.drop_out)
        with torch.cuda.device(x2d.device):
            rc = lib.rrt_encoder_forward_train_f32(C.byref(self._desc), C.byref(w), x2d.data_ptr(), y.data_ptr(),
                                                   x2d.shape[0], stash.data_ptr(), stash.numel(), drop_p,
                                                   self._next_drop_seed() if drop_p > 0 else 0, self._branch_scales(),
                                                   torch.cuda.current_stream(x2d.device).cuda_stream)
        _lib.check(rc, "rrt_encoder_forward_train_f32")
        return y

    # ------------------------------------------------------------------ batch of independent bags
    def _executor(self, n_streams, max_tokens, device):
        key = (n_streams, device)
        ex = getattr(self, "_ex", None)
        self._desc.weights16_valid, self._desc.solo = 0, 0      # per-call fields: the executor sets its own
        if ex is not None and self._ex_key == key and bytes(self._ex_desc) == bytes(self._desc):
            return ex
        self._drop_executor()
        lib = _lib.load()
        h = C.c_void_p()
        with torch.cuda.device(device):
            if os.environ.get("RRT_EXEC_OWN_STREAMS") == "1":        # (experiments: streams created by the library)
                _lib.check(lib.rrt_executor_create(C.byref(self._desc), n_streams, max_tokens, C.byref(h)),
                           "rrt_executor_create")
            else:
                # the process's ONE set of bag streams (torch's stream pool), whatever executors come and go
                pool = _BAG_STREAMS.setdefault(device, [])
                off = int(os.environ.get("RRT_EXEC_POOL_OFFSET", "0"))        # (experiments)
                while len(pool) < n_streams + off:
                    pool.append(torch.cuda.Stream(device))
                # slot 0 of an executor call runs on the caller's stream: slots 1 .. n-1 take the FIRST streams of the list
                hs = [pool[off + n_streams - 1].cuda_stream] + [st.cuda_stream for st in pool[off:off + n_streams - 1]]
                arr = (C.c_void_p * n_streams)(*hs)
                _lib.check(lib.rrt_executor_create_on_streams(C.byref(self._desc), n_streams, arr, max_tokens, C.byref(h)),
                           "rrt_executor_create_on_streams")
        self._ex, self._ex_key, self._ex_desc = h, key, _lib.EncoderDesc.from_buffer_copy(self._desc)
        return h

    def _drop_executor(self):
        ex = getattr(self, "_ex", None)
        if ex is not None:
            _lib.load().rrt_executor_destroy(ex)
            self._ex = None

    def __del__(self):
        try:
            self._drop_executor()
        except Exception:
            pass

    @torch.no_grad()
    def forward_bags(self, bags, streams=4, outs=None):
        """A batch of independent bags (each (N_i, D) or (1, N_i, D), any mix of sizes) -> list of outputs of
        the same shapes.  What the reference does with ``for bag in loader: model(bag)`` (main.py:466-467),
        with ``streams`` bags in flight on the library's own HIP streams (rrt_executor_forward); ordered on
        the current stream like a normal op.  Bags are never mixed.

        ``outs``: optional list of preallocated fp32 outputs (bag shapes).  Called under ``with torch.cuda.stream(s):`` the
        call is asynchronous (the host returns once the launches are queued and prepares the next call while the GPU runs);
        on the process's default stream a call of >= 16 bags blocks the host until the bags are done (see below)."""
        lib = _lib.load()
        n_bags = len(bags)
        if not n_bags:
            return []
        if self._stochastic():          # train() under no_grad: dropout / stochastic depth per bag, one bag at a time
            return [self.forward_bag(b[0]).unsqueeze(0) if b.dim() == 3 else self.forward_bag(b) for b in bags]
        dev = bags[0].device
        D, f32 = self.final_dim, torch.float32
        # Host time per bag matters here: a blocking call leaves the GPU idle from its return to the next call's first launch,
        # and at 50 us of GPU time per bf16 bag every microsecond of Python per bag is 2 % (round 6: the per-bag view /
        # contiguous() / struct-field code measured 3-4 us per bag; this pass is ~1).  Fast path: fp32, contiguous, (N, D) or
        # (1, N, D) on one device -- a (1, N, D) bag IS its (N, D) rows, no view object is made.
        xs = bags
        for b in bags:
            nd = b.dim()
            if (b.dtype is not f32 or b.device != dev or not b.is_contiguous() or b.size(-1) != D
                    or not (nd == 2 or (nd == 3 and b.size(0) == 1))):
                xs = None
                break
        if xs is None:                  # the general route: checks with messages, 16-bit inputs, strided bags
            xs = []
            for b in bags:
                if not b.is_cuda or b.device != dev:
                    raise _lib.RRTHipError("forward_bags: every bag must be on the same HIP device (no CPU fallback)")
                x2 = b[0] if b.dim() == 3 and b.size(0) == 1 else b
                if x2.dim() != 2 or x2.size(1) != D:
                    raise ValueError(f"forward_bags: expected (N, {D}) or (1, N, {D}) bags, got {tuple(b.shape)}")
                if x2.dtype in (torch.bfloat16, torch.float16):
                    x2 = x2.float()
                if x2.dtype != f32:
                    raise NotImplementedError(f"unsupported bag dtype {x2.dtype}")
                xs.append(x2.contiguous())
        elif not dev.type == "cuda":
            raise _lib.RRTHipError("forward_bags: every bag must be on the same HIP device (no CPU fallback)")
        if outs is None:
            ys = [torch.empty_like(x) for x in xs]
        else:
            ys = outs
            if len(ys) != n_bags:
                raise ValueError(f"forward_bags: {n_bags} bags but {len(ys)} outputs")
            for x, y in zip(xs, ys):
                if y.dtype is not f32 or y.device != dev or not y.is_contiguous() or y.numel() != x.numel():
                    raise ValueError("forward_bags: every output must be a contiguous fp32 tensor of its bag's size on the bags' device")
        self._desc.compute = self._compute_mode()
        _warn_hw_queues(int(streams))
        tokens = [x.size(-2) for x in xs]
        ex = self._executor(int(streams), max(tokens), dev)
        # the rrt_bag array, filled through one int64 view (x, y, n_tokens per bag) instead of 3 n ctypes field stores
        arr = (_lib.Bag * n_bags)()
        tab = np.frombuffer(arr, dtype=np.int64).reshape(n_bags, 3)
        tab[:, 0] = [x.data_ptr() for x in xs]
        tab[:, 1] = [y.data_ptr() for y in ys]
        tab[:, 2] = tokens
        w = self._weights()
        with (torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else _NULL_CTX):
            cur = torch.cuda.current_stream(dev)
            in_flight = max(1, min(int(streams), n_bags // 4))      # the executor's rule: one stream per four bags of the call
            # ONE lead stream per process and device (round 6).  The executor carries the first share of a call's bags on the
            # stream the call is ordered on; a process that called from the default stream and later from a stream of its own
            # (or from two of its own) had then used FIVE streams for bags, and from the second caller on every call ran at the
            # one-bag-in-flight rate -- 5.23 -> 4.56 k slides/s fp32, 17.3 -> 12.9 k bf16, in either order
            # (tools/experiments/r06_probe_modcall.py): the chip schedules four queues.  So the first multi-bag call fixes the
            # lead -- the caller's own stream (asynchronous calls), or the side stream of the default-stream case below -- and a
            # call from any OTHER stream runs on the lead behind the caller's earlier work while the HOST waits for it (as the
            # default-stream case always did): blocking instead of asynchronous, but at the four-in-flight rate.
            lead = _LEAD_STREAM.get(dev)
            multi = in_flight >= 2
            if multi and lead is None and cur.cuda_stream != 0:
                lead = _LEAD_STREAM[dev] = cur
            if multi and lead is not None and cur.cuda_stream != 0 and lead.cuda_stream != cur.cuda_stream:
                lead.wait_stream(cur)
                rc = lib.rrt_executor_forward(ex, C.byref(w), arr, n_bags, lead.cuda_stream)
                lead.synchronize()
            elif not (cur.cuda_stream == 0 and in_flight >= 4):
                # the call is ordered on the caller's stream, which carries the first share of the bags itself
                rc = lib.rrt_executor_forward(ex, C.byref(w), arr, n_bags, cur.cuda_stream)
            else:
                # Four bags in flight and the caller on the process's DEFAULT stream (handle 0).  Bags on that stream next
                # to three others run at 4.5 k slides/s instead of 5.1 k (measured, tools/bench_bags.py: with hundreds of
                # launches queued the legacy default stream behaves like one queue more), and a default stream that merely
                # WAITS for the bag streams (event wait) is a fifth active queue with the same cost (4.2-4.5 k).  So the
                # call runs on a side stream, ordered behind the default stream's earlier work, and the HOST waits for it:
                # nothing is parked on the default stream.  Round 6, 256 bags per call: 5.26 k fp32 / 19.3 k bf16 this way,
                # 5.29 k / 19.8-20.2 k from a caller under its own `with torch.cuda.stream(s):` (asynchronous), 5.32 k / 20.4 k
                # for the raw C-ABI loop of bench.py (profiles/r06_probe1.txt).
                side = lead if lead is not None else _SIDE_STREAM.get(dev)
                if side is None:
                    side = _SIDE_STREAM[dev] = torch.cuda.Stream(dev)
                if lead is None:
                    _LEAD_STREAM[dev] = side
                side.wait_stream(cur)
                rc = lib.rrt_executor_forward(ex, C.byref(w), arr, n_bags, side.cuda_stream)
                side.synchronize()
        _lib.check(rc, "rrt_executor_forward")
        # xs / ys are touched on the executor's streams, but the call joins the current stream before it
        # returns, so the caching allocator's stream-ordered reuse of these buffers stays correct
        if xs is bags and outs is None:
            return ys                   # allocated with the bags' own shapes
        return [y.unsqueeze(0) if (b.dim() == 3 and y.dim() == 2) else y for b, y in zip(bags, ys)]

    def _next_drop_seed(self):
        """63-bit seed for this call's dropout masks.  ``drop_seed`` (int) pins it (tests); otherwise it comes from
        torch's default CPU generator."""
        fixed = getattr(self, "drop_seed", None)
        if fixed is not None:
            return int(fixed)
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())

    def _wants_grad(self, x):
        """Autograd path (stash + graph node) whenever gradients are enabled and there is something to
        differentiate -- in train() and in eval() alike, as the reference records a graph in both (eval() only
        switches dropout / stochastic depth off).  Inference proper runs under torch.no_grad(), like the
        reference's validation loops.  In eval() a configuration the backward kernels do not cover falls back
        to the inference path with a warning (its output carries no graph); in train() it raises."""
        if not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return False
        if self.training:
            return True
        n = x.shape[-2] if x.dim() in (2, 3) else x.shape[-1] * x.shape[-2]
        a, b = C.c_size_t(), C.c_size_t()
        if _lib.load().rrt_encoder_train_sizes(C.byref(self._desc), int(n), C.byref(a), C.byref(b)) == 0:
            return True
        if not getattr(self, "_warned_detached", False):
            self._warned_detached = True
            warnings.warn("RRTEncoder.eval() called with gradients enabled on a configuration the backward kernels "
                          "do not cover: running the inference path, the output is detached from the graph "
                          "(wrap inference in torch.no_grad() to silence this)", RuntimeWarning, stacklevel=3)
        return False

    def forward(self, x):
        if self._wants_grad(x):
            return self._forward_impl(x, True)
        with torch.no_grad():
            return self._forward_impl(x, False)

    def forward_bag_train(self, x2d):
        """One bag with an autograd graph (rrt_encoder_forward_train_f32 / rrt_encoder_backward_f32)."""
        if not x2d.is_cuda:
            raise _lib.RRTHipError("rrt_mil_amd.RRTEncoder runs on MI355X only; there is no CPU fallback")
        x2d = x2d.float().contiguous()
        if x2d.shape[1] != self.final_dim:
            raise ValueError(f"expected feature dim {self.final_dim}, got {x2d.shape[1]}")
        return _EncoderFunction.apply(self, x2d, *self.parameters())

    def _forward_impl(self, x, train):
        # rank handling: modules/rrt.py:166-175 and :197-201
        shape_len = 3
        if x.dim() == 2:
            x = x.unsqueeze(0)
            shape_len = 2
        if x.dim() == 4:
            x = x.reshape(x.size(0), x.size(1), -1).transpose(1, 2)
            shape_len = 4
        batch, num_patches, ch = x.shape
        if batch != 1:
            # the reference couples the bags of a batch inside CR-MSA (the regions of all bags share one attention
            # sequence, modules/rmsa.py:316-322): reproduced by rrt_encoder_forward_batch_f32 (inference); every reference
            # trainer uses batch_size = 1, and independent bags belong in forward_bags()
            if train or self._stochastic():
                raise NotImplementedError("batch > 1 with an autograd graph or dropout: the HIP backward covers one bag "
                                          "per call (every reference trainer uses batch_size=1)")
            y = self.forward_batch(x)
        else:
            y = (self.forward_bag_train(x[0]) if train else self.forward_bag(x[0])).unsqueeze(0)
        if shape_len == 2:
            y = y.squeeze(0)
        elif shape_len == 4:
            y = y.transpose(1, 2).reshape(batch, ch, int(num_patches ** 0.5), int(num_patches ** 0.5))
        return y
