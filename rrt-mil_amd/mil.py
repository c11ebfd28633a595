"""RRTMIL -- the reference's R2T-MIL slide classifier around the MI355X encoder.

Mirrors modules/rrt.py:204-246 (RRTMIL) and modules/datten.py:5-101 (ABMIL attention
pooling): ``patch_to_emb`` Linear(input_dim, 512)+act -> Dropout -> RRTEncoder (the HIP
path) -> DAttention pooling -> predictor.  Same constructor, parameter names and
forward signature, so reference checkpoints load with strict=True.

Row f1 of SURVEY.md §8: the prologue / pooling / predictor are the *callers* of the hot
path.  For the form every reference trainer uses -- one bag (1, N, input_dim), eval -- the whole
classifier is ONE C-ABI call (`rrt_mil_forward_f32`): patch_to_emb on the fp32 matrix cores with
the activation in the GEMM epilogue, the encoder, and DAttention pooling + predictor as an online
softmax over token chunks (csrc/mil_pool.hip).  The nn.Module tree below is the parameter store
(reference names) and the composite path for other input ranks.
"""
import ctypes as C

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .encoder import RRTEncoder, initialize_weights


def _act(name):
    return {"gelu": nn.GELU, "relu": nn.ReLU, "tanh": nn.Tanh}.get(name)


class _LibLinear(torch.autograd.Function):
    """nn.Linear on the library's GEMM kernels with a graph (round 3: the classifier's big products -- patch_to_emb
    (N x input_dim -> 512) and DAttention's first Linear(512, 128) -- no longer go through torch / rocBLAS in training):
    forward = rrt_linear_f32, backward = rrt_linear_backward_f32 (dX via the forward GEMM on W^T, dW as the split-K TN
    product with the bias gradient summed inside it).  Operand precision follows autocast like the encoder."""

    @staticmethod
    def forward(ctx, x2d, weight, bias, compute):
        lib = _lib.load()
        x2d = x2d.float().contiguous()
        M, K = x2d.shape
        N = weight.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x2d.device)
        with torch.cuda.device(x2d.device):
            st = torch.cuda.current_stream(x2d.device).cuda_stream
            _lib.check(lib.rrt_linear_f32(x2d.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                          y.data_ptr(), M, N, K, 0, 1.0, compute, st), "rrt_linear_f32")
        ctx.save_for_backward(x2d, weight)
        ctx.has_bias, ctx.compute = bias is not None, compute
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2d, weight = ctx.saved_tensors
        dy = dy.float().contiguous()
        M, K = x2d.shape
        N = weight.shape[0]
        need = C.c_size_t()
        _lib.check(lib.rrt_linear_backward_workspace_size(M, N, K, C.byref(need)), "rrt_linear_backward_workspace_size")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dy.device)
        dx = torch.empty_like(x2d) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        db = torch.empty(N, dtype=torch.float32, device=dy.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        with torch.cuda.device(dy.device):
            st = torch.cuda.current_stream(dy.device).cuda_stream
            _lib.check(lib.rrt_linear_backward_f32(dy.data_ptr(), x2d.data_ptr(), weight.data_ptr(),
                                                   dx.data_ptr() if dx is not None else None,
                                                   dw.data_ptr() if dw is not None else None,
                                                   db.data_ptr() if db is not None else None, M, N, K, ctx.compute,
                                                   ws.data_ptr(), ws.numel(), st), "rrt_linear_backward_f32")
        return dx, dw, db, None


def lib_linear(lin, x, compute):
    """x (..., K) through nn.Linear `lin` on the library kernels when they apply (a plain nn.Linear without forward
    hooks, HIP tensor on the weight's device, fp32 parameters, fp32 / 16-bit input, K a multiple of 32 and -- for the
    input gradient -- N too); otherwise the torch op (a replaced / hooked submodule keeps its own forward)."""
    ok = (type(lin) is nn.Linear and not lin._forward_hooks and not lin._forward_pre_hooks
          and x.is_cuda and lin.weight.device == x.device and lin.weight.dtype == torch.float32
          and x.dtype in (torch.float32, torch.bfloat16, torch.float16)       # (float64 bags keep torch's arithmetic)
          and lin.in_features % 32 == 0 and (not x.requires_grad or lin.out_features % 32 == 0) and x.numel() > 0)
    if not ok:
        return lin(x)
    y = _LibLinear.apply(x.reshape(-1, x.shape[-1]), lin.weight, lin.bias, compute)
    return y.reshape(*x.shape[:-1], lin.out_features)


class _AttnPool(torch.autograd.Function):
    """The pooling of DAttention behind its first Linear + activation (modules/datten.py:28-38, :69-83) as ONE library
    call each way: scores s_n = c_w . h_n + c_b (h = hid_a, or hid_a * hid_b when gated), attn = softmax over the bag,
    pooled = sum_n attn_n y_n  (rrt_attn_pool_f32: the online-softmax chunk kernels of the inference path), and the
    adjoint (rrt_attn_pool_backward_f32: dy, d hid_a, d hid_b, d c_w, d c_b in one pass over y).  Returns
    (pooled [dim], attn [N] normalised, a_raw [N] scores); gradients flowing into the returned attention / raw scores
    are honoured."""

    @staticmethod
    def forward(ctx, y2d, hid_a, hid_b, c_w, c_b):
        lib = _lib.load()
        y2d, hid_a = y2d.float().contiguous(), hid_a.float().contiguous()
        hid_b = hid_b.float().contiguous() if hid_b is not None else None
        n, d = y2d.shape
        hdim = hid_a.shape[1]
        dev = y2d.device
        pooled = torch.empty(d, dtype=torch.float32, device=dev)
        attn = torch.empty(n, dtype=torch.float32, device=dev)
        a_raw = torch.empty(n, dtype=torch.float32, device=dev)
        need = C.c_size_t()
        _lib.check(lib.rrt_attn_pool_workspace_size(n, d, hdim, C.byref(need)), "rrt_attn_pool_workspace_size")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        cw = c_w.reshape(-1).contiguous()
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.rrt_attn_pool_f32(y2d.data_ptr(), hid_a.data_ptr(), hid_b.data_ptr() if hid_b is not None else None,
                                             cw.data_ptr(), c_b.data_ptr() if c_b is not None else None, pooled.data_ptr(),
                                             attn.data_ptr(), a_raw.data_ptr(), n, d, hdim, ws.data_ptr(), ws.numel(), st),
                       "rrt_attn_pool_f32")
        ctx.save_for_backward(y2d, hid_a, hid_b, cw, attn, pooled)
        ctx.has_bias, ctx.cw_shape = c_b is not None, c_w.shape
        ctx.set_materialize_grads(False)
        return pooled, attn, a_raw

    @staticmethod
    def backward(ctx, d_pooled, d_attn, d_raw):
        lib = _lib.load()
        y2d, hid_a, hid_b, cw, attn, pooled = ctx.saved_tensors
        n, d = y2d.shape
        hdim = hid_a.shape[1]
        dev = y2d.device
        d_pooled = (torch.zeros(d, dtype=torch.float32, device=dev) if d_pooled is None else d_pooled.float().contiguous())
        c_ext = None
        if d_attn is not None:
            d_attn = d_attn.float().contiguous()
            c_ext = (attn * d_attn).sum().reshape(1)           # sum_n attn_n d_attn_n (the softmax adjoint's constant)
        if d_raw is not None:
            d_raw = d_raw.float().contiguous()
        dy = torch.empty_like(y2d)
        dha = torch.empty_like(hid_a)
        dhb = torch.empty_like(hid_b) if hid_b is not None else None
        dwcb = torch.empty(hdim + 4, dtype=torch.float32, device=dev)
        need = C.c_size_t()
        _lib.check(lib.rrt_attn_pool_workspace_size(n, d, hdim, C.byref(need)), "rrt_attn_pool_workspace_size")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.rrt_attn_pool_backward_f32(y2d.data_ptr(), hid_a.data_ptr(), p(hid_b), cw.data_ptr(), attn.data_ptr(),
                                                      pooled.data_ptr(), d_pooled.data_ptr(), p(d_attn), p(d_raw), p(c_ext),
                                                      dy.data_ptr(), dha.data_ptr(), p(dhb), dwcb.data_ptr(), n, d, hdim,
                                                      ws.data_ptr(), ws.numel(), st), "rrt_attn_pool_backward_f32")
        dcw = dwcb[:hdim].reshape(ctx.cw_shape)
        dcb = dwcb[hdim:hdim + 1].clone() if ctx.has_bias else None
        return dy, dha, dhb, dcw, dcb


def lib_attn_pool(x, hid_a, hid_b, lin_c):
    """(1, N, D) bag x and its hidden rows (1, N, H) through the HIP pooling when it applies (one bag, a plain final
    nn.Linear(H, 1), H % 4 == 0, D % 32 == 0, fp32); None otherwise (the caller keeps the torch ops)."""
    ok = (type(lin_c) is nn.Linear and lin_c.out_features == 1 and not lin_c._forward_hooks and not lin_c._forward_pre_hooks
          and x.is_cuda and x.dim() == 3 and x.size(0) == 1 and x.dtype == torch.float32 and x.size(1) > 0
          and lin_c.weight.device == x.device and lin_c.weight.dtype == torch.float32
          and hid_a.shape[-1] % 4 == 0 and x.shape[-1] % 32 == 0 and x.size(1) <= 1000000)
    if not ok:
        return None
    pooled, attn, a_raw = _AttnPool.apply(x[0], hid_a[0], hid_b[0] if hid_b is not None else None, lin_c.weight, lin_c.bias)
    return pooled.unsqueeze(0), attn.unsqueeze(0), a_raw.unsqueeze(0)


class Attention(nn.Module):
    """modules/datten.py:5-38."""

    def __init__(self, input_dim=512, act='relu', bias=False, dropout=False):
        super().__init__()
        self.L, self.D, self.K = input_dim, 128, 1
        layers = [nn.Linear(self.L, self.D, bias=bias)]
        if _act(act) is not None:
            layers.append(_act(act)())
        if dropout:
            layers.append(nn.Dropout(0.25))
        layers.append(nn.Linear(self.D, self.K, bias=bias))
        self.attention = nn.Sequential(*layers)

    def forward(self, x, no_norm=False):
        a = self.attention(x).transpose(-1, -2)          # K x N
        a_raw = a.clone()
        a = F.softmax(a, dim=-1)                          # over the N patches
        x = torch.matmul(a, x)
        return (x, a_raw) if no_norm else (x, a)


class AttentionGated(nn.Module):
    """modules/datten.py:40-83."""

    def __init__(self, input_dim=512, act='relu', bias=False, dropout=False):
        super().__init__()
        self.L, self.D, self.K = input_dim, 128, 1
        a = [nn.Linear(self.L, self.D, bias=bias)]
        if _act(act) is not None:
            a.append(_act(act)())
        b = [nn.Linear(self.L, self.D, bias=bias), nn.Sigmoid()]
        if dropout:
            a.append(nn.Dropout(0.25))
            b.append(nn.Dropout(0.25))
        self.attention_a = nn.Sequential(*a)
        self.attention_b = nn.Sequential(*b)
        self.attention_c = nn.Linear(self.D, self.K, bias=bias)

    def forward(self, x, no_norm=False):
        a = self.attention_c(self.attention_a(x).mul(self.attention_b(x))).transpose(-1, -2)
        a_raw = a.clone()
        a = F.softmax(a, dim=-1)
        x = torch.matmul(a, x)
        return (x, a_raw) if no_norm else (x, a)


class DAttention(nn.Module):
    """modules/datten.py:85-101."""

    def __init__(self, input_dim=512, act='relu', gated=False, bias=False, dropout=False):
        super().__init__()
        self.gated = gated
        self.attention = (AttentionGated if gated else Attention)(input_dim, act, bias, dropout)

    def forward(self, x, return_attn=False, no_norm=False, **kwargs):
        x, attn = self.attention(x, no_norm)
        return (x.squeeze(1), attn.squeeze(1)) if return_attn else x.squeeze(1)


class RRTMIL(nn.Module):
    def __init__(self, input_dim=1024, mlp_dim=512, act='relu', n_classes=2, dropout=0.25, pos_pos=0,
                 pos='none', peg_k=7, attn='rmsa', pool='attn', region_num=8, n_layers=2, n_heads=8,
                 drop_path=0., da_act='relu', trans_dropout=0.1, ffn=False, ffn_act='gelu', mlp_ratio=4.,
                 da_gated=False, da_bias=False, da_dropout=False, trans_dim=64, epeg=True,
                 min_region_num=0, qkv_bias=True, **kwargs):
        super().__init__()
        if pool != 'attn':
            raise NotImplementedError("pool='attn' only (the reference's avg-pool branch pools the wrong axis)")
        emb = [nn.Linear(input_dim, 512)]
        if act.lower() == 'relu':
            emb.append(nn.ReLU())
        elif act.lower() == 'gelu':
            emb.append(nn.GELU())
        self.dp = nn.Dropout(dropout) if dropout > 0. else nn.Identity()
        self.patch_to_emb = nn.Sequential(*emb)
        self.online_encoder = RRTEncoder(
            mlp_dim=mlp_dim, pos_pos=pos_pos, pos=pos, peg_k=peg_k, attn=attn, region_num=region_num,
            n_layers=n_layers, n_heads=n_heads, drop_path=drop_path, drop_out=trans_dropout, ffn=ffn,
            ffn_act=ffn_act, mlp_ratio=mlp_ratio, trans_dim=trans_dim, epeg=epeg,
            min_region_num=min_region_num, qkv_bias=qkv_bias, **kwargs)
        self.pool_fn = DAttention(self.online_encoder.final_dim, da_act, gated=da_gated, bias=da_bias,
                                  dropout=da_dropout)
        self.predictor = nn.Linear(self.online_encoder.final_dim, n_classes)
        if self.patch_to_emb[0].out_features != self.online_encoder.final_dim:
            # the reference hard-codes Linear(input_dim, 512) in front of an encoder of width mlp_dim
            # (rrt.py:208,219) and dies with a shape error in the first LayerNorm otherwise
            raise ValueError(f"patch_to_emb emits 512 features but the encoder was built with mlp_dim="
                             f"{self.online_encoder.final_dim} (the reference fails on this combination too)")
        self.apply(initialize_weights)

        self._act_name = act.lower() if act.lower() in ("relu", "gelu") else "none"
        self._da_act = da_act
        self._ws = None

    def __getstate__(self):          # device workspaces and their validity keys stay with the process (deepcopy / pickle)
        st = dict(self.__dict__)
        for k in ("_ws", "_slots", "_w16_key"):
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.__dict__.setdefault("_ws", None)

    # ------------------------------------------------------------------ the one-call HIP path
    def _mil_desc(self, input_dim, solo=True):
        enc = self.online_encoder
        d = _lib.MilDesc()
        C.memmove(C.byref(d.enc), C.byref(enc._desc), C.sizeof(_lib.EncoderDesc))
        d.enc.compute = enc._compute_mode()
        # per-call scheduling hint, never inherited from whatever the encoder's last call left in its descriptor
        # (rrt_encoder_desc.solo also picks between CR-MSA fronts that differ in summation order, ~1e-7: a classifier's logits
        # must not depend on call history): 1 = this slide has the GPU to itself (forward_bag on its own, the module call),
        # 0 = it shares it with the other slides of a forward_bags call -- what the encoder's executor does with
        # solo = (n_streams == 1).  Since round 6 the hint decides more than a microsecond (fp32: the front as 64 blocks that
        # leave the chip to the other slides; 16-bit: the out-projection's tile shape), in both directions.
        d.enc.solo = int(bool(solo))
        d.input_dim = input_dim
        d.emb_act = _lib.ACT_BY_NAME.get(self._act_name, _lib.ACT_NONE)
        d.n_classes = self.predictor.out_features
        att = self.pool_fn.attention
        d.pool_hidden = att.D
        d.pool_act = _lib.ACT_BY_NAME.get(self._da_act, _lib.ACT_NONE)
        d.pool_gated = int(self.pool_fn.gated)
        return d

    def _mil_weights(self):
        enc, p = self.online_encoder, RRTEncoder._ptr
        w = _lib.MilWeights()
        w.enc = enc._weights()
        lin = self.patch_to_emb[0]
        w.emb_w, w.emb_b = p(lin.weight), p(lin.bias)
        att = self.pool_fn.attention
        if self.pool_fn.gated:
            a, b, c = att.attention_a[0], att.attention_b[0], att.attention_c
            w.pool_b_w, w.pool_b_b = p(b.weight), p(b.bias)
        else:
            a, c = att.attention[0], att.attention[-1]
        w.pool_a_w, w.pool_a_b = p(a.weight), p(a.bias)
        w.pool_c_w, w.pool_c_b = p(c.weight), p(c.bias)
        w.pred_w, w.pred_b = p(self.predictor.weight), p(self.predictor.bias)
        return w

    @torch.no_grad()
    def forward_bags(self, bags, streams=4, return_attn=False, no_norm=False):
        """A batch of independent slides (each (N_i, input_dim) or (1, N_i, input_dim), any mix of sizes) -> list of logits
        [(n_classes,) or (1, n_classes)] (and attention rows with return_attn): the reference's validation loop
        ``for bag in loader: model(bag)`` (main.py:466-467) with ``streams`` slides in flight -- each slide is one
        rrt_mil_forward_f32 call with its own workspace on one of the process's bag streams (the same list
        RRTEncoder.forward_bags uses; at most four: the chip schedules four hardware queues).  Ordered like one op of the
        caller's stream: the bag streams wait for it, and the host waits for them before the call returns (the call BLOCKS
        the host; it swaps the module's workspace per stream slot while it runs, so one call at a time per module: not
        re-entrant, not thread-safe)."""
        if not bags:
            return []
        if self.training and (isinstance(self.dp, nn.Dropout) or self.online_encoder._stochastic()):
            return [self(b if b.dim() == 3 else b.unsqueeze(0), return_attn=return_attn, no_norm=no_norm) for b in bags]
        from .encoder import _BAG_STREAMS
        dev = bags[0].device
        if not bags[0].is_cuda:
            raise _lib.RRTHipError("rrt_mil_amd.RRTMIL runs on MI355X only; there is no CPU fallback")
        S = max(1, min(int(streams), 4, len(bags)))
        pool = _BAG_STREAMS.setdefault(dev, [])
        while len(pool) < S:
            pool.append(torch.cuda.Stream(dev))
        slots = self.__dict__.setdefault("_slots", {})
        cur = torch.cuda.current_stream(dev)
        order = sorted(range(len(bags)), key=lambda i: -bags[i].shape[-2])        # big slides first, least loaded stream
        load, outs = [0] * S, [None] * len(bags)
        for st in pool[:S]:
            st.wait_stream(cur)
        ws_was, key_was = self._ws, self.__dict__.get("_w16_key")
        try:
            for i in order:
                s_ = min(range(S), key=lambda t: load[t])
                load[s_] += bags[i].shape[-2]
                b = bags[i]
                x2 = b[0] if b.dim() == 3 else b
                self._ws, self.__dict__["_w16_key"] = slots.get((dev, s_), (None, None))
                with torch.cuda.stream(pool[s_]):
                    o = self.forward_bag(x2, return_attn=return_attn, no_norm=no_norm, solo=(S == 1))
                slots[(dev, s_)] = (self._ws, self.__dict__.get("_w16_key"))
                # the outputs were allocated under the bag stream (the caching allocator ties a block to the stream it was
                # taken on) and are consumed on the caller's: tell the allocator, so that a freed logits / attention row
                # is not handed out again on the bag stream while the caller's stream still reads it
                for t_ in (o if return_attn else (o,)):
                    t_.record_stream(cur)
                if b.dim() == 3:
                    o = tuple(t.unsqueeze(0) for t in o) if return_attn else o.unsqueeze(0)
                outs[i] = o
        finally:
            self._ws, self.__dict__["_w16_key"] = ws_was, key_was
            for st in pool[:S]:
                st.synchronize()      # host wait: nothing is parked on the caller's stream
        return outs

    def forward_bag(self, x2d, return_attn=False, no_norm=False, solo=True):
        """One bag: x2d (N, input_dim) fp32 (or bf16 / fp16) device tensor -> logits (n_classes,) [, attention (N,)].
        ``solo``: the slide has the GPU to itself (forward_bags passes False when it keeps several slides in flight)."""
        lib = _lib.load()
        if not x2d.is_cuda:
            raise _lib.RRTHipError("rrt_mil_amd.RRTMIL runs on MI355X only: move the bag to a 'cuda' (HIP) "
                                   "device; there is no CPU fallback")
        if self.training and (isinstance(self.dp, nn.Dropout) or self.online_encoder._stochastic()):
            raise NotImplementedError("forward_bag is the one-call inference entry (no dropout inside); in train() "
                                      "call the module itself: forward() applies the dropouts and records a graph")
        n, in_dim = x2d.shape
        if in_dim != self.patch_to_emb[0].in_features:
            raise ValueError(f"expected feature dim {self.patch_to_emb[0].in_features}, got {in_dim}")
        d = self._mil_desc(in_dim, solo=solo)
        if x2d.dtype in (torch.bfloat16, torch.float16):
            # 16-bit features (rrt_mil_amd.BagFeeder(dtype=...): half the PCIe bytes of a slide).  When they are of the
            # arithmetic's own type they ARE patch_to_emb's 16-bit operand -- under autocast the reference's first op rounds
            # the fp32 features to exactly these values (rrt.py:208-229), the logits are bit-identical to the fp32-fed call --
            # otherwise they are widened (exactly) and take the fp32 route
            want = {torch.bfloat16: _lib.COMPUTE_BF16, torch.float16: _lib.COMPUTE_F16}[x2d.dtype]
            if d.enc.compute == want and in_dim % 64 == 0:
                d.input16 = want
            else:
                x2d = x2d.float()
        elif x2d.dtype != torch.float32:
            raise NotImplementedError(f"unsupported bag dtype {x2d.dtype}")
        x2d = x2d.contiguous()
        w = self._mil_weights()
        need = C.c_size_t()
        _lib.check(lib.rrt_mil_workspace_size(C.byref(d), n, C.byref(need)), "rrt_mil_workspace_size")
        if self._ws is None or self._ws.device != x2d.device or self._ws.numel() < need.value:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=x2d.device)
        logits = torch.empty(d.n_classes, dtype=torch.float32, device=x2d.device)
        attn = torch.empty(n, dtype=torch.float32, device=x2d.device) if return_attn else None
        with torch.cuda.device(x2d.device):      # kernels launch on the bag's device, whatever the current one is
            stream = torch.cuda.current_stream(x2d.device).cuda_stream
            # reduced-precision modes: the encoder's 16-bit weight images inside this workspace (their place depends on
            # n) are still those of these weights?  (rrt_encoder_desc.weights16_valid, see RRTEncoder.forward_bag)
            key = (self._ws.data_ptr(), d.enc.compute, w.enc.version, stream, n)
            lowp = d.enc.compute != _lib.COMPUTE_F32
            d.enc.weights16_valid = int(lowp and key == getattr(self, "_w16_key", None))
            rc = lib.rrt_mil_forward_f32(C.byref(d), C.byref(w), x2d.data_ptr(), logits.data_ptr(),
                                         attn.data_ptr() if return_attn else None, int(bool(no_norm)), None, n,
                                         self._ws.data_ptr(), self._ws.numel(), stream)
            self._w16_key = key if rc == 0 and lowp else None
        _lib.check(rc, "rrt_mil_forward_f32")
        return (logits, attn) if return_attn else logits

    def forward(self, x, return_attn=False, no_norm=False):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # a graph is wanted (training, or eval()-mode fine-tuning / attribution): the encoder is the HIP
            # autograd Function (forward with stash + backward kernels); the thin layers around it (patch_to_emb,
            # Dropout, DAttention, predictor) are torch ops under autograd
            return self._forward_layers(x, return_attn, no_norm)
        with torch.no_grad():
            if self.training and (isinstance(self.dp, nn.Dropout) or self.online_encoder._stochastic()
                                  or any(isinstance(m, nn.Dropout) for m in self.pool_fn.modules())):
                # train() without a graph: the reference still applies its dropouts
                return self._forward_layers(x, return_attn, no_norm)
            return self._forward_infer(x, return_attn, no_norm)

    def _forward_infer(self, x, return_attn=False, no_norm=False):
        if x.dim() == 3 and x.size(0) == 1 and x.is_cuda:
            # (1, N, input_dim): the form of every reference trainer -> one library call
            out = self.forward_bag(x[0], return_attn=return_attn, no_norm=no_norm)
            if return_attn:
                return out[0].unsqueeze(0), out[1].unsqueeze(0)
            return out.unsqueeze(0)
        # other ranks: the reference's op sequence around the HIP encoder
        return self._forward_layers(x, return_attn, no_norm)

    def _forward_layers(self, x, return_attn=False, no_norm=False):
        enc = self.online_encoder
        if x.is_cuda:
            # the two bag-sized products of the classifier on the library's GEMMs (forward and backward), the
            # activation / dropout / softmax glue as torch ops under autograd
            compute = enc._compute_mode()
            compute = _lib.COMPUTE_F32 if compute == _lib.COMPUTE_F32X3 else compute
            h = lib_linear(self.patch_to_emb[0], x, compute)
            for m in list(self.patch_to_emb)[1:]:
                h = m(h)
            x = self.dp(h.float())                        # (1, N, 512)
        else:
            x = self.dp(self.patch_to_emb(x))
        x = enc(x)                                        # feature re-embedding: the HIP path
        x, a = self._pool(x, no_norm)
        logits = self.predictor(x)
        return (logits, a) if return_attn else logits

    def _pool(self, x, no_norm):
        """DAttention (modules/datten.py:28-38, :69-83) with its Linear(512, 128) layers on the library GEMMs"""
        att = self.pool_fn.attention
        if not x.is_cuda:
            return self.pool_fn(x, return_attn=True, no_norm=no_norm)
        compute = self.online_encoder._compute_mode()
        compute = _lib.COMPUTE_F32 if compute == _lib.COMPUTE_F32X3 else compute

        def seq(mods, t):
            t = lib_linear(mods[0], t, compute)
            for m in list(mods)[1:]:
                t = m(t)
            return t
        if self.pool_fn.gated:
            ha, hb, lin_c = seq(att.attention_a, x), seq(att.attention_b, x), att.attention_c
        else:
            mods = list(att.attention)
            ha, hb, lin_c = seq(mods[:-1], x), None, mods[-1]
        out = lib_attn_pool(x, ha, hb, lin_c)          # scores, softmax over the bag and the weighted sum: HIP, both ways
        if out is not None:
            pooled, a, a_raw = out
            return pooled, (a_raw if no_norm else a)
        s = lin_c(ha.mul(hb) if hb is not None else ha).transpose(-1, -2)   # K x N (batched / odd widths: torch ops)
        a_raw = s.clone()
        a = F.softmax(s, dim=-1)
        pooled = torch.matmul(a, x).squeeze(1)
        return pooled, (a_raw if no_norm else a).squeeze(1)
