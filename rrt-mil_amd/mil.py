"""RRTMIL -- the reference's R2T-MIL slide classifier around the MI355X encoder.

Mirrors modules/rrt.py:204-246 (RRTMIL) and modules/datten.py:5-101 (ABMIL attention
pooling): ``patch_to_emb`` Linear(input_dim, 512)+act -> Dropout -> RRTEncoder (the HIP
path) -> DAttention pooling -> predictor.  Same constructor, parameter names and
forward signature, so reference checkpoints load with strict=True.

Row f1 of SURVEY.md §8: the prologue / pooling / predictor are the *callers* of the hot
path; here they are plain PyTorch-ROCm ops (a [N,in]x[in,512] GEMM and an N x 128 GEMV
chain), not yet fused into the HIP kernels.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .encoder import RRTEncoder, initialize_weights


def _act(name):
    return {"gelu": nn.GELU, "relu": nn.ReLU, "tanh": nn.Tanh}.get(name)


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
        self.apply(initialize_weights)

    @torch.no_grad()
    def forward(self, x, return_attn=False, no_norm=False):
        x = self.dp(self.patch_to_emb(x))                 # (1, N, 512)
        x = self.online_encoder(x)                        # feature re-embedding: the HIP path
        if return_attn:
            x, a = self.pool_fn(x, return_attn=True, no_norm=no_norm)
        else:
            x = self.pool_fn(x)
        logits = self.predictor(x)
        return (logits, a) if return_attn else logits
