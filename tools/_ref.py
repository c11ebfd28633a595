"""Container-only helper: import the real reference (/root/reference) as an oracle.

Never imported by the product, the tests or bench.py -- /root/reference does not
exist on the GPU box.  Used by tools/make_golden.py and tools/make_golden_grad_amp.py.
`timm` is absent from this image; the two symbols the reference imports from it
are stubbed (neither executes on the default path) -- SURVEY.md Appendix A.
"""
import sys
import types

import torch


def load_reference(path="/root/reference"):
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")

        class DropPath(torch.nn.Module):
            def __init__(self, p=0.0):
                super().__init__()
                self.p = p

            def forward(self, x):
                return x

        layers.DropPath = DropPath
        layers.trunc_normal_ = torch.nn.init.trunc_normal_
        timm.models = models
        models.layers = layers
        sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    if path not in sys.path:
        sys.path.insert(0, path)
    from modules.rrt import RRTEncoder, RRTMIL  # noqa: E402
    return RRTEncoder, RRTMIL


def build_reference_encoder(state_np, **cfg):
    """Reference RRTEncoder in eval mode carrying the given numpy state (strict load)."""
    RRTEncoder, _ = load_reference()
    enc = RRTEncoder(**cfg).eval()
    sd = {k: torch.from_numpy(v.copy()) for k, v in state_np.items()}
    enc.load_state_dict(sd, strict=True)
    return enc
