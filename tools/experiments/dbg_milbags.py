import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rrt_mil_amd import synth
from rrt_mil_amd.mil import RRTMIL
torch.manual_seed(3)
mil = RRTMIL(input_dim=256, n_classes=3, da_gated=True, dropout=0.25).to("cuda:0").eval()
mil.online_encoder.compute_dtype = torch.bfloat16
sizes = [3000, 700, 5000, 1, 4096, 2200, 3000]
bags = [torch.from_numpy(synth.bag(n, 256, tag=f"milbags/{i}", nonneg=True)).to("cuda:0") for i, n in enumerate(sizes)]
with torch.no_grad():
    ref = [mil(b.unsqueeze(0), return_attn=True) for b in bags]
    ref2 = [mil(b.unsqueeze(0), return_attn=True) for b in bags]
    for S in (1, 2, 4):
        outs = mil.forward_bags(bags, streams=S, return_attn=True)
        torch.cuda.synchronize()
        for i, ((lg, at), (rl, ra), (rl2, ra2)) in enumerate(zip(outs, ref, ref2)):
            print(f"S={S} bag {i} N={sizes[i]}: seq-vs-seq logits {torch.equal(rl, rl2)} attn {torch.equal(ra, ra2)} | bags-vs-seq logits {torch.equal(lg, rl[0])} "
                  f"({(lg - rl[0]).abs().max().item():.2e}) attn {torch.equal(at, ra[0])} ({(at - ra[0]).abs().max().item():.2e})")
enc = mil.online_encoder
x = torch.randn(5000, 512, device="cuda:0")
with torch.no_grad():
    a = enc(x.unsqueeze(0)); b = enc(x.unsqueeze(0))
    outs = enc.forward_bags([x, x, x, x, x, x, x, x], streams=4)
    torch.cuda.synchronize()
    print("encoder seq-vs-seq", torch.equal(a, b), [torch.equal(o, a[0]) for o in outs], [(o - a[0]).abs().max().item() for o in outs])
