import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rrt_mil_amd import RRTEncoder, synth
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
S = int(sys.argv[4]) if len(sys.argv) > 4 else 4
cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
enc = RRTEncoder(**cfg).eval()
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.encoder_state(**cfg).items()}, strict=True)
enc = enc.to("cuda:0")
if dt == "bf16":
    enc.compute_dtype = torch.bfloat16
enc.solo = False
x = torch.randn(N, 512, device="cuda:0")
with torch.no_grad():
    a = enc(x.unsqueeze(0)); b = enc(x.unsqueeze(0))
    for rep in range(3):
        outs = enc.forward_bags([x] * nb, streams=S)
        torch.cuda.synchronize()
        print(dt, N, nb, S, "seq-vs-seq", torch.equal(a, b), ["ok" if torch.equal(o, a[0]) else f"{(o - a[0]).abs().max().item():.1e}" for o in outs])
