"""Phase timing of the K1 forward roles (debug variant library: -DSAD_K1F_PROF=<block>, via SAD_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sad_amd import functional as Fn, synth
import bench
dev = torch.device("cuda:0")
w = dict(bench.WORKLOADS["cfg3"])
params = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21)
pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
S, lengths, ln_dev, mask, labels = bench.make_inputs(w, 1234, dev)
Sv = S.transpose(0, 1)
bwd = bool(os.environ.get("K1_BWD"))
if bwd:
    Sv.requires_grad_(True)
for it in range(2):
    if bwd:
        loss, _, _ = Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw)
        loss.mean().backward()
    else:
        with torch.no_grad():
            Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw)
    torch.cuda.synchronize()
    print("----")
