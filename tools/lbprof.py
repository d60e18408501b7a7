"""Phase timing of the persistent BPTT (-DSAD_LB_PROF=<group>) or forward recurrence (-DSAD_LF_PROF=<group>;
LB_FWD=1) (debug variant library only,
loaded through SAD_LIB).  Prints 100 MHz wall-clock ticks (10 ns) per step of one workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sad_amd import functional as Fn, synth
Fn.KEEP_LAST = True
import bench
dev = torch.device("cuda:0")
w = dict(bench.WORKLOADS["cfg3"])
if os.environ.get("LB_B"):
    w["B"] = int(os.environ["LB_B"])
params = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21)
pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
S, lengths, ln_dev, mask, labels = bench.make_inputs(w, 1234, dev)
Sv = S.transpose(0, 1).requires_grad_(True)
for it in range(3):
    loss, _, _ = Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw)
    if not os.environ.get("LB_FWD"):
        loss.mean().backward()
    torch.cuda.synchronize()
ws = Fn._Discriminator.last_workspace
v = ws.view("SYNC", torch.int64)[-12:-2].cpu().tolist()
n = max(1, v[6])
names = ["poll", "barrier", "operand loads + K loop", "epilogue + store issue", "store drain", "barrier + signal (+ private stores, fwd)"]
print("steps", v[6], " us/step:")
for nm, x in zip(names, v[:6]):
    print("  %-26s %7.2f" % (nm, x / n / 100.0))
print("  total %.2f" % (sum(v[:6]) / n / 100.0))
if not os.environ.get("LB_FWD"):
    print("  prologue %.2f us, whole kernel (this workgroup) %.2f us, between steps %.2f us/step" % (v[7] / 100.0, v[8] / 100.0, v[9] / n / 100.0))
