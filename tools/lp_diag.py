"""Diagnostic: persistent vs per-step LSTM forward; run-to-run determinism of the persistent path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sad_amd import functional as Fn, synth
Fn.KEEP_LAST = True
dev = torch.device("cuda:0")
B, T, V, E, H = 1024, 32, 500, 300, 512
lengths = synth.make_lengths(B, T, seed=B)
params = synth.make_params(V, E, H, seed=3)
pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
ids = torch.from_numpy(synth.make_token_ids(B, T, V, lengths, seed=5)).to(dev)
ln = torch.from_numpy(lengths)
mask0 = torch.from_numpy(synth.transfer_mask(lengths, T)).to(dev)
lab = torch.from_numpy(synth.make_labels(B, seed=1)).to(dev)
def run(no_persist):
    os.environ["SAD_NO_PERSIST"] = "1" if no_persist else "0"
    loss, logits, weights = Fn.discriminator_loss(ids, ln, mask0.clone(), lab, pw)
    ws = Fn._Discriminator.last_workspace
    torch.cuda.synchronize()
    live = (ws.view("NEED", torch.uint8).reshape(T, B) & 1).bool()
    hs = ws.view("HS", torch.bfloat16).reshape(T, B, 2, -1).float().clone()
    cs = ws.view("CS", torch.float32).reshape(T, B, 2, -1).clone()
    hs[~live] = 0; cs[~live] = 0
    return logits.clone(), hs, cs
ref = run(True)
ref2 = run(True)
print("per-step determinism: Hs equal", torch.equal(ref[1], ref2[1]), "Cs equal", torch.equal(ref[2], ref2[2]))
p1 = run(False); p2 = run(False)
print("persist determinism: Hs equal", torch.equal(p1[1], p2[1]), "Cs equal", torch.equal(p1[2], p2[2]))
for name, a, b in (("Hs", p1[1], ref[1]), ("Cs", p1[2], ref[2]), ("logits", p1[0], ref[0])):
    d = (a - b).abs()
    print(name, "max abs diff %.3e" % d.max().item(), "frac differing %.4f" % (d > 0).float().mean().item(), "max |ref| %.3f" % b.abs().max().item())
d = (p1[2] - ref[2]).abs()          # [T,B,2,Hp]
for dirn in (0, 1):
    per_t = d[:, :, dirn].amax(dim=(1, 2)).cpu().numpy()
    print("dir", dirn, "Cs max diff per t:", np.array2string(per_t, precision=1, max_line_width=200))
