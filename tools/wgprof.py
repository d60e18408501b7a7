"""Phase timing of the weight-gradient core (debug variant library only: SAD_LIB=.../prof.so)."""
import ctypes, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sad_amd import functional as Fn, lib, synth
import bench
dev = torch.device("cuda:0")
w = dict(bench.WORKLOADS["cfg3"])
params = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21)
live = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in params.items() if k in Fn._LIVE_PARAMS}
pw = Fn.pack_weights(live)
S, lengths, ln_dev, mask, labels = bench.make_inputs(w, 1234, dev)
Sv = S.transpose(0, 1).requires_grad_(True)
L = lib.load()
L.sad_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = {"emb": ("embedding.weight",), "cat": ("rnn.weight_ih_l0", "rnn.weight_ih_l0_reverse", "linear_emb.weight"),
         "hh": ("rnn.weight_hh_l0", "rnn.weight_hh_l0_reverse")}
for tag, keep in names.items():
    for k, p in live.items():
        p.requires_grad_(k in keep)
        p.grad = None
    out = (ctypes.c_ulonglong * 8)()
    for it in range(3):
        L.sad_debug_prof(None, 1)
        loss, _, _ = Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw, params=live, pad_id=0)
        loss.mean().backward()
        torch.cuda.synchronize()
    L.sad_debug_prof(out, 0)
    v = list(out)
    nk = max(1, v[4])
    print(tag, "tiles", v[4], "cycles/tile: to_lds(+wait loads) %.0f | issue+flags+probe %.0f | mfma %.0f | barrier %.0f" % tuple(x / nk for x in v[:4]))
