"""Phase timing of the persistent LSTM forward (debug variant library only: SAD_LIB=.../prof.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sad_amd import functional as Fn, lib, synth
import bench
dev = torch.device("cuda:0")
w = dict(bench.WORKLOADS["cfg3"])
params = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21)
pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
S, lengths, ln_dev, mask, labels = bench.make_inputs(w, 1234, dev)
Sv = S.transpose(0, 1).requires_grad_(True)
L = lib.load()
L.sad_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
for it in range(3):
    loss, _, _ = Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw)
    torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
L.sad_debug_prof(out, 0)
v = list(out)
n = max(1, v[6])
names = ["poll", "barrier", "A loads land", "K loop", "epilogue+h store+drain", "barrier+signal+G/Cs stores"]
print("steps", v[6], " cycles/step (100 MHz ticks if constant clock):")
for nm, x in zip(names, v[:6]):
    print("  %-28s %8.0f" % (nm, x / n))
print("  total %.0f" % (sum(v[:6]) / n))
