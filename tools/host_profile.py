#!/usr/bin/env python3
"""Host-side cost of one discriminator step through the drop-in module (B=64, the transfer loop's call
pattern): cProfile over N steps.  At small batches the step is host-bound; this shows where.
usage: python3 tools/host_profile.py [--steps 300]"""
import argparse, cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
import bench
from sad_amd import synth, transfer_models

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=300); a = ap.parse_args()
dev = torch.device("cuda:0")
w = dict(bench.WORKLOADS["yelp64"])
S, lengths, ln_dev, mask, labels = bench.make_inputs(w, 1234, dev, yelp_lengths=True)
params = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21)
cfg = SimpleNamespace(cell="LSTM", enc_bidirectional=True, pad_id=0, dropout=0.0, enc_num_layers=1, hidden_size=w["H"],
                      vocab_size=w["V"], emb_size=w["E"], num_labels=w["L"], freeze_embedding=False, device=str(dev), mask_id=None)
model = transfer_models.SelfAttnRNNClassifier(cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model.to(dev)
for p_ in model.parameters():
    p_.requires_grad = False
len16 = torch.from_numpy(lengths.astype(np.int16))
inv = 1 - labels
Sv = S.transpose(0, 1).requires_grad_(True)
B = w["B"]
def step():
    Sv.grad = None
    logits, _ = model(Sv, len16, mask.clone(), soft=True)
    loss = torch.nn.functional.cross_entropy(logits, inv, reduction="none")
    (loss.sum() / B).backward()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(a.steps): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
