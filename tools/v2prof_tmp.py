import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from sad_amd import synth, transfer_models
dev = torch.device("cuda:0")
w = dict(bench.WORKLOADS["cfg3"])
S, lengths, ln_dev, mask, labels = bench.make_inputs(w, 1234, dev)
cell, bi, layers = "LSTM", True, 2
cfg = SimpleNamespace(cell=cell, enc_bidirectional=bi, pad_id=0, dropout=0.0, enc_num_layers=layers, hidden_size=w["H"],
                      vocab_size=w["V"], emb_size=w["E"], num_labels=w["L"], freeze_embedding=False, device="cuda:0")
m = transfer_models.SelfAttnRNNClassifier(cfg)
p = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21, cell=cell, bidirectional=bi, layers=layers)
m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
m.to(dev)
for q in m.parameters():
    q.requires_grad = False
Sv = S.detach().transpose(0, 1).requires_grad_(True)
import numpy as np
len16 = torch.from_numpy(lengths.astype(np.int16))
for _ in range(8):
    Sv.grad = None
    logits, _ = m(Sv, len16, mask.clone(), soft=True)
    loss = torch.nn.functional.cross_entropy(logits, 1 - labels, reduction="none")
    (loss.sum() / w["B"]).backward()
torch.cuda.synchronize()
