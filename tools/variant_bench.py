"""SURVEY.md §8 f4 variants at the cfg3 shape (frozen discriminator, dense bf16 soft tokens, fwd + bwd to dS):
ms per step of the drop-in module for every cell / direction / depth the reference constructor accepts."""
import json, os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sad_amd import synth, transfer_models

dev = torch.device("cuda:0")
w = dict(bench.WORKLOADS["cfg3"])
S, lengths, ln_dev, mask, labels = bench.make_inputs(w, 1234, dev)
for cell, bi, layers in (("LSTM", True, 1), ("GRU", True, 1), ("LSTM", False, 1), ("GRU", False, 1), ("LSTM", True, 2), ("GRU", True, 2)):
    cfg = SimpleNamespace(cell=cell, enc_bidirectional=bi, pad_id=0, dropout=0.0, enc_num_layers=layers, hidden_size=w["H"],
                          vocab_size=w["V"], emb_size=w["E"], num_labels=w["L"], freeze_embedding=False, device="cuda:0")
    m = transfer_models.SelfAttnRNNClassifier(cfg)
    p = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21, cell=cell, bidirectional=bi, layers=layers)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    m.to(dev)
    for q in m.parameters():
        q.requires_grad = False
    Sv = S.detach().transpose(0, 1).requires_grad_(True)

    def step():
        Sv.grad = None
        loss, _, _ = m.loss(Sv, ln_dev, mask.clone(), labels)
        loss.mean().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(json.dumps({"variant": "%s %s %d-layer" % (cell, "bi" if bi else "uni", layers), "workload": "cfg3",
                      "ms_per_step": round(ms, 3), "sentences_per_s": round(w["B"] / ms * 1e3)}))
