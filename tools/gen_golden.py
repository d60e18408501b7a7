#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Build-container only: imports ``/root/reference/transfer/models.py`` and
``/root/reference/classifier/models.py`` unmodified (pure Python/PyTorch, CPU), feeds them weights
and inputs from this repo's seeded numpy generators (``sad_amd.synth``) and stores inputs +
expected outputs as small ``.npz`` files.  The reference cannot travel to the GPU box; these
vectors are what pins the oracle (``oracle/``) and, through it, the HIP path.

    python tools/gen_golden.py            # rewrites tests/golden/*.npz

Fixtures are data only (arrays + a json header with torch/numpy versions and the seeds).
"""
import argparse
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("SAD_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sad_amd import synth  # noqa: E402


def ref_config(V, E, H, L=2, dropout=0.0, cell="LSTM", bidirectional=True, layers=1):
    return SimpleNamespace(cell=cell, enc_bidirectional=bidirectional, pad_id=0, dropout=dropout,
                           enc_num_layers=layers, hidden_size=H, vocab_size=V, emb_size=E,
                           num_labels=L, freeze_embedding=False, device="cpu", mask_id=None)


def build_ref(which, V, E, H, params, dtype, dropout=0.0, **variant):
    import importlib
    mod = importlib.import_module("%s.models" % which)   # /root/reference/{transfer,classifier}/models.py
    m = mod.SelfAttnRNNClassifier(ref_config(V, E, H, dropout=dropout, **variant))
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    assert list(m.state_dict().keys()) == list(params.keys()), "state_dict layout drifted"
    m = m.to(dtype)          # cast first: load_state_dict copies INTO the existing dtype
    m.load_state_dict(sd)
    m.eval()   # classifier copy: dropout off (p=0 anyway)
    return m


def run_transfer_copy(m, src, lengths, mask, labels, soft, dtype, len_dtype=torch.int16):
    """Calls the reference forward exactly as transfer/models.py:1185-1187 does, then backward of
    mean per-sentence CE.  Returns dict of numpy outputs."""
    B = len(labels)
    if soft:
        S = torch.from_numpy(src).to(dtype).requires_grad_(True)   # [T,B,V] time-major
        arg = S.transpose(0, 1)
    else:
        S = None
        arg = torch.from_numpy(src)
    mask_t = torch.from_numpy(mask.copy())
    ln = torch.from_numpy(np.asarray(lengths)).to(len_dtype)
    logits, weights = m(arg, ln, mask_t, soft=soft)
    loss = F.cross_entropy(logits, torch.from_numpy(labels), reduction="none")
    for p in m.parameters():
        p.grad = None
    (loss.sum() / B).backward()
    out = dict(logits=logits.detach().numpy(), weights=weights.detach().numpy(),
               loss=loss.detach().numpy(), mask_after=mask_t.numpy())
    if soft:
        out["dS"] = S.grad.numpy()    # [T,B,V]
    grads = {}
    none_grads = []
    for k, p in m.named_parameters():
        if p.grad is None:
            none_grads.append(k)
        else:
            grads[k] = p.grad.numpy().copy()
    out["grads"] = grads
    out["none_grads"] = none_grads
    return out


def save(path, header, **arrays):
    arrays["header_json"] = np.frombuffer(json.dumps(header).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrays)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def versions():
    return dict(torch=torch.__version__, numpy=np.__version__, reference=REF)


def tiny_cases(outdir):
    V, E, H, B, T = 41, 10, 8, 6, 9
    lengths = np.array([9, 7, 5, 3, 2, 2], dtype=np.int32)
    params = synth.make_params(V, E, H, seed=11, dtype=np.float64)
    labels = synth.make_labels(B, seed=3)
    mask = synth.transfer_mask(lengths, T)
    for kind in ("dense", "onehot"):
        S = synth.make_soft_tokens(B, T, V, lengths, seed=5, kind=kind, dtype=np.float64)
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            p = {k: v.astype(np.float64 if tag == "f64" else np.float32) for k, v in params.items()}
            m = build_ref("transfer", V, E, H, p, dtype)
            o = run_transfer_copy(m, S, lengths, mask, labels, True, dtype)
            hdr = dict(case="tiny_soft_%s_%s" % (kind, tag), V=V, E=E, H=H, B=B, T=T, L=2,
                       param_seed=11, s_seed=5, label_seed=3, kind=kind,
                       none_grads=o["none_grads"], **versions())
            arrs = dict(S=S, lengths=lengths, mask=mask, labels=labels, logits=o["logits"],
                        weights=o["weights"], loss=o["loss"], mask_after=o["mask_after"], dS=o["dS"])
            for k, g in o["grads"].items():
                arrs["grad:" + k] = g
            for k, v in p.items():
                arrs["param:" + k] = v
            save(os.path.join(outdir, "tiny_soft_%s_%s.npz" % (kind, tag)), hdr, **arrs)

    # hard tokens, both copies
    ids = synth.make_token_ids(B, T, V, lengths, seed=9)
    p64 = params
    mt = build_ref("transfer", V, E, H, p64, torch.float64)
    o = run_transfer_copy(mt, ids, lengths, mask, labels, False, torch.float64, len_dtype=torch.int32)
    arrs = dict(ids=ids, lengths=lengths, mask=mask, labels=labels, logits=o["logits"],
                weights=o["weights"], loss=o["loss"], mask_after=o["mask_after"])
    for k, g in o["grads"].items():
        arrs["grad:" + k] = g
    # classifier copy takes a PAD mask and derives [True | pad[:,2:] | True] itself; no guard.
    # Only rows with len >= 3 are well defined there (len 2 rows give NaN softmax), so use them.
    keep = lengths >= 3
    mc = build_ref("classifier", V, E, H, p64, torch.float64)
    pad_mask = torch.from_numpy(ids[keep] == 0)
    lg, w = mc(torch.from_numpy(ids[keep]), torch.from_numpy(lengths[keep]), pad_mask)
    arrs.update(cls_keep=keep, cls_logits=lg.detach().numpy(), cls_weights=w.detach().numpy())
    # get_masked_src post-processing (classifier/models.py:269-299)
    mc.set_mask_id(7)
    lg2, al, style_src, content_src = mc.get_masked_src(
        torch.from_numpy(ids[keep]), torch.from_numpy(lengths[keep]), pad_mask)
    arrs.update(gms_style_src=style_src.numpy(), gms_content_src=content_src.numpy())
    for k, v in p64.items():
        arrs["param:" + k] = v
    hdr = dict(case="tiny_hard_f64", V=V, E=E, H=H, B=B, T=T, L=2, param_seed=11, id_seed=9,
               label_seed=3, mask_id=7, none_grads=o["none_grads"], **versions())
    save(os.path.join(outdir, "tiny_hard_f64.npz"), hdr, **arrs)


def dropout_case(outdir):
    """Classifier copy in TRAINING mode with dropout p=0.3 on the embedded tokens
    (classifier/models.py:217).  nn.Dropout draws `empty_like(x).bernoulli_(1-p)` from the global
    generator as its first random draw, so the same seed reproduces the keep mask, which is stored
    with the outputs and the parameter gradients of the mean CE loss."""
    V, E, H, B, T, p_drop = 41, 10, 8, 5, 9, 0.3
    lengths = np.array([9, 7, 5, 4, 3], dtype=np.int32)
    params = synth.make_params(V, E, H, seed=11, dtype=np.float64)
    labels = synth.make_labels(B, seed=3)
    ids = synth.make_token_ids(B, T, V, lengths, seed=9)
    m = build_ref("classifier", V, E, H, params, torch.float64, dropout=p_drop)
    m.train()
    pad_mask = torch.from_numpy(ids == 0)
    torch.manual_seed(4321)
    logits, weights = m(torch.from_numpy(ids), torch.from_numpy(lengths), pad_mask)
    loss = F.cross_entropy(logits, torch.from_numpy(labels), reduction="none")
    for q in m.parameters():
        q.grad = None
    (loss.sum() / B).backward()
    torch.manual_seed(4321)
    keep = torch.empty((T, B, E), dtype=torch.float64).bernoulli_(1 - p_drop).numpy() != 0     # [T,B,E]
    arrs = dict(ids=ids, lengths=lengths, labels=labels, keep=keep, logits=logits.detach().numpy(),
                weights=weights.detach().numpy(), loss=loss.detach().numpy())
    none_grads = []
    for k, q in m.named_parameters():
        if q.grad is None:
            none_grads.append(k)
        else:
            arrs["grad:" + k] = q.grad.numpy()
    for k, v in params.items():
        arrs["param:" + k] = v
    hdr = dict(case="tiny_hard_dropout_f64", V=V, E=E, H=H, B=B, T=T, L=2, param_seed=11, id_seed=9, label_seed=3,
               p_drop=p_drop, torch_seed=4321, none_grads=none_grads, **versions())
    save(os.path.join(outdir, "tiny_hard_dropout_f64.npz"), hdr, **arrs)


def yelp_soft_cases(outdir):
    """Transfer-shaped soft call: B=8, T=23, Yelp dims, int16 lengths."""
    V, E, H, B, T = 9353, 300, 512, 8, 23
    lengths = np.array([23, 20, 17, 12, 9, 5, 3, 2], dtype=np.int32)
    params = synth.make_params(V, E, H, seed=21, dtype=np.float32)
    labels = synth.make_labels(B, seed=4)
    mask = synth.transfer_mask(lengths, T)
    for kind in ("dense", "onehot"):
        S = synth.make_soft_tokens(B, T, V, lengths, seed=6, kind=kind, dtype=np.float32)
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            p = {k: v.astype(np.float64 if tag == "f64" else np.float32) for k, v in params.items()}
            m = build_ref("transfer", V, E, H, p, dtype)
            o = run_transfer_copy(m, S.astype(np.float64 if tag == "f64" else np.float32),
                                  lengths, mask, labels, True, dtype)
            dS = o["dS"]
            hdr = dict(case="yelp_soft_%s_%s" % (kind, tag), V=V, E=E, H=H, B=B, T=T, L=2,
                       param_seed=21, s_seed=6, label_seed=4, kind=kind, dS_col_stride=97,
                       none_grads=o["none_grads"], **versions())
            arrs = dict(lengths=lengths, mask=mask, labels=labels, logits=o["logits"],
                        weights=o["weights"], loss=o["loss"], mask_after=o["mask_after"],
                        dS_sample=dS[:, :, ::97].copy(),
                        dS_rowsum=dS.astype(np.float64).sum(axis=2),
                        dS_rowsq=(dS.astype(np.float64) ** 2).sum(axis=2),
                        S_rowdot=(S.astype(np.float64) * np.arange(V)[None, None, :]).sum(axis=2))
            # small param grads only (the big ones are pinned by the tiny case)
            for k in ("output_layer.weight", "output_layer.bias", "linear_attn.weight",
                      "linear_attn.bias", "linear_q.bias", "hidden_bridges.0.bias",
                      "linear_emb.bias", "rnn.bias_ih_l0", "rnn.bias_ih_l0_reverse"):
                arrs["grad:" + k] = o["grads"][k]
            save(os.path.join(outdir, "yelp_soft_%s_%s.npz" % (kind, tag)), hdr, **arrs)


def yelp_hard_case(outdir):
    """Yelp-real hard path: 64 dev sentences through the reference's own
    tokenizer, both copies."""
    from utils import tokenization          # /root/reference/utils/tokenization.py
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        tok = tokenization.BasicTokenizer(os.path.join(REF, "data/yelp/vocab"))
    V = tok.get_vocab_size()
    assert (tok.PAD_ID, tok.SOS_ID, tok.EOS_ID) == (0, 1, 2) and V == 9353
    sents, labs = [], []
    for lab in (0, 1):
        with open(os.path.join(REF, "data/yelp/dev.%d" % lab)) as f:
            lines = f.readlines()[:32]
        for ln in lines:
            sents.append([tok.SOS_ID] + tok.encode(ln) + [tok.EOS_ID])   # classifier/trainers.py:71
            labs.append(lab)
    lens = np.array([len(s) for s in sents], dtype=np.int32)
    order = np.argsort(-lens, kind="stable")
    lens = lens[order]
    labs = np.array(labs, dtype=np.int64)[order]
    T = int(lens.max())
    ids = np.zeros((len(sents), T), dtype=np.int64)
    for r, i in enumerate(order):
        ids[r, :len(sents[i])] = sents[i]
    E, H = 300, 512
    params = synth.make_params(V, E, H, seed=21, dtype=np.float32)
    pad_mask = ids == 0
    mask = synth.classifier_mask(pad_mask)
    arrs = dict(ids=ids, lengths=lens, labels=labs, mask=mask)
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        p = {k: v.astype(np.float64 if tag == "f64" else np.float32) for k, v in params.items()}
        mt = build_ref("transfer", V, E, H, p, dtype)
        o = run_transfer_copy(mt, ids, lens, mask, labs, False, dtype, len_dtype=torch.int32)
        mc = build_ref("classifier", V, E, H, p, dtype)
        lg, w = mc(torch.from_numpy(ids), torch.from_numpy(lens), torch.from_numpy(pad_mask))
        arrs.update({"logits_" + tag: o["logits"], "weights_" + tag: o["weights"],
                     "loss_" + tag: o["loss"], "cls_logits_" + tag: lg.detach().numpy(),
                     "cls_weights_" + tag: w.detach().numpy(),
                     "grad_linear_emb_bias_" + tag: o["grads"]["linear_emb.bias"],
                     "grad_output_layer_weight_" + tag: o["grads"]["output_layer.weight"]})
    hdr = dict(case="yelp_hard_b64", V=V, E=E, H=H, B=64, T=T, L=2, param_seed=21,
               source="data/yelp/dev.{0,1} first 32 lines each", **versions())
    save(os.path.join(outdir, "yelp_hard_b64.npz"), hdr, **arrs)


VARIANTS = (          # tag, cell, bidirectional, layers   (classifier/models.py:102-124)
    ("gru_bi", "GRU", True, 1),
    ("lstm_uni", "LSTM", False, 1),
    ("gru_uni", "GRU", False, 1),
    ("lstm_bi_l2", "LSTM", True, 2),
    ("gru_bi_l2", "GRU", True, 2),
    ("lstm_uni_l3", "LSTM", False, 3),
)


def variant_cases(outdir):
    """GRU / uni-directional / multi-layer discriminators: the transfer copy on dense soft tokens (outputs,
    dS, every parameter gradient) and the classifier copy on hard tokens in eval mode, float64."""
    V, E, H, B, T = 41, 10, 8, 6, 9
    lengths = np.array([9, 7, 5, 3, 3, 2], dtype=np.int32)
    labels = synth.make_labels(B, seed=3)
    mask = synth.transfer_mask(lengths, T)
    S = synth.make_soft_tokens(B, T, V, lengths, seed=5, kind="dense", dtype=np.float64)
    ids = synth.make_token_ids(B, T, V, lengths, seed=9)
    for tag, cell, bi, layers in VARIANTS:
        var = dict(cell=cell, bidirectional=bi, layers=layers)
        params = synth.make_params(V, E, H, seed=13, dtype=np.float64, **var)
        m = build_ref("transfer", V, E, H, params, torch.float64, **var)
        o = run_transfer_copy(m, S, lengths, mask, labels, True, torch.float64)
        arrs = dict(S=S, ids=ids, lengths=lengths, mask=mask, labels=labels, logits=o["logits"], weights=o["weights"],
                    loss=o["loss"], mask_after=o["mask_after"], dS=o["dS"])
        for k, g in o["grads"].items():
            arrs["grad:" + k] = g
        for k, v in params.items():
            arrs["param:" + k] = v
        keep = lengths >= 3
        mc = build_ref("classifier", V, E, H, params, torch.float64, **var)
        lg, w = mc(torch.from_numpy(ids[keep]), torch.from_numpy(lengths[keep]), torch.from_numpy(ids[keep] == 0))
        arrs.update(cls_keep=keep, cls_logits=lg.detach().numpy(), cls_weights=w.detach().numpy())
        hdr = dict(case="variant_%s_f64" % tag, V=V, E=E, H=H, B=B, T=T, L=2, param_seed=13, s_seed=5, id_seed=9,
                   label_seed=3, none_grads=o["none_grads"], **var, **versions())
        save(os.path.join(outdir, "variant_%s_f64.npz" % tag), hdr, **arrs)


def cnn_case(outdir):
    """CNN metric classifier (classifier/models.py:8-73), soft and hard branch, eval mode, float64, with
    d(soft tokens) and the parameter gradients of the mean CE loss."""
    import importlib
    mod = importlib.import_module("classifier.models")
    V, E, B, T, L = 41, 10, 5, 9, 2
    cfg = SimpleNamespace(pad_id=0, dropout=0.0, num_kernels_each_size=3, kernel_sizes=[2, 3, 4, 11], vocab_size=V,
                          emb_size=E, num_labels=L, freeze_embedding=False, device="cpu")
    torch.manual_seed(77)
    m = mod.CNNClassifier(cfg).to(torch.float64)
    m.eval()
    lengths = np.full((B,), T, dtype=np.int32)
    S = synth.make_soft_tokens(B, T, V, lengths, seed=15, kind="dense", dtype=np.float64)     # [T,B,V]
    Sb = torch.from_numpy(np.ascontiguousarray(S.transpose(1, 0, 2))).requires_grad_(True)    # [B,T,V]
    labels = synth.make_labels(B, seed=3)
    logits = m(Sb, soft=True)
    loss = F.cross_entropy(logits, torch.from_numpy(labels))
    loss.backward()
    ids = synth.make_token_ids(B, T, V, lengths, seed=9)
    hard = m(torch.from_numpy(ids))
    arrs = dict(S=Sb.detach().numpy(), labels=labels, logits=logits.detach().numpy(), dS=Sb.grad.numpy(), ids=ids,
                hard_logits=hard.detach().numpy())
    for k, q in m.named_parameters():
        arrs["param:" + k] = q.detach().numpy()
        if q.grad is not None:               # the kernel size above T never runs (classifier/models.py:62,69)
            arrs["grad:" + k] = q.grad.numpy()
    hdr = dict(case="cnn_soft_f64", V=V, E=E, B=B, T=T, L=L, num_kernels_each_size=3, kernel_sizes=[2, 3, 4, 11],
               torch_seed=77, s_seed=15, id_seed=9, label_seed=3, **versions())
    save(os.path.join(outdir, "cnn_soft_f64.npz"), hdr, **arrs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None, help="regenerate one family: dropout | variants")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if a.only == "dropout":
        dropout_case(a.out)
        return
    if a.only == "variants":
        variant_cases(a.out)
        cnn_case(a.out)
        return
    tiny_cases(a.out)
    dropout_case(a.out)
    yelp_soft_cases(a.out)
    yelp_hard_case(a.out)
    variant_cases(a.out)
    cnn_case(a.out)


if __name__ == "__main__":
    main()
