#!/usr/bin/env python3
"""Golden fixture of the CALLER side of the hot path, produced by running the REFERENCE itself.

Build-container only.  Imports ``/root/reference/transfer/models.py`` unmodified, builds a tiny
``TransferModel`` and runs its real ``forward`` once (train mode, noise probabilities 0) on CPU.  What
is recorded, without touching the reference's code:

  * every ``F.gumbel_softmax(logits, tau=1.0, hard=True)`` call of ``soft_decode``
    (transfer/models.py:531): the logits, the Gumbel noise it drew (re-drawn from the saved RNG state),
    and its output — the producer that SURVEY.md §8 f2 fuses;
  * ``soft_decode``'s return values (soft_out [T',B,V], soft_out_len int16; :471-574);
  * the locals of ``TransferModel.forward`` at its return (a ``sys.setprofile`` hook): everything the
    stretch :1172-1212 computes around the discriminator call — sort indices, sorted lengths, the mask
    before/after the call's in-place mutation, logits, style/content alpha, un-sorted logits, and
    ``soft_out_cls_loss`` = CE vs ``1 - lab``;
  * d mean(soft_out_cls_loss) / d soft_out through the discriminator (autograd.grad on the recorded
    ``soft_out``).

    python tools/gen_golden_transfer.py        # rewrites tests/golden/transfer_caller_tiny.npz
"""
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


def main():
    import transfer.models as TM       # the reference, unmodified
    V, E, H, B, T_src, MAXDEC = 57, 12, 16, 7, 9, 6
    cfg = SimpleNamespace(cell="LSTM", enc_bidirectional=True, pad_id=0, sos_id=1, eos_id=2, dropout=0.0,
                          enc_num_layers=1, dec_num_layers=1, hidden_size=H, vocab_size=V, emb_size=E,
                          num_labels=2, freeze_embedding=False, device="cpu", mask_id=None, beam_size=2,
                          max_decoding_len=MAXDEC, mask_prob=0.0, drop_prob=0.0, shuffle_weight=0.0)
    torch.manual_seed(20261111)
    model = TM.TransferModel(cfg)
    # make [eos] likely enough that sentences end at different steps (ragged soft_out_len, incl. short ones)
    with torch.no_grad():
        for name, p in model.decoder.named_parameters():
            if p.dim() == 1 and p.shape[0] == V:
                p[cfg.eos_id] += 2.0
    model.train()
    model.freeze_cls()

    rs = np.random.RandomState(5)
    lengths = np.sort(rs.randint(3, T_src + 1, size=B))[::-1].copy()
    lengths[0] = T_src
    src = np.zeros((B, T_src), dtype=np.int64)
    for b, ln in enumerate(lengths):
        src[b, 0] = cfg.sos_id
        src[b, 1:ln - 1] = rs.randint(3, V, size=ln - 2)
        src[b, ln - 1] = cfg.eos_id
    lab = rs.randint(0, 2, size=B).astype(np.int64)
    src_t = torch.from_numpy(src)
    len_t = torch.from_numpy(lengths.astype(np.int32))
    lab_t = torch.from_numpy(lab)

    # ---- recorders ---------------------------------------------------------------------------------
    steps = []
    orig_gumbel = F.gumbel_softmax

    def rec_gumbel(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        state = torch.get_rng_state()
        out = orig_gumbel(logits, tau=tau, hard=hard, eps=eps, dim=dim)
        after = torch.get_rng_state()
        torch.set_rng_state(state)
        # the draw inside F.gumbel_softmax (torch/nn/functional.py): -empty_like(logits).exponential_().log()
        noise = -torch.empty_like(logits, memory_format=torch.legacy_contiguous_format).exponential_().log()
        torch.set_rng_state(after)
        y_soft = ((logits + noise) / tau).softmax(dim)
        chk = torch.zeros_like(logits).scatter_(dim, y_soft.max(dim, keepdim=True)[1], 1.0) - y_soft.detach() + y_soft
        assert torch.equal(chk, out), "re-drawn Gumbel noise does not reproduce the reference's sample"
        steps.append(dict(logits=logits.detach().numpy().copy(), noise=noise.numpy().copy(), out=out.detach().numpy().copy(),
                          y_soft=y_soft.detach().numpy().copy(), tau=float(tau), hard=bool(hard)))
        return out

    soft = {}
    orig_soft_decode = model.decoder.soft_decode

    def rec_soft_decode(*a, **k):
        out, out_len, align = orig_soft_decode(*a, **k)
        out.retain_grad()
        soft["out"], soft["len"] = out, out_len
        return out, out_len, align

    cls_calls = []
    orig_cls_forward = model.classifier.forward

    def rec_cls(src_, src_len_, mask_, hidden=None, soft=False):
        before = mask_.clone()
        logits, weights = orig_cls_forward(src_, src_len_, mask_, hidden=hidden, soft=soft)
        cls_calls.append(dict(soft=soft, src_shape=tuple(src_.shape), src_stride=tuple(src_.stride()),
                              len=src_len_.clone(), len_dtype=str(src_len_.dtype), mask_before=before,
                              mask_after=mask_.clone(), logits=logits.detach().clone(), weights=weights.detach().clone()))
        return logits, weights

    frames = {}
    fwd_code = TM.TransferModel.forward.__code__

    def prof(frame, event, arg):
        if event == "return" and frame.f_code is fwd_code:
            frames.update({k: v for k, v in frame.f_locals.items() if torch.is_tensor(v)})

    F.gumbel_softmax = rec_gumbel
    model.decoder.soft_decode = rec_soft_decode
    model.classifier.forward = rec_cls
    sys.setprofile(prof)
    try:
        losses, _, _ = model(src_t, len_t, lab_t, src_t == cfg.pad_id)
    finally:
        sys.setprofile(None)
        F.gumbel_softmax = orig_gumbel
    soft_out_cls_loss = losses[3]
    (dsoft,) = torch.autograd.grad(soft_out_cls_loss.mean(), soft["out"], retain_graph=False)

    call2 = [c for c in cls_calls if c["soft"]]
    assert len(call2) == 1 and len(cls_calls) == 3
    c2 = call2[0]
    Tq = soft["out"].shape[0]
    n_soft_steps = Tq - 1 if int(soft["len"].max()) <= Tq else Tq - 1
    soft_steps = steps[:len(steps)]          # soft_decode is the only gumbel_softmax user in forward
    hdr = dict(case="transfer_caller_tiny", V=V, E=E, H=H, B=B, T_src=T_src, max_decoding_len=MAXDEC,
               pad_id=0, sos_id=1, eos_id=2, Tq=int(Tq), n_gumbel_steps=len(soft_steps),
               cls_src_shape=list(c2["src_shape"]), cls_src_stride=list(c2["src_stride"]), cls_len_dtype=c2["len_dtype"],
               tau=soft_steps[0]["tau"], hard=soft_steps[0]["hard"],
               torch=torch.__version__, numpy=np.__version__, reference=REF,
               reference_lines="transfer/models.py:471-574 (soft_decode), :1164-1212 (caller stretch)")
    arrs = dict(
        lab=lab, inv_lab=frames["inv_lab"].numpy(),
        soft_out=soft["out"].detach().numpy(), soft_out_len=soft["len"].numpy(),
        sorted_soft_out_len=frames["sorted_soft_out_len"].numpy(), soft_out_indices=frames["soft_out_indices"].numpy(),
        resorted_soft_out_indices=frames["resorted_soft_out_indices"].numpy(),
        mask_before=c2["mask_before"].numpy(), mask_after=c2["mask_after"].numpy(),
        sorted_logits=c2["logits"].numpy(), sorted_style_alpha_raw=c2["weights"].numpy(),
        sorted_soft_out_content_alpha=frames["sorted_soft_out_content_alpha"].numpy(),
        soft_out_cls_logits=frames["soft_out_cls_logits"].detach().numpy(),
        soft_out_content_alpha=frames["soft_out_content_alpha"].numpy(),
        soft_out_cls_loss=soft_out_cls_loss.detach().numpy(),
        d_soft_out=dsoft.numpy(),
    )
    for i, s in enumerate(soft_steps):
        arrs["step%d_logits" % i] = s["logits"]
        arrs["step%d_noise" % i] = s["noise"]
        arrs["step%d_out" % i] = s["out"]
        arrs["step%d_ysoft" % i] = s["y_soft"]
    for k, v in model.classifier.state_dict().items():
        arrs["param:" + k] = v.numpy()
    arrs["header_json"] = np.frombuffer(json.dumps(hdr).encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "transfer_caller_tiny.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))
    print("soft_out", tuple(soft["out"].shape), "len", soft["len"].tolist(), "sorted", frames["sorted_soft_out_len"].tolist())
    print("all-masked rows fixed by the guard:", int((c2["mask_before"] != c2["mask_after"]).sum()))
    print("soft_out_cls_loss", soft_out_cls_loss.detach().numpy())


if __name__ == "__main__":
    main()
