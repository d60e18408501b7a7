"""ORACLE — test infrastructure only. NOT part of the product path.

CPU restatement (plain numpy) of the CALLER side of the hot path in the reference's transfer step:

  * ``gumbel_softmax_hard``   — ``F.gumbel_softmax(logits, tau, hard=True)`` as ``soft_decode`` uses it
    (transfer/models.py:531) with the Gumbel noise passed in;
  * ``soft_decode_assemble``  — the bookkeeping of ``AttnRNNDecoder.soft_decode`` around the per-step
    samples (transfer/models.py:511-574): alive mask, lengths, [sos] row, optional [eos] row, [pad] fill;
  * ``caller_stretch``        — transfer/models.py:1172-1212: sort by length, pad/sos/eos mask, the
    discriminator call on the sorted batch, NaN -> 0 / ``1 - alpha`` content weights, un-sort, CE vs
    ``1 - lab``; the discriminator itself is ``oracle.sad_oracle.forward`` / ``backward``.

Parity status: PINNED by ``tests/test_caller_golden.py`` against ``tests/golden/transfer_caller_tiny.npz``,
which ``tools/gen_golden_transfer.py`` recorded from the reference's own ``TransferModel.forward``.
Only ``tests/`` may import this file.
"""
import numpy as np

from . import sad_oracle as O


def gumbel_softmax_hard(logits, noise, tau=1.0):
    """torch.nn.functional.gumbel_softmax(hard=True) with the noise given (reference call site
    transfer/models.py:531).  Returns (ret, y_soft, index): ``ret = y_hard - y_soft + y_soft`` evaluated
    left to right in the dtype of ``logits`` — exactly 0 off the arg-max, (1 - p) + p on it."""
    z = (logits + noise) / np.asarray(tau, logits.dtype)
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    y = (e / e.sum(axis=-1, keepdims=True)).astype(logits.dtype)
    idx = y.argmax(axis=-1)
    hard = np.zeros_like(y)
    np.put_along_axis(hard, idx[..., None], 1.0, axis=-1)
    return (hard - y) + y, y, idx


def gumbel_softmax_hard_backward(g, y, tau=1.0):
    """d ret / d logits: the straight-through estimator back-propagates through y_soft only."""
    return y * (g - (g * y).sum(axis=-1, keepdims=True)) / tau


def soft_decode_assemble(step_outputs, V, pad_id, sos_id, eos_id, max_decoding_len):
    """transfer/models.py:511-574 given the per-step samples ``rnn_in`` ([B,V] each, as many as the
    reference's while loop ran).  Returns (out [T',B,V], outputs_len int16 [B])."""
    B = step_outputs[0].shape[0]
    dt = step_outputs[0].dtype
    mask = np.ones((B,), dtype=bool)                       # still decoding
    outputs_len = np.zeros((B,), dtype=np.int16)
    rows = []
    cur = 0
    for rnn_in in step_outputs:
        assert mask.sum() != 0 and cur <= max_decoding_len
        rows.append(rnn_in * mask[:, None].astype(dt))      # :539
        cur += 1
        outputs_len += mask                                 # :542
        mask = (rnn_in.argmax(axis=1) != eos_id) & mask     # :543
    assert mask.sum() == 0 or cur > max_decoding_len        # the while loop's exit condition
    sos = np.zeros((1, B, V), dt)
    sos[..., sos_id] = 1.0
    out = np.concatenate([sos, np.stack(rows, axis=0)], axis=0)   # :548-553
    outputs_len = outputs_len + 1
    if cur > max_decoding_len:                               # :556-566
        add_eos = False
        for b in range(B):
            if mask[b] and out[-1, b].argmax() != eos_id:
                add_eos = True
                outputs_len[b] += 1
        if add_eos:
            eos = np.zeros((1, B, V), dt)
            eos[..., eos_id] = 1.0
            out = np.concatenate([out, eos], axis=0)
    t = np.arange(out.shape[0])[:, None]
    padrow = np.zeros((V,), dt)
    padrow[pad_id] = 1.0
    out[t >= outputs_len[None, :].astype(np.int64)] = padrow     # :570-571
    return out, outputs_len.astype(np.int16)


def caller_stretch(params, soft_out, soft_out_len, inv_lab, want_grad=True):
    """transfer/models.py:1172-1212 around the oracle discriminator.  soft_out [T,B,V] (time-major, as
    soft_decode returns it).  Returns a dict with the reference's variable names."""
    T, B, V = soft_out.shape
    ln = soft_out_len.astype(np.int64)
    idx = np.argsort(-ln, kind="stable")                   # torch.sort(descending=True) is not stable in general;
    sorted_len = ln[idx]                                   # the fixture stores the reference's indices
    return _stretch_with_indices(params, soft_out, ln, inv_lab, idx, want_grad)


def _stretch_with_indices(params, soft_out, ln, inv_lab, idx, want_grad=True):
    T, B, V = soft_out.shape
    sorted_len = ln[idx]
    resort = np.argsort(idx, kind="stable")
    sorted_soft_out = soft_out[:, idx]                     # :1175
    sorted_lab = inv_lab[idx]
    mask = np.arange(T)[None, :] >= (sorted_len[:, None] - 1)   # :1180-1182
    mask[:, 0] = True                                      # :1183
    mask_before = mask.copy()
    p = O.cast_params(params, soft_out.dtype)
    f = O.forward(p, sorted_soft_out.transpose(1, 0, 2), sorted_len, mask, True, labels=sorted_lab)   # :1185-1187 (mask mutated)
    style = f["weights"].copy()
    style[np.isnan(style)] = 0.0                           # :1195
    content = 1.0 - style                                  # :1196
    content[mask] = 0.0                                    # :1197 (the MUTATED mask)
    content = content / (sorted_len - 3 + 1e-4)[:, None].astype(soft_out.dtype)   # :1199
    for i, l in enumerate(sorted_len):                     # :1200-1202
        if l == 2:
            content[i, 1] = 1.0
    out = dict(soft_out_indices=idx, resorted_soft_out_indices=resort, sorted_soft_out_len=sorted_len,
               mask_before=mask_before, mask_after=mask, sorted_logits=f["logits"], sorted_style_alpha_raw=f["weights"],
               sorted_soft_out_content_alpha=content,
               soft_out_cls_logits=f["logits"][resort], soft_out_content_alpha=content[resort],   # :1207-1208
               soft_out_cls_loss=f["loss"][resort])        # :1212: CE(un-sorted logits, inv_lab)
    if want_grad:
        dl = O.ce_dlogits(f["logits"], sorted_lab, np.full((B,), 1.0 / B, soft_out.dtype))
        b = O.backward(p, f["cache"], dl)
        d_sorted = b["dS"].transpose(1, 0, 2)              # [T,B,V] in sorted order
        d = np.zeros_like(soft_out)
        d[:, idx] = d_sorted                               # backward of index_select
        out["d_soft_out"] = d
    return out
