"""Seeded, torch-independent generators for discriminator weights and Yelp-shaped batches.

Everything here is plain numpy (legacy ``RandomState`` streams, which are stable across numpy
versions and machines) so that the build container and the GPU box regenerate bit-identical
weights/inputs from a seed instead of shipping 32 MB checkpoints.

Parameter names and shapes are the 23 tensors of the reference ``SelfAttnRNNClassifier``
(reference ``transfer/models.py:757-829``, twin ``classifier/models.py:80-146``).
"""
from collections import OrderedDict

import numpy as np

GATES = 4  # torch LSTM gate order: i, f, g, o  (reference transfer/models.py:779-785 -> nn.LSTM)


def param_shapes(V, E, H, L=2, cell="LSTM", bidirectional=True, layers=1):
    """Ordered name -> shape map; the order is the reference module's ``state_dict()`` order
    (variants: reference classifier/models.py:102-124)."""
    s = OrderedDict()
    G = GATES if cell == "LSTM" else 3          # GRU gate order: r, z, n
    D = 2 if bidirectional else 1
    s["embedding.weight"] = (V, E)
    for l in range(layers):
        for sfx in ("", "_reverse")[:D]:
            s["rnn.weight_ih_l%d%s" % (l, sfx)] = (G * H, E if l == 0 else D * H)
            s["rnn.weight_hh_l%d%s" % (l, sfx)] = (G * H, H)
            s["rnn.bias_ih_l%d%s" % (l, sfx)] = (G * H,)
            s["rnn.bias_hh_l%d%s" % (l, sfx)] = (G * H,)
    s["output_bridge.weight"] = (H, D * H)
    s["output_bridge.bias"] = (H,)
    for i in range(2 if cell == "LSTM" else 1):
        s["hidden_bridges.%d.weight" % i] = (H, D * H)
        s["hidden_bridges.%d.bias" % i] = (H,)
    s["linear_emb.weight"] = (H, E)
    s["linear_emb.bias"] = (H,)
    s["linear_q.weight"] = (H, H)
    s["linear_q.bias"] = (H,)
    s["linear_attn.weight"] = (1, H)
    s["linear_attn.bias"] = (1,)
    s["output_layer.weight"] = (L, H)
    s["output_layer.bias"] = (L,)
    return s


# Tensors whose value never reaches the logits: grad is None in the reference.
DEAD_PARAMS = ("output_bridge.weight", "output_bridge.bias",
               "hidden_bridges.1.weight", "hidden_bridges.1.bias")


def make_params(V, E, H, L=2, seed=0, emb_std=1.0, pad_id=0, dtype=np.float32, cell="LSTM", bidirectional=True,
                layers=1):
    """torch-default-like magnitudes: embedding ~ N(0, emb_std) with a zero pad row, every other
    tensor ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (LSTM: 1/sqrt(H))."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    shapes = param_shapes(V, E, H, L, cell, bidirectional, layers)
    for name, shape in shapes.items():
        if name == "embedding.weight":
            w = rs.standard_normal(shape) * emb_std
            w[pad_id] = 0.0
        else:
            if name.startswith("rnn."):
                k = 1.0 / np.sqrt(H)
            elif name.endswith(".bias"):
                # fan_in of the matching weight
                k = 1.0 / np.sqrt(shapes[name[:-4] + "weight"][1])
            else:
                k = 1.0 / np.sqrt(shape[1])
            w = rs.uniform(-k, k, size=shape)
        out[name] = np.ascontiguousarray(w.astype(dtype))
    return out


def make_lengths(B, T, seed=0, lo=3, full=False):
    """Sorted-descending sentence lengths (incl. sos/eos) with len[0] == T, as
    ``pack_padded_sequence(enforce_sorted=True)`` demands (reference transfer/models.py:904)."""
    if full:
        return np.full((B,), T, dtype=np.int32)
    rs = np.random.RandomState(seed + 7919)
    ln = rs.randint(lo, T + 1, size=B)
    ln[0] = T
    return np.sort(ln)[::-1].astype(np.int32).copy()


# Sentence lengths (tokens + sos + eos) of the reference's data/yelp/dev.0 + dev.1, 4000 sentences:
# length -> count.  Data, counted in the build container with
#   collections.Counter(len(line.split()) + 2 for f in ("dev.0", "dev.1") for line in open(f))
# ("cfg2 ... lengths from real dev.* (+2 for sos/eos; observed min 3, mean 10.9, max 17)").
YELP_DEV_LENGTH_COUNTS = {3: 12, 4: 90, 5: 195, 6: 227, 7: 340, 8: 348, 9: 323, 10: 320, 11: 324, 12: 346, 13: 327,
                          14: 313, 15: 284, 16: 278, 17: 273}


def yelp_dev_lengths(B, T, seed=0):
    """B sentence lengths drawn from the Yelp dev length histogram (clipped to T), sorted descending."""
    ks = np.array(sorted(YELP_DEV_LENGTH_COUNTS), dtype=np.int64)
    pr = np.array([YELP_DEV_LENGTH_COUNTS[k] for k in ks], dtype=np.float64)
    rs = np.random.RandomState(seed + 611953)
    ln = np.minimum(rs.choice(ks, size=B, p=pr / pr.sum()), T)
    return np.sort(ln)[::-1].astype(np.int32).copy()


def transfer_mask(lengths, T):
    """pad/sos/eos mask exactly as the transfer step builds it for soft outputs
    (reference transfer/models.py:1180-1183): ``arange(T) >= len-1`` and column 0 set."""
    ar = np.arange(T)[None, :]
    m = ar >= (np.asarray(lengths).astype(np.int64)[:, None] - 1)
    m[:, 0] = True
    return m


def classifier_mask(pad_mask):
    """pad+sos+eos mask as the classifier copy derives it from a pad mask
    (reference classifier/models.py:183-187): ``[True | pad[:, 2:] | True]``."""
    B = pad_mask.shape[0]
    ones = np.ones((B, 1), dtype=bool)
    return np.concatenate([ones, pad_mask[:, 2:], ones], axis=1)


def make_soft_tokens(B, T, V, lengths, seed=0, kind="dense", pad_id=0, sos_id=1,
                     dtype=np.float32, time_major=True):
    """Soft tokens S shaped like the generator's output (reference transfer/models.py:531-571).

    kind="dense":  per-token softmax of N(0,1)*3 logits (every vocabulary entry non-zero).
    kind="onehot": straight-through Gumbel output = exact one-hot rows.
    Rows t >= len[b] are one-hot [pad]; row 0 is one-hot [sos].
    Returns an array laid out [T,B,V] (time_major, what soft_decode produces) or [B,T,V].
    """
    rs = np.random.RandomState(seed + 104729)
    S = np.zeros((T, B, V), dtype=dtype)
    lengths = np.asarray(lengths)
    for t in range(T):
        act = lengths > t
        n = int(act.sum())
        if t == 0:
            S[0, :, sos_id] = 1.0
            continue
        if n:
            if kind == "dense":
                z = rs.standard_normal((n, V)).astype(np.float32) * 3.0
                z -= z.max(axis=1, keepdims=True)
                p = np.exp(z)
                p /= p.sum(axis=1, keepdims=True)
                S[t, act] = p.astype(dtype)
            elif kind == "onehot":
                ids = rs.randint(3, V, size=n)
                S[t, np.nonzero(act)[0], ids] = 1.0
            else:
                raise ValueError(kind)
        S[t, ~act, pad_id] = 1.0
    return S if time_major else np.ascontiguousarray(S.transpose(1, 0, 2))


def make_token_ids(B, T, V, lengths, seed=0, pad_id=0, sos_id=1, eos_id=2):
    """Hard-token batch as ``ClassifierPaddingCollate`` emits it (reference utils/data.py:67-84):
    int64 [B,T], ``[sos] w.. [eos] [pad]..``."""
    rs = np.random.RandomState(seed + 15485863)
    ids = np.full((B, T), pad_id, dtype=np.int64)
    for b, ln in enumerate(np.asarray(lengths)):
        ids[b, 0] = sos_id
        if ln > 2:
            ids[b, 1:ln - 1] = rs.randint(3, V, size=ln - 2)
        ids[b, ln - 1] = eos_id
    return ids


def make_labels(B, seed=0, L=2):
    return np.random.RandomState(seed + 32452843).randint(0, L, size=B).astype(np.int64)
