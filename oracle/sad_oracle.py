"""ORACLE — test infrastructure only. NOT part of the product path.

CPU restatement (plain numpy, any float dtype) of the reference's self-attention discriminator
forward and its hand-derived backward.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the shipped path
(``sad_amd`` -> ``libsad.so``) never does and fails loudly without its HIP library.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this file against the fixtures
under ``tests/golden/`` that ``tools/gen_golden.py`` produced by importing the reference itself
(``/root/reference/transfer/models.py`` and ``classifier/models.py``) in the build container.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def guard_mask_inplace(mask):
    """transfer/models.py:865-867 — a row whose mask is all True gets position 1 unmasked.
    Mutates the caller's array exactly like the reference mutates the caller's tensor."""
    allm = mask.all(axis=1)
    if mask.shape[1] > 1:
        mask[allm, 1] = False
    return mask


def embed(params, src, soft):
    """transfer/models.py:895-900.  src: [B,T,V] float (soft) or [B,T] int (hard) -> X [B,T,E]."""
    W = params["embedding.weight"]
    if soft:
        return src @ W
    return W[src]


def lstm_dir_forward(X, lengths, W_ih, W_hh, b_ih, b_hh, reverse):
    """One direction of the packed LSTM, transfer/models.py:779-785,904-906 (torch gate order
    i,f,g,o; only t < len[b] is processed; returns the per-step cache and the final h).

    X: [B,T,E].  Returns (h_final [B,H], cache)."""
    B, T, _ = X.shape
    H = W_hh.shape[1]
    dt = X.dtype
    h = np.zeros((B, H), dt)
    c = np.zeros((B, H), dt)
    Zin = X @ W_ih.T + (b_ih + b_hh)  # [B,T,4H]
    gates = np.zeros((T, B, 4 * H), dt)   # activated i,f,g,o
    cs = np.zeros((T, B, H), dt)          # c_t
    cprev = np.zeros((T, B, H), dt)       # c_{t-1} as seen by step t
    hprev = np.zeros((T, B, H), dt)       # h_{t-1} as seen by step t
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        act = lengths > t
        if not act.any():
            continue
        g = Zin[:, t, :] + h @ W_hh.T
        i = _sigmoid(g[:, 0 * H:1 * H])
        f = _sigmoid(g[:, 1 * H:2 * H])
        gg = np.tanh(g[:, 2 * H:3 * H])
        o = _sigmoid(g[:, 3 * H:4 * H])
        c_new = f * c + i * gg
        h_new = o * np.tanh(c_new)
        hprev[t] = h
        cprev[t] = c
        gates[t] = np.concatenate([i, f, gg, o], axis=1)
        cs[t] = c_new
        a = act[:, None]
        c = np.where(a, c_new, c)
        h = np.where(a, h_new, h)
    return h, dict(gates=gates, c=cs, cprev=cprev, hprev=hprev)


def lstm_dir_backward(dh_final, cache, lengths, W_ih, W_hh, reverse):
    """BPTT for one direction (autograd of nn.LSTM on packed input).  Only the final hidden state
    receives an upstream gradient (transfer/models.py:930,934 consumes h_n only).
    Returns dZ [T,B,4H] (pre-activation gate grads; zero where t >= len[b])."""
    gates, cs, cprev = cache["gates"], cache["c"], cache["cprev"]
    T, B, H4 = gates.shape
    H = H4 // 4
    dt = gates.dtype
    dZ = np.zeros((T, B, H4), dt)
    dh = dh_final.astype(dt).copy()
    dc = np.zeros((B, H), dt)
    order = range(T) if reverse else range(T - 1, -1, -1)   # reverse of the forward order
    for t in order:
        act = (lengths > t)[:, None]
        if not act.any():
            continue
        i = gates[t][:, 0 * H:1 * H]
        f = gates[t][:, 1 * H:2 * H]
        g = gates[t][:, 2 * H:3 * H]
        o = gates[t][:, 3 * H:4 * H]
        tc = np.tanh(cs[t])
        do = dh * tc
        dct = dc + dh * o * (1.0 - tc * tc)
        di = dct * g
        dg = dct * i
        df = dct * cprev[t]
        dz = np.concatenate([di * i * (1 - i), df * f * (1 - f), dg * (1 - g * g), do * o * (1 - o)], axis=1)
        dz = np.where(act, dz, 0.0)
        dZ[t] = dz
        dh = np.where(act, dz @ W_hh, dh)
        dc = np.where(act, dct * f, dc)
    return dZ


def forward(params, src, lengths, mask, soft, labels=None, mutate_mask=True, guard=True, drop=None):
    """Full discriminator forward (transfer/models.py:885-953; attention :841-883;
    per-sentence CE = nn.CrossEntropyLoss(reduction='none'), transfer/models.py:990,1212).

    src: [B,T,V] float (soft) or [B,T] int;  lengths [B] (sorted desc, >= 2);  mask [B,T] bool,
    True = pad/sos/eos.  ``guard`` applies the all-masked-row fix of the transfer copy (the
    classifier copy, classifier/models.py:161-197, has none).
    ``drop`` = (keep [B,T,E] bool, scale) applies inverted dropout to the embedded tokens
    (classifier copy in training mode, classifier/models.py:217).
    Returns dict(logits [B,L], weights [B,T], loss [B] or None, cache)."""
    lengths = np.asarray(lengths).astype(np.int64)
    if not mutate_mask:
        mask = mask.copy()
    if guard:
        guard_mask_inplace(mask)
    X = embed(params, src, soft)
    if drop is not None:
        X = X * (np.asarray(drop[0]) * drop[1]).astype(X.dtype)
    dt = X.dtype
    h_f, cf = lstm_dir_forward(X, lengths, params["rnn.weight_ih_l0"], params["rnn.weight_hh_l0"],
                               params["rnn.bias_ih_l0"], params["rnn.bias_hh_l0"], False)
    h_b, cb = lstm_dir_forward(X, lengths, params["rnn.weight_ih_l0_reverse"],
                               params["rnn.weight_hh_l0_reverse"], params["rnn.bias_ih_l0_reverse"],
                               params["rnn.bias_hh_l0_reverse"], True)
    hcat = np.concatenate([h_f, h_b], axis=1)                                 # :929-934
    q = hcat @ params["hidden_bridges.0.weight"].T + params["hidden_bridges.0.bias"]
    Kk = X @ params["linear_emb.weight"].T + params["linear_emb.bias"]        # :925
    u = q @ params["linear_q.weight"].T + params["linear_q.bias"]             # :855
    mid = np.tanh(Kk + u[:, None, :])
    s = mid @ params["linear_attn.weight"][0] + params["linear_attn.bias"][0]  # :858
    s = np.where(mask, -np.inf, s)                                            # :868
    with np.errstate(invalid="ignore"):
        m = s.max(axis=1, keepdims=True)
        e = np.exp(s - m)
        a = (e / e.sum(axis=1, keepdims=True)).astype(dt)                     # :871
    ctx = np.einsum("bt,bth->bh", a, Kk)                                      # :880
    logits = ctx @ params["output_layer.weight"].T + params["output_layer.bias"]  # :949
    loss = None
    if labels is not None:
        z = logits - logits.max(axis=1, keepdims=True)
        lse = np.log(np.exp(z).sum(axis=1))
        loss = lse - z[np.arange(len(labels)), labels]
    cache = dict(X=X, cf=cf, cb=cb, hcat=hcat, q=q, u=u, Kk=Kk, mid=mid, a=a, ctx=ctx,
                 mask=mask, lengths=lengths, soft=soft, src=src, drop=drop)
    return dict(logits=logits, weights=a, loss=loss, cache=cache)


def ce_dlogits(logits, labels, dloss):
    """d loss_b / d logits_b for per-sentence CE, scaled by the upstream dloss [B]."""
    z = logits - logits.max(axis=1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(axis=1, keepdims=True)
    p[np.arange(len(labels)), labels] -= 1.0
    return p * np.asarray(dloss)[:, None]


def backward(params, cache, dlogits, dweights=None, want_param_grads=False):
    """Hand-derived backward of :func:`forward` (what autograd runs for
    transfer/trainers.py:376 through transfer/models.py:1185-1187).

    Returns dict(dX [B,T,E], dS [B,T,V] (soft only), grads {name: array} if requested)."""
    X, Kk, mid, a, ctx = cache["X"], cache["Kk"], cache["mid"], cache["a"], cache["ctx"]
    lengths = cache["lengths"]
    B, T, E = X.shape
    H = Kk.shape[2]
    dt = X.dtype
    Wo = params["output_layer.weight"]
    wa = params["linear_attn.weight"][0]
    dlogits = dlogits.astype(dt)
    dctx = dlogits @ Wo
    da = np.einsum("bh,bth->bt", dctx, Kk)
    if dweights is not None:
        da = da + dweights
    a0 = np.nan_to_num(a, nan=0.0)
    ds = a0 * (da - (a0 * da).sum(axis=1, keepdims=True))
    dpre = ds[:, :, None] * wa[None, None, :] * (1.0 - mid * mid)
    dKk = a0[:, :, None] * dctx[:, None, :] + dpre
    du = dpre.sum(axis=1)
    dq = du @ params["linear_q.weight"]
    dhcat = dq @ params["hidden_bridges.0.weight"]
    dZf = lstm_dir_backward(dhcat[:, :H], cache["cf"], lengths, params["rnn.weight_ih_l0"],
                            params["rnn.weight_hh_l0"], False)
    dZb = lstm_dir_backward(dhcat[:, H:], cache["cb"], lengths, params["rnn.weight_ih_l0_reverse"],
                            params["rnn.weight_hh_l0_reverse"], True)
    dX = dKk @ params["linear_emb.weight"]
    dX = dX + np.einsum("tbg,ge->bte", dZf, params["rnn.weight_ih_l0"])
    dX = dX + np.einsum("tbg,ge->bte", dZb, params["rnn.weight_ih_l0_reverse"])
    dXd = dX                                    # gradient w.r.t. the (dropped) LSTM / key input
    if cache.get("drop") is not None:           # ... and w.r.t. the embedding output in front of the dropout
        dX = dX * (np.asarray(cache["drop"][0]) * cache["drop"][1]).astype(dt)
    out = dict(dX=dX, dXd=dXd, dKk=dKk, dZf=dZf, dZb=dZb, du=du, dq=dq, dhcat=dhcat, dctx=dctx)
    if cache["soft"]:
        out["dS"] = dX @ params["embedding.weight"].T
    if want_param_grads:
        g = {}
        g["output_layer.weight"] = dlogits.T @ ctx
        g["output_layer.bias"] = dlogits.sum(axis=0)
        g["linear_attn.weight"] = np.einsum("bt,bth->h", ds, mid)[None, :]
        g["linear_attn.bias"] = ds.sum()[None]
        g["linear_q.weight"] = du.T @ cache["q"]
        g["linear_q.bias"] = du.sum(axis=0)
        g["hidden_bridges.0.weight"] = dq.T @ cache["hcat"]
        g["hidden_bridges.0.bias"] = dq.sum(axis=0)
        g["linear_emb.weight"] = np.einsum("bth,bte->he", dKk, X)
        g["linear_emb.bias"] = dKk.sum(axis=(0, 1))
        for sfx, dZ, c in (("", dZf, cache["cf"]), ("_reverse", dZb, cache["cb"])):
            g["rnn.weight_ih_l0" + sfx] = np.einsum("tbg,bte->ge", dZ, X)
            g["rnn.weight_hh_l0" + sfx] = np.einsum("tbg,tbh->gh", dZ, c["hprev"])
            g["rnn.bias_ih_l0" + sfx] = dZ.sum(axis=(0, 1))
            g["rnn.bias_hh_l0" + sfx] = dZ.sum(axis=(0, 1))
        if cache["soft"]:
            g["embedding.weight"] = np.einsum("btv,bte->ve", cache["src"], dX)
        else:
            ge = np.zeros_like(params["embedding.weight"])
            np.add.at(ge, cache["src"].reshape(-1), dX.reshape(-1, E))
            g["embedding.weight"] = ge
        out["grads"] = g
    return out


def cast_params(params, dtype):
    return {k: v.astype(dtype) for k, v in params.items()}


def fwd_bwd(params, src, lengths, mask, soft, labels, dloss=None, dtype=np.float64,
            want_param_grads=False):
    """Convenience: forward + per-sentence CE + backward with upstream dloss (default 1/B, i.e.
    the ``.mean()`` of transfer/trainers.py:370)."""
    p = cast_params(params, dtype)
    s = src.astype(dtype) if soft else src
    f = forward(p, s, lengths, mask, soft, labels=labels, mutate_mask=False)
    B = len(labels)
    if dloss is None:
        dloss = np.full((B,), 1.0 / B, dtype)
    dlog = ce_dlogits(f["logits"], labels, dloss)
    b = backward(p, f["cache"], dlog, want_param_grads=want_param_grads)
    return f, b
