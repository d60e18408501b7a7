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


def variant(params):
    """(cell, num_directions, num_layers) read off the state_dict layout
    (classifier/models.py:102-124: getattr(nn, cell), enc_bidirectional, enc_num_layers)."""
    H = params["rnn.weight_hh_l0"].shape[1]
    rows = params["rnn.weight_hh_l0"].shape[0]
    cell = {4 * H: "LSTM", 3 * H: "GRU"}[rows]
    D = 2 if "rnn.weight_ih_l0_reverse" in params else 1
    layers = 0
    while "rnn.weight_ih_l%d" % layers in params:
        layers += 1
    return cell, D, layers


def rnn_dir_forward(X, lengths, W_ih, W_hh, b_ih, b_hh, reverse, cell="LSTM"):
    """One direction of one layer of the packed RNN, transfer/models.py:779-785,904-906 (torch gate
    order i,f,g,o for the LSTM and r,z,n for the GRU; only t < len[b] is processed).

    X: [B,T,in].  Returns (h_final [B,H], hs [T,B,H] (zeros for t >= len), cache)."""
    B, T, _ = X.shape
    H = W_hh.shape[1]
    dt = X.dtype
    NG = 4 if cell == "LSTM" else 3
    h = np.zeros((B, H), dt)
    c = np.zeros((B, H), dt)
    Zx = X @ W_ih.T + b_ih                 # [B,T,NG*H]
    gates = np.zeros((T, B, NG * H), dt)   # activated gates
    cs = np.zeros((T, B, H), dt)           # c_t (LSTM)
    cprev = np.zeros((T, B, H), dt)        # c_{t-1} as seen by step t
    hprev = np.zeros((T, B, H), dt)        # h_{t-1} as seen by step t
    hn_pre = np.zeros((T, B, H), dt)       # GRU: W_hn h + b_hn
    hs = np.zeros((T, B, H), dt)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        act = lengths > t
        if not act.any():
            continue
        zh = h @ W_hh.T + b_hh
        if cell == "LSTM":
            g = Zx[:, t, :] + zh
            i = _sigmoid(g[:, 0 * H:1 * H])
            f = _sigmoid(g[:, 1 * H:2 * H])
            gg = np.tanh(g[:, 2 * H:3 * H])
            o = _sigmoid(g[:, 3 * H:4 * H])
            c_new = f * c + i * gg
            h_new = o * np.tanh(c_new)
            gates[t] = np.concatenate([i, f, gg, o], axis=1)
        else:
            zx = Zx[:, t, :]
            r = _sigmoid(zx[:, 0:H] + zh[:, 0:H])
            z = _sigmoid(zx[:, H:2 * H] + zh[:, H:2 * H])
            n = np.tanh(zx[:, 2 * H:] + r * zh[:, 2 * H:])
            h_new = (1.0 - z) * n + z * h
            c_new = c
            gates[t] = np.concatenate([r, z, n], axis=1)
            hn_pre[t] = zh[:, 2 * H:]
        hprev[t] = h
        cprev[t] = c
        cs[t] = c_new
        a = act[:, None]
        c = np.where(a, c_new, c)
        h = np.where(a, h_new, h)
        hs[t] = np.where(a, h_new, 0.0)
    return h, hs, dict(gates=gates, c=cs, cprev=cprev, hprev=hprev, hn_pre=hn_pre, cell=cell)


def rnn_dir_backward(dh_final, dhs, cache, lengths, W_hh, reverse):
    """BPTT for one direction of one layer (autograd of nn.LSTM / nn.GRU on packed input).
    dh_final [B,H] or None: gradient of the final hidden state (transfer/models.py:930,934);
    dhs [T,B,H] or None: gradient of the per-step outputs (from the layer above).
    Returns (dZx, dZh) [T,B,NG*H]: pre-activation gradients of the input half (feeds d input,
    dW_ih, db_ih) and of the hidden half (feeds dh_prev, dW_hh, db_hh); equal for the LSTM, and
    differing in the candidate gate for the GRU (its hidden half sits inside r * (.))."""
    gates, cs, cprev, hprev = cache["gates"], cache["c"], cache["cprev"], cache["hprev"]
    cell = cache["cell"]
    T, B, GH = gates.shape
    H = hprev.shape[2]
    dt = gates.dtype
    dZx = np.zeros((T, B, GH), dt)
    dZh = np.zeros((T, B, GH), dt)
    dh = np.zeros((B, H), dt) if dh_final is None else dh_final.astype(dt).copy()
    dc = np.zeros((B, H), dt)
    order = range(T) if reverse else range(T - 1, -1, -1)   # reverse of the forward order
    for t in order:
        act = (lengths > t)[:, None]
        if not act.any():
            continue
        dht = dh if dhs is None else dh + np.where(act, dhs[t], 0.0)
        if cell == "LSTM":
            i = gates[t][:, 0 * H:1 * H]
            f = gates[t][:, 1 * H:2 * H]
            g = gates[t][:, 2 * H:3 * H]
            o = gates[t][:, 3 * H:4 * H]
            tc = np.tanh(cs[t])
            do = dht * tc
            dct = dc + dht * o * (1.0 - tc * tc)
            zx = np.concatenate([dct * g * i * (1 - i), dct * cprev[t] * f * (1 - f), dct * i * (1 - g * g),
                                 do * o * (1 - o)], axis=1)
            zh = zx
            dh_direct = 0.0
            dc = np.where(act, dct * f, dc)
        else:
            r = gates[t][:, 0:H]
            z = gates[t][:, H:2 * H]
            n = gates[t][:, 2 * H:]
            da = dht * (1.0 - z) * (1.0 - n * n)
            dr = da * cache["hn_pre"][t] * r * (1 - r)
            dz = dht * (hprev[t] - n) * z * (1 - z)
            zx = np.concatenate([dr, dz, da], axis=1)
            zh = np.concatenate([dr, dz, da * r], axis=1)
            dh_direct = dht * z
        zx = np.where(act, zx, 0.0)
        zh = np.where(act, zh, 0.0)
        dZx[t] = zx
        dZh[t] = zh
        dh = np.where(act, zh @ W_hh + dh_direct, dh)
    return dZx, dZh


def lstm_dir_forward(X, lengths, W_ih, W_hh, b_ih, b_hh, reverse):
    """The shipped configuration's direction pass (kept under its round-1 name)."""
    h, _, cache = rnn_dir_forward(X, lengths, W_ih, W_hh, b_ih, b_hh, reverse, "LSTM")
    return h, cache


def lstm_dir_backward(dh_final, cache, lengths, W_ih, W_hh, reverse):
    return rnn_dir_backward(dh_final, None, cache, lengths, W_hh, reverse)[0]


_SFX = ("", "_reverse")


def rnn_forward(params, X, lengths, layer_drop=None):
    """All layers and directions (nn.LSTM / nn.GRU on the packed batch).  ``layer_drop`` = list of
    (keep [B,T,D*H] bool, scale) per layer boundary: the inter-layer dropout nn.LSTM applies in
    training mode (classifier/models.py:108).  Returns (final hidden of the LAST layer [B, D*H] —
    hidden.view(-1, D, B, H)[-1] of classifier/models.py:245,256 — and the per-layer caches)."""
    cell, D, layers = variant(params)
    inp = X
    caches = []
    for l in range(layers):
        hf, hs, cc = [], [], []
        for d in range(D):
            k = "_l%d%s" % (l, _SFX[d])
            h, o, c = rnn_dir_forward(inp, lengths, params["rnn.weight_ih" + k], params["rnn.weight_hh" + k],
                                      params["rnn.bias_ih" + k], params["rnn.bias_hh" + k], d == 1, cell)
            hf.append(h)
            hs.append(o.transpose(1, 0, 2))
            cc.append(c)
        caches.append(dict(inp=inp, dirs=cc))
        inp = np.concatenate(hs, axis=2)                  # [B,T,D*H] input of the next layer
        if layer_drop is not None and l + 1 < layers:
            inp = inp * (np.asarray(layer_drop[l][0]) * layer_drop[l][1]).astype(inp.dtype)
    return np.concatenate(hf, axis=1), caches


def rnn_backward(params, caches, dhcat, lengths, layer_drop=None, want_param_grads=False):
    """Backward of :func:`rnn_forward` given d(final hidden of the last layer).  Returns
    (d input of layer 0 [B,T,E], per-layer per-direction dZ_x, grads)."""
    cell, D, layers = variant(params)
    H = params["rnn.weight_hh_l0"].shape[1]
    grads = {}
    dZs = [None] * layers
    dout = None                                           # d per-step outputs of the current layer [B,T,D*H]
    for l in range(layers - 1, -1, -1):
        inp = caches[l]["inp"]
        dinp = np.zeros_like(inp)
        dZs[l] = []
        for d in range(D):
            k = "_l%d%s" % (l, _SFX[d])
            dfin = dhcat[:, d * H:(d + 1) * H] if l == layers - 1 else None
            dhs = None if dout is None else dout[:, :, d * H:(d + 1) * H].transpose(1, 0, 2)
            c = caches[l]["dirs"][d]
            dZx, dZh = rnn_dir_backward(dfin, dhs, c, lengths, params["rnn.weight_hh" + k], d == 1)
            dZs[l].append(dZx)
            dinp = dinp + np.einsum("tbg,ge->bte", dZx, params["rnn.weight_ih" + k])
            if want_param_grads:
                grads["rnn.weight_ih" + k] = np.einsum("tbg,bte->ge", dZx, inp)
                grads["rnn.weight_hh" + k] = np.einsum("tbg,tbh->gh", dZh, c["hprev"])
                grads["rnn.bias_ih" + k] = dZx.sum(axis=(0, 1))
                grads["rnn.bias_hh" + k] = dZh.sum(axis=(0, 1))
        if l > 0 and layer_drop is not None:
            dinp = dinp * (np.asarray(layer_drop[l - 1][0]) * layer_drop[l - 1][1]).astype(dinp.dtype)
        dout = dinp
    return dout, dZs, grads


def forward(params, src, lengths, mask, soft, labels=None, mutate_mask=True, guard=True, drop=None,
            layer_drop=None):
    """Full discriminator forward (transfer/models.py:885-953; attention :841-883;
    per-sentence CE = nn.CrossEntropyLoss(reduction='none'), transfer/models.py:990,1212).

    src: [B,T,V] float (soft) or [B,T] int;  lengths [B] (sorted desc, >= 2);  mask [B,T] bool,
    True = pad/sos/eos.  ``guard`` applies the all-masked-row fix of the transfer copy (the
    classifier copy, classifier/models.py:161-197, has none).
    ``drop`` = (keep [B,T,E] bool, scale) applies inverted dropout to the embedded tokens
    (classifier copy in training mode, classifier/models.py:217); ``layer_drop`` the inter-layer
    dropout of a multi-layer RNN (see :func:`rnn_forward`).
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
    hcat, rnn_caches = rnn_forward(params, X, lengths, layer_drop)             # :929-934
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
    cache = dict(X=X, rnn=rnn_caches, layer_drop=layer_drop, hcat=hcat, q=q, u=u, Kk=Kk, mid=mid, a=a, ctx=ctx,
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
    dXr, dZs, rnn_grads = rnn_backward(params, cache["rnn"], dhcat, lengths, cache.get("layer_drop"), want_param_grads)
    dX = dKk @ params["linear_emb.weight"] + dXr
    dXd = dX                                    # gradient w.r.t. the (dropped) LSTM / key input
    if cache.get("drop") is not None:           # ... and w.r.t. the embedding output in front of the dropout
        dX = dX * (np.asarray(cache["drop"][0]) * cache["drop"][1]).astype(dt)
    out = dict(dX=dX, dXd=dXd, dKk=dKk, dZf=dZs[0][0], dZb=dZs[0][-1], dZs=dZs, du=du, dq=dq, dhcat=dhcat, dctx=dctx)
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
        g.update(rnn_grads)
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
