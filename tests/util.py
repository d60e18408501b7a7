"""Shared helpers for the test-suite (golden loading, tolerances)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    hdr = json.loads(bytes(d.pop("header_json")).decode())
    params = {k[len("param:"):]: v for k, v in d.items() if k.startswith("param:")}
    grads = {k[len("grad:"):]: v for k, v in d.items() if k.startswith("grad:")}
    return hdr, d, params, grads


def relerr(a, b, floor=1e-30):
    """max |a-b| / max(max|b|, floor) — scale-aware max error (floor: for quantities that are
    analytically zero, e.g. the gradient of linear_attn.bias under a shift-invariant softmax)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), floor))


# ---------------------------------------------------------------------------------------------
# GPU-side helpers (import torch lazily so the CPU-only tests stay light)
# ---------------------------------------------------------------------------------------------
def gate_cols(H):
    """index array: packed gate column (libsad layout) for torch row g*H + unit."""
    unit = np.arange(H)
    return np.concatenate([(unit >> 4) * 64 + g * 16 + (unit & 15) for g in range(4)])


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32, in numpy."""
    a = np.ascontiguousarray(x, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(a.shape)


def run_sad(params, src, lengths, mask, labels=None, soft=True, s_dtype="f32", time_major=True,
            dloss=None, fused_loss=True, guard=True, want_ws=False, dweights=None, param_grads=False,
            frozen=(), pad_id=None, drop=None):
    """Run libsad forward (+ backward when soft) on cuda:0 through the Python binding.

    src: numpy [B,T,V] float / [B,T] int64 (batch-major, like the oracle takes it).
    drop: (keep [T,B,E] bool numpy, scale) dropout on the embedded tokens.
    param_grads: also back-propagate into the parameters (all but the names in `frozen`);
    out["grads"] = {name: numpy}.
    Returns dict of numpy arrays (logits, weights, loss, mask_after, dS [B,T,V], ws fields)."""
    import torch
    from sad_amd import functional as Fn
    Fn.KEEP_LAST = True           # debug hook: keep the call's workspace reachable for inspection
    dev = torch.device("cuda:0")
    tp = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in params.items()}
    pw = Fn.pack_weights(tp)
    live = None
    if param_grads:
        live = {k: tp[k] for k in Fn._LIVE_PARAMS}
        for k, v in live.items():
            v.requires_grad_(k not in frozen)
    B, T = src.shape[:2]
    if soft:
        tdt = torch.float32 if s_dtype == "f32" else torch.bfloat16
        S = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32))
        if time_major:   # storage [T,B,V] viewed as [B,T,V] — what the transfer loop passes
            S = S.transpose(0, 1).contiguous().to(dev).to(tdt).transpose(0, 1)
        else:
            S = S.to(dev).to(tdt)
        S.requires_grad_(True)
        x = S
    else:
        x = torch.from_numpy(np.ascontiguousarray(src, dtype=np.int64)).to(dev)
    m = torch.from_numpy(mask.copy()).to(dev)
    ln = torch.from_numpy(np.asarray(lengths).astype(np.int32))
    out = {}
    if drop is not None:
        drop = (torch.from_numpy(np.ascontiguousarray(drop[0])).to(dev), float(drop[1]))
    if labels is not None and fused_loss:
        lab = torch.from_numpy(np.asarray(labels, dtype=np.int64)).to(dev)
        loss, logits, weights = Fn.discriminator_loss(x, ln, m, lab, pw, guard=guard, params=live, pad_id=pad_id, drop=drop)
    else:
        logits, weights = Fn.discriminator(x, ln, m, pw, guard=guard, params=live, pad_id=pad_id, drop=drop)
        loss = None
        if labels is not None:
            lab = torch.from_numpy(np.asarray(labels, dtype=np.int64)).to(dev)
            loss = torch.nn.functional.cross_entropy(logits, lab, reduction="none")
    ws = Fn._Discriminator.last_workspace
    if (soft or param_grads) and loss is not None:
        up = torch.full((B,), 1.0 / B, device=dev) if dloss is None else torch.from_numpy(
            np.asarray(dloss, dtype=np.float32)).to(dev)
        tot = (loss * up).sum()
        if dweights is not None:
            tot = tot + (weights * torch.from_numpy(np.asarray(dweights, dtype=np.float32)).to(dev)).sum()
        tot.backward()
        if soft:
            out["dS"] = S.grad.float().cpu().numpy()
            out["dS_strides"] = tuple(S.grad.stride())
            out["S_strides"] = tuple(S.stride())
        if param_grads:
            out["grads"] = {k: (None if v.grad is None else v.grad.cpu().numpy()) for k, v in live.items()}
    torch.cuda.synchronize()
    ws.check()                    # raises if a hand-off wait of the persistent recurrences timed out
    out.update(logits=logits.detach().cpu().numpy(), weights=weights.detach().cpu().numpy(),
               loss=None if loss is None else loss.detach().cpu().numpy(), mask_after=m.cpu().numpy())
    if want_ws:
        from sad_amd import lib
        pd = lib.padded_dims(ws.d)
        Ep, Hp, Gp, Ncat = pd["Ep"], pd["Hp"], pd["Gp"], pd["Ncat"]
        f = lambda name, dt: ws.view(name, dt)
        import ctypes
        Gact = f("G", torch.float32).cpu().numpy().reshape(T, B, 2 * Gp + 64)[:, :, :2 * Gp].reshape(T, B, 2, Gp)
        if lib.load().sad_uses_half_gates(ctypes.byref(ws.d)) == 1:
            # persistent recurrences: activated gates are fp16 [T,B,2,Hp,(i,f,g,o)]; bring them to the
            # permuted-column layout of G: pc = (unit/16)*64 + gate*16 + unit%16
            Bp = (B + 15) // 16 * 16       # records: [T][2][B/16][Hp/4][16 sentences][4 units][4 gates]
            ga = f("GA", torch.float16).float().cpu().numpy().reshape(T, 2, Bp // 16, Hp // 4, 16, 4, 4)
            ga = ga.transpose(0, 2, 4, 1, 3, 5, 6).reshape(T, Bp, 2, Hp, 4)[:, :B]
            Gact = ga.reshape(T, B, 2, Hp // 16, 16, 4).transpose(0, 1, 2, 3, 5, 4).reshape(T, B, 2, Gp).copy()
        out["ws"] = dict(
            pd=pd,
            LENS=f("LENS", torch.int32).cpu().numpy(),
            NEED=f("NEED", torch.uint8).cpu().numpy().reshape(T, B),
            X=f("X", torch.bfloat16).float().cpu().numpy().reshape(T, B, Ep),
            G=Gact,
            CS=f("CS", torch.float32).cpu().numpy().reshape(T, B, 2, Hp),
            HS=f("HS", torch.bfloat16).float().cpu().numpy().reshape(T, B, 2, Hp),
            KK=f("KK", torch.float32).cpu().numpy().reshape(T, B, Hp),
            Q=f("Q", torch.float32).cpu().numpy().reshape(B, Hp),
            U=f("U", torch.float32).cpu().numpy().reshape(B, Hp),
            CTX=f("CTX", torch.float32).cpu().numpy().reshape(B, Hp),
            DZ=f("DZ", torch.bfloat16).float().cpu().numpy().reshape(T, B, Ncat),
            DX=f("DX", torch.bfloat16).float().cpu().numpy().reshape(T, B, Ep),
            DU=f("DU", torch.float32).cpu().numpy().reshape(B, Hp),
            DQ=f("DQ", torch.float32).cpu().numpy().reshape(B, Hp),
            DHFIN=f("DHFIN", torch.float32).cpu().numpy().reshape(B, 2, Hp),
        )
    return out


def stage_errors(ws, f, b, H, E):
    """Scale-aware max error of every workspace stage against oracle intermediates.
    f, b: outputs of oracle forward()/backward() (batch-major).  Returns ordered dict name->err."""
    from collections import OrderedDict
    c = f["cache"]
    lengths = c["lengths"]
    T, B = ws["NEED"].shape
    Gp, Hp = ws["pd"]["Gp"], ws["pd"]["Hp"]
    gc = gate_cols(H)
    act = (np.arange(T)[:, None] < lengths[None, :])            # [T,B]
    errs = OrderedDict()

    def rel(a, r, sel=None):
        a = np.asarray(a, np.float64)
        r = np.asarray(r, np.float64)
        if sel is not None:
            a, r = a[sel], r[sel]
        return float(np.max(np.abs(a - r)) / max(np.max(np.abs(r)), 1e-30)) if r.size else 0.0

    errs["X"] = rel(ws["X"][:, :, :E], c["X"].transpose(1, 0, 2), act)
    for d in (0, 1):
        cd = c["rnn"][0]["dirs"][d]                              # layer 0, direction d
        errs["G%d" % d] = rel(ws["G"][:, :, d, :][:, :, gc], cd["gates"], act)
        errs["CS%d" % d] = rel(ws["CS"][:, :, d, :H], cd["c"], act)
        g = cd["gates"]
        h = g[:, :, 3 * H:4 * H] * np.tanh(cd["c"])
        errs["HS%d" % d] = rel(ws["HS"][:, :, d, :H], h, act)
    att = ~c["mask"].T                                           # [T,B] attention-active
    errs["KK"] = rel(ws["KK"][:, :, :H], c["Kk"].transpose(1, 0, 2), att | act)
    errs["Q"] = rel(ws["Q"][:, :H], c["q"])
    errs["U"] = rel(ws["U"][:, :H], c["u"])
    errs["CTX"] = rel(ws["CTX"][:, :H], c["ctx"])
    if b is not None:
        errs["DZ_kk"] = rel(ws["DZ"][:, :, 2 * Gp:2 * Gp + H], b["dKk"].transpose(1, 0, 2))
        errs["DU"] = rel(ws["DU"][:, :H], b["du"])
        errs["DQ"] = rel(ws["DQ"][:, :H], b["dq"])
        errs["DHFIN"] = rel(np.concatenate([ws["DHFIN"][:, 0, :H], ws["DHFIN"][:, 1, :H]], axis=1), b["dhcat"])
        errs["DZ_f"] = rel(ws["DZ"][:, :, 0:Gp][:, :, gc], b["dZf"])
        errs["DZ_b"] = rel(ws["DZ"][:, :, Gp:2 * Gp][:, :, gc], b["dZb"])
        errs["DX"] = rel(ws["DX"][:, :, :E], b["dX"].transpose(1, 0, 2))
        # padding columns / rows must be exact zeros
        errs["DX_pad"] = float(np.abs(ws["DX"][:, :, E:]).max()) if ws["DX"].shape[2] > E else 0.0
        errs["DZ_inactive"] = float(np.abs(ws["DZ"][~act]).max()) if (~act).any() else 0.0
    return errs
