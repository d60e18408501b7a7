"""GPU parity at the BASELINE.json sizes, and robustness of the C-ABI path (``pytest -m gpu``).

* BASELINE ``configs[2]`` (cfg3: B=1024, T=32, V=9353, dense bf16 soft tokens) at its FULL size — the
  exact benchmarked path (stream-K K1 with the ragged last k-tile of V=9353, two-per-CU recurrence):
  a 16-sentence subsample against the oracle run on just those sentences (sentences are independent,
  SURVEY.md §8 e1).
* BASELINE ``configs[4]`` per-GPU slice (cfg5 "stress": B=512, T=128, V=50 000): an 8-sentence
  subsample against the oracle + size-independent properties at full size (exact zeros for t >= len,
  linearity in dloss, half-batch == full batch, repeatability, hand-off timeout words clear).
* a forced hand-off timeout must end in NaN outputs and a RuntimeError, never in silent garbage;
* repeated calls do not grow device memory (no ctx <-> output reference cycles);
* stage entry points reproduce the full pass; dlogits and dloss arriving together.

Tolerances: as stated in tests/test_gpu_parity.py (bf16 MFMA inputs, fp32 accumulation: 2e-2 scale-aware
max error on logits / weights / loss, 3e-2 on a bf16-stored dS).
"""
import ctypes

import numpy as np
import pytest

from oracle import sad_oracle as O
from sad_amd import synth
from tests.util import relerr

pytestmark = pytest.mark.gpu

TOL_FWD, TOL_BWD = 2e-2, 3e-2


def _device_batch(w, seed):
    """The bench's synthetic batch (dense softmax soft tokens built on the device)."""
    import torch
    import bench
    dev = torch.device("cuda:0")
    S, lengths, ln_dev, mask, labels = bench.make_inputs(w, seed, dev)
    return dev, S, lengths, ln_dev, mask, labels


def _oracle_subsample(params, S_tbv, lengths, mask, labels, sel, B):
    """Oracle forward + backward (fp64) on the sentences `sel` of a time-major device batch."""
    import torch
    Ssel = S_tbv[:, torch.as_tensor(sel, device=S_tbv.device)].float().cpu().numpy().transpose(1, 0, 2)   # [b,T,V]
    p = O.cast_params(params, np.float64)
    msel = mask[torch.as_tensor(sel, device=mask.device)].cpu().numpy()
    lab = labels[torch.as_tensor(sel, device=labels.device)].cpu().numpy()
    f = O.forward(p, Ssel.astype(np.float64), lengths[sel], msel.copy(), True, labels=lab)
    dl = O.ce_dlogits(f["logits"], lab, np.full((len(sel),), 1.0 / B))
    b = O.backward(p, f["cache"], dl)
    return f, b


def _step(Fn, S_tbv, ln, mask, labels, pw, scale=1.0, lo=None, hi=None):
    B = S_tbv.shape[1]
    sl = slice(lo, hi)
    Sv = S_tbv[:, sl].transpose(0, 1).detach().requires_grad_(True)     # [b,T,V] view of time-major storage
    loss, logits, w = Fn.discriminator_loss(Sv, ln[sl], mask[sl].clone(), labels[sl], pw)
    (loss.sum() * (scale / B)).backward()
    return loss.detach(), logits.detach(), w.detach(), Sv.grad


def test_cfg3_full_size_dense_subsample_vs_oracle():
    import torch
    import bench
    from sad_amd import functional as Fn
    Fn.KEEP_LAST = True
    w = dict(bench.WORKLOADS["cfg3"])
    dev, S, lengths, ln_dev, mask, labels = _device_batch(w, 1234)
    B, T, V = w["B"], w["T"], w["V"]
    params = synth.make_params(V, w["E"], w["H"], w["L"], seed=21)
    pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
    loss, logits, wts, dS = _step(Fn, S, ln_dev, mask, labels, pw)
    torch.cuda.synchronize()
    Fn._Discriminator.last_workspace.check()
    sel = np.linspace(0, B - 1, 16).astype(int)
    f, b = _oracle_subsample(params, S, lengths, mask, labels, sel, B)
    ts = torch.as_tensor(sel, device=dev)
    errs = dict(logits=relerr(logits[ts].cpu().numpy(), f["logits"]), weights=relerr(wts[ts].cpu().numpy(), f["weights"]),
                loss=relerr(loss[ts].cpu().numpy(), f["loss"]), dS=relerr(dS[ts].float().cpu().numpy(), b["dS"]))
    print("\n[cfg3 full size, dense bf16 S, 16-sentence subsample vs oracle]", errs)
    assert errs["logits"] < TOL_FWD and errs["weights"] < TOL_FWD and errs["loss"] < TOL_FWD, errs
    assert errs["dS"] < TOL_BWD, errs
    pad = torch.arange(T, device=dev)[None, :] >= ln_dev[:, None]
    assert bool((dS[pad] == 0).all())
    # run-to-run: every summation order on this path is fixed
    loss2, logits2, _, dS2 = _step(Fn, S, ln_dev, mask, labels, pw)
    assert torch.equal(logits, logits2) and torch.equal(dS, dS2)


def test_stress_slice_cfg5():
    """BASELINE configs[4], one GPU's share: B=512, T=128, V=50 000, E=300, H=512, bf16 soft tokens."""
    import torch
    import bench
    from sad_amd import functional as Fn
    Fn.KEEP_LAST = True
    w = dict(bench.WORKLOADS["stress"])
    dev, S, lengths, ln_dev, mask, labels = _device_batch(w, 4321)
    B, T, V = w["B"], w["T"], w["V"]
    params = synth.make_params(V, w["E"], w["H"], w["L"], seed=22)
    pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
    loss, logits, wts, dS = _step(Fn, S, ln_dev, mask, labels, pw)
    torch.cuda.synchronize()
    Fn._Discriminator.last_workspace.check()          # hand-off timeout words of both recurrences
    assert bool(torch.isfinite(logits).all()) and bool(torch.isfinite(dS.float()).all())
    # --- oracle on 8 sentences spread over the batch (long and short ones)
    sel = np.linspace(0, B - 1, 8).astype(int)
    f, b = _oracle_subsample(params, S, lengths, mask, labels, sel, B)
    ts = torch.as_tensor(sel, device=dev)
    errs = dict(logits=relerr(logits[ts].cpu().numpy(), f["logits"]), weights=relerr(wts[ts].cpu().numpy(), f["weights"]),
                loss=relerr(loss[ts].cpu().numpy(), f["loss"]), dS=relerr(dS[ts].float().cpu().numpy(), b["dS"]))
    print("\n[stress slice B=512 T=128 V=50000, 8-sentence subsample vs oracle]", errs)
    # a 128-step bf16 recurrence: same stated tolerance as T=100 in test_long_sequences_vs_oracle
    assert errs["logits"] < TOL_FWD and errs["weights"] < TOL_FWD and errs["loss"] < TOL_FWD, errs
    assert errs["dS"] < TOL_BWD, errs
    del f, b
    # --- properties at full size
    assert torch.allclose(wts.sum(dim=1), torch.ones(B, device=dev), atol=1e-5)
    pad = torch.arange(T, device=dev)[None, :] >= ln_dev[:, None]
    assert bool((dS[pad] == 0).all()) and bool((dS[~pad].float().abs().sum() > 0))
    _, logits_r, _, dS_r = _step(Fn, S, ln_dev, mask, labels, pw)
    assert torch.equal(logits, logits_r) and torch.equal(dS, dS_r)              # repeatable bit for bit
    del dS_r
    _, _, _, dS2 = _step(Fn, S, ln_dev, mask, labels, pw, scale=2.0)
    assert torch.equal(dS2.float(), 2.0 * dS.float())                           # linear in dloss, bit-exact
    del dS2
    # data parallelism: the two half batches scored separately (what 2 ranks do).  Dense soft tokens:
    # the stream-K split points of K1 move with the batch, so sums are regrouped -> rounding-level
    # differences (fp32 outputs 2e-3, bf16 dS one ulp = 8e-3)
    h = B // 2
    for lo, hi in ((0, h), (h, B)):
        _, plg, pw_, pdS = _step(Fn, S, ln_dev, mask, labels, pw, lo=lo, hi=hi)      # same 1/B_full upstream scale
        assert relerr(plg.cpu().numpy(), logits[lo:hi].cpu().numpy()) < 2e-3
        assert relerr(pw_.cpu().numpy(), wts[lo:hi].cpu().numpy()) < 2e-3
        d = (pdS.float() - dS[lo:hi].float()).abs().max().item() / dS[lo:hi].float().abs().max().item()
        assert d < 1e-2, d
        del pdS
    torch.cuda.synchronize()
    Fn._Discriminator.last_workspace.check()


def test_handoff_timeout_is_loud():
    """A workgroup of a persistent recurrence is withheld (debug setting) and the wait bound made tiny:
    its group's hand-off waits time out.  The call must come back (bounded spins) with NaN logits /
    loss (forward) resp. NaN d(soft tokens) (BPTT), and Workspace.check() must raise; the next sad_forward
    refuses once with the reason (host-mapped time-out word), the call after it, with the settings restored, is
    clean and identical to the call before."""
    import torch
    from sad_amd import functional as Fn, lib
    Fn.KEEP_LAST = True
    dev = torch.device("cuda:0")
    B, T, V, E, H = 256, 12, 300, 300, 512
    lengths = synth.make_lengths(B, T, seed=5)
    params = synth.make_params(V, E, H, seed=3)
    pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
    S0 = torch.from_numpy(synth.make_soft_tokens(B, T, V, lengths, seed=5, kind="dense", time_major=False)).to(dev)
    ln = torch.from_numpy(lengths)
    mask0 = torch.from_numpy(synth.transfer_mask(lengths, T)).to(dev)
    lab = torch.from_numpy(synth.make_labels(B, seed=1)).to(dev)
    d = lib.dims(B, T, V, E, H, 2)
    assert lib.load().sad_uses_half_gates(ctypes.byref(d)) == 1, "persistent recurrences expected at this shape"

    def run():
        S = S0.clone().requires_grad_(True)
        loss, logits, w = Fn.discriminator_loss(S, ln, mask0.clone(), lab, pw)
        ws = Fn._Discriminator.last_workspace
        loss.mean().backward()
        torch.cuda.synchronize()
        return loss.detach(), logits.detach(), S.grad, ws

    loss0, logits0, dS0, ws0 = run()
    ws0.check()
    assert bool(torch.isfinite(logits0).all()) and bool(torch.isfinite(dS0).all())
    try:
        lib.debug_set(lib.DBG_SPIN_LIMIT, 2000)
        lib.debug_set(lib.DBG_WITHHOLD_FWD, 0)
        loss1, logits1, dS1, ws1 = run()
        assert ws1.handoff_timeouts()[0] != 0
        assert bool(torch.isnan(logits1).all()) and bool(torch.isnan(loss1).all())      # whole batch poisoned
        with pytest.raises(RuntimeError, match="timed out"):
            ws1.check()
        lib.debug_set(lib.DBG_WITHHOLD_FWD, -1)
        # the recurrence also raised the library's host-mapped word: the next sad_forward on the device refuses
        # with the reason (no synchronisation involved) and clears it
        with pytest.raises(RuntimeError, match="EARLIER call on this device timed out"):
            run()
        lib.debug_set(lib.DBG_WITHHOLD_BWD, 0)
        loss2, logits2, dS2, ws2 = run()
        assert torch.equal(logits2, logits0)                                            # the forward was clean
        assert ws2.handoff_timeouts() == (0, 1) or ws2.handoff_timeouts()[1] != 0
        assert bool(torch.isnan(dS2).any())                                             # the stalled group's sentences
        with pytest.raises(RuntimeError, match="BPTT"):
            ws2.check()
    finally:
        lib.debug_set(lib.DBG_WITHHOLD_FWD, -1)
        lib.debug_set(lib.DBG_WITHHOLD_BWD, -1)
        lib.debug_set(lib.DBG_SPIN_LIMIT, 0)
    with pytest.raises(RuntimeError, match="BPTT of an EARLIER call"):
        run()
    loss3, logits3, dS3, ws3 = run()
    ws3.check()
    assert torch.equal(logits3, logits0) and torch.equal(dS3, dS0)


def test_repeated_calls_do_not_leak_device_memory():
    """outputs kept on ctx closed a cycle output -> grad_fn -> ctx -> output that the
    GC cannot see; every call leaked its workspace.  Forward (+ backward, and forward only) in a loop
    must leave torch.cuda.memory_allocated() flat."""
    import gc
    import torch
    from sad_amd import functional as Fn
    Fn.KEEP_LAST = False
    Fn._Discriminator.last_workspace = None
    Fn._Discriminator.last_grad_bucket = None
    dev = torch.device("cuda:0")
    B, T, V, E, H = 64, 23, 1000, 300, 512
    lengths = synth.make_lengths(B, T, seed=2)
    params = synth.make_params(V, E, H, seed=2)
    tp = {k: torch.from_numpy(v).to(dev) for k, v in params.items()}
    pw = Fn.pack_weights(tp)
    live = {k: tp[k].requires_grad_(True) for k in Fn._LIVE_PARAMS}
    S0 = torch.from_numpy(synth.make_soft_tokens(B, T, V, lengths, seed=2, kind="dense", time_major=False)).to(dev)
    ln = torch.from_numpy(lengths)
    mask0 = torch.from_numpy(synth.transfer_mask(lengths, T)).to(dev)
    lab = torch.from_numpy(synth.make_labels(B, seed=2)).to(dev)

    def one(i):
        S = S0.clone().requires_grad_(True)
        if i % 3 == 0:      # forward only (outputs dropped without a backward)
            Fn.discriminator(S, ln, mask0.clone(), pw)
        elif i % 3 == 1:    # frozen step
            loss, _, _ = Fn.discriminator_loss(S, ln, mask0.clone(), lab, pw)
            loss.mean().backward()
        else:               # unfrozen step
            loss, _, _ = Fn.discriminator_loss(S, ln, mask0.clone(), lab, pw, params=live, pad_id=0)
            loss.mean().backward()
            for p in live.values():
                p.grad = None

    for i in range(3):
        one(i)
    gc.collect()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for i in range(30):
        one(i)
    torch.cuda.synchronize()
    grown = torch.cuda.memory_allocated() - base
    ws_bytes = lib_ws_bytes(B, T, V, E, H)
    assert grown < ws_bytes // 2, "device memory grew by %d bytes over 30 calls (one workspace = %d)" % (grown, ws_bytes)
    Fn.KEEP_LAST = True


def lib_ws_bytes(B, T, V, E, H):
    from sad_amd import lib
    d = lib.dims(B, T, V, E, H, 2)
    return int(lib.load().sad_workspace_bytes(ctypes.byref(d)))


def test_soft_embedding_results_survive_the_next_call():
    """with bf16 soft tokens the result used to be a view of the shared per-shape
    workspace, overwritten by the next call of the same shape."""
    import torch
    from sad_amd import functional as Fn
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    V, E = 777, 300
    W = torch.randn((V, E), generator=g, device=dev) * 0.05
    for dtype in (torch.bfloat16, torch.float32):
        S1 = torch.softmax(torch.randn((5, 64, V), generator=g, device=dev), dim=-1).to(dtype)
        S2 = torch.softmax(torch.randn((5, 64, V), generator=g, device=dev) * 3, dim=-1).to(dtype)
        o1 = Fn.soft_embedding(S1, W)
        keep = o1.clone()
        o2 = Fn.soft_embedding(S2, W)
        assert torch.equal(o1, keep) and not torch.equal(o1, o2)
        ref = (S1.float() @ W.to(torch.bfloat16).float())
        assert relerr(o1.float().cpu().numpy(), ref.cpu().numpy()) < 6e-3


def test_stage_entry_points_and_joint_upstream_gradients():
    """sad_lstm_fwd / sad_attnpool_fwd re-run on a filled workspace reproduce the full pass bit for bit
    (SURVEY.md §8 b3 stage entries), sad_attnpool_bwd + sad_lstm_bwd reproduce DZ; and a backward
    that receives BOTH dlogits and dloss (fused CE plus another head on the logits) matches the oracle."""
    import torch
    from sad_amd import functional as Fn, lib
    Fn.KEEP_LAST = True
    dev = torch.device("cuda:0")
    B, T, V, E, H = 96, 19, 400, 300, 512
    lengths = synth.make_lengths(B, T, seed=11)
    params = synth.make_params(V, E, H, seed=12)
    pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
    S_np = synth.make_soft_tokens(B, T, V, lengths, seed=13, kind="dense", time_major=False)
    mask_np = synth.transfer_mask(lengths, T)
    lab_np = synth.make_labels(B, seed=14)
    R = np.random.RandomState(15).standard_normal((B, 2)).astype(np.float32) * 0.3
    S = torch.from_numpy(S_np).to(dev).requires_grad_(True)
    lab = torch.from_numpy(lab_np).to(dev)
    loss, logits, w = Fn.discriminator_loss(S, torch.from_numpy(lengths), torch.from_numpy(mask_np.copy()).to(dev), lab, pw)
    ws = Fn._Discriminator.last_workspace
    (loss.mean() + (logits * torch.from_numpy(R).to(dev)).sum()).backward()
    torch.cuda.synchronize()
    ws.check()
    # oracle: dlogits = CE part (1/B) + R
    p = O.cast_params(params, np.float64)
    f = O.forward(p, S_np.astype(np.float64), lengths, mask_np.copy(), True, labels=lab_np)
    dl = O.ce_dlogits(f["logits"], lab_np, np.full((B,), 1.0 / B)) + R.astype(np.float64)
    b = O.backward(p, f["cache"], dl)
    errs = dict(logits=relerr(logits.detach().cpu().numpy(), f["logits"]), dS=relerr(S.grad.cpu().numpy(), b["dS"]))
    print("\n[dlogits + dloss together]", errs)
    assert errs["logits"] < TOL_FWD and errs["dS"] < TOL_BWD, errs

    # --- stage entries on the filled workspace
    L = lib.load()
    d = ws.d
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    assert L.sad_uses_half_gates(ctypes.byref(d)) == 1
    hs0 = ws.view("HS", torch.bfloat16).clone()
    cs0 = ws.view("CS", torch.float32).clone()
    dz0 = ws.view("DZ", torch.bfloat16).clone()
    live = (ws.view("NEED", torch.uint8).reshape(T, B) & 1).bool()
    ws.view("HS", torch.bfloat16).zero_()
    # the recurrence consumed the pre-activation records of GA (it overwrites them with the activated gates): the
    # in-projection stage rebuilds them from X, then the recurrence stage can run again
    ga0 = ws.view("GA", torch.int16).clone()               # (bit patterns: rows nobody writes hold whatever was there)
    lib.check(L.sad_inproj_fwd(ctypes.byref(d), ptr(pw.buf), ptr(ws.buf), st), "sad_inproj_fwd")
    lib.check(L.sad_lstm_fwd(ctypes.byref(d), ptr(pw.buf), ptr(ws.buf), st), "sad_lstm_fwd")
    torch.cuda.synchronize()
    ws.check()
    hp = ws.d.H + (-ws.d.H) % 32
    assert torch.equal(ws.view("HS", torch.bfloat16).reshape(T, B, 2 * hp)[live], hs0.reshape(T, B, 2 * hp)[live])
    assert torch.equal(ws.view("CS", torch.float32).reshape(T, B, 2 * hp)[live], cs0.reshape(T, B, 2 * hp)[live])
    assert torch.equal(ws.view("GA", torch.int16), ga0)       # activated gate records, bit for bit
    lg2 = torch.empty_like(logits)
    w2 = torch.empty_like(w)
    ls2 = torch.empty_like(loss)
    lib.check(L.sad_attnpool_fwd(ctypes.byref(d), ptr(pw.buf), ptr(lab), ptr(lg2), ptr(w2), ptr(ls2), ptr(ws.buf), st),
              "sad_attnpool_fwd")
    torch.cuda.synchronize()
    assert torch.equal(lg2, logits.detach()) and torch.equal(w2, w.detach()) and torch.equal(ls2, loss.detach())
    a = lib.SadBwdArgs()
    dlg = torch.from_numpy(R).to(dev).contiguous()
    a.dlogits, a.labels, a.dloss_scale = dlg.data_ptr(), lab.data_ptr(), 1.0 / B
    a.logits, a.weights = lg2.data_ptr(), w2.data_ptr()
    ws.view("DZ", torch.bfloat16).zero_()
    lib.check(L.sad_attnpool_bwd(ctypes.byref(d), ptr(pw.buf), ctypes.byref(a), ptr(ws.buf), st), "sad_attnpool_bwd")
    lib.check(L.sad_lstm_bwd(ctypes.byref(d), ptr(pw.buf), ptr(ws.buf), st), "sad_lstm_bwd")
    torch.cuda.synchronize()
    ws.check()
    assert torch.equal(ws.view("DZ", torch.bfloat16), dz0)


@pytest.mark.parametrize("V", [9, 33, 64, 95, 130, 1000, 4097])
def test_soft_token_gradient_alignment_edges(V):
    """K1' (d soft tokens, bf16, E = 300 -> the A-stationary kernel with 64-byte-aligned stores): vocabulary sizes
    below / across / far above one 64-column tile, row pitches of every alignment class, a storage offset that
    moves the base address off the 64-byte grid, time-major and batch-major layouts.  Every element of dS must
    match the reference contraction dX . W^T (bf16 inputs, fp32 accumulation, bf16 result), and the elements
    around the tensor must stay untouched (the aligned stores clip at the row ends)."""
    import torch
    from sad_amd import functional as Fn
    dev = torch.device("cuda:0")
    E = 300
    g = torch.Generator(device=dev)
    g.manual_seed(V)
    W = (torch.randn((V, E), generator=g, device=dev) * 0.1)
    for (T, B, batch_major, off) in ((5, 37, False, 0), (3, 300, False, 3), (7, 19, True, 5)):
        guard = 64
        flat = torch.full((guard + off + T * B * V + guard,), 7.0, dtype=torch.bfloat16, device=dev)
        body = flat[guard + off: guard + off + T * B * V]
        if batch_major:
            S = body.view(B, T, V)
        else:
            S = body.view(T, B, V).transpose(0, 1)           # [B,T,V] view of time-major storage
        S.copy_(torch.softmax(torch.randn((B, T, V), generator=g, device=dev), -1).to(torch.bfloat16))
        S = S.detach().requires_grad_(True)
        G = torch.randn((B, T, E), generator=g, device=dev).to(torch.bfloat16)
        Fn.soft_embedding(S, W).backward(G)
        want = (G.float() @ W.to(torch.bfloat16).float().t())
        got = S.grad.float()
        assert got.shape == want.shape
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 1e-2 * scale + 1e-6, (V, T, B, batch_major, off)
        # the gradient keeps the layout of S, so it is a view-shaped tensor of its own: nothing outside may be written;
        # check the guard bands of the INPUT storage too (soft tokens are read-only for the kernels)
        assert bool((flat[:guard + off] == 7.0).all()) and bool((flat[-guard:] == 7.0).all())


@pytest.mark.parametrize("V", [9, 64, 95, 4097])
def test_soft_token_gradient_writes_stay_inside_rows(V):
    """Same kernel through the stage entry sad_softemb_bwd with the OUTPUT inside a guarded buffer: the aligned
    stores start up to 31 elements in front of a row's first column and must clip there, at V, and at the column
    ranges of the workgroups — nothing around the tensor may change, every element inside must."""
    import ctypes
    import torch
    from sad_amd import functional as Fn, lib as L
    dev = torch.device("cuda:0")
    T, B, E = 6, 70, 300
    g = torch.Generator(device=dev)
    g.manual_seed(100 + V)
    W = torch.randn((V, E), generator=g, device=dev) * 0.1
    pw = Fn._se_packed(W)
    d, ws = Fn._se_workspace(B, T, V, E, dev)
    Ep = L.padded_dims(d)["Ep"]
    dx = ws.view("DX", torch.bfloat16)[:T * B * Ep].view(T * B, Ep)
    dx.zero_()
    dx[:, :E] = torch.randn((T * B, E), generator=g, device=dev).to(torch.bfloat16)
    want = dx[:, :E].float() @ W.to(torch.bfloat16).float().t()
    for off in (0, 1, 13, 32):
        guard = 96
        flat = torch.full((guard + off + T * B * V + guard,), 3.0, dtype=torch.bfloat16, device=dev)
        dS = flat[guard + off: guard + off + T * B * V].view(T, B, V)
        L.check(L.load().sad_softemb_bwd(ctypes.byref(d), Fn._ptr(pw.buf), Fn._ptr(dS), L.SRC_BF16, dS.stride(1), dS.stride(0),
                                         Fn._ptr(ws.buf), Fn._stream_ptr()), "sad_softemb_bwd")
        torch.cuda.synchronize()
        assert bool((flat[:guard + off] == 3.0).all()), (V, off, "wrote in front of the tensor")
        assert bool((flat[guard + off + T * B * V:] == 3.0).all()), (V, off, "wrote behind the tensor")
        got = dS.reshape(T * B, V).float()
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 1e-2 * scale + 1e-6, (V, off)
