"""sad_forward_noalloc (include/sad.h): the allocation-free forward, through the C-ABI.

Its kernels are sad_forward's; only the workspace differs (a static device array inside the library
instead of a caller-provided buffer), so every output must equal sad_forward's BIT FOR BIT — checked on
two calls in a row, the second with a smaller input (the shared array then holds the first call's
leftovers at other offsets), for hard tokens and for soft tokens, plus the size limit's error path.
Collected last (file name) so that a failure here hides no other test.
"""
import ctypes

import numpy as np
import pytest

from sad_amd import synth

pytestmark = pytest.mark.gpu


def _run_both(lib, Fn, torch, pw, params, B, T, V, E, H, soft, seed):
    dev = torch.device("cuda:0")
    L = lib.load()
    lengths = synth.make_lengths(B, T, seed=seed, lo=2)
    mask_np = synth.transfer_mask(lengths, T)
    labels = torch.from_numpy(synth.make_labels(B, seed=seed)).to(dev)
    ln = torch.from_numpy(lengths).to(dev)
    if soft:
        src = torch.from_numpy(synth.make_soft_tokens(B, T, V, lengths, seed=seed, kind="dense", time_major=False)).to(dev)
        kind = lib.SRC_F32
    else:
        src = torch.from_numpy(synth.make_token_ids(B, T, V, lengths, seed=seed)).to(dev)
        kind = lib.SRC_IDS
    d = lib.dims(B, T, V, E, H, 2)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for noalloc in (False, True):
        mask = torch.from_numpy(mask_np.copy()).to(dev)
        logits = torch.full((B, 2), 7.0, device=dev)
        weights = torch.full((B, T), 7.0, device=dev)
        loss = torch.full((B,), 7.0, device=dev)
        a = lib.SadFwdArgs()
        a.src, a.src_kind = src.data_ptr(), kind
        a.src_stride_b, a.src_stride_t = src.stride(0), src.stride(1)
        a.lengths = ln.data_ptr()
        a.mask, a.mask_stride_b, a.mask_stride_t = mask.data_ptr(), mask.stride(0), mask.stride(1)
        a.guard = 1
        a.labels, a.logits, a.weights, a.loss = labels.data_ptr(), logits.data_ptr(), weights.data_ptr(), loss.data_ptr()
        if noalloc:
            torch.cuda.synchronize()
            before = torch.cuda.memory_allocated()
            lib.check(L.sad_forward_noalloc(ctypes.byref(d), ctypes.c_void_p(pw.buf.data_ptr()), ctypes.byref(a), st),
                      "sad_forward_noalloc")
            torch.cuda.synchronize()
            assert torch.cuda.memory_allocated() == before
        else:
            ws = Fn.Workspace(d, dev)
            lib.check(L.sad_forward(ctypes.byref(d), ctypes.c_void_p(pw.buf.data_ptr()), ctypes.byref(a),
                                    ctypes.c_void_p(ws.buf.data_ptr()), st), "sad_forward")
            torch.cuda.synchronize()
            ws.check()
        outs.append((logits, weights, loss, mask))
    (lg0, w0, ls0, m0), (lg1, w1, ls1, m1) = outs
    assert bool(torch.isfinite(lg0).all()) and bool(torch.isfinite(ls0).all())
    assert torch.equal(lg0, lg1) and torch.equal(w0, w1) and torch.equal(ls0, ls1) and torch.equal(m0, m1)
    return lg1


def test_forward_noalloc_matches_forward_two_calls_in_a_row():
    import torch
    from sad_amd import functional as Fn, lib
    dev = torch.device("cuda:0")
    V, E, H = 9353, 300, 512
    params = synth.make_params(V, E, H, seed=31)
    pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
    # first call: the transfer/train.sh scoring shape (hard tokens, B=64, T=23); second: a smaller input
    _run_both(lib, Fn, torch, pw, params, 64, 23, V, E, H, soft=False, seed=1)
    _run_both(lib, Fn, torch, pw, params, 7, 9, V, E, H, soft=False, seed=2)
    # soft tokens (stream-K soft-embedding kernels use the tile list and slabs inside the scratch), big then small
    _run_both(lib, Fn, torch, pw, params, 48, 20, V, E, H, soft=True, seed=3)
    _run_both(lib, Fn, torch, pw, params, 5, 6, V, E, H, soft=True, seed=4)
    # beyond the static scratch: refused with a clear error, nothing launched
    L = lib.load()
    d = lib.dims(1024, 32, V, E, H, 2)
    assert L.sad_workspace_bytes(ctypes.byref(d)) > (192 << 20)
    a = lib.SadFwdArgs()
    rc = L.sad_forward_noalloc(ctypes.byref(d), ctypes.c_void_p(pw.buf.data_ptr()), ctypes.byref(a), None)
    assert rc == -3 and "static scratch" in lib.last_error()
    with pytest.raises(RuntimeError, match="static scratch"):
        lib.check(rc, "sad_forward_noalloc")
