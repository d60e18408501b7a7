"""Pins the oracle (oracle/sad_oracle.py, oracle/torch_ref.py) to golden vectors produced by the
reference itself (tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import sad_oracle as O
from oracle.torch_ref import TorchRefDiscriminator
from sad_amd import synth
from tests.util import load_golden, relerr


@pytest.mark.parametrize("kind", ["dense", "onehot"])
@pytest.mark.parametrize("tag,tol", [("f64", 1e-12), ("f32", 2e-5)])
def test_tiny_soft_full(kind, tag, tol):
    hdr, d, params, grads = load_golden("tiny_soft_%s_%s.npz" % (kind, tag))
    dt = np.float64 if tag == "f64" else np.float32
    S_bt = np.ascontiguousarray(d["S"].transpose(1, 0, 2)).astype(dt)   # oracle is batch-major
    mask = d["mask"].copy()
    p = O.cast_params(params, dt)
    f = O.forward(p, S_bt, d["lengths"], mask, True, labels=d["labels"])
    # bit-exact mask semantics incl. the in-place mutation (transfer/models.py:865-867)
    assert np.array_equal(mask, d["mask_after"])
    assert (mask != d["mask"]).sum() == (d["lengths"] == 2).sum() > 0
    assert relerr(f["logits"], d["logits"]) < tol
    assert relerr(f["weights"], d["weights"]) < tol
    assert relerr(f["loss"], d["loss"]) < tol
    B = len(d["labels"])
    dlog = O.ce_dlogits(f["logits"], d["labels"], np.full((B,), 1.0 / B, dt))
    b = O.backward(p, f["cache"], dlog, want_param_grads=True)
    assert relerr(b["dS"].transpose(1, 0, 2), d["dS"]) < tol * 10
    # dS is exactly zero on padded steps (SURVEY.md §3.2 probe)
    for bi, ln in enumerate(d["lengths"]):
        assert np.all(b["dS"][bi, ln:] == 0)
    assert set(hdr["none_grads"]) == set(synth.DEAD_PARAMS)
    for k, g in grads.items():
        assert relerr(b["grads"][k], g, floor=1e-3) < tol * 20, k
    assert set(grads) | set(hdr["none_grads"]) == set(params)


def test_tiny_hard_both_copies():
    hdr, d, params, grads = load_golden("tiny_hard_f64.npz")
    mask = d["mask"].copy()
    f = O.forward(params, d["ids"], d["lengths"], mask, False, labels=d["labels"])
    assert np.array_equal(mask, d["mask_after"])
    assert relerr(f["logits"], d["logits"]) < 1e-12
    assert relerr(f["weights"], d["weights"]) < 1e-12
    B = len(d["labels"])
    dlog = O.ce_dlogits(f["logits"], d["labels"], np.full((B,), 1.0 / B))
    b = O.backward(params, f["cache"], dlog, want_param_grads=True)
    for k, g in grads.items():
        assert relerr(b["grads"][k], g, floor=1e-3) < 1e-10, k
    # classifier copy: pad mask -> [True | pad[:,2:] | True], no guard (classifier/models.py:183-190)
    keep = d["cls_keep"]
    ids = d["ids"][keep]
    cm = synth.classifier_mask(ids == 0)
    f2 = O.forward(params, ids, d["lengths"][keep], cm, False, guard=False)
    assert relerr(f2["logits"], d["cls_logits"]) < 1e-12
    assert relerr(f2["weights"], d["cls_weights"]) < 1e-12


@pytest.mark.parametrize("kind", ["dense", "onehot"])
def test_yelp_soft_b8(kind):
    """Transfer-shaped call at Yelp dims; inputs regenerated from the seeds in the header."""
    for tag, dt, tol in (("f64", np.float64, 1e-10), ("f32", np.float32, 5e-4)):
        hdr, d, _, grads = load_golden("yelp_soft_%s_%s.npz" % (kind, tag))
        V, E, H, B, T = (hdr[k] for k in "VEHBT")
        params = O.cast_params(synth.make_params(V, E, H, seed=hdr["param_seed"]), dt)
        S = synth.make_soft_tokens(B, T, V, d["lengths"], seed=hdr["s_seed"], kind=kind)
        # same inputs as the generator saw
        assert np.allclose((S.astype(np.float64) * np.arange(V)).sum(axis=2), d["S_rowdot"])
        S_bt = np.ascontiguousarray(S.transpose(1, 0, 2)).astype(dt)
        mask = d["mask"].copy()
        f = O.forward(params, S_bt, d["lengths"], mask, True, labels=d["labels"])
        assert np.array_equal(mask, d["mask_after"])
        assert relerr(f["logits"], d["logits"]) < tol
        assert relerr(f["weights"], d["weights"]) < tol
        assert relerr(f["loss"], d["loss"]) < tol
        dlog = O.ce_dlogits(f["logits"], d["labels"], np.full((B,), 1.0 / B, dt))
        b = O.backward(params, f["cache"], dlog, want_param_grads=True)
        dS = b["dS"].transpose(1, 0, 2)
        assert relerr(dS[:, :, ::hdr["dS_col_stride"]], d["dS_sample"]) < tol * 10
        assert relerr(dS.astype(np.float64).sum(axis=2), d["dS_rowsum"]) < tol * 10
        assert relerr((dS.astype(np.float64) ** 2).sum(axis=2), d["dS_rowsq"]) < tol * 10
        for k, g in grads.items():
            assert relerr(b["grads"][k], g, floor=1e-3) < tol * 20, k


def test_yelp_hard_b64():
    hdr, d, _, _ = load_golden("yelp_hard_b64.npz")
    V, E, H = hdr["V"], hdr["E"], hdr["H"]
    params = O.cast_params(synth.make_params(V, E, H, seed=hdr["param_seed"]), np.float64)
    mask = d["mask"].copy()
    f = O.forward(params, d["ids"], d["lengths"], mask, False, labels=d["labels"])
    assert relerr(f["logits"], d["logits_f64"]) < 1e-10
    assert relerr(f["weights"], d["weights_f64"]) < 1e-10
    assert relerr(f["logits"], d["cls_logits_f64"]) < 1e-10   # the two copies agree
    assert relerr(f["logits"], d["logits_f32"]) < 1e-4


def test_torch_ref_matches_golden():
    """The stock-torch restatement timed as cpu_baseline is the same function."""
    hdr, d, params, _ = load_golden("tiny_soft_dense_f64.npz")
    V, E, H = hdr["V"], hdr["E"], hdr["H"]
    m = TorchRefDiscriminator(V, E, H).double().load_numpy(params)
    S = torch.from_numpy(d["S"]).requires_grad_(True)       # [T,B,V]
    mask = torch.from_numpy(d["mask"].copy())
    logits, w = m(S.transpose(0, 1), torch.from_numpy(d["lengths"]), mask, soft=True)
    assert np.array_equal(mask.numpy(), d["mask_after"])
    assert relerr(logits.detach().numpy(), d["logits"]) < 1e-12
    assert relerr(w.detach().numpy(), d["weights"]) < 1e-12
    B = len(d["labels"])
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(d["labels"]), reduction="none")
    (loss.sum() / B).backward()
    assert relerr(S.grad.numpy(), d["dS"]) < 1e-11


def test_classifier_dropout_training_step():
    """Classifier copy in training mode with dropout on the embedded tokens (classifier/models.py:217):
    the fixture holds the keep mask the reference drew, its outputs and parameter gradients."""
    hdr, d, params, grads = load_golden("tiny_hard_dropout_f64.npz")
    ids, lengths = d["ids"], d["lengths"]
    mask = synth.classifier_mask(ids == 0)
    drop = (d["keep"].transpose(1, 0, 2), 1.0 / (1.0 - hdr["p_drop"]))
    f = O.forward(params, ids, lengths, mask, False, labels=d["labels"], guard=False, drop=drop)
    assert relerr(f["logits"], d["logits"]) < 1e-12
    assert relerr(f["weights"], d["weights"]) < 1e-12
    B = len(d["labels"])
    dlog = O.ce_dlogits(f["logits"], d["labels"], np.full((B,), 1.0 / B))
    b = O.backward(params, f["cache"], dlog, want_param_grads=True)
    assert len(grads) == 19
    for k, g in grads.items():
        assert relerr(b["grads"][k], g, floor=1e-3) < 1e-10, k


VARIANT_TAGS = ["gru_bi", "lstm_uni", "gru_uni", "lstm_bi_l2", "gru_bi_l2", "lstm_uni_l3"]


@pytest.mark.parametrize("tag", VARIANT_TAGS)
def test_variant_oracle(tag):
    """GRU / uni-directional / multi-layer discriminators (reference
    classifier/models.py:102-124,243-260) — the generalized oracle against the reference's outputs, dS and
    every parameter gradient."""
    hdr, d, params, grads = load_golden("variant_%s_f64.npz" % tag)
    assert O.variant(params) == (hdr["cell"], 2 if hdr["bidirectional"] else 1, hdr["layers"])
    S_bt = np.ascontiguousarray(d["S"].transpose(1, 0, 2))
    mask = d["mask"].copy()
    f = O.forward(params, S_bt, d["lengths"], mask, True, labels=d["labels"])
    assert np.array_equal(mask, d["mask_after"])
    assert relerr(f["logits"], d["logits"]) < 1e-12
    assert relerr(f["weights"], d["weights"]) < 1e-12
    assert relerr(f["loss"], d["loss"]) < 1e-12
    B = len(d["labels"])
    dlog = O.ce_dlogits(f["logits"], d["labels"], np.full((B,), 1.0 / B))
    b = O.backward(params, f["cache"], dlog, want_param_grads=True)
    assert relerr(b["dS"].transpose(1, 0, 2), d["dS"]) < 1e-11
    dead = {k for k in params if k.startswith("output_bridge.") or k.startswith("hidden_bridges.1.")}
    assert set(hdr["none_grads"]) == dead
    for k, g in grads.items():
        assert relerr(b["grads"][k], g, floor=1e-3) < 1e-10, k
    assert set(grads) | dead == set(params)
    keep = d["cls_keep"]
    ids = d["ids"][keep]
    f2 = O.forward(params, ids, d["lengths"][keep], synth.classifier_mask(ids == 0), False, guard=False)
    assert relerr(f2["logits"], d["cls_logits"]) < 1e-12
    assert relerr(f2["weights"], d["cls_weights"]) < 1e-12
