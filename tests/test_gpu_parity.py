"""GPU parity tests: libsad (through the C-ABI, via the ctypes binding) vs the oracle and the golden
fixtures generated from the reference.  Run on a real MI355X: ``pytest -m gpu``.

Stated tolerances (scale-aware max error, |a-b|_max / |b|_max):
  * the kernels multiply in bf16 with fp32 accumulation (weights and GEMM-input activations are
    rounded to bf16, gates/cell state/softmax stay fp32) — BASELINE.json's "bf16" configuration;
  * BF16_TOL_FWD = 2e-2 for logits / attention weights / loss, BF16_TOL_BWD = 2e-2 for d(soft
    tokens) against the fp32/fp64 REFERENCE outputs;
  * token / mask indexing (mask mutation, need bits, zero rows of dS, padding) is bit-exact.
"""
import numpy as np
import pytest

from oracle import sad_oracle as O
from sad_amd import synth
from tests.util import bf16_round, load_golden, relerr, run_sad, stage_errors

pytestmark = pytest.mark.gpu

BF16_TOL_FWD = 2e-2
BF16_TOL_BWD = 2e-2


def _report(title, errs):
    print("\n[%s]" % title)
    for k, v in errs.items():
        print("   %-12s %.3e" % (k, v))


def _oracle(params, S_bt, lengths, mask, labels, soft=True, dloss=None, dweights=None, guard=True):
    p = O.cast_params(params, np.float64)
    f = O.forward(p, S_bt.astype(np.float64) if soft else S_bt, lengths, mask.copy(), soft, labels=labels, guard=guard)
    B = len(labels)
    dl = O.ce_dlogits(f["logits"], labels, np.full((B,), 1.0 / B) if dloss is None else dloss)
    b = O.backward(p, f["cache"], dl, dweights=dweights)
    return f, b


def _oracle_grads(params, src, lengths, mask, labels, soft=True, guard=True, dtype=np.float64):
    p = O.cast_params(params, dtype)
    f = O.forward(p, src.astype(dtype) if soft else src, lengths, mask.copy(), soft, labels=labels, guard=guard)
    B = len(labels)
    dl = O.ce_dlogits(f["logits"], labels, np.full((B,), 1.0 / B))
    return f, O.backward(p, f["cache"], dl, want_param_grads=True)


def _grad_errors(got, want):
    errs = {}
    for k, w in want.items():
        g = got[k]
        assert g is not None and g.shape == w.shape, k
        assert np.isfinite(g).all(), k
        # linear_attn.bias has an analytically zero gradient (softmax shift invariance): floor the scale
        # scale-aware error relative to the tensor's own largest entry; linear_attn.bias has an
        # analytically zero gradient (softmax shift invariance), so only its absolute size is checked
        errs[k] = relerr(g, w, floor=1e-3 if k == "linear_attn.bias" else 1e-12)
    return errs


BF16_TOL_WGRAD = 2e-2


@pytest.mark.parametrize("kind", ["dense", "onehot"])
def test_param_grads_tiny_golden(kind):
    """Unfrozen mode: all 19 live parameter gradients vs the reference's autograd (golden fixture
    written by tools/gen_golden.py from the imported reference module, mean CE loss)."""
    hdr, d, params, want = load_golden("tiny_soft_%s_f32.npz" % kind)
    S_bt = np.ascontiguousarray(d["S"].transpose(1, 0, 2))
    out = run_sad(params, S_bt, d["lengths"], d["mask"], d["labels"], param_grads=True)
    assert len(want) == 19
    errs = _grad_errors(out["grads"], want)
    errs["dS"] = relerr(out["dS"].transpose(1, 0, 2), d["dS"])
    _report("param grads tiny golden %s" % kind, errs)
    assert max(errs.values()) < BF16_TOL_WGRAD, errs


def test_param_grads_tiny_hard_golden():
    """Hard tokens (classifier training, classifier/trainers.py:367): gradients incl. the embedding
    scatter-add; the padding_idx row stays zero."""
    hdr, d, params, want = load_golden("tiny_hard_f64.npz")
    out = run_sad(params, d["ids"], d["lengths"], d["mask"], d["labels"], soft=False, param_grads=True, pad_id=0)
    assert len(want) == 19
    errs = _grad_errors(out["grads"], want)
    _report("param grads tiny hard golden", errs)
    assert max(errs.values()) < BF16_TOL_WGRAD, errs
    assert not out["grads"]["embedding.weight"][0].any()


@pytest.mark.parametrize("s_dtype", ["f32", "bf16"])
def test_param_grads_yelp_dims(s_dtype):
    """Yelp dims (V=9353, E=300, H=512): the shipped-size kernel configurations (one 320-wide
    tile over E).  Small tensors vs the reference's golden gradients, every tensor vs the oracle."""
    hdr, d, _, gold = load_golden("yelp_soft_dense_f32.npz")
    V, E, H, B, T = (hdr[k] for k in "VEHBT")
    params = synth.make_params(V, E, H, seed=hdr["param_seed"])
    S = synth.make_soft_tokens(B, T, V, d["lengths"], seed=hdr["s_seed"], kind="dense", time_major=False)
    out = run_sad(params, S, d["lengths"], d["mask"], d["labels"], s_dtype=s_dtype, param_grads=True)
    f, b = _oracle_grads(params, S, d["lengths"], d["mask"], d["labels"])
    errs = _grad_errors(out["grads"], b["grads"])
    for k, w in gold.items():
        errs["golden:" + k] = relerr(out["grads"][k], w, floor=1e-3)
    _report("param grads yelp dims %s" % s_dtype, errs)
    assert max(errs.values()) < BF16_TOL_WGRAD, errs


def test_param_grads_yelp_hard_b64():
    """Classifier-training shape on real Yelp sentences (hard tokens, B=64) vs the oracle."""
    hdr, d, _, _ = load_golden("yelp_hard_b64.npz")
    params = synth.make_params(hdr["V"], hdr["E"], hdr["H"], seed=hdr["param_seed"])
    out = run_sad(params, d["ids"], d["lengths"], d["mask"], d["labels"], soft=False, param_grads=True, pad_id=0)
    f, b = _oracle_grads(params, d["ids"], d["lengths"], d["mask"], d["labels"], soft=False)
    errs = _grad_errors(out["grads"], b["grads"])
    _report("param grads yelp hard b64", errs)
    assert max(errs.values()) < BF16_TOL_WGRAD, errs


def test_classifier_dropout_training_golden():
    """Classifier copy, training mode, dropout p=0.3 on the embedded tokens (classifier/models.py:217),
    with the keep mask the REFERENCE drew: outputs and all 19 parameter gradients vs its autograd."""
    hdr, d, params, want = load_golden("tiny_hard_dropout_f64.npz")
    mask = synth.classifier_mask(d["ids"] == 0)
    drop = (d["keep"], 1.0 / (1.0 - hdr["p_drop"]))
    out = run_sad(params, d["ids"], d["lengths"], mask, d["labels"], soft=False, guard=False, param_grads=True,
                  pad_id=0, drop=drop)
    errs = _grad_errors(out["grads"], want)
    errs["logits"] = relerr(out["logits"], d["logits"])
    errs["weights"] = relerr(out["weights"], d["weights"])
    _report("classifier dropout training golden", errs)
    assert max(errs.values()) < BF16_TOL_WGRAD, errs
    # the module draws its own mask in train() mode and none in eval() mode
    import torch
    from types import SimpleNamespace
    from sad_amd import classifier_models
    cfg = SimpleNamespace(cell="LSTM", enc_bidirectional=True, pad_id=0, dropout=hdr["p_drop"], enc_num_layers=1,
                          hidden_size=hdr["H"], vocab_size=hdr["V"], emb_size=hdr["E"], num_labels=2,
                          freeze_embedding=False, device="cuda:0", mask_id=None)
    mc = classifier_models.SelfAttnRNNClassifier(cfg)
    mc.load_state_dict({k: torch.from_numpy(v.astype(np.float32)) for k, v in params.items()})
    mc.to("cuda:0")
    ids = torch.from_numpy(d["ids"]).cuda()
    ln = torch.from_numpy(d["lengths"])
    mc.eval()
    with torch.no_grad():
        e1, _ = mc(ids, ln, ids == 0)
        e2, _ = mc(ids, ln, ids == 0)
    assert torch.equal(e1, e2)
    mc.train()
    t1, _ = mc(ids, ln, ids == 0)
    t2, _ = mc(ids, ln, ids == 0)
    assert not torch.equal(t1, t2) and torch.isfinite(t1).all()
    torch.nn.functional.cross_entropy(t1, torch.from_numpy(d["labels"]).cuda()).backward()
    assert mc.embedding.weight.grad is not None and torch.isfinite(mc.embedding.weight.grad).all()


def test_param_grads_ragged_vs_oracle_and_frozen_subset():
    """Odd sizes (H, E not multiples of the tile sizes, ragged lengths, several row splits) vs the
    oracle; then the same call with the embedding frozen (freeze_emb) and one bias frozen: frozen
    tensors get no gradient, the others are unchanged up to atomics' summation order."""
    V, E, H, B, T = 203, 44, 24, 37, 13
    lengths = np.sort(np.random.RandomState(5).randint(2, T + 1, size=B))[::-1].astype(np.int32)
    lengths[0] = T
    params = synth.make_params(V, E, H, seed=21)
    S = synth.make_soft_tokens(B, T, V, lengths, seed=9, kind="dense", time_major=False)
    labels = synth.make_labels(B, seed=4)
    mask = synth.transfer_mask(lengths, T)
    out = run_sad(params, S, lengths, mask, labels, param_grads=True)
    f, b = _oracle_grads(params, S, lengths, mask, labels)
    errs = _grad_errors(out["grads"], b["grads"])
    errs["dS"] = relerr(out["dS"], b["dS"])
    _report("param grads ragged", errs)
    assert max(errs.values()) < BF16_TOL_WGRAD, errs
    frozen = ("embedding.weight", "linear_q.bias")
    out2 = run_sad(params, S, lengths, mask, labels, param_grads=True, frozen=frozen)
    for k, v in out2["grads"].items():
        if k in frozen:
            assert v is None
        else:
            assert relerr(v, out["grads"][k], floor=1e-6) < 1e-4, k
    assert np.array_equal(out2["dS"], out["dS"])


def test_stages_tiny_exact_inputs():
    """Every workspace stage against the oracle on inputs that are exactly representable in bf16
    (one-hot S, bf16-rounded weights): the GEMM stages must then agree to fp32-accumulate accuracy,
    which separates indexing bugs (O(1)) from rounding (1e-3)."""
    V, E, H, B, T = 41, 10, 8, 6, 9
    lengths = np.array([9, 7, 5, 3, 2, 2], dtype=np.int32)
    params = {k: bf16_round(v) for k, v in synth.make_params(V, E, H, seed=11).items()}
    S = synth.make_soft_tokens(B, T, V, lengths, seed=5, kind="onehot", time_major=False)
    labels = synth.make_labels(B, seed=3)
    mask = synth.transfer_mask(lengths, T)
    out = run_sad(params, S, lengths, mask, labels, want_ws=True)
    f, b = _oracle(params, S, lengths, mask, labels)
    errs = stage_errors(out["ws"], f, b, H, E)
    errs["logits"] = relerr(out["logits"], f["logits"])
    errs["weights"] = relerr(out["weights"], f["weights"])
    errs["loss"] = relerr(out["loss"], f["loss"])
    errs["dS"] = relerr(out["dS"], b["dS"])
    _report("tiny exact-input stages", errs)
    assert np.array_equal(out["ws"]["LENS"], lengths)
    need = (np.arange(T)[:, None] < lengths[None, :]) * 1 + (~f["cache"]["mask"].T) * 2
    assert np.array_equal(out["ws"]["NEED"], need.astype(np.uint8))
    assert errs["X"] == 0.0                      # exact gather through the GEMM
    for k in ("G0", "G1", "KK"):
        assert errs[k] < 2e-2, k                 # h is bf16 inside the recurrence
    assert errs["DX_pad"] == 0.0 and errs["DZ_inactive"] == 0.0
    for k, v in errs.items():
        assert v < 5e-2, (k, v)


@pytest.mark.parametrize("kind", ["dense", "onehot"])
@pytest.mark.parametrize("time_major", [True, False])
def test_tiny_golden(kind, time_major):
    hdr, d, params, _ = load_golden("tiny_soft_%s_f32.npz" % kind)
    S_bt = np.ascontiguousarray(d["S"].transpose(1, 0, 2))
    out = run_sad(params, S_bt, d["lengths"], d["mask"], d["labels"], time_major=time_major, want_ws=True)
    f, b = _oracle(params, S_bt, d["lengths"], d["mask"], d["labels"])
    _report("tiny golden %s tm=%s stages" % (kind, time_major), stage_errors(out["ws"], f, b, hdr["H"], hdr["E"]))
    # bit-exact mask semantics incl. in-place mutation of the caller's tensor
    assert np.array_equal(out["mask_after"], d["mask_after"])
    errs = dict(logits=relerr(out["logits"], d["logits"]), weights=relerr(out["weights"], d["weights"]),
                loss=relerr(out["loss"], d["loss"]), dS=relerr(out["dS"].transpose(1, 0, 2), d["dS"]))
    _report("tiny golden %s tm=%s" % (kind, time_major), errs)
    assert errs["logits"] < BF16_TOL_FWD and errs["weights"] < BF16_TOL_FWD and errs["loss"] < BF16_TOL_FWD
    assert errs["dS"] < BF16_TOL_BWD
    assert out["dS_strides"] == out["S_strides"]          # dS keeps the layout of S
    for bi, ln in enumerate(d["lengths"]):
        assert np.all(out["dS"][bi, ln:] == 0)            # exact zeros on padded steps
        assert np.all(out["weights"][bi][d["mask_after"][bi]] == 0)


def test_tiny_unfused_loss_and_dweights():
    """Module-style use: logits returned, CE taken by torch, plus an upstream gradient on the
    attention weights (dlogits / dweights inputs of sad_backward)."""
    hdr, d, params, _ = load_golden("tiny_soft_dense_f32.npz")
    S_bt = np.ascontiguousarray(d["S"].transpose(1, 0, 2))
    B, T = S_bt.shape[:2]
    dw = np.random.RandomState(0).standard_normal((B, T)).astype(np.float32) * 0.1
    out = run_sad(params, S_bt, d["lengths"], d["mask"], d["labels"], fused_loss=False, dweights=dw)
    f, b = _oracle(params, S_bt, d["lengths"], d["mask"], d["labels"], dweights=dw.astype(np.float64))
    errs = dict(logits=relerr(out["logits"], f["logits"]), dS=relerr(out["dS"], b["dS"]))
    _report("tiny unfused + dweights", errs)
    assert errs["logits"] < BF16_TOL_FWD and errs["dS"] < BF16_TOL_BWD


@pytest.mark.parametrize("kind", ["dense", "onehot"])
@pytest.mark.parametrize("s_dtype", ["f32", "bf16"])
def test_yelp_soft_b8_golden(kind, s_dtype):
    """Transfer-shaped call at Yelp dims (B=8, T=23, V=9353, E=300, H=512): golden logits /
    weights / loss / dS samples + checksums from the reference."""
    hdr, d, _, _ = load_golden("yelp_soft_%s_f32.npz" % kind)
    V, E, H, B, T = (hdr[k] for k in "VEHBT")
    params = synth.make_params(V, E, H, seed=hdr["param_seed"])
    S = synth.make_soft_tokens(B, T, V, d["lengths"], seed=hdr["s_seed"], kind=kind, time_major=False)
    out = run_sad(params, S, d["lengths"], d["mask"], d["labels"], s_dtype=s_dtype, want_ws=True)
    f, b = _oracle(params, S, d["lengths"], d["mask"], d["labels"])
    _report("yelp b8 %s %s stages" % (kind, s_dtype), stage_errors(out["ws"], f, b, H, E))
    assert np.array_equal(out["mask_after"], d["mask_after"])
    dS = out["dS"].transpose(1, 0, 2)     # golden is [T,B,V]
    errs = dict(logits=relerr(out["logits"], d["logits"]), weights=relerr(out["weights"], d["weights"]),
                loss=relerr(out["loss"], d["loss"]),
                dS_sample=relerr(dS[:, :, ::hdr["dS_col_stride"]], d["dS_sample"]),
                dS_rowsum=relerr(dS.astype(np.float64).sum(axis=2), d["dS_rowsum"]),
                dS_rowsq=relerr((dS.astype(np.float64) ** 2).sum(axis=2), d["dS_rowsq"]),
                dS_full_vs_oracle=relerr(out["dS"], b["dS"]))
    _report("yelp b8 %s %s" % (kind, s_dtype), errs)
    assert errs["logits"] < BF16_TOL_FWD and errs["weights"] < BF16_TOL_FWD and errs["loss"] < BF16_TOL_FWD
    tol_b = BF16_TOL_BWD if s_dtype == "f32" else BF16_TOL_BWD * 1.5   # + bf16 rounding of dS itself
    assert errs["dS_sample"] < tol_b and errs["dS_full_vs_oracle"] < tol_b
    assert errs["dS_rowsq"] < 2 * tol_b
    for bi, ln in enumerate(d["lengths"]):
        assert np.all(out["dS"][bi, ln:] == 0)


def test_yelp_hard_b64_golden():
    """Hard-token path on real Yelp sentences (both copies of the reference agree on it)."""
    hdr, d, _, _ = load_golden("yelp_hard_b64.npz")
    params = synth.make_params(hdr["V"], hdr["E"], hdr["H"], seed=hdr["param_seed"])
    out = run_sad(params, d["ids"], d["lengths"], d["mask"], d["labels"], soft=False)
    errs = dict(logits=relerr(out["logits"], d["logits_f32"]), weights=relerr(out["weights"], d["weights_f32"]),
                loss=relerr(out["loss"], d["loss_f32"]))
    _report("yelp hard b64", errs)
    assert all(v < BF16_TOL_FWD for v in errs.values())
    assert np.all(out["weights"][d["mask"]] == 0)


def test_module_dropin_both_flavours():
    """The nn.Module mirrors: same state_dict keys as the fixtures' parameter set, transfer and
    classifier forward signatures, get_masked_src (bit-exact index output)."""
    import torch
    from types import SimpleNamespace
    from sad_amd import classifier_models, transfer_models
    hdr, d, params, _ = load_golden("tiny_hard_f64.npz")
    V, E, H = hdr["V"], hdr["E"], hdr["H"]
    cfg = SimpleNamespace(cell="LSTM", enc_bidirectional=True, pad_id=0, dropout=0.0, enc_num_layers=1,
                          hidden_size=H, vocab_size=V, emb_size=E, num_labels=2, freeze_embedding=False,
                          device="cuda:0", mask_id=None)
    sd = {k: torch.from_numpy(v.astype(np.float32)) for k, v in params.items()}
    mt = transfer_models.SelfAttnRNNClassifier(cfg)
    assert list(mt.state_dict().keys()) == list(params.keys())
    mt.load_state_dict(sd)
    mt.to("cuda:0")
    for p in mt.parameters():
        p.requires_grad = False
    ids = torch.from_numpy(d["ids"]).cuda()
    mask = torch.from_numpy(d["mask"].copy()).cuda()
    logits, w = mt(ids, torch.from_numpy(d["lengths"]), mask)
    assert np.array_equal(mask.cpu().numpy(), d["mask_after"])
    assert relerr(logits.cpu().numpy(), d["logits"]) < BF16_TOL_FWD
    assert relerr(w.cpu().numpy(), d["weights"]) < BF16_TOL_FWD

    mc = classifier_models.SelfAttnRNNClassifier(cfg)
    mc.load_state_dict(sd)
    mc.to("cuda:0").eval()
    keep = d["cls_keep"]
    idk = torch.from_numpy(d["ids"][keep]).cuda()
    with torch.no_grad():
        lg, w2 = mc(idk, torch.from_numpy(d["lengths"][keep]), idk == 0)
        assert relerr(lg.cpu().numpy(), d["cls_logits"]) < BF16_TOL_FWD
        assert relerr(w2.cpu().numpy(), d["cls_weights"]) < BF16_TOL_FWD
        mc.set_mask_id(hdr["mask_id"])
        _, alpha, style_src, content_src = mc.get_masked_src(idk, torch.from_numpy(d["lengths"][keep]), idk == 0)
    # the threshold split is exact unless a weight sits within bf16 noise of 1/(len-2)
    thr = 1.0 / (d["lengths"][keep] - 2.0)
    margin = np.abs(d["cls_weights"] - thr[:, None]).min()
    if margin > 0.05:
        assert np.array_equal(style_src.cpu().numpy(), d["gms_style_src"])
        assert np.array_equal(content_src.cpu().numpy(), d["gms_content_src"])
    # one classifier-training step through the module API (reference classifier/trainers.py:367-373):
    # gradients land in .grad of the live parameters, dead ones stay None, clip + optimizer step work,
    # and the next forward uses the updated weights (packed image is refreshed).
    _, _, _, gold = load_golden("tiny_hard_f64.npz")
    for p in mt.parameters():
        p.requires_grad = True
    opt = torch.optim.SGD(mt.parameters(), lr=0.5)
    lab = torch.from_numpy(d["labels"]).cuda()
    mask = torch.from_numpy(d["mask"].copy()).cuda()
    logits, _ = mt(ids, torch.from_numpy(d["lengths"]), mask)
    loss = torch.nn.functional.cross_entropy(logits, lab)
    opt.zero_grad()
    loss.backward()
    got = {k: p.grad for k, p in mt.named_parameters()}
    for k in synth.DEAD_PARAMS:
        assert got[k] is None
    for k, w_ in gold.items():
        assert relerr(got[k].cpu().numpy(), w_, floor=1e-3) < BF16_TOL_WGRAD, k
    torch.nn.utils.clip_grad_norm_(mt.parameters(), 5.0)
    opt.step()
    new_params = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in mt.state_dict().items()}
    mask = torch.from_numpy(d["mask"].copy()).cuda()
    with torch.no_grad():
        logits2, _ = mt(ids, torch.from_numpy(d["lengths"]), mask)
    f2 = O.forward(new_params, d["ids"], d["lengths"], d["mask"].copy(), False, labels=d["labels"])
    assert relerr(logits2.cpu().numpy(), f2["logits"]) < BF16_TOL_FWD
    assert relerr(logits2.cpu().numpy(), d["logits"]) > 1e-3      # the step really moved the weights
    # freeze_emb(): no embedding gradient, the rest still trains
    mt.freeze_emb()
    opt.zero_grad(set_to_none=True)
    mask = torch.from_numpy(d["mask"].copy()).cuda()
    logits, _ = mt(ids, torch.from_numpy(d["lengths"]), mask)
    torch.nn.functional.cross_entropy(logits, lab).backward()
    assert mt.embedding.weight.grad is None and mt.linear_q.weight.grad is not None


def _cfg3_inputs(B, T=32, V=9353, seed=1234):
    lengths = synth.make_lengths(B, T, seed=seed)
    S = synth.make_soft_tokens(B, T, V, lengths, seed=seed, kind="dense", time_major=False)
    return lengths, S, synth.transfer_mask(lengths, T), synth.make_labels(B, seed=seed)


def test_cfg3_subsample_vs_oracle_and_shard_invariance():
    """BASELINE config-3 shape (T=32, V=9353, E=300, H=512) at B=256 in bf16 soft tokens:
      * 12 sentences spread over the batch are checked against the oracle run on just those
        sentences (sentences are independent);
      * size-independent properties: attention weights sum to 1 and vanish under the mask, dS rows
        are exactly zero for t >= len, dS is linear in dloss, and scoring the two halves of the
        batch separately (what 2-GPU data parallelism does) reproduces the full-batch result.
    At this batch the K1 / in-projection-backward GEMMs run split-K with fp32 atomics (few row
    tiles), whose summation order is not fixed, so the last two properties are checked to 2e-3 (fp32 outputs) /
    1e-2 (bf16 dS, one ulp = 4e-3) here and BIT-EXACTLY at the bench batch in test_full_batch_bit_exact_properties."""
    B, T, V, E, H = 256, 32, 9353, 300, 512
    params = synth.make_params(V, E, H, seed=21)
    lengths, S, mask, labels = _cfg3_inputs(B)
    S = bf16_round(S)      # the bf16 input path sees exactly these values
    full = run_sad(params, S, lengths, mask, labels, s_dtype="bf16")
    sel = np.linspace(0, B - 1, 12).astype(int)
    f, b = _oracle(params, S[sel], lengths[sel], mask[sel], labels[sel], dloss=np.full((12,), 1.0 / B))
    errs = dict(logits=relerr(full["logits"][sel], f["logits"]), weights=relerr(full["weights"][sel], f["weights"]),
                loss=relerr(full["loss"][sel], f["loss"]), dS=relerr(full["dS"][sel], b["dS"]))
    _report("cfg3 B=256 subsample vs oracle", errs)
    assert errs["logits"] < BF16_TOL_FWD and errs["weights"] < BF16_TOL_FWD and errs["loss"] < BF16_TOL_FWD
    assert errs["dS"] < BF16_TOL_BWD * 1.5
    assert np.allclose(full["weights"].sum(axis=1), 1.0, atol=1e-5)
    assert np.all(full["weights"][full["mask_after"]] == 0)
    for bi, ln in enumerate(lengths):
        assert np.all(full["dS"][bi, ln:] == 0)
    twice = run_sad(params, S, lengths, mask, labels, s_dtype="bf16", dloss=np.full((B,), 2.0 / B))
    assert relerr(twice["dS"], 2.0 * full["dS"]) < 1e-2
    h = B // 2
    for lo, hi in ((0, h), (h, B)):
        part = run_sad(params, S[lo:hi], lengths[lo:hi], mask[lo:hi], labels[lo:hi], s_dtype="bf16",
                       dloss=np.full((hi - lo,), 1.0 / B))
        assert relerr(part["logits"], full["logits"][lo:hi]) < 2e-3
        assert relerr(part["weights"], full["weights"][lo:hi]) < 2e-3
        assert relerr(part["dS"], full["dS"][lo:hi]) < 1e-2      # dS is stored in bf16: one ulp = 4e-3


def test_full_batch_bit_exact_properties():
    """BASELINE config-3 at its FULL size (B=1024, T=32, V=9353, bf16 one-hot soft tokens built on
    the device): properties that need no oracle, checked bit for bit —
      * data parallelism: the two half batches scored separately == the full batch (logits,
        attention weights, dS), which is the N-rank == 1-rank invariant of the weak-scaling bench;
      * linearity: dloss * 2 gives exactly 2 * dS;
      * dS is exactly zero for t >= len and exactly reproducible run to run."""
    import torch
    from sad_amd import functional as Fn
    B, T, V, E, H = 1024, 32, 9353, 300, 512
    dev = torch.device("cuda:0")
    params = synth.make_params(V, E, H, seed=21)
    pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
    lengths = synth.make_lengths(B, T, seed=77)
    ids = torch.from_numpy(synth.make_token_ids(B, T, V, lengths, seed=77)).to(dev)
    S_store = torch.zeros((T, B, V), dtype=torch.bfloat16, device=dev)       # time-major storage
    S_store.scatter_(2, ids.t().unsqueeze(2), 1.0)
    labels = torch.from_numpy(synth.make_labels(B, seed=77)).to(dev)
    mask0 = torch.from_numpy(synth.transfer_mask(lengths, T)).to(dev)
    ln = torch.from_numpy(lengths)

    def run(lo, hi, scale):
        S = S_store[:, lo:hi].transpose(0, 1).detach().requires_grad_(True)   # strided [b,T,V] view
        loss, logits, w = Fn.discriminator_loss(S, ln[lo:hi], mask0[lo:hi].clone(), labels[lo:hi], pw)
        (loss.sum() * (scale / B)).backward()
        return logits.detach(), w.detach(), S.grad

    lg, w, dS = run(0, B, 1.0)
    lg2, w2, dS2 = run(0, B, 1.0)
    assert torch.equal(lg, lg2) and torch.equal(dS, dS2)                      # reproducible
    _, _, dS_twice = run(0, B, 2.0)
    assert torch.equal(dS_twice.float(), 2.0 * dS.float())                    # linear, bit-exact
    pad = torch.arange(T, device=dev)[None, :] >= ln.to(dev)[:, None]
    assert bool((dS[pad] == 0).all())
    assert bool((dS[~pad].float().abs().sum() > 0))
    h = B // 2
    for lo, hi in ((0, h), (h, B)):
        plg, pw_, pdS = run(lo, hi, 1.0)
        assert torch.equal(plg, lg[lo:hi]) and torch.equal(pw_, w[lo:hi]) and torch.equal(pdS, dS[lo:hi])


def test_persistent_recurrence_matches_per_step_kernels():
    """The persistent LSTM kernels (forward and BPTT; in-launch hand-off between workgroups) against
    the one-launch-per-step kernels on the same inputs: hidden / cell states, dZ, outputs and dS
    agree to fp32 rounding, and repeated persistent runs are BIT-identical (a stale hand-off read
    would show as a difference), at full batch (one workgroup per CU), at a ragged batch (partial
    sentence tile), at H not a multiple of 64, and at each of the three sentence-tile sizes the kernels choose
    (lstm_fwd_tile / lstm_bwd_tile)."""
    import torch
    from sad_amd import functional as Fn, lib
    Fn.KEEP_LAST = True
    dev = torch.device("cuda:0")
    # sentence tiles of the persistent kernels at H = 512: 128 at B = 1024, 64 at B = 300, 32 at B = 64
    for (B, T, V, E, H) in ((1024, 32, 500, 300, 512), (200, 17, 300, 44, 96), (64, 23, 300, 300, 512), (300, 9, 300, 300, 512)):
        lengths = synth.make_lengths(B, T, seed=B)
        params = synth.make_params(V, E, H, seed=3)
        pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
        S = torch.from_numpy(synth.make_soft_tokens(B, T, V, lengths, seed=5, kind="dense", time_major=False)).to(dev)
        ln = torch.from_numpy(lengths)
        mask0 = torch.from_numpy(synth.transfer_mask(lengths, T)).to(dev)
        lab = torch.from_numpy(synth.make_labels(B, seed=1)).to(dev)

        def run(no_persist):
            lib.debug_set(lib.DBG_NO_PERSIST, 1 if no_persist else 0)
            try:
                Sg = S.clone().requires_grad_(True)
                loss, logits, weights = Fn.discriminator_loss(Sg, ln, mask0.clone(), lab, pw)
                ws = Fn._Discriminator.last_workspace
                loss.mean().backward()
                torch.cuda.synchronize()
                ws.check()
                live = (ws.view("NEED", torch.uint8).reshape(T, B) & 1).bool()      # t < len: rows the LSTM writes
                hs = ws.view("HS", torch.bfloat16).reshape(T, B, -1)[live].float()
                cs = ws.view("CS", torch.float32).reshape(T, B, -1)[live]
                dz = ws.view("DZ", torch.bfloat16).reshape(T, B, -1).float()
                return dict(logits=logits.detach().clone(), weights=weights.detach().clone(), Hs=hs.clone(), Cs=cs.clone(),
                            DZ=dz.clone(), dS=Sg.grad.clone())
            finally:
                lib.debug_set(lib.DBG_NO_PERSIST, 0)

        ref = run(True)
        first = run(False)
        # the two paths run the same arithmetic up to the compiler's fma contraction in the cell update and
        # the summation order of the K-split BPTT GEMM: rounding-level differences that grow slowly over
        # the steps (measured 1.7e-4 on c after 32 steps)
        for name in ref:
            scale = max(ref[name].abs().max().item(), 1e-20)
            d = (first[name] - ref[name]).abs().max().item() / scale
            tol = 1e-2 if name in ("Hs", "DZ", "dS") else 1e-3       # bf16-stored quantities: one ulp
            assert d < tol, (B, T, H, name, d)
        assert not first["DZ"].isnan().any()
        # a stale or torn hand-off read would make the persistent path differ from run to run: every dh of
        # the BPTT is summed by one wave in a fixed order and the stream-K partial tiles are added in a
        # fixed order, so forward AND backward must repeat bit for bit
        for rep in range(4):
            got = run(False)
            for name in ("logits", "weights", "Hs", "Cs", "DZ"):
                assert torch.equal(got[name], first[name]), (B, T, H, rep, name)
            if E == 300:     # fast-path GEMMs: fixed summation order.  Other E: split-K with fp32 atomics
                assert torch.equal(got["dS"], first["dS"]), (B, T, H, rep)
            else:
                scale = max(first["dS"].abs().max().item(), 1e-20)
                assert (got["dS"] - first["dS"]).abs().max().item() / scale < 1e-2, (B, T, H, rep)


def test_long_sequences_vs_oracle():
    """T = 100 at the shipped E/H (fast-path kernels, persistent recurrence with 100 hand-offs per
    group, two sentence tiles, one of them partial): forward, dS and all parameter gradients vs the
    fp64 oracle.  Errors of a bf16 recurrence grow with T; the stated tolerance still holds."""
    V, E, H, B, T = 300, 300, 512, 150, 100
    lengths = synth.make_lengths(B, T, seed=77)
    params = synth.make_params(V, E, H, seed=8)
    S = synth.make_soft_tokens(B, T, V, lengths, seed=2, kind="dense", time_major=False)
    labels = synth.make_labels(B, seed=6)
    mask = synth.transfer_mask(lengths, T)
    out = run_sad(params, S, lengths, mask, labels, param_grads=True)
    f, b = _oracle_grads(params, S, lengths, mask, labels)
    errs = _grad_errors(out["grads"], b["grads"])
    errs.update(logits=relerr(out["logits"], f["logits"]), weights=relerr(out["weights"], f["weights"]),
                loss=relerr(out["loss"], f["loss"]), dS=relerr(out["dS"], b["dS"]))
    _report("long sequences T=100", errs)
    assert max(errs.values()) < BF16_TOL_BWD, errs


def test_large_batch_dense_soft_tokens_vs_oracle():
    """M = B*T above the small-batch switch with DENSE soft tokens, ragged sorted lengths and a vocabulary
    that is not a multiple of the k-tile (the stream-K form of K1: partial tiles of several workgroups
    are added up by the fix-up kernel; three-role pipeline with LDS-direct W loads; ragged last k-tile;
    two-workgroups-per-CU recurrence).  Checked: the embedded tokens X of every needed row against
    S . W_emb in fp64, and forward / dS of 16 sentences against the oracle run on just those."""
    B, T, V, E, H = 640, 30, 1111, 300, 512
    lengths = np.sort(synth.make_lengths(B, T, seed=9))[::-1].copy()
    params = synth.make_params(V, E, H, seed=4)
    S = bf16_round(synth.make_soft_tokens(B, T, V, lengths, seed=3, kind="dense", time_major=False))
    labels = synth.make_labels(B, seed=2)
    mask = synth.transfer_mask(lengths, T)
    for s_dtype in ("bf16", "f32"):
        out = run_sad(params, S, lengths, mask, labels, s_dtype=s_dtype, want_ws=True)
        X = out["ws"]["X"][:, :, :E]                                     # [T,B,E]
        Wb = bf16_round(params["embedding.weight"]).astype(np.float64)
        ref = np.einsum("btv,ve->tbe", S.astype(np.float64), Wb)
        alive = (np.arange(T)[:, None] < lengths[None, :])
        errX = np.abs(X - ref)[alive].max() / np.abs(ref[alive]).max()
        sel = np.linspace(0, B - 1, 16).astype(int)
        f, b = _oracle(params, S[sel], lengths[sel], mask[sel], labels[sel], dloss=np.full((16,), 1.0 / B))
        errs = dict(X=errX, logits=relerr(out["logits"][sel], f["logits"]), weights=relerr(out["weights"][sel], f["weights"]),
                    loss=relerr(out["loss"][sel], f["loss"]), dS=relerr(out["dS"][sel], b["dS"]))
        _report("large batch dense soft tokens (%s)" % s_dtype, errs)
        assert errX < 6e-3, errX                                         # bf16 storage of X: half an ulp = 2^-9
        assert errs["logits"] < BF16_TOL_FWD and errs["weights"] < BF16_TOL_FWD and errs["loss"] < BF16_TOL_FWD
        assert errs["dS"] < BF16_TOL_BWD * 1.5


def test_soft_embedding_sibling_contractions():
    """``soft_tokens @ embedding.weight`` as the generator calls it — one decoder step
    ``[1,B,V]`` (transfer/models.py:254), the content-alignment product on ``[T,B,V]`` (:1221), a
    batch-major strided view and a plain 2-D matrix; fp32 and bf16 soft tokens.  Forward, d soft_tokens
    and d weight against the fp64 product of the bf16-rounded operands (stated bf16 tolerance)."""
    import torch
    from sad_amd import functional as Fn
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    V, E = 1111, 300
    W = (rng.standard_normal((V, E)) * 0.05).astype(np.float32)
    cases = [("decoder step", (1, 64, V), False), ("content align", (30, 640, V), False),
             ("batch-major view", (30, 96, V), True), ("matrix", (257, V), False)]
    for name, shape, as_view in cases:
        for dtype in (torch.float32, torch.bfloat16):
            logits = rng.standard_normal(shape).astype(np.float32) * 2.0
            S_np = np.exp(logits - logits.max(-1, keepdims=True))
            S_np = bf16_round((S_np / S_np.sum(-1, keepdims=True)).astype(np.float32))
            G_np = rng.standard_normal(shape[:-1] + (E,)).astype(np.float32)
            S = torch.from_numpy(S_np).to(dev).to(dtype)
            if as_view:      # storage [T,B,V], handed over as [B,T,V]
                S = S.transpose(0, 1)
                S_np, G_np = S_np.swapaxes(0, 1), G_np.swapaxes(0, 1)
            S = S.detach().requires_grad_(True)
            Wt = torch.from_numpy(W).to(dev).requires_grad_(True)
            out = Fn.soft_embedding(S, Wt)
            assert tuple(out.shape) == tuple(S_np.shape[:-1]) + (E,) and out.dtype == dtype
            out.backward(torch.from_numpy(np.ascontiguousarray(G_np)).to(dev).to(dtype))
            Wb, Gb = bf16_round(W).astype(np.float64), bf16_round(G_np).astype(np.float64)
            S64 = S_np.astype(np.float64)
            ref = S64 @ Wb
            ref_dS = Gb @ Wb.T
            ref_dW = S64.reshape(-1, V).T @ Gb.reshape(-1, E)
            errs = dict(out=relerr(out.detach().float().cpu().numpy(), ref),
                        dS=relerr(S.grad.float().cpu().numpy(), ref_dS),
                        dW=relerr(Wt.grad.cpu().numpy(), ref_dW))
            _report("soft_embedding %s %s" % (name, str(dtype).split(".")[-1]), errs)
            assert errs["out"] < 6e-3 and errs["dS"] < 1e-2 and errs["dW"] < 1e-2, (name, dtype, errs)
            assert S.grad.stride() == S.stride() or not as_view


def test_ragged_and_edge_shapes():
    """Edge cases: B=1, all lengths equal to T, len==2 rows (all-masked guard), a batch and T that
    are not multiples of any tile size, more than 256 time steps, unsorted lengths given on the device (no host check)."""
    import torch
    V, E, H = 203, 36, 40
    params = synth.make_params(V, E, H, seed=5)
    # (T = 300: the attention kernels work through the steps in rounds of 256)
    for B, T, lens in ((1, 4, [4]), (5, 7, [7] * 5), (3, 5, [2, 2, 2]), (67, 33, None), (3, 300, [300, 270, 2])):
        lengths = synth.make_lengths(B, T, seed=B) if lens is None else np.array(lens, np.int32)
        S = synth.make_soft_tokens(B, T, V, lengths, seed=B, kind="dense", time_major=False)
        mask = synth.transfer_mask(lengths, T)
        labels = synth.make_labels(B, seed=B)
        out = run_sad(params, S, lengths, mask, labels)
        f, b = _oracle(params, S, lengths, mask, labels)
        errs = dict(logits=relerr(out["logits"], f["logits"]), weights=relerr(out["weights"], f["weights"]),
                    dS=relerr(out["dS"], b["dS"]))
        _report("edge B=%d T=%d" % (B, T), errs)
        assert np.array_equal(out["mask_after"], f["cache"]["mask"])
        assert errs["logits"] < BF16_TOL_FWD and errs["weights"] < BF16_TOL_FWD and errs["dS"] < BF16_TOL_BWD
    # error paths surface as RuntimeError (what the reference's trainers catch)
    from sad_amd import functional as Fn
    pw = Fn.pack_weights({k: torch.from_numpy(v).cuda() for k, v in params.items()})
    S = torch.zeros(2, 3, V, device="cuda")
    m = torch.zeros(2, 3, dtype=torch.bool, device="cuda")
    with pytest.raises(RuntimeError):
        Fn.discriminator(S, torch.tensor([2, 3]), m, pw)        # unsorted CPU lengths
    with pytest.raises(RuntimeError):
        Fn.discriminator(S.cpu(), torch.tensor([3, 2]), m, pw)  # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        Fn.discriminator(torch.zeros(2, 3, V + 1, device="cuda"), torch.tensor([3, 2]), m, pw)
