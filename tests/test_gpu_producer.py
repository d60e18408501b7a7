"""GPU tests of the fused producer and the caller stretch (§8 a12), through the
C-ABI, against the fixture recorded from the reference's own TransferModel.forward
(tests/golden/transfer_caller_tiny.npz, tools/gen_golden_transfer.py) and the pinned caller oracle.

Bars: arg-max indices, lengths, masks, zero patterns bit-exact; the straight-through value (1 - p) + p within
2 ulp of the reference's; y_soft / gradients of the producer 1e-5 (fp32 softmax, other exp implementation);
everything that passes through the discriminator within its stated bf16 tolerance (2e-2 / 3e-2)."""
import numpy as np
import pytest

from oracle import caller_oracle as C
from sad_amd import synth
from tests.test_caller_golden import load_caller
from tests.util import relerr

pytestmark = pytest.mark.gpu


def _classifier(hdr, params, dev):
    import torch
    from types import SimpleNamespace
    from sad_amd import transfer_models
    cfg = SimpleNamespace(cell="LSTM", enc_bidirectional=True, pad_id=0, dropout=0.0, enc_num_layers=1,
                          hidden_size=hdr["H"], vocab_size=hdr["V"], emb_size=hdr["E"], num_labels=2,
                          freeze_embedding=False, device=str(dev), mask_id=None)
    m = transfer_models.SelfAttnRNNClassifier(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m.to(dev)
    for p in m.parameters():
        p.requires_grad = False
    return m


@pytest.mark.parametrize("s_dtype", ["f32", "bf16"])
@pytest.mark.parametrize("side_channel", [True, False])
def test_producer_and_caller_stretch_vs_reference_fixture(s_dtype, side_channel):
    import torch
    from sad_amd import producer
    hdr, d, params = load_caller()
    dev = torch.device("cuda:0")
    B, V, n = hdr["B"], hdr["V"], hdr["n_gumbel_steps"]
    dt = torch.float32 if s_dtype == "f32" else torch.bfloat16
    asm = producer.SoftTokenAssembler(B, V, hdr["max_decoding_len"], hdr["pad_id"], hdr["sos_id"], hdr["eos_id"], dev, dtype=dt,
                                      tau=hdr["tau"])
    logits, alive_hist = [], []
    alive = np.ones((B,), bool)
    for i in range(n):
        assert asm.keep_decoding()                                     # the reference's loop condition held here
        lg = torch.from_numpy(d["step%d_logits" % i]).to(dev).requires_grad_(True)
        rnn = asm.step(lg, torch.from_numpy(d["step%d_noise" % i]).to(dev))
        ref = d["step%d_out" % i]
        got = rnn.detach().cpu().numpy()
        assert np.array_equal(got != 0, ref != 0)                      # one non-zero per row, same position
        assert np.abs(got - ref).max() <= 2.4e-7
        assert np.array_equal(asm.last_ids.cpu().numpy(), ref.argmax(axis=1))
        assert relerr(asm._y_soft[i].cpu().numpy(), d["step%d_ysoft" % i]) < 1e-5
        logits.append(lg)
        alive_hist.append(alive.copy())
        alive = alive & (ref.argmax(axis=1) != hdr["eos_id"])
    assert not asm.keep_decoding()                                     # ... and ended the loop here
    soft = asm.finalize()
    S = soft.S.detach().float().cpu().numpy()
    assert S.shape == d["soft_out"].shape
    assert np.array_equal(S != 0, d["soft_out"] != 0)
    assert np.abs(S - d["soft_out"]).max() <= (2.4e-7 if s_dtype == "f32" else 4e-3)
    assert np.array_equal(soft.lens.cpu().numpy(), d["soft_out_len"].astype(np.int32))
    assert np.array_equal(soft.ids.cpu().numpy(), d["soft_out"].argmax(axis=2))
    assert np.abs(soft.val.cpu().numpy() - d["soft_out"].max(axis=2)).max() <= 2.4e-7
    mask_unsorted = d["mask_before"][d["resorted_soft_out_indices"]]
    assert np.array_equal(soft.mask.cpu().numpy(), mask_unsorted)

    # ---- the caller stretch around the drop-in discriminator (transfer/models.py:1172-1212)
    cls = _classifier(hdr, params, dev)
    inv_lab = torch.from_numpy(d["inv_lab"]).to(dev)
    r = producer.score_soft_tokens(cls, soft, inv_lab, use_side_channel=side_channel)
    assert np.array_equal(r["sorted_soft_out_len"].cpu().numpy(), d["sorted_soft_out_len"])
    # the mask the discriminator mutated in place (all-masked rows: position 1 unmasked), in the fixture's order
    order = d["soft_out_indices"]
    back = np.argsort(r["soft_out_indices"].cpu().numpy(), kind="stable")
    mask_after_unsorted = r["sorted_mask"].cpu().numpy()[back]
    assert np.array_equal(mask_after_unsorted[order], d["mask_after"])
    errs = dict(loss=relerr(r["soft_out_cls_loss"].detach().cpu().numpy(), d["soft_out_cls_loss"]),
                logits=relerr(r["soft_out_cls_logits"].detach().cpu().numpy(), d["soft_out_cls_logits"]),
                content_alpha=relerr(r["soft_out_content_alpha"].cpu().numpy(), d["soft_out_content_alpha"]))
    r["soft_out_cls_loss"].mean().backward()
    # d logits_t through S only (the decoder that would also consume rnn_in is out of scope): the pinned
    # oracle formula on the reference's own d soft_out and y_soft
    for i in range(n):
        g = d["d_soft_out"][i + 1] * alive_hist[i][:, None]
        want = C.gumbel_softmax_hard_backward(g.astype(np.float64), d["step%d_ysoft" % i].astype(np.float64), hdr["tau"])
        errs["dlogits%d" % i] = relerr(logits[i].grad.cpu().numpy(), want, floor=1e-8)
    print("\n[producer + caller stretch %s side_channel=%s]" % (s_dtype, side_channel), errs)
    assert errs["loss"] < 2e-2 and errs["logits"] < 2e-2 and errs["content_alpha"] < 2e-2, errs
    assert max(v for k, v in errs.items() if k.startswith("dlogits")) < 4e-2, errs


def test_rnn_in_gradient_path_and_side_channel_equals_gemm():
    """(a) the gradient that arrives through the un-masked sample (next decoder input) follows the
    straight-through formula; (b) at Yelp dims the side-channel gather gives bit-identical embedded
    tokens, logits and d(soft tokens) to the dense GEMM on the same one-hot rows."""
    import torch
    from sad_amd import functional as Fn, producer
    Fn.KEEP_LAST = True
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    B, V, E, H, MAXDEC = 96, 9353, 300, 512, 20
    asm = producer.SoftTokenAssembler(B, V, MAXDEC, 0, 1, 2, dev, dtype=torch.bfloat16)
    first = None
    while asm.keep_decoding():
        lg = (torch.randn((B, V), generator=g, device=dev) * 2.0)
        lg[:, 2] += 2.0 + 0.4 * asm.steps                               # [eos] more and more likely: ragged lengths
        lg.requires_grad_(True)
        rnn = asm.step(lg, producer.gumbel_noise_like(lg))
        if first is None:
            first = (lg, rnn, asm._y_soft[0])
    soft = asm.finalize()
    lens = soft.lens.cpu().numpy()
    assert lens.min() >= 2 and lens.max() <= MAXDEC + 3 and len(set(lens.tolist())) > 3
    lg, rnn, y = first
    G = torch.randn((B, V), generator=g, device=dev)
    rnn.backward(G)
    want = C.gumbel_softmax_hard_backward(G.double().cpu().numpy(), y.double().cpu().numpy(), 1.0)
    assert relerr(lg.grad.cpu().numpy(), want) < 1e-5
    # (b)
    params = synth.make_params(V, E, H, seed=5)
    pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
    sorted_len, idx = torch.sort(soft.lens, descending=True)
    S = torch.index_select(soft.S.detach(), 1, idx)
    T = S.shape[0]
    mask = torch.arange(T, device=dev)[None, :] >= (sorted_len[:, None] - 1)
    mask[:, 0] = True
    lab = torch.from_numpy(synth.make_labels(B, seed=5)).to(dev)
    outs = []
    for onehot in (None, (torch.index_select(soft.ids, 1, idx).t(), torch.index_select(soft.val, 1, idx).t())):
        Sv = S.transpose(0, 1).detach().requires_grad_(True)
        loss, logits, w = Fn.discriminator_loss(Sv, sorted_len, mask.clone(), lab, pw, onehot=onehot)
        ws = Fn._Discriminator.last_workspace
        X = ws.view("X", torch.bfloat16).clone()
        loss.mean().backward()
        torch.cuda.synchronize()
        ws.check()
        outs.append((X, logits.detach(), Sv.grad))
    live = (ws.view("NEED", torch.uint8).reshape(T, B) != 0).reshape(-1)
    Xa, Xb = outs[0][0].reshape(T * B, -1)[live], outs[1][0].reshape(T * B, -1)[live]
    assert torch.equal(Xa, Xb)
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
