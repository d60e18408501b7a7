"""Pins oracle/caller_oracle.py (the caller side of the hot path: Gumbel straight-through producer,
soft_decode bookkeeping, sort / mask / un-sort / CE(1-lab) / content-alpha stretch) against the fixture
recorded from the reference's own TransferModel.forward (tools/gen_golden_transfer.py)."""
import json
import os

import numpy as np

from oracle import caller_oracle as C
from tests.util import GOLDEN, relerr


def load_caller():
    z = np.load(os.path.join(GOLDEN, "transfer_caller_tiny.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    hdr = json.loads(bytes(d.pop("header_json")).decode())
    params = {k[len("param:"):]: v for k, v in d.items() if k.startswith("param:")}
    return hdr, d, params


def test_gumbel_straight_through_matches_reference_samples():
    hdr, d, _ = load_caller()
    assert hdr["hard"] and hdr["tau"] == 1.0
    for i in range(hdr["n_gumbel_steps"]):
        ret, y, idx = C.gumbel_softmax_hard(d["step%d_logits" % i], d["step%d_noise" % i], hdr["tau"])
        ref = d["step%d_out" % i]
        assert np.array_equal(idx, ref.argmax(axis=1))                 # index: bit-exact
        assert np.array_equal(ret != 0, ref != 0)                      # exact zeros off the arg-max
        assert np.abs(ret - ref).max() <= 2.4e-7                       # (1 - p) + p: within 2 ulp of 1
        assert relerr(y, d["step%d_ysoft" % i]) < 1e-6


def test_soft_decode_assembly_matches_reference():
    hdr, d, _ = load_caller()
    steps = [d["step%d_out" % i] for i in range(hdr["n_gumbel_steps"])]
    out, out_len = C.soft_decode_assemble(steps, hdr["V"], hdr["pad_id"], hdr["sos_id"], hdr["eos_id"], hdr["max_decoding_len"])
    assert out_len.dtype == np.int16 and np.array_equal(out_len, d["soft_out_len"])
    assert out.shape == d["soft_out"].shape and np.array_equal(out, d["soft_out"])


def test_caller_stretch_matches_reference():
    hdr, d, params = load_caller()
    ln = d["soft_out_len"].astype(np.int64)
    o = C._stretch_with_indices({k: v.astype(np.float64) for k, v in params.items()}, d["soft_out"].astype(np.float64), ln,
                                d["inv_lab"], d["soft_out_indices"])
    assert np.array_equal(o["sorted_soft_out_len"], d["sorted_soft_out_len"])
    assert np.array_equal(o["resorted_soft_out_indices"], d["resorted_soft_out_indices"])
    assert np.array_equal(o["mask_before"], d["mask_before"]) and np.array_equal(o["mask_after"], d["mask_after"])
    assert (d["mask_before"] != d["mask_after"]).sum() == 2             # the fixture exercises the all-masked guard
    for k in ("sorted_logits", "soft_out_cls_logits", "soft_out_cls_loss", "sorted_soft_out_content_alpha",
              "soft_out_content_alpha", "d_soft_out"):
        assert relerr(o[k], d[k]) < 2e-5, k                             # fp32 reference vs fp64 restatement
    # own sort (ties may come out in another order): everything un-sorted is order independent
    o2 = C.caller_stretch({k: v.astype(np.float64) for k, v in params.items()}, d["soft_out"].astype(np.float64),
                          d["soft_out_len"], d["inv_lab"])
    assert np.array_equal(o2["sorted_soft_out_len"], d["sorted_soft_out_len"])
    for k in ("soft_out_cls_logits", "soft_out_cls_loss", "soft_out_content_alpha", "d_soft_out"):
        assert relerr(o2[k], d[k]) < 2e-5, k
