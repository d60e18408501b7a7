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
