"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/sad.h declares,
and its size/argument logic behaves (no compute call is made — there is no GPU here)."""
import ctypes
import os
import re

import pytest

from sad_amd import lib
from tests.util import ROOT


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "sad.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sad_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    l = lib.load()
    declared = _header_symbols()
    assert declared, "no declarations parsed from include/sad.h"
    assert sorted(lib.EXPORTS) == declared
    for name in declared:
        assert hasattr(l, name), name
    assert l.sad_version() == 300


def test_layout_and_padded_dims():
    l = lib.load()
    d = lib.dims(1024, 32, 9353, 300, 512, 2)
    pd = lib.padded_dims(d)
    assert pd == dict(Ep=320, Hp=512, Gp=2048, Vp=9408, Ncat=4608)
    total = l.sad_workspace_bytes(ctypes.byref(d))
    end = 0
    for name in lib.WS_FIELDS:
        off, nb = lib.workspace_field(d, name)
        assert off % 256 == 0 and off >= end and nb > 0
        end = off + nb
    assert end <= total
    M = 1024 * 32
    assert lib.workspace_field(d, "G")[1] == M * (2 * 2048 + 64) * 4
    assert lib.workspace_field(d, "DZ")[1] == M * 4608 * 2
    # packed image: both orientations of the embedding dominate
    assert l.sad_packed_bytes(ctypes.byref(d)) > 2 * 9408 * 320 * 2
    # tiny / odd dims pad up
    assert lib.padded_dims(lib.dims(6, 9, 41, 10, 8, 2)) == dict(Ep=32, Hp=32, Gp=128, Vp=64, Ncat=288)


def test_argument_errors_are_reported_not_crashed():
    l = lib.load()
    bad = lib.dims(0, 32, 9353, 300, 512, 2)
    assert l.sad_workspace_bytes(ctypes.byref(bad)) == 0
    assert "positive" in lib.last_error()
    d = lib.dims(4, 2000, 100, 10, 8, 2)          # T beyond the supported maximum
    assert l.sad_packed_bytes(ctypes.byref(d)) == 0
    assert "maximum" in lib.last_error()
    d = lib.dims(4, 8, 100, 10, 8, 2)
    a = lib.SadFwdArgs()
    rc = l.sad_forward(ctypes.byref(d), None, ctypes.byref(a), None, None)   # NULL packed: refused before any launch
    assert rc < 0 and "NULL" in lib.last_error()
    with pytest.raises(RuntimeError, match="libsad"):
        lib.check(rc, "sad_forward")


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libsad.so")
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        lib.load()


def test_cpu_tensors_are_refused():
    """The product path never falls back to CPU / stock torch ops."""
    import torch
    from sad_amd import functional as Fn

    class FakePW:
        V, E, H, L = 50, 8, 8, 2
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Fn.discriminator(torch.zeros(2, 3, 50), torch.tensor([3, 2]), torch.zeros(2, 3, dtype=torch.bool), FakePW())


def test_soft_embedding_has_no_cpu_path():
    """The sibling contraction runs on the HIP kernels only: CPU tensors are refused."""
    import pytest
    import torch
    from sad_amd import functional as Fn
    with pytest.raises(RuntimeError, match="no CPU path"):
        Fn.soft_embedding(torch.rand(2, 3, 7), torch.rand(7, 4))

