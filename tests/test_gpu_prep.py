"""Bit-exact check of the prep stage through the C-ABI (``pytest -m gpu``): clamped lengths, the
all-masked-row guard applied in place on the caller's mask (reference transfer/models.py:865-867), the
per-token need bits, the three compacted row lists, the list of non-empty 128-row tiles and the zeroed
hand-off words — against a numpy restatement, for the single-launch kernel (k_prep_lists) and for the
three-kernel path taken at large M or B, contiguous and strided masks, ragged sizes around every
chunk width the kernels use (16 rows per thread, 128-row tiles, 1024-row segments).
"""
import ctypes

import numpy as np
import pytest

from sad_amd import lib

pytestmark = pytest.mark.gpu


def _expected(lens, mask, guard, B, T):
    ln = np.clip(lens, 0, T).astype(np.int32)
    m = mask.astype(bool).copy()
    if guard and T > 1:
        m[m.all(axis=1), 1] = False
    t = np.arange(T)[:, None]
    need = ((t < ln[None, :]).astype(np.uint8) | (2 * (~m.T).astype(np.uint8))).reshape(-1)        # [t*B + b]
    M = B * T
    idx = np.arange(M)
    bit0 = (need & 1) != 0
    nxt = np.zeros(M, bool)
    nxt[:M - B] = bit0[B:]
    lists = [idx[need != 0], idx[(idx >= B) & bit0], idx[nxt]]
    nt = (M + 127) // 128
    pad = np.zeros(nt * 128, np.uint8)
    pad[:M] = need
    tiles = np.nonzero(pad.reshape(nt, 128).any(axis=1))[0]
    return ln, m, need, lists, tiles


CASES = [
    # B, T, E, layout, guard
    (1, 1, 300, "contig", 1),
    (3, 5, 300, "contig", 1),
    (64, 20, 300, "contig", 1),
    (7, 33, 300, "strided", 1),
    (129, 16, 300, "contig", 0),
    (1024, 32, 300, "contig", 1),          # cfg3
    (1024, 32, 300, "transposed", 1),
    (1500, 17, 300, "contig", 1),          # B above one thread per sentence, M % 16 != 0
    (2048, 32, 300, "offset", 1),          # M = 65536: the largest single-launch case; mask not 16-byte aligned
    (2049, 3, 300, "contig", 1),           # B above the single-launch limit
    (100, 700, 300, "contig", 1),          # M above the single-launch limit
    (64, 20, 16, "contig", 1),             # no stream-K tile list in the workspace
]


@pytest.mark.parametrize("B,T,E,layout,guard", CASES)
def test_prep_lists_bit_exact(B, T, E, layout, guard):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(B * 1000 + T)
    lens = rng.integers(-2, T + 3, size=B).astype(np.int32)
    lens[rng.integers(0, B)] = 0
    mask = rng.random((B, T)) < 0.6
    for b in rng.integers(0, B, size=max(1, B // 7)):
        mask[b] = True                                           # rows the guard applies to
    if B > 4:
        mask[3] = False
    # a stretch of sentences with no needed rows at all: empty 128-row tiles when B >= 128
    if B >= 256:
        lens[:192] = 0
        mask[:192, 2:] = True

    if layout == "contig":
        mdev = torch.from_numpy(mask).to(dev)
    elif layout == "strided":
        big = torch.ones((B, 2 * T + 3), dtype=torch.bool, device=dev)
        mdev = big[:, 1:2 * T + 1:2]
        mdev.copy_(torch.from_numpy(mask).to(dev))
    elif layout == "transposed":
        mdev = torch.from_numpy(np.ascontiguousarray(mask.T)).to(dev).t()
    else:                                                        # contiguous, base address 1 byte past an aligned one
        flat = torch.ones((B * T + 1,), dtype=torch.bool, device=dev)
        mdev = flat[1:].view(B, T)
        mdev.copy_(torch.from_numpy(mask).to(dev))
    ldev = torch.from_numpy(lens).to(dev)

    d = lib.dims(B, T, 50, E, 32, 2)
    L = lib.load()
    nbytes = L.sad_workspace_bytes(ctypes.byref(d))
    ws = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.check(L.sad_prep(ctypes.byref(d), ctypes.c_void_p(ldev.data_ptr()), ctypes.c_void_p(mdev.data_ptr()),
                         mdev.stride(0), mdev.stride(1), guard, ctypes.c_void_p(ws.data_ptr()), st), "sad_prep")
    torch.cuda.synchronize()

    def field(name, dtype):
        off, nb = lib.workspace_field(d, name)
        return ws[off:off + nb].cpu().numpy().view(dtype)

    ln, m2, need, lists, tiles = _expected(lens, mask, guard, B, T)
    M = B * T
    assert np.array_equal(field("LENS", np.int32)[:B], ln)
    assert np.array_equal(mdev.cpu().numpy(), m2), "guard not applied in place as the reference does"
    assert np.array_equal(field("NEED", np.uint8)[:M], need)
    nrows = field("NROWS", np.int32)
    rows = field("ROWS", np.int32).reshape(3, M)
    for l in range(3):
        assert nrows[l] == len(lists[l]), (l, nrows[l], len(lists[l]))
        assert np.array_equal(rows[l, :nrows[l]], lists[l]), "row list %d" % l
    assert not field("SYNC", np.uint32).any(), "hand-off words not zeroed"
    if E == 300:
        tl = field("K1TILES", np.int32)
        assert tl[0] == len(tiles)
        assert np.array_equal(tl[1:1 + tl[0]], tiles)
    else:
        assert lib.workspace_field(d, "K1TILES")[1] == 0
