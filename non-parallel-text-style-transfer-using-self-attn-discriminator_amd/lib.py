"""ctypes binding of libsad.so (the C-ABI declared in include/sad.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/build.sh`` into
``<package>/_native/libsad.so``.  There is no fallback: if the library is missing, every entry point
raises ``RuntimeError`` (the product path never routes through the oracle or through stock torch
ops).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAD_LIB") or os.path.join(_HERE, "_native", "libsad.so")   # SAD_LIB: A/B builds

SRC_IDS, SRC_F32, SRC_BF16 = 0, 1, 2

WS_FIELDS = ("LENS", "NEED", "X", "G", "CS", "HS", "KK", "Q", "U", "CTX", "DZ", "DX", "DU", "DQ",
             "DHFIN", "DC", "ACCF", "DL", "DWA", "DSSUM", "ROWS", "NROWS", "SYNC", "DZX", "HX", "K1TILES", "K1SLABS", "GA")

# every symbol include/sad.h declares (tests/test_cabi.py checks the .so exports all of them)
EXPORTS = ("sad_version", "sad_last_error", "sad_padded_dims", "sad_packed_bytes",
           "sad_workspace_bytes", "sad_workspace_field", "sad_pack_weights", "sad_forward",
           "sad_backward", "sad_param_grads_accumulate", "sad_prep", "sad_softemb_fwd", "sad_softemb_bwd", "sad_uses_half_gates",
           "sad_inproj_fwd", "sad_inproj_bwd", "sad_lstm_fwd", "sad_lstm_bwd", "sad_attnpool_fwd", "sad_attnpool_bwd", "sad_debug_set", "sad_forward_noalloc",
           "sad_gumbel_st_step", "sad_gumbel_st_step_bwd", "sad_soft_finalize")

DBG_NO_PERSIST, DBG_SPIN_LIMIT, DBG_WITHHOLD_FWD, DBG_WITHHOLD_BWD, DBG_WITHHOLD_UPPER_ONLY, DBG_WGRAD_ATOMICS = 0, 1, 2, 3, 4, 5


class SadDims(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("B", "T", "V", "E", "H", "L", "cell")]


class SadWeights(ctypes.Structure):
    _fields_ = [
        ("embedding_weight", c_void_p),
        ("rnn_weight_ih", c_void_p * 2),
        ("rnn_weight_hh", c_void_p * 2),
        ("rnn_bias_ih", c_void_p * 2),
        ("rnn_bias_hh", c_void_p * 2),
        ("hidden_bridge0_weight", c_void_p),
        ("hidden_bridge0_bias", c_void_p),
        ("linear_emb_weight", c_void_p),
        ("linear_emb_bias", c_void_p),
        ("linear_q_weight", c_void_p),
        ("linear_q_bias", c_void_p),
        ("linear_attn_weight", c_void_p),
        ("linear_attn_bias", c_void_p),
        ("output_layer_weight", c_void_p),
        ("output_layer_bias", c_void_p),
    ]


class SadParamGrads(ctypes.Structure):
    """sad_param_grads: same tensors as sad_weights, writable; NULL = frozen tensor."""
    _fields_ = SadWeights._fields_ + [("final_hidden_workspace", c_void_p)]


class SadRnnLayer(ctypes.Structure):
    """sad_rnn_layer: one RNN layer above the first."""
    _fields_ = [("packed", c_void_p), ("workspace", c_void_p), ("drop_keep", c_void_p), ("drop_scale", c_float)]


class SadFwdArgs(ctypes.Structure):
    _fields_ = [
        ("src", c_void_p), ("src_kind", c_int32), ("src_stride_b", c_int64), ("src_stride_t", c_int64),
        ("lengths", c_void_p),
        ("mask", c_void_p), ("mask_stride_b", c_int64), ("mask_stride_t", c_int64), ("guard", c_int32),
        ("labels", c_void_p),
        ("logits", c_void_p), ("weights", c_void_p), ("loss", c_void_p),
        ("drop_keep", c_void_p), ("drop_scale", c_float),
        ("onehot_ids", c_void_p), ("onehot_val", c_void_p), ("onehot_stride_b", c_int64), ("onehot_stride_t", c_int64),
        ("upper", ctypes.POINTER(SadRnnLayer)), ("n_upper", c_int32),
    ]


class SadGumbelArgs(ctypes.Structure):
    _fields_ = [
        ("B", c_int32), ("V", c_int32),
        ("logits", c_void_p), ("logits_stride", c_int64), ("noise", c_void_p), ("noise_stride", c_int64),
        ("tau", c_float), ("alive", c_void_p), ("eos_id", c_int32),
        ("s_row", c_void_p), ("s_kind", c_int32), ("s_stride_b", c_int64),
        ("rnn_in", c_void_p), ("y_soft", c_void_p), ("idx", c_void_p), ("val", c_void_p),
        ("lens", c_void_p), ("alive_out", c_void_p),
    ]


class SadBwdArgs(ctypes.Structure):
    _fields_ = [
        ("dlogits", c_void_p), ("labels", c_void_p), ("dloss", c_void_p), ("dloss_scale", c_float),
        ("dweights", c_void_p), ("logits", c_void_p), ("weights", c_void_p),
        ("dsrc", c_void_p), ("dsrc_kind", c_int32), ("dsrc_stride_b", c_int64), ("dsrc_stride_t", c_int64),
        ("drop_keep", c_void_p), ("drop_scale", c_float),
        ("upper", ctypes.POINTER(SadRnnLayer)), ("n_upper", c_int32),
    ]


_lib = None


def load():
    """Load libsad.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it brings its own copy of the HIP runtime, and libsad must bind to that one — loaded the other
    # way round the process holds two runtimes and libsad's first launch fails with "no ROCm-capable device"
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsad.so not found at %s — the HIP extension is required (no CPU/eager fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`bash <package>/csrc/build.sh`." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.sad_version.restype = c_int
    lib.sad_last_error.restype = c_char_p
    lib.sad_padded_dims.argtypes = [POINTER(SadDims), POINTER(c_int32 * 5)]
    lib.sad_packed_bytes.argtypes = [POINTER(SadDims)]
    lib.sad_packed_bytes.restype = c_size_t
    lib.sad_workspace_bytes.argtypes = [POINTER(SadDims)]
    lib.sad_workspace_bytes.restype = c_size_t
    lib.sad_workspace_field.argtypes = [POINTER(SadDims), c_int, POINTER(c_size_t), POINTER(c_size_t)]
    lib.sad_pack_weights.argtypes = [POINTER(SadDims), POINTER(SadWeights), c_void_p, c_void_p]
    lib.sad_forward.argtypes = [POINTER(SadDims), c_void_p, POINTER(SadFwdArgs), c_void_p, c_void_p]
    lib.sad_backward.argtypes = [POINTER(SadDims), c_void_p, POINTER(SadBwdArgs), c_void_p, c_void_p]
    lib.sad_forward_noalloc.argtypes = [POINTER(SadDims), c_void_p, POINTER(SadFwdArgs), c_void_p]
    lib.sad_param_grads_accumulate.argtypes = [POINTER(SadDims), POINTER(SadParamGrads), c_void_p, c_int32, c_int64,
                                               c_int64, c_int32, c_void_p, c_void_p]
    lib.sad_uses_half_gates.argtypes = [POINTER(SadDims)]
    lib.sad_uses_half_gates.restype = c_int
    lib.sad_prep.argtypes = [POINTER(SadDims), c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p]
    lib.sad_softemb_fwd.argtypes = [POINTER(SadDims), c_void_p, c_void_p, c_int32, c_int64, c_int64, c_void_p, c_void_p]
    lib.sad_softemb_bwd.argtypes = [POINTER(SadDims), c_void_p, c_void_p, c_int32, c_int64, c_int64, c_void_p, c_void_p]
    lib.sad_lstm_fwd.argtypes = [POINTER(SadDims), c_void_p, c_void_p, c_void_p]
    lib.sad_inproj_fwd.argtypes = [POINTER(SadDims), c_void_p, c_void_p, c_void_p]
    lib.sad_inproj_bwd.argtypes = [POINTER(SadDims), c_void_p, c_void_p, c_void_p]
    lib.sad_lstm_bwd.argtypes = [POINTER(SadDims), c_void_p, c_void_p, c_void_p]
    lib.sad_attnpool_fwd.argtypes = [POINTER(SadDims), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.sad_attnpool_bwd.argtypes = [POINTER(SadDims), c_void_p, POINTER(SadBwdArgs), c_void_p, c_void_p]
    lib.sad_debug_set.argtypes = [c_int32, c_int64]
    lib.sad_gumbel_st_step.argtypes = [POINTER(SadGumbelArgs), c_void_p]
    lib.sad_gumbel_st_step_bwd.argtypes = [c_int32, c_int32, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_float,
                                           c_void_p, c_int64, c_void_p]
    lib.sad_soft_finalize.argtypes = [c_int32] * 8 + [c_void_p, c_int32, c_int64, c_int64] + [c_void_p] * 7
    for name in EXPORTS:
        getattr(lib, name)   # AttributeError here = header/library drift
    _lib = lib
    return lib


def last_error():
    return load().sad_last_error().decode("utf-8", "replace")


def check(rc, what):
    """Map a negative return code to the RuntimeError the reference's callers catch
    (reference classifier/trainers.py:375-382 special-cases 'out of memory' in the message)."""
    if rc != 0:
        raise RuntimeError("libsad %s failed (%d): %s" % (what, rc, last_error()))


def debug_set(key, value):
    """Process-wide test / diagnosis settings of the library (include/sad.h, sad_debug_set)."""
    check(load().sad_debug_set(int(key), int(value)), "sad_debug_set")


CELL_LSTM, CELL_GRU = 0, 1
MAX_UPPER_LAYERS = 7       # SAD_MAX_UPPER_LAYERS


def dims(B, T, V, E, H, L, cell=CELL_LSTM):
    return SadDims(int(B), int(T), int(V), int(E), int(H), int(L), int(cell))


def padded_dims(d):
    out = (c_int32 * 5)()
    check(load().sad_padded_dims(ctypes.byref(d), ctypes.byref(out)), "sad_padded_dims")
    return dict(Ep=out[0], Hp=out[1], Gp=out[2], Vp=out[3], Ncat=out[4])


def workspace_field(d, name):
    off, nb = c_size_t(), c_size_t()
    check(load().sad_workspace_field(ctypes.byref(d), WS_FIELDS.index(name), ctypes.byref(off), ctypes.byref(nb)),
          "sad_workspace_field")
    return off.value, nb.value
