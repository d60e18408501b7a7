"""torch.autograd bridge to libsad.so.

PyTorch is plumbing here (device memory, streams, autograd graph); every arithmetic step of the
discriminator runs in the hand-written HIP kernels behind the C-ABI of include/sad.h.

    pw = pack_weights(named_params, V, E, H, L)                    # fp32 state_dict -> packed image
    logits, weights = discriminator(src, src_len, mask, pw, soft=True)            # module API
    loss, logits, weights = discriminator_loss(src, src_len, mask, labels, pw)    # fused CE

``src`` is ``[B,T,V]`` soft tokens (fp32/bf16, any batch/time strides with a contiguous vocabulary
axis — e.g. the transfer loop's ``sorted_soft_out.transpose(0,1)``, reference
transfer/models.py:1186) or ``[B,T]`` int64 ids.  The bool ``mask`` is mutated in place exactly as
the reference mutates it (transfer/models.py:865-867).
"""
import contextlib
import ctypes

import numpy as np
import torch

from . import lib as _lib

_LIVE_PARAMS = (
    "embedding.weight",
    "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0",
    "rnn.weight_ih_l0_reverse", "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0_reverse", "rnn.bias_hh_l0_reverse",
    "hidden_bridges.0.weight", "hidden_bridges.0.bias",
    "linear_emb.weight", "linear_emb.bias", "linear_q.weight", "linear_q.bias",
    "linear_attn.weight", "linear_attn.bias", "output_layer.weight", "output_layer.bias",
)


_RNN_PARAMS = _LIVE_PARAMS[1:9]      # the eight rnn.* tensors: all a layer above the first owns


def _raw_stream_fallback(index):
    return torch.cuda.current_stream(index).cuda_stream


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", _raw_stream_fallback)


def _stream_ptr():
    # the raw stream of the current device straight from the C layer: torch.cuda.current_stream() resolves its default
    # device through torch.cuda.is_available(), which reads the environment on every call (measured on the GPU box:
    # the longest single item of the module path's host time at batch 64)
    return ctypes.c_void_p(_RAW_STREAM(torch.cuda.current_device()))


_NULL_CTX = contextlib.nullcontext()


def _on(dev):
    """Context that makes `dev` the current device; nothing when it already is (the switch costs host time)."""
    return _NULL_CTX if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class PackedWeights:
    """Device image of the discriminator weights in the kernels' layout (bf16, padded to MFMA tile
    multiples, LSTM gate rows permuted, plus pre-transposed copies for the backward GEMMs)."""

    def __init__(self, V, E, H, L, device, cell=_lib.CELL_LSTM):
        self.V, self.E, self.H, self.L, self.cell = int(V), int(E), int(H), int(L), int(cell)
        self.device = torch.device(device)
        self.upper = []      # PackedWeights of the RNN layers above the first (dims V=1, E=2H)
        d = _lib.dims(1, 1, V, E, H, L, cell)
        nbytes = _lib.load().sad_packed_bytes(ctypes.byref(d))
        if nbytes == 0:
            raise RuntimeError("libsad sad_packed_bytes failed: " + _lib.last_error())
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.key = None

    def pack(self, params):
        """params: mapping name -> fp32 contiguous CUDA tensor (state_dict names)."""
        w = _lib.SadWeights()
        keep = []

        def p(name):
            t = params[name].detach()
            if t.dtype != torch.float32 or not t.is_cuda:
                raise RuntimeError("parameter %s must be a float32 tensor on the HIP device" % name)
            t = t.contiguous()
            keep.append(t)
            return t.data_ptr()

        w.embedding_weight = p("embedding.weight")
        for i, sfx in enumerate(("", "_reverse")):
            w.rnn_weight_ih[i] = p("rnn.weight_ih_l0" + sfx)
            w.rnn_weight_hh[i] = p("rnn.weight_hh_l0" + sfx)
            w.rnn_bias_ih[i] = p("rnn.bias_ih_l0" + sfx)
            w.rnn_bias_hh[i] = p("rnn.bias_hh_l0" + sfx)
        w.hidden_bridge0_weight = p("hidden_bridges.0.weight")
        w.hidden_bridge0_bias = p("hidden_bridges.0.bias")
        w.linear_emb_weight = p("linear_emb.weight")
        w.linear_emb_bias = p("linear_emb.bias")
        w.linear_q_weight = p("linear_q.weight")
        w.linear_q_bias = p("linear_q.bias")
        w.linear_attn_weight = p("linear_attn.weight")
        w.linear_attn_bias = p("linear_attn.bias")
        w.output_layer_weight = p("output_layer.weight")
        w.output_layer_bias = p("output_layer.bias")
        d = _lib.dims(1, 1, self.V, self.E, self.H, self.L, self.cell)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().sad_pack_weights(ctypes.byref(d), ctypes.byref(w), _ptr(self.buf), _stream_ptr()),
                       "sad_pack_weights")
        return self


_zero_cache = {}


def _upper_canonical(rnn_params, H, L, dev):
    """The 19-tensor mapping of an RNN layer above the first: its own rnn.* tensors (input width 2H), zeros
    for everything else (include/sad.h, sad_rnn_layer)."""
    key = (H, L, str(dev))
    z = _zero_cache.get(key)
    if z is None:
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        z = {"embedding.weight": f(1, 2 * H), "hidden_bridges.0.weight": f(H, 2 * H), "hidden_bridges.0.bias": f(H),
             "linear_emb.weight": f(H, 2 * H), "linear_emb.bias": f(H), "linear_q.weight": f(H, H),
             "linear_q.bias": f(H), "linear_attn.weight": f(1, H), "linear_attn.bias": f(1),
             "output_layer.weight": f(L, H), "output_layer.bias": f(L)}
        _zero_cache[key] = z
    out = dict(z)
    out.update({k: rnn_params[k] for k in _RNN_PARAMS})
    return out


def pack_weights(params, V=None, E=None, H=None, L=None, cell=_lib.CELL_LSTM, upper=()):
    """Build a PackedWeights from a name->tensor mapping (a module's ``state_dict()`` or
    ``dict(named_parameters())``) in the kernels' canonical layout: bidirectional, four gate slots
    (``cell=CELL_GRU``: include/sad.h).  ``upper``: one mapping of the eight ``rnn.*_l0*`` names per RNN layer
    above the first (input width 2H)."""
    emb = params["embedding.weight"]
    V = emb.shape[0] if V is None else V
    E = emb.shape[1] if E is None else E
    H = params["rnn.weight_hh_l0"].shape[1] if H is None else H
    L = params["output_layer.weight"].shape[0] if L is None else L
    pw = PackedWeights(V, E, H, L, emb.device, cell).pack(params)
    for up in upper:
        pw.upper.append(PackedWeights(1, 2 * H, H, L, emb.device, cell).pack(_upper_canonical(up, H, L, emb.device)))
    return pw


def _lengths_to_device(src_len, B, device):
    """Reference callers pass CPU int16/int32/int64 lengths (transfer/models.py:1172, utils/data.py:84);
    sortedness is what ``pack_padded_sequence(enforce_sorted=True)`` enforces (transfer/models.py:904)."""
    if not torch.is_tensor(src_len):
        src_len = torch.as_tensor(src_len)
    if src_len.numel() != B:
        raise RuntimeError("src_len has %d entries for a batch of %d" % (src_len.numel(), B))
    if not src_len.is_cuda:
        # numpy on the tensor's own memory: the same checks as torch ops cost ~25 us of host time per call, which
        # is what a batch-64 step through the module API waits for
        ln = src_len.detach().contiguous().numpy()
        if B > 1 and bool((ln[1:] > ln[:-1]).any()):
            raise RuntimeError("`lengths` array must be sorted in decreasing order when `enforce_sorted` is True.")
        if B > 0 and int(ln.min()) <= 0:
            raise RuntimeError("Length of all samples has to be greater than 0, but found an element in "
                               "'lengths' that is <= 0")
        return torch.from_numpy(ln.astype(np.int32, copy=False)).to(device=device, non_blocking=True)
    return src_len.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()


class Workspace:
    """Per-call activation/scratch buffer (torch caching allocator; libsad itself never allocates)."""

    def __init__(self, d, device):
        nbytes = _lib.load().sad_workspace_bytes(ctypes.byref(d))
        if nbytes == 0:
            raise RuntimeError("libsad sad_workspace_bytes failed: " + _lib.last_error())
        self.d = d
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.upper = []      # workspaces of the RNN layers above the first (each runs its own recurrences)

    def view(self, name, dtype):
        off, nb = _lib.workspace_field(self.d, name)
        return self.buf[off:off + nb].view(dtype)

    def handoff_timeouts(self):
        """(forward, backward) timeout words of the persistent recurrences.  SYNCHRONISES (reads device
        memory): call it where a sync happens anyway (tests, benchmarks, a logging interval)."""
        f = b = 0
        for w in [self] + list(self.upper):          # a multi-layer model: any layer's recurrences
            words = w.view("SYNC", torch.int32)[-32:-30].tolist()
            f, b = f | int(words[0]), b | int(words[1])
        return f, b

    def check(self):
        """Raise RuntimeError if a hand-off wait inside a persistent recurrence timed out during the
        calls that used this workspace (the outputs of such a call are NaN by construction; this gives
        the reason).  The usual cause: the recurrence's workgroups were not all resident because another
        kernel occupied CUs of the device (another stream or process, CU masking)."""
        f, b = self.handoff_timeouts()
        if f or b:
            raise RuntimeError(
                "libsad: a hand-off wait inside the persistent LSTM %s timed out — its workgroups were not "
                "co-resident (another kernel running on the device?). logits / loss / d(soft tokens) of this "
                "call are NaN. Run nothing else on the device during the discriminator step, or set "
                "SAD_NO_PERSIST=1." % ("forward recurrence" if f else "BPTT"))


def _src_kind(src):
    if src.dim() == 2:
        if src.dtype != torch.int64:
            raise RuntimeError("token ids must be int64, got %s" % src.dtype)
        return _lib.SRC_IDS
    if src.dim() != 3:
        raise RuntimeError("src must be [B,T] ids or [B,T,V] soft tokens")
    if src.dtype == torch.float32:
        return _lib.SRC_F32
    if src.dtype == torch.bfloat16:
        return _lib.SRC_BF16
    raise RuntimeError("soft tokens must be float32 or bfloat16, got %s" % src.dtype)


def _is_dense(shape, strides):
    """True if (shape, strides) is a permutation of a contiguous layout with a unit vocabulary stride."""
    if strides[-1] != 1:
        return False
    order = sorted(range(len(shape)), key=lambda i: strides[i])
    expect = 1
    for i in order:
        if shape[i] != 1 and strides[i] != expect:
            return False
        expect *= shape[i]
    return True


def _grad_struct(views):
    """name -> tensor (or missing = frozen)  ->  sad_param_grads."""
    g = _lib.SadParamGrads()

    def p(name):
        t = views.get(name)
        return t.data_ptr() if t is not None else None

    g.embedding_weight = p("embedding.weight")
    for i, sfx in enumerate(("", "_reverse")):
        g.rnn_weight_ih[i] = p("rnn.weight_ih_l0" + sfx)
        g.rnn_weight_hh[i] = p("rnn.weight_hh_l0" + sfx)
        g.rnn_bias_ih[i] = p("rnn.bias_ih_l0" + sfx)
        g.rnn_bias_hh[i] = p("rnn.bias_hh_l0" + sfx)
    g.hidden_bridge0_weight = p("hidden_bridges.0.weight")
    g.hidden_bridge0_bias = p("hidden_bridges.0.bias")
    g.linear_emb_weight = p("linear_emb.weight")
    g.linear_emb_bias = p("linear_emb.bias")
    g.linear_q_weight = p("linear_q.weight")
    g.linear_q_bias = p("linear_q.bias")
    g.linear_attn_weight = p("linear_attn.weight")
    g.linear_attn_bias = p("linear_attn.bias")
    g.output_layer_weight = p("output_layer.weight")
    g.output_layer_bias = p("output_layer.bias")
    return g


_N_FIXED = 10   # forward() arguments in front of *params

# Data-parallel hook: callable(span, tag) or None.  When set, the backward calls it with contiguous fp32
# spans of the flat gradient bucket as soon as the kernels that fill them are enqueued on the current
# stream ("rest" = every live tensor but embedding.weight, then "embedding"; "all" when embedding.weight
# is frozen) -- see dp.OverlappedGradReducer.
GRAD_READY_HOOK = None


def _upper_grads(lib, d_up, ws_up, views, n_up):
    """Parameter gradients of the RNN layers above the first (their own dims / workspaces)."""
    for i in range(n_up):
        pre = "upper%d." % i
        gu = _grad_struct({n: views[pre + n] for n in _RNN_PARAMS if pre + n in views})
        _lib.check(lib.sad_param_grads_accumulate(ctypes.byref(d_up), ctypes.byref(gu), None, _lib.SRC_IDS, 0, 0,
                                                  -1, _ptr(ws_up[i].buf), _stream_ptr()),
                   "sad_param_grads_accumulate(upper layer)")

# Debug / profiling hook (tests, bench.py, tools): when True the last call's Workspace and flat
# gradient bucket stay reachable as _Discriminator.last_workspace / .last_grad_bucket.  Off by
# default: the hook would keep one full workspace alive for the life of the process.
KEEP_LAST = False


class _Discriminator(torch.autograd.Function):
    """forward(src, lengths, mask, labels, pw, guard, pad_id, drop, onehot, layer_drop, *params).

    ``layer_drop`` is None or one (keep [T,B,2H] bool/uint8, scale) per RNN layer above the first: nn.LSTM's
    inter-layer dropout in training mode (reference classifier/models.py:108).

    ``onehot`` is None or (ids int64 [B,T], val float32 [B,T]) — the side channel of straight-through
    one-hot soft tokens (producer.SoftTokenAssembler): K1 forward gathers instead of contracting.

    ``drop`` is None or (keep [T,B,E] bool/uint8 CUDA tensor, scale): inverted dropout on the embedded
    tokens with the caller's mask (classifier copy in training mode, reference classifier/models.py:217).

    ``params`` is either empty (frozen discriminator: the transfer loop, reference
    transfer/trainers.py:216-223) or the 19 live parameters in ``_LIVE_PARAMS`` order followed by the eight
    ``_RNN_PARAMS`` of every RNN layer above the first; those that
    require grad receive gradients from sad_param_grads_accumulate (classifier training, reference
    classifier/trainers.py:367-373)."""

    @staticmethod
    def forward(ctx, src, lengths_dev, mask, labels, pw, guard, pad_id, drop, onehot, layer_drop, *params):
        if not src.is_cuda:
            raise RuntimeError("sad_amd runs on the HIP device only (got a CPU tensor); there is no CPU fallback")
        n_up = len(pw.upper)
        if params and len(params) != len(_LIVE_PARAMS) + n_up * len(_RNN_PARAMS):
            raise RuntimeError("expected %d live parameters, got %d" % (len(_LIVE_PARAMS) + n_up * len(_RNN_PARAMS),
                                                                        len(params)))
        kind = _src_kind(src)
        B, T = src.shape[0], src.shape[1]
        if kind != _lib.SRC_IDS:
            if src.shape[2] != pw.V:
                raise RuntimeError("soft tokens have vocabulary %d, weights %d" % (src.shape[2], pw.V))
            if src.stride(2) != 1:
                src = src.contiguous()
        if mask.dtype != torch.bool or mask.shape != (B, T) or mask.device != src.device:
            raise RuntimeError("mask must be a bool [B,T] tensor on the same device as src")
        dev = src.device
        d = _lib.dims(B, T, pw.V, pw.E, pw.H, pw.L, pw.cell)
        ws = Workspace(d, dev)
        # RNN layers above the first: own dims / image / workspace each (include/sad.h, sad_rnn_layer)
        d_up = _lib.dims(B, T, 1, 2 * pw.H, pw.H, pw.L, pw.cell)
        ws_up = [Workspace(d_up, dev) for _ in range(n_up)]
        ws.upper = ws_up
        upper = (_lib.SadRnnLayer * max(n_up, 1))()
        keeps_up = []
        for i in range(n_up):
            upper[i].packed, upper[i].workspace = pw.upper[i].buf.data_ptr(), ws_up[i].buf.data_ptr()
            if layer_drop is not None:
                keep, scale = layer_drop[i]
                keep = (keep.view(torch.uint8) if keep.dtype == torch.bool else keep.to(torch.uint8)).to(dev).contiguous()
                if tuple(keep.shape) != (T, B, 2 * pw.H):
                    raise RuntimeError("inter-layer dropout keep mask must be [T,B,2H] = %s, got %s"
                                       % ((T, B, 2 * pw.H), tuple(keep.shape)))
                keeps_up.append(keep)
                upper[i].drop_keep, upper[i].drop_scale = keep.data_ptr(), float(scale)
        logits = torch.empty((B, pw.L), dtype=torch.float32, device=dev)
        weights = torch.empty((B, T), dtype=torch.float32, device=dev)
        loss = torch.empty((B,), dtype=torch.float32, device=dev) if labels is not None else None
        if labels is not None:
            labels = labels.to(device=dev, dtype=torch.int64).contiguous()
        a = _lib.SadFwdArgs()
        a.src, a.src_kind = src.data_ptr(), kind
        a.src_stride_b, a.src_stride_t = src.stride(0), src.stride(1)
        a.lengths = lengths_dev.data_ptr()
        a.mask, a.mask_stride_b, a.mask_stride_t = mask.data_ptr(), mask.stride(0), mask.stride(1)
        a.guard = 1 if guard else 0
        a.labels = labels.data_ptr() if labels is not None else None
        a.logits, a.weights = logits.data_ptr(), weights.data_ptr()
        a.loss = loss.data_ptr() if loss is not None else None
        if drop is not None:
            keep, scale = drop
            keep = keep.to(device=dev).contiguous().view(torch.uint8) if keep.dtype == torch.bool else keep.to(
                device=dev, dtype=torch.uint8).contiguous()
            if tuple(keep.shape) != (T, B, pw.E):
                raise RuntimeError("dropout keep mask must be [T,B,E] = %s, got %s" % ((T, B, pw.E), tuple(keep.shape)))
            a.drop_keep, a.drop_scale = keep.data_ptr(), float(scale)
            drop = (keep, float(scale))
        if onehot is not None:
            if kind == _lib.SRC_IDS:
                raise RuntimeError("the one-hot side channel belongs to soft tokens, not to token ids")
            oh_ids, oh_val = onehot
            if (oh_ids.dtype != torch.int64 or oh_val.dtype != torch.float32 or tuple(oh_ids.shape) != (B, T)
                    or tuple(oh_val.shape) != (B, T) or oh_ids.stride() != oh_val.stride() or oh_ids.device != dev):
                raise RuntimeError("onehot = (ids int64 [B,T], val float32 [B,T]) with equal strides on the device of src")
            a.onehot_ids, a.onehot_val = oh_ids.data_ptr(), oh_val.data_ptr()
            a.onehot_stride_b, a.onehot_stride_t = oh_ids.stride(0), oh_ids.stride(1)
        if n_up:
            a.upper, a.n_upper = upper, n_up
        with _on(dev):
            _lib.check(_lib.load().sad_forward(ctypes.byref(d), _ptr(pw.buf), ctypes.byref(a), _ptr(ws.buf),
                                               _stream_ptr()), "sad_forward")
        ctx.set_materialize_grads(False)
        any_param = any(ctx.needs_input_grad[_N_FIXED:])
        if kind == _lib.SRC_IDS and not any_param:
            ctx.mark_non_differentiable(*([logits, weights] + ([loss] if loss is not None else [])))
        # Outputs and inputs go through save_for_backward: kept on ctx directly they would close a
        # cycle output -> grad_fn -> ctx -> output that Python's GC cannot see (the grad_fn is held by
        # C++ shared_ptrs of every output), and each call would leak its whole workspace.
        # The forward input is needed again for d embedding.weight (S^T . dX / scatter-add by id).
        keep_src = src if (params and ctx.needs_input_grad[_N_FIXED]) else None
        ctx.save_for_backward(logits, weights, labels, keep_src)
        ctx.sad = (d, ws, pw, kind, src.shape, src.stride(), src.dtype,
                   -1 if pad_id is None else int(pad_id), [tuple(p.shape) for p in params], drop)
        ctx.sad_upper = (d_up, ws_up, upper, keeps_up)
        if KEEP_LAST:
            _Discriminator.last_workspace = ws   # introspection hook for tests / profiling
        if loss is None:
            return logits, weights
        return logits, weights, loss

    @staticmethod
    def backward(ctx, dlogits, dweights, dloss=None):
        d, ws, pw, kind, shape, strides, dtype, pad_id, pshapes, drop = ctx.sad
        d_up, ws_up, upper, keeps_up = ctx.sad_upper
        n_up = len(ws_up)
        logits, weights, labels, src = ctx.saved_tensors
        n_out = _N_FIXED + len(pshapes)
        want_src = kind != _lib.SRC_IDS and ctx.needs_input_grad[0]
        want_p = [bool(x) for x in ctx.needs_input_grad[_N_FIXED:]]
        if not want_src and not any(want_p):
            return (None,) * n_out
        dev = logits.device
        B = shape[0]
        # (dlogits and dloss may both arrive: k_attn_bwd adds the CE term to the given dlogits)
        if dlogits is None and dloss is None:
            if dweights is None:
                return (None,) * n_out
            dlogits = torch.zeros_like(logits)
        dsrc = None
        if want_src:
            # dS keeps the memory layout of S (time-major in the transfer loop) so the generator's
            # autograd consumes it without a transposing copy.
            if _is_dense(shape, strides):
                dsrc = torch.empty_strided(shape, strides, dtype=dtype, device=dev)
            else:
                dsrc = torch.empty(shape, dtype=dtype, device=dev)
        a = _lib.SadBwdArgs()
        keep = dlogits = dlogits.to(torch.float32).contiguous() if dlogits is not None else None
        a.dlogits = dlogits.data_ptr() if dlogits is not None else None
        a.labels = labels.data_ptr() if labels is not None else None
        if dloss is not None:
            dloss = dloss.to(torch.float32).contiguous()
            a.dloss = dloss.data_ptr()
        a.dloss_scale = 0.0
        if dweights is not None:
            dweights = dweights.to(torch.float32).contiguous()
            a.dweights = dweights.data_ptr()
        a.logits, a.weights = logits.data_ptr(), weights.data_ptr()
        if dsrc is not None:
            a.dsrc, a.dsrc_kind = dsrc.data_ptr(), kind
            a.dsrc_stride_b, a.dsrc_stride_t = dsrc.stride(0), dsrc.stride(1)
        if drop is not None:
            a.drop_keep, a.drop_scale = drop[0].data_ptr(), drop[1]
        if n_up:
            a.upper, a.n_upper = upper, n_up
        pgrads = [None] * len(pshapes)
        with _on(dev):
            lib = _lib.load()
            _lib.check(lib.sad_backward(ctypes.byref(d), _ptr(pw.buf), ctypes.byref(a), _ptr(ws.buf), _stream_ptr()),
                       "sad_backward")
            if any(want_p):
                # one flat zero-filled bucket, one view per live tensor: the kernels accumulate into
                # it, and a data-parallel caller can all-reduce the bucket as a single message.
                sizes = [int(torch.Size(s).numel()) if w else 0 for s, w in zip(pshapes, want_p)]
                flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
                names = list(_LIVE_PARAMS) + ["upper%d.%s" % (i, n) for i in range(n_up) for n in _RNN_PARAMS]
                views, off = {}, 0
                for i, (name, s, w) in enumerate(zip(names, pshapes, want_p)):
                    if w:
                        pgrads[i] = flat[off:off + sizes[i]].view(s)
                        views[name] = pgrads[i]
                        off += sizes[i]
                sp = src.data_ptr() if src is not None else None
                sb, st_ = (strides[0], strides[1])
                hook = GRAD_READY_HOOK
                # With a hook (data-parallel training, dp.OverlappedGradReducer) the gradients are written in two
                # stages -- everything but embedding.weight, then embedding.weight (the longest kernel, first in
                # the bucket) -- and each span is handed to the hook as soon as its kernels are enqueued, so the
                # first all-reduce message travels beside k_wgrad_emb.
                stages = [views] if hook is None or "embedding.weight" not in views else [
                    {k: v for k, v in views.items() if k != "embedding.weight"}, {"embedding.weight": views["embedding.weight"]}]
                for si, stage in enumerate(stages):
                    g = _grad_struct(stage)
                    if n_up:
                        g.final_hidden_workspace = ws_up[-1].buf.data_ptr()
                    _lib.check(lib.sad_param_grads_accumulate(ctypes.byref(d), ctypes.byref(g), sp, kind, sb, st_, pad_id,
                                                              _ptr(ws.buf), _stream_ptr()), "sad_param_grads_accumulate")
                    if si == 0 and len(stages) == 2:
                        _upper_grads(lib, d_up, ws_up, views, n_up)
                        hook(flat[sizes[0]:], "rest")
                if len(stages) == 2:
                    hook(flat[:sizes[0]], "embedding")
                else:
                    _upper_grads(lib, d_up, ws_up, views, n_up)
                    if hook is not None:
                        hook(flat, "all")
                if KEEP_LAST:
                    _Discriminator.last_grad_bucket = flat
        del keep
        return (dsrc,) + (None,) * (_N_FIXED - 1) + tuple(pgrads)


def _live(params, upper_params=()):
    """Mapping / None -> tuple of the 19 live tensors in kernel order, then the eight rnn.* tensors of every
    layer above the first (empty = frozen)."""
    if not params:
        return ()
    out = [params[n] for n in _LIVE_PARAMS]
    for up in upper_params:
        out += [up[n] for n in _RNN_PARAMS]
    return tuple(out)


def discriminator(src, src_len, mask, pw, guard=True, params=None, pad_id=None, drop=None, onehot=None,
                  layer_drop=None, upper_params=()):
    """(logits [B,L], attention weights [B,T]) — the reference module's return value.

    ``params`` (name -> nn.Parameter, optional) makes the call differentiable w.r.t. the parameters
    that require grad; without it the discriminator is frozen (gradient flows to soft tokens only)."""
    lengths_dev = _lengths_to_device(src_len, src.shape[0], src.device)
    return _Discriminator.apply(src, lengths_dev, mask, None, pw, guard, pad_id, drop, onehot, layer_drop,
                                *_live(params, upper_params))


def discriminator_loss(src, src_len, mask, labels, pw, guard=True, params=None, pad_id=None, drop=None, onehot=None,
                       layer_drop=None, upper_params=()):
    """(per-sentence CE loss [B], logits, weights) with the cross entropy fused into the attention
    kernels — what the transfer step computes at reference transfer/models.py:1185-1212."""
    lengths_dev = _lengths_to_device(src_len, src.shape[0], src.device)
    logits, weights, loss = _Discriminator.apply(src, lengths_dev, mask, labels, pw, guard, pad_id, drop, onehot,
                                                 layer_drop, *_live(params, upper_params))
    return loss, logits, weights


# ------------------------------------------------------------------------------------------------
# Sibling soft contractions of the generator: ``soft_tokens @ embedding.weight`` in
# the decoder step (transfer/models.py:254) and in the content-alignment loss (:1221).  The same GEMM
# shape as the discriminator's K1 / K1', so the same kernels run it through the stage entry points
# of the C-ABI (sad_prep with full lengths -> every row needed, sad_softemb_fwd, sad_softemb_bwd,
# sad_param_grads_accumulate for d weight).
# ------------------------------------------------------------------------------------------------
_SE_H, _SE_L = 16, 2          # smallest discriminator shape the layouts accept: only the embedding is used
_se_cache = {}


def _se_packed(weight):
    """bf16 images of the embedding matrix (both orientations), cached per tensor version."""
    key = (weight.data_ptr(), tuple(weight.shape), str(weight.device))
    hit = _se_cache.get(key)
    if hit is not None and hit[0] == weight._version:
        return hit[1]
    V, E = weight.shape
    dev = weight.device
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    H, L = _SE_H, _SE_L
    params = {"embedding.weight": weight.detach()}
    for sfx in ("", "_reverse"):
        params["rnn.weight_ih_l0" + sfx] = z(4 * H, E)
        params["rnn.weight_hh_l0" + sfx] = z(4 * H, H)
        params["rnn.bias_ih_l0" + sfx] = z(4 * H)
        params["rnn.bias_hh_l0" + sfx] = z(4 * H)
    params.update({"hidden_bridges.0.weight": z(H, 2 * H), "hidden_bridges.0.bias": z(H),
                   "linear_emb.weight": z(H, E), "linear_emb.bias": z(H), "linear_q.weight": z(H, H),
                   "linear_q.bias": z(H), "linear_attn.weight": z(1, H), "linear_attn.bias": z(1),
                   "output_layer.weight": z(L, H), "output_layer.bias": z(L)})
    pw = PackedWeights(V, E, H, L, dev).pack(params)
    if len(_se_cache) > 8:
        _se_cache.clear()
    _se_cache[key] = (weight._version, pw)
    return pw


_se_ws = {}


def _se_workspace(B, T, V, E, dev):
    """One workspace per shape, shared by all calls (a decoder runs this once per time step and keeps
    every call alive until the backward): it only carries the need bits (every row needed, written
    once by sad_prep), X (copied out by forward) and DX (filled by backward right before use), so
    calls on one stream can share it."""
    key = (B, T, V, E, str(dev))
    hit = _se_ws.get(key)
    if hit is not None:
        return hit
    d = _lib.dims(B, T, V, E, _SE_H, _SE_L)
    ws = Workspace(d, dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    mask = torch.zeros((B, T), dtype=torch.uint8, device=dev)
    with _on(dev):
        _lib.check(_lib.load().sad_prep(ctypes.byref(d), _ptr(lens), _ptr(mask), mask.stride(0), mask.stride(1), 0,
                                        _ptr(ws.buf), _stream_ptr()), "sad_prep")
    if len(_se_ws) > 4:
        _se_ws.clear()
    _se_ws[key] = (d, ws)
    return d, ws


class _SoftEmbedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, weight):
        if not S.is_cuda or not weight.is_cuda:
            raise RuntimeError("soft_embedding needs tensors on the HIP device (there is no CPU path)")
        if weight.dtype != torch.float32 or weight.dim() != 2:
            raise RuntimeError("weight must be a float32 [V, E] matrix")
        if S.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError("soft tokens must be float32 or bfloat16, got %s" % S.dtype)
        if S.shape[-1] != weight.shape[0]:
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (%s and %s)" % (tuple(S.shape), tuple(weight.shape)))
        lead = tuple(S.shape[:-1])
        S3 = S if S.dim() == 3 else S.reshape(-1, 1, S.shape[-1])
        if S3.stride(2) != 1:
            S3 = S3.contiguous()
        # rows are numbered m = t*B + b: the dimension with the larger stride plays "t", so the result
        # is laid out like the input
        swap = S3.stride(0) < S3.stride(1)
        Sv = S3.transpose(0, 1) if swap else S3               # [t][b][V]
        T, B, V = Sv.shape
        E = weight.shape[1]
        dev = S.device
        pw = _se_packed(weight)
        d, ws = _se_workspace(B, T, V, E, dev)
        kind = _lib.SRC_F32 if S.dtype == torch.float32 else _lib.SRC_BF16
        with _on(dev):
            lib = _lib.load()
            _lib.check(lib.sad_softemb_fwd(ctypes.byref(d), _ptr(pw.buf), _ptr(Sv), kind, Sv.stride(1), Sv.stride(0),
                                           _ptr(ws.buf), _stream_ptr()), "sad_softemb_fwd")
        Ep = _lib.padded_dims(d)["Ep"]
        # always a COPY out of the shared per-shape workspace (a bf16 `.to()` would be a view that the next
        # call of the same shape overwrites)
        X = ws.view("X", torch.bfloat16)[:T * B * Ep].view(T, B, Ep)[:, :, :E].to(S.dtype, copy=True)
        ctx.save_for_backward(Sv, weight)
        ctx.meta = (d, ws, pw, kind, swap, lead, Ep)
        out = X.transpose(0, 1) if swap else X
        return out.reshape(lead + (E,))

    @staticmethod
    def backward(ctx, dout):
        Sv, weight = ctx.saved_tensors
        d, ws, pw, kind, swap, lead, Ep = ctx.meta
        T, B, V = Sv.shape
        E = weight.shape[1]
        dev = Sv.device
        g3 = dout.reshape((B, T, E) if swap else (T, B, E))
        if swap:
            g3 = g3.transpose(0, 1)
        dx = ws.view("DX", torch.bfloat16)[:T * B * Ep].view(T, B, Ep)
        dx[:, :, :E] = g3
        if Ep > E:
            dx[:, :, E:] = 0
        dS = dW = None
        with _on(dev):
            lib = _lib.load()
            if ctx.needs_input_grad[0]:
                dSv = torch.empty_strided(Sv.shape, Sv.stride(), dtype=Sv.dtype, device=dev) if _is_dense(
                    tuple(Sv.shape), tuple(Sv.stride())) else torch.empty(Sv.shape, dtype=Sv.dtype, device=dev)
                _lib.check(lib.sad_softemb_bwd(ctypes.byref(d), _ptr(pw.buf), _ptr(dSv), kind, dSv.stride(1), dSv.stride(0),
                                               _ptr(ws.buf), _stream_ptr()), "sad_softemb_bwd")
                dS = (dSv.transpose(0, 1) if swap else dSv).reshape(lead + (V,))
            if ctx.needs_input_grad[1]:
                dW = torch.zeros((V, E), dtype=torch.float32, device=dev)
                g = _lib.SadParamGrads()
                g.embedding_weight = dW.data_ptr()
                _lib.check(lib.sad_param_grads_accumulate(ctypes.byref(d), ctypes.byref(g), _ptr(Sv), kind, Sv.stride(1),
                                                          Sv.stride(0), -1, _ptr(ws.buf), _stream_ptr()),
                           "sad_param_grads_accumulate")
        return dS, dW


def soft_embedding(soft_tokens, weight):
    """``soft_tokens @ weight`` for soft tokens ``[..., V]`` (float32 or bfloat16, unit stride along V)
    and an embedding matrix ``[V, E]`` (float32), differentiable in both — the generator's sibling
    contractions of the discriminator's soft branch (reference transfer/models.py:254 decoder step,
    :1221 content alignment).  bf16 MFMA inputs, fp32 accumulation; the result is rounded to bf16
    once (the kernels' activation format) and returned in the dtype of ``soft_tokens``."""
    return _SoftEmbedding.apply(soft_tokens, weight)
