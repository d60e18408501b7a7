"""Producer side of the hot path and the caller stretch around it (§8 a12).

``SoftTokenAssembler`` replaces what ``AttnRNNDecoder.soft_decode`` (reference transfer/models.py:471-574)
does AROUND the decoder step: per step ``F.gumbel_softmax(logits, tau=1.0, hard=True)`` (:531), the masked
copy appended to ``output_list`` (:539), ``outputs_len`` / ``mask`` bookkeeping (:542-543), then the ``cat``
with the [sos] row, the optional [eos] row and the [pad] fill (:546-571).  Here one kernel per step writes the
sample straight into its row of the time-major soft-token tensor ``S[T,B,V]`` in the dtype the
discriminator consumes (bf16 by default: half the bytes of the reference's fp32 tensor, no ``cat`` /
``clone`` copies) together with an exact (index, value) side channel, and one kernel at the end writes the
[sos]/[eos]/[pad] rows, the final lengths and the pad+sos+eos mask of transfer/models.py:1180-1183.
Gradients flow back to every step's logits through the straight-through estimator (dense softmax
Jacobian), from ``S`` and from the un-masked sample ``rnn_in`` that feeds the next decoder step.

``score_soft_tokens`` restates the caller stretch transfer/models.py:1172-1212 around the drop-in
discriminator (sort by length, call, NaN -> 0 / 1 - alpha, un-sort, CE vs ``1 - lab``).  The decoder
itself (the generator) stays the reference's stock PyTorch code: out of scope.
"""
import ctypes

import torch

from . import lib as _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _raw_stream_fallback(index):
    return torch.cuda.current_stream(index).cuda_stream


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", _raw_stream_fallback)


def _stream():
    # the raw stream of the current device straight from the C layer: torch.cuda.current_stream() resolves its default
    # device through torch.cuda.is_available(), which reads the environment on every call (measured on the GPU box:
    # the longest single item of the module path's host time at batch 64)
    return ctypes.c_void_p(_RAW_STREAM(torch.cuda.current_device()))


def _kind(dtype):
    if dtype == torch.float32:
        return _lib.SRC_F32
    if dtype == torch.bfloat16:
        return _lib.SRC_BF16
    raise RuntimeError("soft tokens are float32 or bfloat16, got %s" % dtype)


def gumbel_noise_like(logits):
    """The draw inside ``F.gumbel_softmax`` (torch/nn/functional.py): same generator consumption."""
    return -torch.empty_like(logits, memory_format=torch.legacy_contiguous_format).exponential_().log()


def _step_backward(meta, y_soft, alive, d_s_row, g_rnn):
    """dlogits of one decode step (sad_gumbel_st_step_bwd).  `meta` = (B, V, kind, dtype, tau, device): plain
    values, so that no autograd context has to keep the assembler alive."""
    B, V, kind, dtype, tau, device = meta
    dl = torch.empty((B, V), dtype=torch.float32, device=device)
    if d_s_row is not None and (d_s_row.stride(1) != 1 or d_s_row.dtype != dtype):
        d_s_row = d_s_row.to(dtype).contiguous()
    with torch.cuda.device(device):
        _lib.check(_lib.load().sad_gumbel_st_step_bwd(
            B, V, _ptr(d_s_row), kind, d_s_row.stride(0) if d_s_row is not None else 0, _ptr(g_rnn),
            _ptr(alive), _ptr(y_soft), tau, _ptr(dl), dl.stride(0), _stream()), "sad_gumbel_st_step_bwd")
    return dl


# Neither autograd function keeps the assembler on its context: an output that the assembler also stored, or a
# context that points back at the assembler whose `_logits` hold the decoder graph, closes a reference cycle
# through autograd's C++ nodes (output -> grad_fn -> ctx -> assembler -> ...), which Python's collector only
# breaks late, if at all -- every decode call would pin S[T_max,B,V] and two fp32 [B,V] buffers per step.
# The contexts hold what the backward needs (softmax, alive bytes) as saved tensors and plain values.
class _GumbelStep(torch.autograd.Function):
    """One decode step: kernel launch + the dense un-masked sample as the autograd output."""

    @staticmethod
    def forward(ctx, logits, noise, asm):
        y, alive, r = asm._launch_step(logits, noise, True)
        ctx.save_for_backward(y, alive)
        ctx.meta = asm._meta()
        return r

    @staticmethod
    def backward(ctx, g_rnn):
        if g_rnn is None:
            return None, None, None
        y, alive = ctx.saved_tensors
        return _step_backward(ctx.meta, y, alive, None, g_rnn.contiguous().float()), None, None


class _Assemble(torch.autograd.Function):
    """S as a function of every step's logits: backward hands row t+1 of dS to step t."""

    @staticmethod
    def forward(ctx, asm, T, *logits):
        ctx.n, ctx.meta = len(logits), asm._meta()
        ctx.save_for_backward(*(asm._y_soft + asm._alive_hist))
        return asm.S[:T]

    @staticmethod
    def backward(ctx, dS):
        saved = ctx.saved_tensors
        grads = []
        for t in range(ctx.n):
            grads.append(_step_backward(ctx.meta, saved[t], saved[ctx.n + t], dS[t + 1], None)
                         if ctx.needs_input_grad[2 + t] else None)
        return (None, None) + tuple(grads)


class SoftTokens:
    """What the assembler hands to the caller stretch: ``S`` [T,B,V] time-major (autograd-connected to the
    step logits), ``lens`` int32 [B] on the device, ``mask`` bool [B,T] (pad/sos/eos, before the
    discriminator's in-place guard), side channel ``ids`` int64 [T,B] / ``val`` float32 [T,B]."""

    def __init__(self, S, lens, mask, ids, val):
        self.S, self.lens, self.mask, self.ids, self.val = S, lens, mask, ids, val


class SoftTokenAssembler:
    def __init__(self, B, V, max_decoding_len, pad_id, sos_id, eos_id, device, dtype=torch.bfloat16, tau=1.0):
        self.B, self.V, self.max_decoding_len = int(B), int(V), int(max_decoding_len)
        self.pad_id, self.sos_id, self.eos_id, self.tau = int(pad_id), int(sos_id), int(eos_id), float(tau)
        self.device, self.dtype = torch.device(device), dtype
        if self.device.type != "cuda":
            raise RuntimeError("SoftTokenAssembler runs on the HIP device only (there is no CPU path)")
        self.kind = _kind(dtype)
        self.T_max = self.max_decoding_len + 3              # [sos] + (max_decoding_len + 1) samples + optional [eos]
        self.S = torch.empty((self.T_max, B, V), dtype=dtype, device=self.device)
        self.ids = torch.zeros((self.T_max, B), dtype=torch.int64, device=self.device)
        self.val = torch.zeros((self.T_max, B), dtype=torch.float32, device=self.device)
        self.lens_steps = torch.zeros((B,), dtype=torch.int32, device=self.device)
        self.alive = torch.ones((B,), dtype=torch.uint8, device=self.device)
        self.steps = 0
        self._alive_hist, self._y_soft, self._logits = [], [], []

    def _meta(self):
        return (self.B, self.V, self.kind, self.dtype, self.tau, self.device)

    # ---- per step ------------------------------------------------------------------------------
    def step(self, logits, noise=None, want_rnn_in=True):
        """logits [B,V] float32 of one decode step -> the un-masked straight-through sample ``rnn_in``
        [B,V] float32 (next decoder input; None when ``want_rnn_in`` is False).  The masked sample lands in
        row ``steps + 1`` of ``S``; ``last_ids`` / ``last_val`` are its side channel."""
        if self.steps > self.max_decoding_len:
            raise RuntimeError("more decode steps than max_decoding_len + 1")
        if logits.dim() != 2 or tuple(logits.shape) != (self.B, self.V) or logits.dtype != torch.float32 or not logits.is_cuda:
            raise RuntimeError("logits must be a float32 [B,V] tensor on the HIP device")
        if noise is None:
            noise = gumbel_noise_like(logits)
        out = None
        if want_rnn_in:
            out = _GumbelStep.apply(logits, noise, self)
        else:
            self._launch_step(logits, noise, False)      # the gradient reaches the logits through S only (finalize)
        self._logits.append(logits)
        return out

    def _launch_step(self, logits, noise, want_rnn_in):
        t = self.steps
        lg = logits.detach()
        lg = lg if lg.stride(1) == 1 else lg.contiguous()
        nz = noise.detach().float()
        nz = nz if nz.stride(1) == 1 else nz.contiguous()
        alive_before = self.alive.clone()        # the kernel updates `alive` in place; the backward needs the old bytes
        self._alive_hist.append(alive_before)
        y = torch.empty((self.B, self.V), dtype=torch.float32, device=self.device)
        r = torch.empty((self.B, self.V), dtype=torch.float32, device=self.device) if want_rnn_in else None
        a = _lib.SadGumbelArgs()
        a.B, a.V = self.B, self.V
        a.logits, a.logits_stride, a.noise, a.noise_stride = lg.data_ptr(), lg.stride(0), nz.data_ptr(), nz.stride(0)
        a.tau, a.alive, a.eos_id = self.tau, alive_before.data_ptr(), self.eos_id
        row = self.S[t + 1]
        a.s_row, a.s_kind, a.s_stride_b = row.data_ptr(), self.kind, row.stride(0)
        a.rnn_in = r.data_ptr() if r is not None else None
        a.y_soft, a.idx, a.val = y.data_ptr(), self.ids[t + 1].data_ptr(), self.val[t + 1].data_ptr()
        a.lens, a.alive_out = self.lens_steps.data_ptr(), self.alive.data_ptr()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().sad_gumbel_st_step(ctypes.byref(a), _stream()), "sad_gumbel_st_step")
        self._y_soft.append(y)
        self.steps += 1
        return y, alive_before, r

    @property
    def last_ids(self):
        return self.ids[self.steps]

    @property
    def last_val(self):
        return self.val[self.steps]

    def any_alive(self):
        """``torch.sum(mask) != 0`` of the reference's while loop (transfer/models.py:519): synchronises,
        exactly like the reference does every step."""
        return bool(self.alive.any().item())

    def keep_decoding(self):
        return self.any_alive() and self.steps <= self.max_decoding_len

    # ---- after the last step ---------------------------------------------------------------------
    def finalize(self):
        over_max = self.steps > self.max_decoding_len                      # transfer/models.py:556
        add_eos = bool(over_max and self.any_alive())                      # :557-566 (alive => last token != eos)
        T = self.steps + 1 + (1 if add_eos else 0)
        lens = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        mask = torch.empty((self.B, T), dtype=torch.uint8, device=self.device)
        ids, val = self.ids[:T], self.val[:T]
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().sad_soft_finalize(
                self.B, T, self.V, self.steps, 1 if over_max else 0, self.pad_id, self.sos_id, self.eos_id,
                _ptr(self.S), self.kind, self.S.stride(0), self.S.stride(1), _ptr(self.ids), _ptr(self.val),
                _ptr(self.lens_steps), _ptr(self.alive), _ptr(lens), _ptr(mask), _stream()), "sad_soft_finalize")
        S = _Assemble.apply(self, T, *self._logits)
        # the graph now owns what its backward needs; drop the assembler's references to the decoder graph
        self._logits, self._y_soft, self._alive_hist = [], [], []
        return SoftTokens(S, lens, mask.view(torch.bool), ids, val)


def score_soft_tokens(classifier, soft, inv_lab, use_side_channel=True):
    """The caller stretch of the transfer step around the discriminator, reference
    transfer/models.py:1172-1212: sort by length (descending), call the discriminator on the sorted soft
    tokens with the pad+sos+eos mask, content weights from the detached attention weights (NaN -> 0,
    ``1 - alpha``, masked, normalised, ``len == 2`` rows), un-sort, per-sentence CE against ``inv_lab``.

    ``soft``: SoftTokens (or any object with S [T,B,V], lens, and optionally mask / ids / val).
    Returns dict(soft_out_cls_loss [B], soft_out_cls_logits [B,L], soft_out_content_alpha [B,T],
    sorted_mask (after the discriminator's in-place guard), soft_out_indices)."""
    S, lens = soft.S, soft.lens
    T, B = S.shape[0], S.shape[1]
    dev = S.device
    sorted_len, idx = torch.sort(lens.to(dev), dim=0, descending=True)             # :1172-1173
    _, resort = torch.sort(idx, dim=0)                                             # :1174
    sorted_S = torch.index_select(S, 1, idx)                                       # :1175
    sorted_lab = torch.index_select(inv_lab.to(dev), 0, idx)                       # :1177
    ar = torch.arange(T, device=dev)
    mask = ar.unsqueeze(0).expand(B, -1) >= (sorted_len.unsqueeze(1).expand(-1, T) - 1)   # :1180-1182
    mask[:, 0] = True                                                              # :1183
    onehot = None
    if use_side_channel and getattr(soft, "ids", None) is not None:
        onehot = (torch.index_select(soft.ids, 1, idx).transpose(0, 1), torch.index_select(soft.val, 1, idx).transpose(0, 1))
    kw = dict(onehot=onehot) if onehot is not None else {}
    logits, alpha = classifier(sorted_S.transpose(0, 1), sorted_len, mask, soft=True, **kw)   # :1185-1187
    with torch.no_grad():                                                          # :1193-1202
        style = alpha.clone().detach()
        style[torch.isnan(style)] = 0.0
        content = 1.0 - style
        content[mask] = 0.0
        content /= (sorted_len - 3 + 1e-4).unsqueeze(-1)
        content[sorted_len == 2, 1] = 1.0
    out_logits = torch.index_select(logits, 0, resort)                             # :1207
    out_content = torch.index_select(content, 0, resort)                           # :1208
    loss = torch.nn.functional.cross_entropy(out_logits, inv_lab.to(dev), reduction="none")   # :1212
    return dict(soft_out_cls_loss=loss, soft_out_cls_logits=out_logits, soft_out_content_alpha=out_content,
                sorted_mask=mask, soft_out_indices=idx, sorted_soft_out_len=sorted_len)
