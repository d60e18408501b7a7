"""Batch data-parallel helpers: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The discriminator step shards over the batch with NO data-path collective in the transfer loop's
frozen mode: every sentence is independent in the forward and in d(soft tokens),
so each rank scores its own contiguous slice of the length-sorted batch and keeps its slice of dS.
The only exchanges are (a) optionally one scalar (the mean loss, for logging) and (b), when the
discriminator parameters are being trained, one sum all-reduce of the live parameter gradients
flattened into a single bucket (the reference has no distributed code; this is new in this repo).
"""
import torch
import torch.distributed as dist


def shard_bounds(B, rank, world):
    """Contiguous [lo, hi) slice of a batch of B sentences for `rank` (remainder to low ranks)."""
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(rank, world, *tensors, dim=0):
    """Slice every tensor along `dim` to this rank's shard.  A length-sorted batch stays sorted
    inside each shard, which is all ``pack_padded_sequence`` semantics need."""
    out = []
    for t in tensors:
        lo, hi = shard_bounds(t.shape[dim], rank, world)
        out.append(t.narrow(dim, lo, hi - lo))
    return out


def global_mean_loss(local_loss_sum, local_count):
    """Mean per-sentence loss over all ranks (one 2-element all-reduce, logging only)."""
    v = torch.stack([local_loss_sum.detach().float().reshape(()),
                     torch.as_tensor(float(local_count), device=local_loss_sum.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return v[0] / v[1]


def _as_one_bucket(grads, n):
    """If the gradients are contiguous fp32 views laid out back to back in one storage (what the
    discriminator's backward produces: one zero-filled bucket, one view per live tensor), return that
    span as a flat tensor sharing their memory, else None."""
    g0 = grads[0]
    if g0.dtype != torch.float32:
        return None
    base = g0.untyped_storage().data_ptr()
    off = g0.storage_offset()
    for g in grads:
        if (g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != base
                or g.storage_offset() != off):
            return None
        off += g.numel()
    return torch.as_strided(g0, (n,), (1,), g0.storage_offset())


def allreduce_grads(params, bucket=None):
    """Sum-all-reduce the gradients of `params` (those with a grad) as ONE flat fp32 bucket:
    xGMI is point-to-point, so one large collective beats many small ones.  With the local loss
    scaled by 1/global_B the summed gradients equal the single-process ``.mean()`` gradients.
    Returns the flat bucket (reusable across steps)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return bucket
    n = sum(g.numel() for g in grads)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    flat = _as_one_bucket(grads, n)
    if flat is not None:            # the backward already wrote the gradients back to back: no copies
        if multi:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat
    if bucket is None or bucket.numel() != n or bucket.device != grads[0].device:
        bucket = torch.empty(n, dtype=torch.float32, device=grads[0].device)
    o = 0
    for g in grads:
        bucket[o:o + g.numel()].copy_(g.reshape(-1))
        o += g.numel()
    if multi:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    o = 0
    for g in grads:
        g.copy_(bucket[o:o + g.numel()].view_as(g))
        o += g.numel()
    return bucket


class OverlappedGradReducer:
    """Sum-all-reduce of the discriminator's gradient bucket in TWO messages, the first one issued while
    the backward is still running ("launch on a side stream as soon as BPTT finishes the
    LSTM wgrads, embedding wgrad (11.2 MB) last").

    ``functional._Discriminator.backward`` writes the parameter gradients in two stages into one flat
    bucket ``[embedding.weight | everything else]``: first every tensor but ``embedding.weight``
    (`k_wgrad_cat`, `k_wgrad_hh`, the dense layers and the small reductions), then
    ``embedding.weight`` (`k_wgrad_emb`, the longest weight-gradient kernel).  With this object
    installed as ``functional.GRAD_READY_HOOK`` the first span is handed to the collective as soon as
    its kernels are enqueued: ``dist.all_reduce(..., async_op=True)`` runs on the process group's own
    stream (RCCL) behind an event of the compute stream, i.e. beside `k_wgrad_emb`, which is an
    ordinary (non-persistent) launch.  The persistent recurrences need every CU (co-residency), so
    nothing may still be in flight when the next forward starts: call :meth:`finish` before the
    optimizer step / the next ``sad_pack_weights`` — it makes the compute stream wait for both
    messages.

    Stream / event order per step (what ``bench.py --dry-run`` and tests/test_bench_dryrun.py replay on
    gloo):  backward kernels -> hook("rest") -> k_wgrad_emb -> hook("embedding") -> finish() ->
    optimizer -> sad_pack_weights -> next forward.
    """

    def __init__(self, group=None):
        self.group = group
        self.pending = []
        self.log = []          # (tag, numel) per issued message, for tests

    def _multi(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def __call__(self, span, tag):
        """The hook: `span` is a contiguous fp32 view of the bucket whose kernels have been enqueued."""
        self.log.append((tag, int(span.numel())))
        if span.numel() and self._multi():
            self.pending.append(dist.all_reduce(span, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the current stream wait for every message issued since the last call."""
        for w in self.pending:
            w.wait()
        n = len(self.pending)
        self.pending = []
        return n
