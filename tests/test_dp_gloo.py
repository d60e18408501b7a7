"""Data-parallel host logic on CPU with the gloo backend, world_size 2.

The kernels cannot run here; what is covered is the sharding arithmetic and the collectives the
N>1 path uses: shard bounds tile the batch exactly and keep each shard length-sorted, the flat
gradient bucket all-reduce reproduces the single-process mean gradient, and the logging all-reduce
gives the global mean.  The "N ranks == 1 rank on the concatenated batch" invariant of the kernels
themselves is the bit-exact shard test in tests/test_gpu_parity.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sad_amd import dp, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_tile_the_batch():
    for B in (1, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 8):
            cuts = [dp.shard_bounds(B, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
    lengths = torch.from_numpy(synth.make_lengths(64, 23, seed=1))
    S = torch.zeros(64, 23, 5)
    for r in range(2):
        ln, s = dp.shard_batch(r, 2, lengths, S)
        assert s.shape[0] == ln.shape[0] == 32
        assert bool((ln[1:] <= ln[:-1]).all())       # still sorted descending inside the shard


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        B, D = 10, 6
        x = torch.randn(B, D)
        w = torch.nn.Parameter(torch.ones(D, 3))
        v = torch.nn.Parameter(torch.full((3,), 0.5))
        dead = torch.nn.Parameter(torch.zeros(2))      # grad stays None, like output_bridge.*
        lo, hi = dp.shard_bounds(B, rank, world)
        loss_b = ((x[lo:hi] @ w) * v).sum(dim=1) ** 2          # per-sentence loss
        (loss_b.sum() / B).backward()                          # local loss scaled by 1/global_B
        dp.allreduce_grads([w, v, dead])
        # second form: gradients that already sit back to back in one bucket (what the discriminator's
        # backward returns) are reduced in place, without the gather/scatter copies
        flat = torch.cat([w.grad.reshape(-1), v.grad.reshape(-1)]).clone() / world   # undo the sum: per-rank share
        w2, v2 = torch.nn.Parameter(w.detach().clone()), torch.nn.Parameter(v.detach().clone())
        w2.grad, v2.grad = flat[:w.numel()].view_as(w), flat[w.numel():].view_as(v)
        out = dp.allreduce_grads([w2, v2])
        assert out.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
        assert torch.allclose(w2.grad, w.grad) and torch.allclose(v2.grad, v.grad)
        # third form: the two-message reducer the backward's hook drives (dp.OverlappedGradReducer): spans of one
        # bucket, issued asynchronously in the hook order, complete after finish()
        flat3 = torch.cat([w.grad.reshape(-1), v.grad.reshape(-1)]).clone() / world
        red = dp.OverlappedGradReducer()
        red(flat3[w.numel():], "rest")
        red(flat3[:w.numel()], "embedding")
        assert red.finish() == 2 and red.finish() == 0
        assert red.log == [("rest", v.numel()), ("embedding", w.numel())]
        assert torch.allclose(flat3[:w.numel()].view_as(w), w.grad) and torch.allclose(flat3[w.numel():], v.grad)
        mean = dp.global_mean_loss(loss_b.sum(), hi - lo)
        q.put((rank, w.grad.numpy().copy(), v.grad.numpy().copy(), dead.grad is None, float(mean)))
    finally:
        dist.destroy_process_group()


def test_grad_allreduce_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    x = torch.randn(10, 6)
    w = torch.nn.Parameter(torch.ones(6, 3))
    v = torch.nn.Parameter(torch.full((3,), 0.5))
    loss_b = ((x @ w) * v).sum(dim=1) ** 2
    loss_b.mean().backward()
    for rank, gw, gv, dead_none, mean in res:
        assert np.allclose(gw, w.grad.numpy(), rtol=1e-5, atol=1e-6)
        assert np.allclose(gv, v.grad.numpy(), rtol=1e-5, atol=1e-6)
        assert dead_none
        assert abs(mean - float(loss_b.mean())) < 1e-4
