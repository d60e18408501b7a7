#!/usr/bin/env python3
"""Benchmark of the discriminator hot path: fwd+bwd sentences/s on synthetic Yelp-shaped batches.

    python bench.py [--gpus N --steps K --warmup W]            # N=1 directly
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # N>1 (driver)

One "step" = one pass of the hot path over one batch already resident in HBM: forward on soft
tokens (K1 soft-embedding GEMM, K2 BiLSTM, K3 keys, K4 bridge, K5 attention/head/CE vs labels) +
backward to d(soft tokens) with frozen parameters — what reference transfer/models.py:1185-1212 +
transfer/trainers.py:376 run per training step.  Weak scaling: every rank owns a full per-GPU batch
(independent sentences; no data-path collective in frozen mode, DESIGN.md "Multi-GPU").

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel,
timed live with events on the launch stream) and `cpu_baseline` (stock-torch port of the reference
op sequence from oracle/torch_ref.py on the host cores; N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[2]/[3]: "Synthetic Yelp-shaped batches (vocab ~10k, len 32, batch 1024), bf16"
    "cfg3": dict(B=1024, T=32, V=9353, E=300, H=512, L=2, s_dtype="bf16"),
    # configs[1]: transfer/train.sh shape, fp32 soft tokens as the generator emits them
    "yelp64": dict(B=64, T=23, V=9353, E=300, H=512, L=2, s_dtype="f32"),
    # configs[4] per-GPU slice of the stress shape
    "stress": dict(B=512, T=128, V=50000, E=300, H=512, L=2, s_dtype="bf16"),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_PEAK_TFS = 2500.0      # dense bf16 MFMA (spec)


def make_inputs(w, seed, dev):
    """Synthetic batch of SURVEY.md §8 d2, generated on the device (seeded): dense softmax soft
    tokens, one-hot [sos] at t=0 and one-hot [pad] for t >= len; sorted-descending lengths."""
    from sad_amd import synth
    B, T, V = w["B"], w["T"], w["V"]
    lengths = synth.make_lengths(B, T, seed=seed)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    dt = torch.bfloat16 if w["s_dtype"] == "bf16" else torch.float32
    S = torch.empty((T, B, V), dtype=dt, device=dev)          # time-major storage, as soft_decode emits
    for t in range(T):
        z = torch.randn((B, V), generator=g, device=dev, dtype=torch.float32) * 3.0
        S[t] = torch.softmax(z, dim=1).to(dt)
    ln = torch.from_numpy(lengths).to(dev)
    pad = torch.arange(T, device=dev)[:, None] >= ln[None, :]  # [T,B]
    S[pad] = 0
    S[:, :, 0][pad] = 1
    S[0] = 0
    S[0, :, 1] = 1
    mask = torch.from_numpy(synth.transfer_mask(lengths, T)).to(dev)
    labels = torch.from_numpy(synth.make_labels(B, seed=seed)).to(dev)
    return S, lengths, ln.to(torch.int32), mask, labels


def time_kernel(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()     # events on torch's current stream == the stream libsad launches on
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def cpu_baseline(w, params, S, lengths, mask, labels, threads):
    """Stock-torch port of the reference op sequence on the host cores, fp32, bounded sample."""
    from oracle.torch_ref import TorchRefDiscriminator, fwd_bwd_step
    torch.set_num_threads(threads)
    sel = torch.arange(0, w["B"], max(1, w["B"] // 128))[:128]            # every k-th sentence: same length mix
    m = TorchRefDiscriminator(w["V"], w["E"], w["H"], w["L"]).load_numpy(params)
    for p in m.parameters():
        p.requires_grad = False
    Sc = S[:, sel.to(S.device)].float().cpu().transpose(0, 1)              # [b,T,V] view of time-major storage
    lc = torch.from_numpy(lengths)[sel]
    mc, yc = mask[sel.to(mask.device)].cpu(), labels[sel.to(labels.device)].cpu()
    fwd_bwd_step(m, Sc, lc, mc, yc)
    n, t0 = 0, time.perf_counter()
    while n < 3 or time.perf_counter() - t0 < 10.0:
        fwd_bwd_step(m, Sc, lc, mc, yc)
        n += 1
        if n >= 20:
            break
    dt = time.perf_counter() - t0
    return dict(value=round(len(sel) * n / dt, 2), unit="sentences/s", cores=threads, kind="port",
                sample="%d of the %d sentences (every %d-th, same length mix) x %d steps, fp32, "
                       "oracle/torch_ref.py (stock nn.LSTM/matmul op sequence of the reference)"
                       % (len(sel), w["B"], max(1, w["B"] // 128), n))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--s-dtype", default=None, choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" == RCCL on ROCm
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from sad_amd import functional as Fn, lib, synth
    w = dict(WORKLOADS[a.workload])
    if a.s_dtype:
        w["s_dtype"] = a.s_dtype
    params = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21)
    pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
    S, lengths, ln_dev, mask, labels = make_inputs(w, 1234 + rank, dev)
    Sv = S.transpose(0, 1).requires_grad_(True)           # [B,T,V] view, exactly like the transfer loop
    B, T, V = w["B"], w["T"], w["V"]

    def step():
        Sv.grad = None
        loss, _, _ = Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw)
        loss.mean().backward()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # --- unfrozen mode: + parameter gradients, + one all-reduce of the flat grad
    # bucket over RCCL when N > 1, + re-packing the weights (an optimizer step invalidates the image).
    live = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in params.items() if k in Fn._LIVE_PARAMS}

    def step_unfrozen():
        Sv.grad = None
        for p_ in live.values():
            p_.grad = None
        pw.pack(live)
        loss, _, _ = Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw, params=live, pad_id=0)
        (loss.sum() / (B * world)).backward()
        if world > 1:
            dist.all_reduce(Fn._Discriminator.last_grad_bucket)
        return loss

    u_steps = max(1, min(a.steps, 10))
    for _ in range(2):
        step_unfrozen()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(u_steps):
        step_unfrozen()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el_u = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el_u], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el_u = float(tt.item())
    # leave a frozen step's workspace behind for the per-kernel timing below
    step()
    torch.cuda.synchronize()

    out = None
    if rank == 0:
        # --- roofline of the dominant kernels, timed in isolation on the launch stream ----------
        d = lib.dims(B, T, V, w["E"], w["H"], w["L"])
        ws = Fn._Discriminator.last_workspace                 # need bits + DX of the last step
        kind = lib.SRC_BF16 if w["s_dtype"] == "bf16" else lib.SRC_F32
        esz = 2 if w["s_dtype"] == "bf16" else 4
        L = lib.load()
        st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        dS = torch.empty_like(S)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        f_fwd = lambda: lib.check(L.sad_softemb_fwd(ctypes.byref(d), p(pw.buf), p(S), kind, V, B * V, p(ws.buf), st()), "softemb_fwd")
        f_bwd = lambda: lib.check(L.sad_softemb_bwd(ctypes.byref(d), p(pw.buf), p(dS), kind, V, B * V, p(ws.buf), st()), "softemb_bwd")
        t_fwd, t_bwd = time_kernel(f_fwd), time_kernel(f_bwd)
        need_rows = int((np.arange(T)[:, None] < lengths[None, :]).sum())
        M = B * T
        Ep = lib.padded_dims(d)["Ep"]
        bytes_fwd = need_rows * V * esz + M * Ep * 2           # read needed S rows once + write X
        bytes_bwd = M * V * esz + need_rows * Ep * 2           # write every dS row + read dX
        flops_each = 2.0 * need_rows * V * w["E"]
        k_fwd = dict(kernel="k_softemb_fwd", seconds=t_fwd, GBps=bytes_fwd / t_fwd / 1e9, TFLOPs=flops_each / t_fwd / 1e12)
        k_bwd = dict(kernel="k_softemb_bwd", seconds=t_bwd, GBps=bytes_bwd / t_bwd / 1e9, TFLOPs=flops_each / t_bwd / 1e12)
        dom, dom_bytes = (k_bwd, bytes_bwd) if t_bwd >= t_fwd else (k_fwd, bytes_fwd)
        roofline = dict(bound="hbm", kernel=dom["kernel"], achieved=round(dom["GBps"], 1), peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=round(dom["GBps"] / HBM_PEAK_GBS, 4), traffic=None,
                        algorithmic_bytes=dom_bytes, avg_launch_us=round(dom["seconds"] * 1e6, 1),
                        mfma_TFLOPs=round(dom["TFLOPs"], 1), mfma_frac=round(dom["TFLOPs"] / MFMA_PEAK_TFS, 4),
                        other=dict(kernel=(k_fwd if dom is k_bwd else k_bwd)["kernel"],
                                   avg_launch_us=round((k_fwd if dom is k_bwd else k_bwd)["seconds"] * 1e6, 1),
                                   GBps=round((k_fwd if dom is k_bwd else k_bwd)["GBps"], 1),
                                   mfma_TFLOPs=round((k_fwd if dom is k_bwd else k_bwd)["TFLOPs"], 1)))
        # HBM traffic of the dominant kernel from the committed PMC passes of THIS command
        # (tools/pmc_bench.sh -> profiles/pmc_bench_traffic.json; FETCH_SIZE doubled per the gfx950 note)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_bench_traffic.json")))
            key = [k for k in pmc if k.startswith("k_softemb_bwd" if dom is k_bwd else "k_softemb_fwd")]
            if key and a.workload == "cfg3" and w["s_dtype"] == "bf16":
                roofline["traffic"] = pmc[key[0]]["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            cpu = cpu_baseline(w, params, S, lengths, mask, labels, threads)
        out = {
            "metric": "discriminator fwd+bwd sentences/sec (Yelp shape)",
            "value": round(world * B * a.steps / el, 1), "unit": "sentences/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s: synthetic Yelp-shaped soft tokens, B=%d per GPU, T=%d, V=%d, E=%d, H=%d, "
                                   "len~U{3..T} sorted, S %s, frozen discriminator (no param grads, no collective)"
                                   % (a.workload, B, T, V, w["E"], w["H"], w["s_dtype"]),
                       "global_batch": world * B, "seq_len": T, "parallelism": "dp%d" % world,
                       "mfma_inputs": "bf16", "accumulate": "fp32"},
            "roofline": roofline, "cpu_baseline": cpu,
            "unfrozen": {"value": round(world * B * u_steps / el_u, 1), "unit": "sentences/s", "steps": u_steps,
                         "ms_per_step": round(el_u / u_steps * 1e3, 4),
                         "what": "same batch, + all 19 parameter gradients + weight re-pack"
                                 + (" + all-reduce of the %.1f MB fp32 grad bucket (RCCL)" % (
                                     Fn._Discriminator.last_grad_bucket.numel() * 4 / 1e6) if world > 1 else "")},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
