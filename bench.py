#!/usr/bin/env python3
"""Benchmark of the discriminator hot path: fwd+bwd sentences/s on synthetic Yelp-shaped batches.

    python bench.py [--gpus N --steps K --warmup W]            # N=1 directly; N>1: starts N rank processes itself
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # N>1 (driver)
    python bench.py --gpus 2 --dry-run                         # the N-rank harness on CPU/gloo, kernels skipped

One "step" = one pass of the hot path over one batch already resident in HBM: forward on soft
tokens (K1 soft-embedding GEMM, K2 BiLSTM, K3 keys, K4 bridge, K5 attention/head/CE vs labels) +
backward to d(soft tokens) with frozen parameters — what reference transfer/models.py:1185-1212 +
transfer/trainers.py:376 run per training step.

  --workload cfg3|yelp64|stress   BASELINE.json configs[2]/[3] (default), configs[1], configs[4] per-GPU slice
  --scaling weak|strong           weak: every rank owns a full per-GPU batch (default);
                                  strong: the SAME global batch split over the ranks (BASELINE configs[3]:
                                  1024 sentences -> 512 / 256 / 128 per GPU)
  --api fused|module              fused: functional.discriminator_loss (device lengths, CE inside the
                                  attention kernel); module: the drop-in SelfAttnRNNClassifier.forward(soft=True)
                                  with CPU int16 lengths + F.cross_entropy(1 - lab), as the transfer loop calls it

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (the longest kernel
of the step), `kernels` (every kernel class of SURVEY.md §8 d3, each timed in isolation through its
C-ABI stage entry point with events on the launch stream) and `cpu_baseline` (stock-torch port of the
reference op sequence from oracle/torch_ref.py on the host cores; N=1 only).
"""
import argparse
import ctypes
import hashlib
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


def source_id():
    """Hash of the kernel sources + header: ties committed PMC counter files to the build they were
    taken from (bench refuses a traffic file from another build)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "non-parallel-text-style-transfer-using-self-attn-discriminator_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hpp", ".hip")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "sad.h"), "rb").read())
    return h.hexdigest()[:16]


def make_inputs(w, seed, dev, yelp_lengths=False):
    """Synthetic batch of SURVEY.md §8 d2, generated on the device (seeded): dense softmax soft
    tokens, one-hot [sos] at t=0 and one-hot [pad] for t >= len; sorted-descending lengths."""
    from sad_amd import synth
    B, T, V = w["B"], w["T"], w["V"]
    lengths = synth.yelp_dev_lengths(B, T, seed=seed) if yelp_lengths else synth.make_lengths(B, T, seed=seed)
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


def time_launches(fn, iters=10, warm=2):
    """Average duration of back-to-back launches, events on torch's current stream == the stream
    libsad launches on."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, params, S, lengths, mask, labels):
    """Stock-torch port of the reference op sequence on the host cores, fp32, bounded sample."""
    from oracle.torch_ref import TorchRefDiscriminator, fwd_bwd_step
    host = os.cpu_count() or 1
    cap_name = "SAD_CPU_THREADS" if os.environ.get("SAD_CPU_THREADS") else ("OMP_NUM_THREADS" if os.environ.get("OMP_NUM_THREADS") else None)
    threads = min(host, int(os.environ[cap_name])) if cap_name else host
    torch.set_num_threads(threads)
    B = w["B"]
    # the whole batch when one CPU step stays in the tens of seconds, else every k-th sentence (same length mix)
    n_s = B if B * w["T"] * w["V"] <= 1024 * 32 * 9353 else 16
    sel = torch.arange(0, B, max(1, B // n_s))[:n_s]
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
                host_cpus=host, cpu_model=cpu_model(),
                thread_cap=("%s=%s" % (cap_name, os.environ[cap_name])) if cap_name else None,
                sample="%d of the %d sentences (%s) as one batch x %d steps, fp32, oracle/torch_ref.py "
                       "(stock nn.LSTM/matmul op sequence of the reference)"
                       % (len(sel), B, "the whole batch" if len(sel) == B else "every %d-th, same length mix" % (B // n_s), n))


def kernel_table(w, lib, Fn, pw, ws, S, lengths, labels, logits, weights):
    """Every kernel class of SURVEY.md §8 d3 timed in isolation on the step's own workspace through the
    stage entry points of the C-ABI.  Algorithmic work: DESIGN.md §4."""
    B, T, V, E, H = w["B"], w["T"], w["V"], w["E"], w["H"]
    d = lib.dims(B, T, V, E, H, w["L"])
    pd = lib.padded_dims(d)
    Ep, Hp, Gp, Ncat = pd["Ep"], pd["Hp"], pd["Gp"], pd["Ncat"]
    L = lib.load()
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    kind = lib.SRC_BF16 if w["s_dtype"] == "bf16" else lib.SRC_F32
    esz = 2 if w["s_dtype"] == "bf16" else 4
    M = B * T
    rows = int((np.arange(T)[:, None] < lengths[None, :]).sum())          # token rows with t < len
    rows_att = int(np.maximum(lengths.astype(np.int64) - 2, 1).sum())       # attention-active rows (mask: pad/sos/eos)
    half_gates = L.sad_uses_half_gates(ctypes.byref(d)) == 1               # gate pre-activations leave as fp16 records
    dS = torch.empty_like(S)
    lg, wt = torch.empty_like(logits), torch.empty_like(weights)
    ls = torch.empty((B,), dtype=torch.float32, device=S.device)
    a = lib.SadBwdArgs()
    a.labels, a.dloss_scale = labels.data_ptr(), 1.0 / B
    a.logits, a.weights = logits.data_ptr(), weights.data_ptr()
    D = ctypes.byref(d)
    calls = [
        ("K1 soft-embedding fwd (k_softemb_fwd_sk + k_softemb_fwd_fix)", "mfma+hbm",
         lambda: L.sad_softemb_fwd(D, p(pw.buf), p(S), kind, V, B * V, p(ws.buf), st()),
         2.0 * rows * V * E, rows * V * esz + M * Ep * 2),
        ("K2a+K3 in-projection fwd (k_inproj_fwd_rs3)", "mfma",
         lambda: L.sad_inproj_fwd(D, p(pw.buf), p(ws.buf), st()),
         2.0 * rows * E * 9 * H, rows * Ep * 2 + rows * (2 * Gp * (2 if half_gates else 4) + Hp * 4)),
        ("K2b LSTM recurrence fwd (k_lstm_fwd_persist2)", "latency",
         lambda: L.sad_lstm_fwd(D, p(pw.buf), p(ws.buf), st()),
         2.0 * rows * H * 4 * H * 2, 0),
        ("K5 attention+pool+head+CE fwd (k_attn_fwd)", "hbm",
         lambda: L.sad_attnpool_fwd(D, p(pw.buf), p(labels), p(lg), p(wt), p(ls), p(ws.buf), st()),
         0.0, rows_att * Hp * 4),
        ("K5' attention bwd (k_attn_bwd)", "hbm",
         lambda: L.sad_attnpool_bwd(D, p(pw.buf), ctypes.byref(a), p(ws.buf), st()),
         0.0, rows_att * Hp * 4 + M * Hp * 2),
        ("K2' LSTM BPTT (k_lstm_bwd_persist)", "latency",
         lambda: L.sad_lstm_bwd(D, p(pw.buf), p(ws.buf), st()),
         2.0 * rows * H * 4 * H * 2, 0),
        ("K2a'+K3' in-projection bwd (K1 pipeline on dZ)", "mfma",
         lambda: L.sad_inproj_bwd(D, p(pw.buf), p(ws.buf), st()),
         2.0 * rows * E * 9 * H, rows * Ncat * 2 + M * Ep * 2),
        ("K1' soft-embedding bwd (k_softemb_bwd_ws2)", "hbm",
         lambda: L.sad_softemb_bwd(D, p(pw.buf), p(dS), kind, V, B * V, p(ws.buf), st()),
         2.0 * rows * V * E, M * V * esz + rows * Ep * 2),
    ]
    out = []
    for name, bound, fn, flops, nbytes in calls:
        def call(fn=fn, name=name):
            lib.check(fn(), name)
        sec = time_launches(call)
        e = dict(kernel=name, bound=bound, avg_launch_us=round(sec * 1e6, 1))
        if flops:
            e["TFLOPs"] = round(flops / sec / 1e12, 1)
            e["mfma_frac"] = round(flops / sec / 1e12 / MFMA_PEAK_TFS, 4)
        if nbytes:
            e["algorithmic_bytes"] = int(nbytes)
            e["GBps"] = round(nbytes / sec / 1e9, 1)
            e["hbm_frac"] = round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4)
        if bound == "latency":
            e["us_per_time_step"] = round(sec * 1e6 / T, 2)
        out.append(e)
    ws.check()
    return out


def pmc_counters(kernel_name):
    """Counters of the committed --pmc passes of THIS command (tools/pmc_bench.sh ->
    profiles/pmc_bench_traffic.json), only if they were taken from the same sources."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_bench_traffic.json")))
    except (OSError, ValueError):
        return None, "no profiles/pmc_bench_traffic.json"
    if pmc.get("_source_id") != source_id():
        return None, "profiles/pmc_bench_traffic.json was taken from another build (%s != %s)" % (pmc.get("_source_id"), source_id())
    for k, v in pmc.items():
        if not k.startswith("_") and k.split("<")[0] in kernel_name:
            return v, None
    return None, "kernel not in the counter file"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one process per GPU).  Without WORLD_SIZE in the environment and N > 1 this process "
                         "starts the N rank processes itself; under torch.distributed.run it must equal WORLD_SIZE")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--api", default="fused", choices=["fused", "module"])
    ap.add_argument("--s-dtype", default=None, choices=["bf16", "f32"])
    ap.add_argument("--onehot", action="store_true",
                    help="straight-through one-hot soft tokens (what the transfer loop really emits) with the "
                         "(index, value) side channel of the fused producer: K1 forward becomes a gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-kernel table (counter passes)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="unfrozen mode at N > 1: one all-reduce of the whole bucket after the backward instead of two "
                         "messages, the first beside k_wgrad_emb (dp.OverlappedGradReducer)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU, no kernels: process-group init (gloo), sharding for --scaling, barriers, max-over-ranks "
                         "timing, the gradient-bucket all-reduce in the product's hook order and the rank-0 JSON line, "
                         "with a stand-in step.  The printed value is NOT a measurement ('dry_run': true)")
    return ap.parse_args(argv)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh interpreters, started before
    anything in this process has touched the GPU; this parent never does) with the environment
    torch.distributed.run would give them, wait, and return the worst exit code.  Rank 0's JSON line goes to
    this process's stdout (inherited)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        for p_ in procs:
            r_ = p_.wait()
            if r_ != 0 and rc == 0:
                rc = r_
                for q_ in procs:                      # a dead rank leaves the others in a collective: end them
                    if q_.poll() is None:
                        q_.terminate()
    except KeyboardInterrupt:
        for q_ in procs:
            if q_.poll() is None:
                q_.terminate()
        rc = 130
    return rc


def dry_workload(a, w, rank, world, dev):
    """Stand-ins for --dry-run: the real lengths / shard arithmetic and a gradient bucket of the real size, but
    no soft tokens and no kernels.  Returns (step, step_unfrozen, B_local, lengths)."""
    from sad_amd import dp, functional as Fn, synth
    if a.scaling == "strong":
        lo, hi = dp.shard_bounds(w["B"], rank, world)
        if hi - lo < 1:
            raise SystemExit("strong scaling: fewer sentences than ranks")
        lengths = (synth.yelp_dev_lengths if a.workload == "yelp64" else synth.make_lengths)(w["B"], w["T"], seed=1234)[lo:hi]
    else:
        lengths = (synth.yelp_dev_lengths if a.workload == "yelp64" else synth.make_lengths)(w["B"], w["T"], seed=1234 + rank)
    shapes = synth.param_shapes(w["V"], w["E"], w["H"], w["L"])
    sizes = [int(np.prod(shapes[n])) for n in Fn._LIVE_PARAMS]
    work = torch.ones((256, 256), dtype=torch.float32, device=dev)

    def step():
        return (work @ work).sum()

    reducer = None if a.no_overlap else dp.OverlappedGradReducer()
    holders = [torch.nn.Parameter(torch.empty(n_, device=dev)) for n_ in sizes]

    def step_unfrozen():
        if reducer is not None:
            reducer.finish()                          # what precedes the optimizer / sad_pack_weights
        step()
        # the backward's bucket: [embedding.weight | rest], every rank contributes rank + 1
        flat = torch.full((sum(sizes),), float(rank + 1), dtype=torch.float32, device=dev)
        if reducer is not None:                       # the hook order of functional._Discriminator.backward
            reducer(flat[sizes[0]:], "rest")
            step()                                    # stands for k_wgrad_emb running beside the first message
            reducer(flat[:sizes[0]], "embedding")
            reducer.finish()
        else:
            off = 0
            for p_, n_ in zip(holders, sizes):
                p_.grad = flat[off:off + n_]
                off += n_
            flat = dp.allreduce_grads(holders)
        return flat

    return step, step_unfrozen, len(lengths), lengths, reducer


def main(argv=None):
    a = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (a.gpus or 1) > 1:
        # no launcher: become the launcher.  Nothing above has initialised the GPU in this process.
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:] if argv is None else argv))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus is not None and a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch as `python -m torch.distributed.run --nproc-per-node %d "
                         "... bench.py --gpus %d`, or without a launcher as `python bench.py --gpus %d`"
                         % (a.gpus, world, a.gpus, a.gpus, a.gpus))
    if world > 1:
        # "nccl" == RCCL on ROCm; the dry run rehearses the same harness on gloo
        dist.init_process_group("gloo" if a.dry_run else "nccl", rank=rank, world_size=world)
    if a.dry_run:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)

    def sync():
        if not a.dry_run:
            torch.cuda.synchronize()

    from sad_amd import dp, functional as Fn, synth
    w = dict(WORKLOADS[a.workload])
    if a.s_dtype:
        w["s_dtype"] = a.s_dtype
    global_B = w["B"] * world if a.scaling == "weak" else w["B"]
    reducer = None
    if a.dry_run:
        step, step_unfrozen, B, lengths, reducer = dry_workload(a, w, rank, world, dev)
        w["B"] = B
        T, V = w["T"], w["V"]
    else:
        from sad_amd import lib
        Fn.KEEP_LAST = True                                       # the per-kernel table re-uses the step's workspace
        if a.scaling == "strong":
            lo, hi = dp.shard_bounds(w["B"], rank, world)             # contiguous slice of the SAME sorted batch
            if hi - lo < 1:
                raise SystemExit("strong scaling: fewer sentences than ranks")
        params = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21)
        tparams = {k: torch.from_numpy(v).to(dev) for k, v in params.items()}
        pw = Fn.pack_weights(tparams)
        if a.scaling == "strong":
            # every rank generates the same global batch (same seed) and keeps its shard
            S, lengths, ln_dev, mask, labels = make_inputs(w, 1234, dev, yelp_lengths=a.workload == "yelp64")
            S = S[:, lo:hi].contiguous()
            lengths, ln_dev, mask, labels = lengths[lo:hi], ln_dev[lo:hi].contiguous(), mask[lo:hi].contiguous(), labels[lo:hi].contiguous()
            w["B"] = hi - lo
        else:
            S, lengths, ln_dev, mask, labels = make_inputs(w, 1234 + rank, dev, yelp_lengths=a.workload == "yelp64")
        B, T, V = w["B"], w["T"], w["V"]
        onehot = None
        if a.onehot:
            # exact one-hot rows + their side channel, as producer.SoftTokenAssembler emits them
            ids_tb = torch.from_numpy(synth.make_token_ids(B, T, V, lengths, seed=1234 + rank)).to(dev).t().contiguous()   # [T,B]
            S.zero_()
            S.scatter_(2, ids_tb.unsqueeze(2), 1.0)
            onehot = (ids_tb.t(), torch.ones((T, B), dtype=torch.float32, device=dev).t())
        Sv = S.transpose(0, 1).requires_grad_(True)           # [B,T,V] view, exactly like the transfer loop

        if a.api == "module":
            # the drop-in module as reference transfer/models.py:1172-1212 drives it: CPU int16 lengths,
            # bool mask, soft=True, un-fused CE against the inverted label
            from types import SimpleNamespace
            from sad_amd import transfer_models
            cfg = SimpleNamespace(cell="LSTM", enc_bidirectional=True, pad_id=0, dropout=0.0, enc_num_layers=1,
                                  hidden_size=w["H"], vocab_size=V, emb_size=w["E"], num_labels=w["L"],
                                  freeze_embedding=False, device=str(dev), mask_id=None)
            model = transfer_models.SelfAttnRNNClassifier(cfg)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
            model.to(dev)
            for p_ in model.parameters():                     # TransferModel.freeze_cls (transfer/models.py:995-999)
                p_.requires_grad = False
            len16 = torch.from_numpy(lengths.astype(np.int16))
            inv_lab = 1 - labels

            def step():
                Sv.grad = None
                logits, _ = model(Sv, len16, mask.clone(), soft=True, onehot=onehot)
                loss = torch.nn.functional.cross_entropy(logits, inv_lab, reduction="none")
                (loss.sum() / global_B).backward()
                return loss
        else:
            def step():
                Sv.grad = None
                loss, _, _ = Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw, onehot=onehot)
                (loss.sum() / global_B).backward()
                return loss

        # --- unfrozen mode: + parameter gradients, + the all-reduce of the flat grad bucket
        # over RCCL when N > 1, + re-packing the weights (an optimizer step invalidates the image).  At N > 1
        # the bucket travels in two messages (dp.OverlappedGradReducer): everything but embedding.weight is
        # handed to RCCL while k_wgrad_emb (the longest weight-gradient kernel) still runs, embedding.weight
        # after it; both are waited for before the next sad_pack_weights, so no collective kernel is in flight
        # when the persistent recurrences (which need every CU) start.  DESIGN.md "Multi-GPU".
        live = {k: tparams[k].clone().requires_grad_(True) for k in Fn._LIVE_PARAMS}
        if world > 1 and not a.no_overlap:
            reducer = dp.OverlappedGradReducer()

        def step_unfrozen():
            Sv.grad = None
            for p_ in live.values():
                p_.grad = None
            if reducer is not None:
                reducer.finish()                              # the previous step's messages, before the re-pack
            pw.pack(live)
            loss, _, _ = Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw, params=live, pad_id=0)
            Fn.GRAD_READY_HOOK = reducer
            try:
                (loss.sum() / global_B).backward()
            finally:
                Fn.GRAD_READY_HOOK = None
            if reducer is not None:
                reducer.finish()
                return Fn._Discriminator.last_grad_bucket
            return dp.allreduce_grads(list(live.values()))       # the grads are views of ONE bucket: reduced in place

    def timed(fn, steps, warm):
        for _ in range(warm):
            fn()
        sync()
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        if world > 1:
            dist.barrier()
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    el = timed(step, a.steps, a.warmup)
    u_steps = max(1, min(a.steps, 10))
    el_u = timed(step_unfrozen, u_steps, 2)
    bucket = step_unfrozen()
    kernels, roofline, cpu, extra = None, None, None, {}
    if a.dry_run:
        # every rank filled its bucket with rank + 1: the reduced bucket must hold 1 + 2 + ... + world everywhere
        want = world * (world + 1) / 2.0
        ok = bool((bucket == want).all())
        extra = {"dry_run": True, "note": "harness rehearsal on CPU/gloo with a stand-in step: NOT a measurement",
                 "bucket_allreduce_ok": ok, "bucket_numel": int(bucket.numel()),
                 "messages_per_step": [list(x) for x in (reducer.log[-2:] if reducer is not None else [("all", int(bucket.numel()))])],
                 "local_batch": B, "backend": "gloo" if world > 1 else None}
        if not ok:
            raise SystemExit("dry run: the all-reduced gradient bucket is wrong on rank %d" % rank)
    else:
        pw.pack(tparams)
        # leave a frozen step's workspace behind for the per-kernel timing below
        Sv.grad = None
        loss_, logits_, weights_ = Fn.discriminator_loss(Sv, ln_dev, mask, labels, pw)
        (loss_.sum() / global_B).backward()
        torch.cuda.synchronize()
        ws = Fn._Discriminator.last_workspace
        ws.check()                                               # hand-off timeout words of the recurrences

    if rank == 0:
        if not a.dry_run and not a.no_kernels:
            kernels = kernel_table(w, lib, Fn, pw, ws, S, lengths, labels, logits_.detach(), weights_.detach())
            dom = max(kernels, key=lambda e: e["avg_launch_us"])
            if dom["bound"] == "hbm" or (dom["bound"] == "mfma+hbm" and dom.get("hbm_frac", 0) >= dom.get("mfma_frac", 0)):
                roofline = dict(bound="hbm", kernel=dom["kernel"], achieved=dom["GBps"], peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=dom["hbm_frac"], algorithmic_bytes=dom["algorithmic_bytes"])
            else:
                roofline = dict(bound="mfma", kernel=dom["kernel"], achieved=dom["TFLOPs"], peak=MFMA_PEAK_TFS,
                                unit="TFLOP/s", frac=dom["mfma_frac"])
                if dom["bound"] == "latency":
                    roofline["note"] = ("strictly sequential recurrence: %d dependent steps of %.2f us; MFMA peak is the "
                                        "only roof it has, hand-off latency is what binds it" % (T, dom["us_per_time_step"]))
            roofline["avg_launch_us"] = dom["avg_launch_us"]
            cnt, why = pmc_counters(dom["kernel"])
            roofline["traffic"] = cnt["hbm_bytes_per_launch"] if cnt and a.workload == "cfg3" and w["s_dtype"] == "bf16" and world == 1 else None
            if roofline["traffic"] is None:
                roofline["traffic_note"] = why or "counters are taken on the default workload only"
            elif "mfma_busy_frac" in cnt:
                roofline["mfma_busy_frac_pmc"] = cnt["mfma_busy_frac"]
        if not a.dry_run and world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(w, params, S, lengths, mask, labels)
        units = (B * world if a.scaling == "weak" else global_B)
        out = {
            "metric": "discriminator fwd+bwd sentences/sec (Yelp shape)",
            "value": round(units * a.steps / el, 1), "unit": "sentences/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None,
            # the arithmetic type of the contractions; fp32 soft tokens are rounded to bf16 on their way into the MFMAs
            "dtype": "bf16" if w["s_dtype"] == "bf16" else "bf16 (fp32 soft tokens / d(soft tokens) in HBM)",
            "data": "synthetic",
            "config": {"workload": "%s: synthetic Yelp-shaped soft tokens, B=%d per GPU, T=%d, V=%d, E=%d, H=%d, "
                                   "len %s sorted, S %s%s, frozen discriminator (no param grads, no collective), api=%s"
                                   % (a.workload, B, T, V, w["E"], w["H"],
                                      "from data/yelp/dev.* (+sos/eos)" if a.workload == "yelp64" else "~U{3..T}",
                                      w["s_dtype"], " straight-through one-hot + side channel" if a.onehot else " dense", a.api),
                       "global_batch": units, "seq_len": T, "parallelism": "dp%d" % world,
                       "mfma_inputs": "bf16", "accumulate": "fp32"},
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
            "whole_step": {"algorithmic_TFLOP": round(step_flops(w, lengths) / 1e12, 4),
                           "mfma_frac": round(step_flops(w, lengths) / (el / a.steps) / 1e12 / MFMA_PEAK_TFS, 4)},
            "unfrozen": {"value": round(units * u_steps / el_u, 1), "unit": "sentences/s", "steps": u_steps,
                         "ms_per_step": round(el_u / u_steps * 1e3, 4),
                         "what": "same batch, + all 19 parameter gradients + weight re-pack"
                                 + ((" + all-reduce of the %.1f MB fp32 grad bucket (%s), %s"
                                     % (bucket.numel() * 4 / 1e6, "gloo" if a.dry_run else "RCCL",
                                        "one message after the backward" if reducer is None else
                                        "two messages, the first beside k_wgrad_emb")) if world > 1 else "")},
            "source_id": source_id(),
        }
        if not a.dry_run:
            out["sad_version"] = lib.load().sad_version()
        out.update(extra)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def step_flops(w, lengths):
    """Algorithmic flops of one frozen fwd+bwd step over the rows with t < len (SURVEY.md §8 d3;
    dead ops excluded): forward K1 + K2a + K2b + K3 per token, backward the same again."""
    rows = float((np.arange(w["T"])[:, None] < lengths[None, :]).sum())
    V, E, H = w["V"], w["E"], w["H"]
    per_token = 2.0 * V * E + 2.0 * E * 8 * H + 2.0 * H * 8 * H + 2.0 * E * H
    per_sentence = 2.0 * 2 * H * H + 2.0 * H * H + 4.0 * H
    return 2.0 * (rows * per_token + len(lengths) * per_sentence)


if __name__ == "__main__":
    main()
