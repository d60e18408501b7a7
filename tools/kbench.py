#!/usr/bin/env python3
"""Micro-benchmark of the two soft-embedding kernels in isolation (events on the launch stream).
usage: python tools/kbench.py [--V 9353] [--B 1024] [--T 32] [--dtype bf16|f32] [--full]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sad_amd import functional as Fn, lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--V", type=int, default=9353); ap.add_argument("--B", type=int, default=1024)
ap.add_argument("--T", type=int, default=32); ap.add_argument("--E", type=int, default=300)
ap.add_argument("--H", type=int, default=512)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--full", action="store_true")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, T, V = a.B, a.T, a.V
params = synth.make_params(V, a.E, a.H, 2, seed=21)
pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
lengths = synth.make_lengths(B, T, seed=1234, full=a.full)
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
S = (torch.rand((T, B, V), device=dev) / V * 2).to(dt)
dS = torch.empty_like(S)
mask = torch.from_numpy(synth.transfer_mask(lengths, T)).to(dev)
ln = torch.from_numpy(lengths).to(dev)
d = lib.dims(B, T, V, a.E, a.H, 2)
ws = Fn.Workspace(d, dev)
ws.buf.zero_()
L = lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.check(L.sad_prep(ctypes.byref(d), p(ln), p(mask), mask.stride(0), mask.stride(1), 1, p(ws.buf), st()), "prep")
kind = lib.SRC_BF16 if a.dtype == "bf16" else lib.SRC_F32
esz = 2 if a.dtype == "bf16" else 4
ws.view("DX", torch.bfloat16).normal_()
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e-3
tf = timeit(lambda: lib.check(L.sad_softemb_fwd(ctypes.byref(d), p(pw.buf), p(S), kind, V, B * V, p(ws.buf), st()), "f"))
tb = timeit(lambda: lib.check(L.sad_softemb_bwd(ctypes.byref(d), p(pw.buf), p(dS), kind, V, B * V, p(ws.buf), st()), "b"))
rows = int((np.arange(T)[:, None] < lengths[None, :]).sum())
M = B * T
print("V=%d B=%d T=%d %s full=%s rows=%d/%d | fwd %.1f us  %.0f GB/s  %.0f TF/s | bwd %.1f us  %.0f GB/s  %.0f TF/s" % (
    V, B, T, a.dtype, a.full, rows, M, tf * 1e6, rows * V * esz / tf / 1e9, 2.0 * rows * V * a.E / tf / 1e12,
    tb * 1e6, M * V * esz / tb / 1e9, 2.0 * rows * V * a.E / tb / 1e12))
