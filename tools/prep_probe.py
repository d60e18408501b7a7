#!/usr/bin/env python3
"""Probe of the prep stage (need bits + row lists + tile list): the same call back to back (warm
instruction cache and L2) and with an unrelated large kernel between calls (as inside a step).
Run under `rocprofv3 --kernel-trace --output-format csv`; the per-launch durations are in the trace.
usage: python3 tools/prep_probe.py [--B 1024] [--T 32]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sad_amd import lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1024); ap.add_argument("--T", type=int, default=32)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, T = a.B, a.T
lengths = synth.make_lengths(B, T, seed=1234)
mask = torch.from_numpy(synth.transfer_mask(lengths, T)).to(dev)
ln = torch.from_numpy(lengths).to(dev)
d = lib.dims(B, T, 9353, 300, 512, 2)
L = lib.load()
ws = torch.zeros((L.sad_workspace_bytes(ctypes.byref(d)),), dtype=torch.uint8, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
prep = lambda: lib.check(L.sad_prep(ctypes.byref(d), p(ln), p(mask), mask.stride(0), mask.stride(1), 1, p(ws), st()), "prep")
x = torch.randn((4096, 4096), device=dev, dtype=torch.bfloat16)
for _ in range(5): prep()
torch.cuda.synchronize()
for _ in range(5):
    y = x @ x
    z = torch.softmax(y.float(), dim=1)
    prep()
torch.cuda.synchronize()
print("done")
