"""Role ablations of the three wave-specialised GEMM kernels at cfg3: which role sets the pace of a tile?

    ABL_KERNEL=k1|ip|k1b python tools/k1abl.py build   # here: variant libraries tools/abl/libsad_<kernel>_abl<bits>.so
    ABL_KERNEL=k1|ip|k1b python tools/k1abl.py         # on the GPU box: times the stage entry point with every variant

    k1  (default)  K1  k_softemb_fwd_sk, -DSAD_K1_ABL:  1 no MFMA, 2 no LDS fragment reads + no MFMA, 4 no S global loads,
                   8 no S LDS writes, 16 no W LDS-direct loads                     -> profiles/r03_k1_role_ablation.txt
    ip             in-projection forward k_inproj_fwd_rs3, -DSAD_IP_ABL: 1 no MFMA, 2 no fragment reads + no MFMA,
                   4 no epilogue, 8 no W loads, 16 no W LDS writes                 -> profiles/r03_inproj_fwd_ablation.txt
    k1b            K1' k_softemb_bwd_ws2, -DSAD_K1B_ABL: 1 no global stores, 2 no MFMA, 4 no W_emb loads
                   (K1B_FULL=1: every row needed)                                  -> profiles/r03_k1b_role_ablation.txt
    K1_V=<V>       vocabulary size of the probe (row alignment of S)

Variant results are garbage by construction; only the launch time is read.  The variant libraries are selected through
SAD_LIB, one process per variant."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, "tools", "abl")
KERNEL = os.environ.get("ABL_KERNEL", "k1")          # k1: -DSAD_K1_ABL, sad_softemb_fwd;  ip: -DSAD_IP_ABL, sad_inproj_fwd
VARIANTS = {"k1": [0, 1, 3, 4, 12, 16, 28, 19, 7, 15, 31], "ip": [0, 1, 3, 4, 8, 24, 7, 27, 31], "k1b": [0, 1, 2, 4, 3, 6, 7]}[KERNEL]
NAMES = {0: "full kernel", 1: "no MFMA", 3: "no fragment reads, no MFMA", 4: "no S loads", 12: "no S loads, no S LDS writes",
         16: "no W loads", 28: "no loads at all (S, W), no S LDS writes", 19: "W stream only feeds nobody: no W loads, no compute",
         7: "S LDS writes + W loads only (no S loads, no compute)", 15: "W loads only", 31: "barriers only"}
if KERNEL == "k1b":    # bits: 1 no global stores, 2 no MFMA, 4 no W loads
    NAMES = {0: "full kernel", 1: "no global stores", 2: "no MFMA", 4: "no W loads", 3: "no stores, no MFMA", 6: "stores only (no MFMA, no W loads)",
             7: "barriers + LDS traffic only"}
if KERNEL == "ip":     # bits: 1 no MFMA, 2 no fragment reads + no MFMA, 4 no epilogue, 8 no W loads, 16 no W LDS writes
    NAMES = {0: "full kernel", 1: "no MFMA", 3: "no fragment reads, no MFMA", 4: "no epilogue (no output stores)", 8: "no W loads",
             24: "no W loads, no W LDS writes", 7: "W staging only (no compute, no epilogue)", 27: "epilogue only", 31: "barriers only"}


def build():
    os.makedirs(ABL, exist_ok=True)
    src = os.path.join(ROOT, "non-parallel-text-style-transfer-using-self-attn-discriminator_amd", "csrc", "sad_api.hip")
    procs = []
    for v in VARIANTS:
        out = os.path.join(ABL, "libsad_%s_abl%d.so" % (KERNEL, v))
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                       "-Wno-unused-function", "-DSAD_%s_ABL=%d" % ({"k1": "K1", "ip": "IP", "k1b": "K1B"}[KERNEL], v), "-o", out, src]))
        if len(procs) == 4:
            for p in procs:
                assert p.wait() == 0
            procs = []
    for p in procs:
        assert p.wait() == 0


def one():
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    from sad_amd import functional as Fn, synth, lib
    import bench
    dev = torch.device("cuda:0")
    w = dict(bench.WORKLOADS["cfg3"])
    if os.environ.get("K1_V"):
        w["V"] = int(os.environ["K1_V"])            # alignment probe: row pitch of S in bytes = 2 V
    params = synth.make_params(w["V"], w["E"], w["H"], w["L"], seed=21)
    pw = Fn.pack_weights({k: torch.from_numpy(v).to(dev) for k, v in params.items()})
    S, lengths, ln_dev, mask, labels = bench.make_inputs(w, 1234, dev)
    Fn.KEEP_LAST = True
    with torch.no_grad():
        Fn.discriminator_loss(S.transpose(0, 1), ln_dev, mask, labels, pw)
    ws = Fn._Discriminator.last_workspace
    L = lib.load()
    d = lib.dims(w["B"], w["T"], w["V"], w["E"], w["H"], w["L"])
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    if KERNEL == "k1b":
        full = os.environ.get("K1B_FULL")           # every row needed (lengths = T)
        dS = torch.empty_like(S)
        if full:
            ws.view("NEED", torch.uint8).fill_(1)
        fn = lambda: L.sad_softemb_bwd(ctypes.byref(d), p(pw.buf), p(dS), lib.SRC_BF16, w["V"], w["B"] * w["V"], p(ws.buf), st)
    elif KERNEL == "k1":
        fn = lambda: L.sad_softemb_fwd(ctypes.byref(d), p(pw.buf), p(S), lib.SRC_BF16, w["V"], w["B"] * w["V"], p(ws.buf), st)
    else:
        fn = lambda: L.sad_inproj_fwd(ctypes.byref(d), p(pw.buf), p(ws.buf), st)
    sec = bench.time_launches(fn, iters=20, warm=3)
    print("%8.1f us" % (sec * 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for v in VARIANTS:
            env = dict(os.environ, SAD_LIB=os.path.join(ABL, "libsad_%s_abl%d.so" % (KERNEL, v)))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True, timeout=300)
            print("ABL=%2d %-58s %s" % (v, NAMES[v], r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-300:]), flush=True)
