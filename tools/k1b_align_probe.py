"""K1' (d soft tokens) alone at vocabulary sizes whose rows are / are not 128-byte aligned: is the row pitch
(V = 9353 -> 18706 B) what costs the write bandwidth?  Prints us per launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sad_amd import functional as Fn, lib as L
dev = torch.device("cuda:0")
T, B, E = 32, 896, 300
for V in (9353, 9344, 9376, 9360, 9352):
    W = torch.randn(V, E, device=dev) * 0.05
    pw = Fn._se_packed(W)
    d, ws = Fn._se_workspace(B, T, V, E, dev)
    Ep = L.padded_dims(d)["Ep"]
    ws.view("DX", torch.bfloat16)[:T * B * Ep].normal_()
    dS = torch.empty((T, B, V), dtype=torch.bfloat16, device=dev)
    lib = L.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        L.check(lib.sad_softemb_bwd(ctypes.byref(d), Fn._ptr(pw.buf), Fn._ptr(dS), L.SRC_BF16, dS.stride(1), dS.stride(0),
                                    Fn._ptr(ws.buf), st), "bwd")
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): run()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 10 * 1e3
    print("V=%d pitch=%d B (mod 128 = %d): %.1f us, %.0f GB/s written" % (V, 2 * V, (2 * V) % 128, us, T * B * V * 2 / us / 1e3))
