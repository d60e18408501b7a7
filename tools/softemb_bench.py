"""Sibling soft contractions: sad_amd.functional.soft_embedding vs the stock torch matmul
the reference runs, forward + backward (d soft_tokens and d weight), at the content-alignment shape
[T,B,V] @ [V,E] of config 3 and at one decoder step [1,B,V].  Prints one JSON line per shape."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sad_amd import functional as Fn
dev = torch.device("cuda:0")
torch.manual_seed(0)
V, E = 9353, 300
W = (torch.randn(V, E, device=dev) * 0.05).requires_grad_(True)


def timeit(f, n=10, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, shape in (("content_align", (32, 1024, V)), ("decoder_step", (1, 1024, V))):
    for dtype in (torch.bfloat16, torch.float32):
        S = torch.softmax(torch.randn(shape, device=dev), -1).to(dtype).requires_grad_(True)
        G = torch.randn(shape[:-1] + (E,), device=dev).to(dtype)

        def ours():
            S.grad = None; W.grad = None
            Fn.soft_embedding(S, W).backward(G)

        def stock():
            S.grad = None; W.grad = None
            (S @ W.to(dtype)).backward(G)

        t_o, t_s = timeit(ours), timeit(stock)
        rows = shape[0] * shape[1]
        print(json.dumps({"op": "soft_embedding fwd+bwd", "shape": name, "rows": rows, "V": V, "E": E,
                          "dtype": str(dtype).split(".")[-1], "hip_us": round(t_o, 1), "torch_matmul_us": round(t_s, 1),
                          "rows_per_s": round(rows / t_o * 1e6)}))
