"""Producer timing: producer.SoftTokenAssembler against the reference's stock
sequence around the decoder step (transfer/models.py:531-571: F.gumbel_softmax(hard=True) per step, masked copy,
length / mask bookkeeping, then cat with the [sos] row + optional [eos] row + [pad] fill), same logits, on one GPU.

Both sides get the same pre-computed per-step logits (the decoder itself is out of scope) and both are timed forward
only and forward + backward (d soft tokens = ones -> d logits of every step).  Prints one JSON line per batch size:
    python tools/producer_bench.py > profiles/r03_producer_bench.jsonl
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sad_amd import producer  # noqa: E402

PAD, SOS, EOS = 0, 1, 2


def stock_soft_decode(logits_steps, max_decoding_len):
    """The bookkeeping of AttnRNNDecoder.soft_decode around the decoder call, restated with stock torch ops."""
    B, V = logits_steps[0].shape
    dev = logits_steps[0].device
    mask = torch.ones((B,), dtype=torch.bool, device=dev)
    outputs_len = torch.zeros((B,), dtype=torch.int16, device=dev)
    output_list = []
    cur = 0
    while cur <= max_decoding_len and cur < len(logits_steps):     # (the reference also syncs on torch.sum(mask) != 0)
        out = F.gumbel_softmax(logits_steps[cur], tau=1.0, hard=True)
        idx = out.argmax(dim=1)
        output_list.append(out * mask.unsqueeze(1))
        outputs_len = outputs_len + mask.to(torch.int16)
        mask = mask & (idx != EOS)
        cur += 1
    sos = torch.zeros((1, B, V), device=dev)
    sos[:, :, SOS] = 1.0
    outputs = torch.cat([sos] + [o.unsqueeze(0) for o in output_list], 0)
    eos = torch.zeros((1, B, V), device=dev)
    eos[:, :, EOS] = 1.0
    outputs = torch.cat([outputs, eos * mask.view(1, B, 1)], 0)
    outputs_len = outputs_len + 1 + mask.to(torch.int16)
    T = outputs.shape[0]
    pad = torch.arange(T, device=dev).view(T, 1) >= outputs_len.view(1, B)
    outputs = outputs.clone()
    outputs[:, :, PAD] = outputs[:, :, PAD] + pad.float()
    return outputs, outputs_len


def fused_soft_decode(logits_steps, max_decoding_len, dtype):
    B, V = logits_steps[0].shape
    asm = producer.SoftTokenAssembler(B, V, max_decoding_len, PAD, SOS, EOS, logits_steps[0].device, dtype=dtype)
    for lg in logits_steps:
        asm.step(lg, want_rnn_in=False)
    return asm.finalize()


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    V, MAXDEC = 9353, 20
    for B in (64, 1024):
        g = torch.Generator(device=dev)
        g.manual_seed(B)
        steps = [(torch.randn((B, V), generator=g, device=dev) * 2.0).requires_grad_(True) for _ in range(MAXDEC + 1)]

        def stock_fwd():
            with torch.no_grad():
                stock_soft_decode(steps, MAXDEC)

        def stock_fb():
            S, _ = stock_soft_decode(steps, MAXDEC)
            S.sum().backward()
            for s_ in steps:
                s_.grad = None

        out = {"B": B, "V": V, "decode_steps": MAXDEC + 1,
               "stock_fwd_ms": round(timed(stock_fwd), 3), "stock_fwd_bwd_ms": round(timed(stock_fb), 3)}
        for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            def fused_fwd(dt=dt):
                with torch.no_grad():
                    fused_soft_decode(steps, MAXDEC, dt)

            def fused_fb(dt=dt):
                soft = fused_soft_decode(steps, MAXDEC, dt)
                soft.S.backward(torch.ones_like(soft.S))
                for s_ in steps:
                    s_.grad = None

            out["fused_%s_fwd_ms" % name] = round(timed(fused_fwd), 3)
            out["fused_%s_fwd_bwd_ms" % name] = round(timed(fused_fb), 3)
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        S, _ = stock_soft_decode(steps, MAXDEC)
        out["stock_peak_extra_MB"] = round((torch.cuda.max_memory_allocated() - base) / 1e6, 1)
        del S
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        soft = fused_soft_decode(steps, MAXDEC, torch.bfloat16)
        out["fused_bf16_peak_extra_MB"] = round((torch.cuda.max_memory_allocated() - base) / 1e6, 1)
        del soft
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
