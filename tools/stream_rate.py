#!/usr/bin/env python3
"""What a plain streaming read reaches on this box, next to the large-N assembly (GPU box): python tools/stream_rate.py
torch.sum over N bytes of float64 (one pass, read-only) and torch's device copy, by HIP events; sizes as in pnp_n10000_1k."""
import torch
dev = torch.device("cuda:0")
for mb in (64, 400, 1600):
    n = mb * 1024 * 1024 // 8
    x = torch.rand(n, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    for name, fn, nbytes in (("sum (read)", lambda: x.sum(), 8 * n), ("copy (read+write)", lambda: y.copy_(x), 16 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{mb} MB {name}: {ms:.4f} ms  {nbytes / ms / 1e9:.2f} TB/s")
