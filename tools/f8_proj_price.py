#!/usr/bin/env python3
"""What an fp8 x fp8 proj would cost with the existing kernels (VERDICT r2 #3 asked for an e4m3 proj input; the attention kernel does
not emit one): the fp8 RESID kernel on proj's shapes against the product's bf16-activation x e4m3-weight launch (GPU box only)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat

dev = "cuda"


def timed(fn):
    best = 1e9
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    return best


for t in (197, 173, 152, 121, 87):
    M, N, K = 256 * t, 768, 768
    xq = torch.randint(0, 120, (M, K), device=dev, dtype=torch.uint8)
    xb = torch.randn(M, K, device=dev).to(torch.bfloat16)
    wq = torch.randint(0, 120, ((N + 255) // 256 * 256, K), device=dev, dtype=torch.uint8)
    xs = torch.rand(M, device=dev) / 64 + 0.01
    ws = torch.rand(N, device=dev) / 64 + 0.01
    b = torch.randn(N, device=dev)
    resid = torch.randn(1, M, N, device=dev)
    t8 = timed(lambda: ops.linear(xq.view(1, M, K), wq, N, b, nat.EPI_BIAS_RESID, resid=resid, w_scale=ws, x_scale=xs))
    tb = timed(lambda: ops.linear(xb.view(1, M, K), wq, N, b, nat.EPI_BIAS_RESID, resid=resid, w_scale=ws))
    print(f"M={M:6d}: fp8 x fp8 {t8:7.1f} us   bf16 x e4m3-weights (product) {tb:7.1f} us", flush=True)
