#!/usr/bin/env python3
"""Does the epilogue's bias load (an ordinary global load queued behind the next tile's DMA) cost time?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K, epi in [("qkv", 50432, 2304, 768, nat.EPI_BIAS), ("fc1", 50432, 3072, 768, nat.EPI_BIAS_GELU)]:
    x = torch.randn(1, M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(N, device=dev)
    for r in range(2):
        print(name, "with bias %.1f us" % t(lambda: ops.linear(x, w, N, b, epi)), "| without bias %.1f us" % t(lambda: ops.linear(x, w, N, None, epi)))
