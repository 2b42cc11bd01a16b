#!/usr/bin/env python3
"""General-head-dim attention (attn_bf16_dgen) next to the tuned D = 64 kernels at the same token count and
head count: microseconds per launch and TFLOP/s (GPU box only)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops

dev = "cuda"
for B, N, H in [(64, 257, 16), (256, 197, 12), (256, 87, 12)]:
    row = []
    for D in (64, 80, 128, 32):
        qkv = torch.randn(B, N, 3 * H * D, device=dev).to(torch.bfloat16)
        for _ in range(3):
            ops.attention(qkv, None, H, D ** -0.5)
        best = 1e9
        for r in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attention(qkv, None, H, D ** -0.5)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        fl = 4.0 * B * H * N * N * D
        row.append(f"D={D}: {best*1e3:7.1f} us {fl/(best*1e-3)/1e12:6.1f} TF")
    print(f"B={B} N={N} H={H}  " + "   ".join(row), flush=True)
