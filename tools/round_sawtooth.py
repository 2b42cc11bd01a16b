#!/usr/bin/env python3
"""Time of the wide-output GEMM launches (QKV: bias, FC1: bias + GELU; 256x256 tiling) against the number of row tiles, across
the points where the tile count passes a whole number of rounds of 256 CUs (GPU box only).  us per launch and us per round."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat

dev = "cuda"
for name, N, epi in (("qkv", 2304, nat.EPI_BIAS), ("fc1", 3072, nat.EPI_BIAS_GELU)):
    w = ops.pack_weight((torch.randn(N, 768, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(N, device=dev)
    for t in list(range(83, 93)) + list(range(148, 156)) + list(range(168, 176)) + [197]:
        M = 256 * t
        x = torch.randn(M, 768, device=dev).to(torch.bfloat16)
        best = 1e9
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(x.view(1, M, 768), w, N, b, epi)
            e1.record()
            torch.cuda.synchronize()
            if r:
                best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
        rounds = t * (N // 256) / 256
        print(f"{name} rows {t:3d} x 256: tiles {t * (N // 256):5d} = {rounds:5.2f} rounds  {best:7.1f} us  {best / rounds:6.2f} us/round", flush=True)
