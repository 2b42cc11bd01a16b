#!/usr/bin/env python3
"""fp8 x fp8 QKV launches (bias epilogue, bf16 output): 256x128 (tiling 1) vs 256x256 (tiling 2) per token count, ViT-B dims at
batch 256 and 512 (GPU box only).  us per launch, min over repetitions."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat

dev = "cuda"
for B in (256, 512):
    for t in (197, 173, 152, 121, 87):
        M, N, K = B * t, 2304, 768
        xq = torch.randint(0, 120, (M, K), device=dev, dtype=torch.uint8)
        wq = torch.randint(0, 120, ((N + 255) // 256 * 256, K), device=dev, dtype=torch.uint8)
        xs = torch.rand(M, device=dev) / 64 + 0.01
        ws = torch.rand(N, device=dev) / 64 + 0.01
        b = torch.randn(N, device=dev)
        res = {}
        for til in (1, 2):
            nat.lib().rajni_debug_force_f8_tiling(til)
            best = 1e9
            for r in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    y = ops.linear(xq, wq, N, b, nat.EPI_BIAS, w_scale=ws, x_scale=xs)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
            res[til] = best
        if B == 256 and t == 197:      # the two tilings give the same bits? (same fp32 sums in a different order: no - report max diff)
            nat.lib().rajni_debug_force_f8_tiling(1); y1 = ops.linear(xq, wq, N, b, nat.EPI_BIAS, w_scale=ws, x_scale=xs).float()
            nat.lib().rajni_debug_force_f8_tiling(2); y2 = ops.linear(xq, wq, N, b, nat.EPI_BIAS, w_scale=ws, x_scale=xs).float()
            print(f"max |256x128 - 256x256| = {(y1 - y2).abs().max().item():.4g} of max {y1.abs().max().item():.4g}")
        print(f"M={M:6d}: 256x128 {res[1]:7.1f} us   256x256 {res[2]:7.1f} us   ({2 * M * N * K / res[2] / 1e6:.0f} TF on 256x256)", flush=True)
nat.lib().rajni_debug_force_f8_tiling(0)
