#!/usr/bin/env python3
"""A/B of the persistent GEMM grid: one workgroup per CU vs the smallest grid with the same number of
tile rounds (rajni_debug_set_gemm_balanced_grid), on every GEMM shape of the ViT-B README schedule."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat

dev = "cuda"
B = 256
shapes = []
for n in (197, 173, 152, 121, 87):
    M = B * n
    shapes += [(f"qkv_{n}", M, 2304, 768, nat.EPI_BIAS), (f"fc1_{n}", M, 3072, 768, nat.EPI_BIAS_GELU),
               (f"proj_{n}", M, 768, 768, nat.EPI_BIAS_RESID), (f"fc2_{n}", M, 768, 3072, nat.EPI_BIAS_RESID)]
tot = {0: 0.0, 1: 0.0}
for name, M, N, K, epi in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(N, device=dev)
    resid = torch.randn(1, M, N, device=dev) if epi == nat.EPI_BIAS_RESID else None
    times = {0: [], 1: []}
    for r in range(6):
        for m in (0, 1):
            nat.lib().rajni_debug_set_gemm_balanced_grid(m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(x.view(1, M, K), w, N, b, epi, resid=resid)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[m].append(e0.elapsed_time(e1) / 5 * 1e3)
    a, bb = min(times[0]), min(times[1])
    tot[0] += a; tot[1] += bb
    print(f"{name:10s} per-CU grid {a:7.1f} us   balanced {bb:7.1f} us   {100*(bb/a-1):+.1f}%", flush=True)
nat.lib().rajni_debug_set_gemm_balanced_grid(0)
print("sum", tot)
