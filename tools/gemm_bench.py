#!/usr/bin/env python3
"""Micro-benchmark of rajni_linear on the ViT-B/16 shapes, per forced tiling (GPU box only).
Interleaved rounds in ONE process (guide rule 24), random data (rule 25), HIP-event timing."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat

dev = "cuda"
shapes = [("qkv", 50432, 2304, 768, nat.EPI_BIAS), ("fc1", 50432, 3072, 768, nat.EPI_BIAS_GELU),
          ("proj", 50432, 768, 768, nat.EPI_BIAS_RESID), ("fc2", 50432, 768, 3072, nat.EPI_BIAS_RESID),
          ("fc2_87", 22272, 768, 3072, nat.EPI_BIAS_RESID), ("qkv_121", 30976, 2304, 768, nat.EPI_BIAS)]
modes = [int(m) for m in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,4,5".split(","))]
FP8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"   # second argument "fp8": e4m3 weights + per-row scale
rounds = 5
res = {}
for name, M, N, K, epi in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w, wsc = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16), None
    w, wsc = ops.pack_weight_fp8(w, torch.bfloat16, dev) if FP8 else (ops.pack_weight(w), None)
    b = torch.randn(N, device=dev)
    resid = torch.randn(1, M, N, device=dev) if epi == nat.EPI_BIAS_RESID else None
    out = None
    times = {m: [] for m in modes}
    for r in range(rounds + 1):
        for m in modes:
            nat.lib().rajni_debug_force_gemm_tiling(m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y = ops.linear(x.view(1, M, K), w, N, b, epi, resid=resid, w_scale=wsc)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[m].append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * M * N * K
    res[name] = {m: round(fl / (min(t) * 1e-3) / 1e12, 1) for m, t in times.items()}
    print(name, M, N, K, {m: f"{min(t)*1e3:.1f}us {fl/(min(t)*1e-3)/1e12:.0f}TF (med {fl/(sorted(t)[len(t)//2]*1e-3)/1e12:.0f})" for m, t in times.items()}, flush=True)
nat.lib().rajni_debug_force_gemm_tiling(0)
print(json.dumps(res))
