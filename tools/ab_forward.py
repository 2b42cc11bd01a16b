#!/usr/bin/env python3
"""A/B of a debug switch on the whole forward (README schedule), interleaved rounds in one process:
    python tools/ab_forward.py nblock [model] [batch]   ->  ms per forward with the switch at each setting."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
import rajni_amd
from rajni_amd import timm_shaped as ts, _native as nat

SWITCHES = {"nblock": (lambda v: nat.lib().rajni_debug_set_gemm_nblock_bytes(v), [0, 1600 * 1024]),
            "nblockscan": (lambda v: nat.lib().rajni_debug_set_gemm_nblock_bytes(v), [1600 * 1024, 800 * 1024, 1200 * 1024, 2400 * 1024, 3200 * 1024]),
            "stagger": (lambda v: nat.lib().rajni_debug_set_resid_stagger(v), [1, 0, 2])}
name = sys.argv[1] if len(sys.argv) > 1 else "nblock"
model_name = sys.argv[2] if len(sys.argv) > 2 else "vit_base_patch16_224"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
setter, values = SWITCHES[name]
sched = {3: {"keep_ratio": 0.88, "update": True}, 4: {"keep_ratio": 0.88, "update": True},
         7: {"keep_ratio": 0.80, "update": True}, 8: {"keep_ratio": 0.72, "update": True}}
cfg = ts.CONFIGS[model_name]
m = rajni_amd.RAJNIViTWrapper(ts.create_model(cfg, seed=0).to(torch.bfloat16).cuda(), sched).eval()
x = torch.randn(B, 3, cfg.img_size, cfg.img_size, device="cuda").to(torch.bfloat16)
for _ in range(3):
    m(x)
res = {v: [] for v in values}
for r in range(5):
    for v in values:
        setter(v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            m(x)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 10 * 1e3)
setter(values[0] if name == 'nblockscan' else values[-1])
for v in values:
    print(f"{name}={v}: min {min(res[v]):.3f} ms  median {sorted(res[v])[len(res[v]) // 2]:.3f} ms  -> {B / min(res[v]) * 1e3:.0f} img/s")
