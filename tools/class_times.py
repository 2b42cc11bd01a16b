#!/usr/bin/env python3
"""us per forward by kernel class (HIP events on the launch stream, rajni_profile_*) for any model / batch / schedule / format:
    python tools/class_times.py vit_large_patch16_384 64 '{"4":{"keep_ratio":0.7},"12":{"keep_ratio":0.5},"20":{"keep_ratio":0.3}}' [model|fp8|fp8_mfma] [fp32|bf16]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch, rajni_amd
from rajni_amd import timm_shaped as ts, _native as nat
name = sys.argv[1] if len(sys.argv) > 1 else "vit_base_patch16_224"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sched = {3: {"keep_ratio": 0.88}, 4: {"keep_ratio": 0.88}, 7: {"keep_ratio": 0.80}, 8: {"keep_ratio": 0.72}}
if len(sys.argv) > 3 and sys.argv[3] not in ("", "readme"):
    sched = {int(k): v for k, v in json.loads(sys.argv[3]).items()}
cfg = ts.CONFIGS[name]
m = rajni_amd.RAJNIViTWrapper(ts.create_model(cfg, seed=0).to(torch.bfloat16).cuda(), sched).eval()
m.set_weight_format(sys.argv[4] if len(sys.argv) > 4 else "model")
if len(sys.argv) > 5 and sys.argv[5] == "bf16":
    m.set_residual_dtype(torch.bfloat16)
x = torch.randn(B, 3, cfg.img_size, cfg.img_size, device="cuda").to(torch.bfloat16)
for _ in range(5): m(x)
ts_ = []
for r in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m(x)
    torch.cuda.synchronize(); ts_.append((time.perf_counter() - t0) / 5 * 1e3)
print(f"{name} batch {B}: {min(ts_):.3f} ms per forward = {B / min(ts_) * 1e3:.0f} img/s; token counts {m.get_last_stats()['token_counts']}")
nat.profile_reset(); nat.profile_enable(0xFFFF)
for _ in range(5): m(x)
torch.cuda.synchronize(); nat.profile_enable(0)
pr = nat.profile_collect()
tot = sum(v["ms"] for v in pr.values()) / 5 * 1e3
for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["ms"]):
    us = v["ms"] / 5 * 1e3
    extra = f"{v['flops'] / (v['ms'] * 1e-3) / 1e12:7.0f} TF" if v["flops"] else f"{v['bytes'] / (v['ms'] * 1e-3) / 1e9:7.0f} GB/s"
    print(f"  {k:36s} {v['launches'] // 5:4d} launches  {us:8.0f} us  {100 * us / tot:5.1f} %  avg {us / (v['launches'] / 5):7.1f} us  {extra}")
print(f"  sum of kernels {tot:.0f} us")
