#!/usr/bin/env python3
"""attention kernel timing on the forward's stage shapes (B=256, ViT-B heads)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops
B, H = 256, 12
for N, Np in ((197, 197), (197, 173), (173, 152), (152, 152), (152, 121), (121, 87), (87, 87)):
    qkv = torch.randn(B, N, 3 * H * 64, device="cuda").to(torch.bfloat16)
    idx = None
    if Np != N:
        idx = torch.stack([torch.cat([torch.zeros(1, dtype=torch.int64), 1 + torch.randperm(N - 1)[: Np - 1].sort().values]) for _ in range(B)]).to(torch.int32).cuda()
    f = lambda: ops.attention(qkv, idx, H, 0.125)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = B * (Np * 3 * H * 64 * 2 + Np * H * 64 * 2) / 1e6
    print(f"N={N} Np={Np}: {us:6.1f} us   {mb / us:6.2f} TB/s (q,k,v of kept rows + out)", flush=True)
