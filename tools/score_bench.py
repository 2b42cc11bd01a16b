#!/usr/bin/env python3
"""score + select kernel timing on the forward's stage shapes: fused, importance only, selection only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

from rajni_amd import _native as nat
two = "--two-pass" in sys.argv
nat.lib().rajni_debug_force_score_two_pass(1 if two else 0)
print("layout:", "two-pass (forced)" if two else "one-pass where it fits")
for (B, N, H, keep) in [(256, 197, 12, 172), (256, 173, 12, 151), (256, 152, 12, 120), (256, 121, 12, 86), (64, 577, 16, 403)]:
    qkv = torch.randn(B, N, 3 * H * 64, device="cuda").to(torch.bfloat16)
    sc = ops.importance(qkv, H)
    print(f"B={B} N={N} H={H}: fused {t(lambda: ops.score_select(qkv, H, keep)):6.1f} us   "
          f"importance {t(lambda: ops.importance(qkv, H)):6.1f} us   select {t(lambda: ops.select_topk(sc, keep)):6.1f} us", flush=True)
