#!/usr/bin/env python3
"""Where does the proj GEMM (K=768, N=768, RESID epilogue) spend its time?  Variants: no residual
(bias only), bf16 residual stream, fp32 residual stream, gathered rows."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat
dev = "cuda"
B, N, C = 256, 197, 768
M = B * N
x = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
w = ops.pack_weight((torch.randn(C, C, device=dev) * 0.05).to(torch.bfloat16))
b = torch.randn(C, device=dev)
r32 = torch.randn(B, N, C, device=dev)
r16 = r32.to(torch.bfloat16)
idx = torch.arange(N, device=dev, dtype=torch.int32).repeat(B, 1)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mode in (1, 5, 4):
    nat.lib().rajni_debug_force_gemm_tiling(mode)
    print("tiling", mode,
          "bias only %.1f us" % t(lambda: ops.linear(x, w, C, b, nat.EPI_BIAS)),
          "| resid bf16 %.1f" % t(lambda: ops.linear(x, w, C, b, nat.EPI_BIAS_RESID, resid=r16)),
          "| resid fp32 %.1f" % t(lambda: ops.linear(x, w, C, b, nat.EPI_BIAS_RESID, resid=r32)),
          "| resid fp32 gathered %.1f" % t(lambda: ops.linear(x, w, C, b, nat.EPI_BIAS_RESID, resid=r32, r_idx=idx)))
nat.lib().rajni_debug_force_gemm_tiling(0)
