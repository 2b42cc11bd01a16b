#!/usr/bin/env python3
"""Residual GEMMs with K > N (fc2): 256x128 (tiling 5) vs 256x256 (tiling 4) per token count, for the dispatcher's
round-count rule (GPU box only).  Prints the measured times and what the rule would pick."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat
dev = "cuda"

def eff_rounds(tiles):      # full rounds + a partial round whose tiles run faster the emptier the chip is
    full, frac = tiles // 256, tiles / 256 - tiles // 256
    return full + (0.6 + 0.4 * frac if frac else 0.0)

def rule(M, N):
    tw = math.ceil(M / 256) * math.ceil(N / 256)
    tm = math.ceil(M / 256) * math.ceil(N / 128)
    return 4 if 1.83 * eff_rounds(tw) < eff_rounds(tm) else 5

cases = [(256 * t, 768, 3072) for t in (197, 173, 152, 121, 87)] + [(64 * t, 1024, 4096) for t in (577, 404, 202, 61)] + \
        [(512 * t, 768, 3072) for t in (197, 152, 87)] + [(128 * t, 768, 3072) for t in (197, 152, 87)] + \
        [(64 * t, 1280, 5120) for t in (257, 205, 143, 86)] + \
        [(256 * t, 768, 768) for t in (197, 173, 152, 121, 87)] + [(64 * t, 1024, 1024) for t in (577, 404, 202, 61)] + \
        [(512 * t, 768, 768) for t in (197, 152)]
agree = 0
for M, N, K in cases:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(N, device=dev); resid = torch.randn(1, M, N, device=dev)
    res = {}
    for r in range(4):
        for til in (5, 4):
            nat.lib().rajni_debug_force_gemm_tiling(til)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.linear(x.view(1, M, K), w, N, b, nat.EPI_BIAS_RESID, resid=resid)
            e1.record(); torch.cuda.synchronize()
            if r: res.setdefault(til, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    t5, t4 = min(res[5]), min(res[4])
    best, pick = (4 if t4 < t5 else 5), rule(M, N)
    agree += best == pick or abs(t4 - t5) / min(t4, t5) < 0.02
    print(f"M={M:6d} N={N} K={K}: 256x128 {t5:7.1f} us  256x256 {t4:7.1f} us  best {best}  rule {pick}  {'' if best == pick else ('(within 2 %)' if abs(t4-t5)/min(t4,t5) < 0.02 else 'MISS %.1f%%' % (100*abs(t4-t5)/min(t4,t5)))}", flush=True)
nat.lib().rajni_debug_force_gemm_tiling(0)
print(f"rule agrees (or is within 2 %) on {agree} of {len(cases)}")
