#!/usr/bin/env python3
"""Experiment: do the residual-epilogue GEMMs (proj, fc2) gain from de-synchronising their epilogues' HBM bursts?
Every other workgroup of an XCD sleeps `units` x 8192 cycles before its first tile (rajni_debug_set_resid_stagger).
Interleaved rounds in one process, HIP-event timing, fp32 residual stream in place."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat
dev = "cuda"
units = [0, 1, 2, 3, 4, 6, 8]
for name, M, N, K in [("proj_197", 50432, 768, 768), ("proj_152", 38912, 768, 768), ("fc2_197", 50432, 768, 3072),
                      ("fc2_152", 38912, 768, 3072), ("fc2_87", 22272, 768, 3072)]:
    x = torch.randn(1, M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(N, device=dev)
    resid = torch.randn(1, M, N, device=dev)
    best = {u: 1e9 for u in units}
    for r in range(6):
        for u in units:
            nat.lib().rajni_debug_set_resid_stagger(u)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(x, w, N, b, nat.EPI_BIAS_RESID, resid=resid, out=resid.view(M, N))
            e1.record(); torch.cuda.synchronize()
            if r: best[u] = min(best[u], e0.elapsed_time(e1) / 5 * 1e3)
    nat.lib().rajni_debug_set_resid_stagger(1)
    print(name, "  ".join(f"{u}: {t:.1f}us" for u, t in best.items()), flush=True)
