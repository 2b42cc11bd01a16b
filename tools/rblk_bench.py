#!/usr/bin/env python3
"""Row super-blocks of the persistent tile order (rajni_debug_set_gemm_row_superblock) x N-block size on the ViT-B
GEMM shapes: microseconds per launch (GPU box only).  Correctness of every order is checked against order 0."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat
dev = "cuda"
shapes = []
for tok in (197, 152, 87):
    M = tok * 256
    shapes += [(f"qkv_{tok}", M, 2304, 768, nat.EPI_BIAS), (f"fc1_{tok}", M, 3072, 768, nat.EPI_BIAS_GELU),
               (f"fc2_{tok}", M, 768, 3072, nat.EPI_BIAS_RESID)]
rvals = [0, 4, 6, 8, 12, 16, 32]
nvals = [1600 * 1024, -2, -3]
for name, M, N, K, epi in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(N, device=dev)
    resid = torch.randn(1, M, N, device=dev) if epi == nat.EPI_BIAS_RESID else None
    nat.lib().rajni_debug_set_gemm_row_superblock(0); nat.lib().rajni_debug_set_gemm_nblock_bytes(1600 * 1024)
    ref = ops.linear(x.view(1, M, K), w, N, b, epi, resid=resid).clone()
    for nv in nvals:
        nat.lib().rajni_debug_set_gemm_nblock_bytes(nv)
        t = {v: [] for v in rvals}
        for r in range(4):
            for v in rvals:
                nat.lib().rajni_debug_set_gemm_row_superblock(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    y = ops.linear(x.view(1, M, K), w, N, b, epi, resid=resid)
                e1.record(); torch.cuda.synchronize()
                if r: t[v].append(e0.elapsed_time(e1) / 5 * 1e3)
                elif not torch.equal(y, ref): print("MISMATCH", name, nv, v)
        print(f"{name:9s} nblk={'auto' if nv > 0 else -nv}: " + "  ".join(f"rb{v}:{min(t[v]):6.1f}" for v in rvals), flush=True)
nat.lib().rajni_debug_set_gemm_row_superblock(0); nat.lib().rajni_debug_set_gemm_nblock_bytes(1600 * 1024)
