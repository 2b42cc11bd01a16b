#!/usr/bin/env python3
"""rajni_linear (EPI_BIAS, bf16) on large square shapes next to torch.matmul (hipBLASLt) on the same box:
where the GEMM main loop stands apart from the ViT shapes' short K and tile-count effects (GPU box only).
Tiling 6 is the four-wave experiment: it falls back to tiling 4 unless the library was built with
-DRAJNI_GEMM_WIDE4."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat

dev = "cuda"
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (50432, 2304, 768), (50432, 2304, 3072), (50432, 768, 3072)]
for M, N, K in shapes:
    x = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    wp = ops.pack_weight(w)
    b = torch.zeros(N, device=dev)
    fl = 2.0 * M * N * K
    def timed(fn, n=10):
        best = 1e9
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if r:
                best = min(best, e0.elapsed_time(e1) / n)
        return fl / (best * 1e-3) / 1e12
    out = {}
    for m in (4, 5, 6):
        nat.lib().rajni_debug_force_gemm_tiling(m)
        out[f"rajni tiling {m}"] = round(timed(lambda: ops.linear(x.view(1, M, K), wp, N, b, nat.EPI_BIAS)))
    y6 = ops.linear(x.view(1, M, K), wp, N, b, nat.EPI_BIAS).float()   # tiling 6 still forced: check it
    ref = torch.matmul(x, w.t()).float()
    out["max|tiling6 - torch|/scale"] = float((y6.view(M, N) - ref).abs().max() / ref.abs().max())
    nat.lib().rajni_debug_force_gemm_tiling(0)
    out["torch.matmul"] = round(timed(lambda: torch.matmul(x, w.t())))
    print(M, N, K, out, flush=True)
