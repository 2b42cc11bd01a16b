"""N-block size of the persistent GEMM tile order (rajni_debug_set_gemm_nblock_bytes) on the ViT-B GEMM shapes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat
dev = "cuda"
shapes = []
for tok in (197, 152, 87):
    M = tok * 256
    shapes += [(f"qkv_{tok}", M, 2304, 768, nat.EPI_BIAS), (f"fc1_{tok}", M, 3072, 768, nat.EPI_BIAS_GELU),
               (f"proj_{tok}", M, 768, 768, nat.EPI_BIAS_RESID), (f"fc2_{tok}", M, 768, 3072, nat.EPI_BIAS_RESID)]
vals = [0, -1, -2, -3, -4, -6]
for name, M, N, K, epi in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(N, device=dev)
    resid = torch.randn(1, M, N, device=dev) if epi == nat.EPI_BIAS_RESID else None
    t = {v: [] for v in vals}
    for r in range(5):
        for v in vals:
            nat.lib().rajni_debug_set_gemm_nblock_bytes(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(x.view(1, M, K), w, N, b, epi, resid=resid)
            e1.record(); torch.cuda.synchronize()
            if r: t[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"{name:10s} " + "  ".join(f"nb{-v if v else 'off'}:{min(t[v]):6.1f}" for v in vals), flush=True)
