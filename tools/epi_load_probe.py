#!/usr/bin/env python3
"""Is the fp32-stream residual epilogue bound by the chip's HBM bandwidth (all CUs bursting at once) or by one CU's own
load / store path?  Stamps of the fp8 x fp8 fc2-shaped launch (256x128 tiles, -DRAJNI_GEMM_STAMPS build) with 30, 120, 252
workgroups of ONE tile each against the full 4-round launch: epilogue cycles per tile by number of CUs bursting together."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import numpy as np, torch
from rajni_amd import ops, _native as nat
dev = "cuda"
nat.lib().rajni_debug_force_f8_tiling(1)
N, K = 768, 3072
for f32 in (True, False):
    for M in (1280, 5120, 10752, 44288):
        xq = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev)
        wq = torch.randint(0, 120, (768, K), dtype=torch.uint8, device=dev)
        xs, ws = torch.rand(M, device=dev) / 64 + 0.01, torch.rand(N, device=dev) / 64 + 0.01
        b = torch.randn(N, device=dev)
        resid = torch.randn(1, M, N, device=dev)
        resid = resid if f32 else resid.to(torch.bfloat16)
        ntile = ((M + 255) // 256) * 6
        st = torch.zeros(ntile * 4, dtype=torch.int64, device=dev)
        run = lambda: ops.linear(xq.view(1, M, K), wq, N, b, nat.EPI_BIAS_RESID, resid=resid, w_scale=ws, x_scale=xs)
        for _ in range(3):
            run()
        nat.lib().rajni_debug_set_gemm_stamps(st.data_ptr())
        run(); torch.cuda.synchronize()
        nat.lib().rajni_debug_set_gemm_stamps(None)
        t = st.cpu().numpy().reshape(ntile, 4).astype(np.float64)
        # one-tile workgroups write no stamp record (`more` is false after their only tile): take ts from the tiles that did
        t = t[t[:, 2] > 0]
        if len(t) == 0:
            print(f"stream {'fp32' if f32 else 'bf16'} M={M}: no stamped tiles"); continue
        print(f"stream {'fp32' if f32 else 'bf16'} M={M} tiles={ntile}: K loop {np.median(t[:, 1] - t[:, 0]):.0f}  epilogue {np.median(t[:, 2] - t[:, 1]):.0f} "
              f"(p10 {np.percentile(t[:, 2] - t[:, 1], 10):.0f} p90 {np.percentile(t[:, 2] - t[:, 1], 90):.0f}) cycles, {len(t)} stamped tiles", flush=True)
