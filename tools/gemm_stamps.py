#!/usr/bin/env python3
"""Diagnostic: per-workgroup s_memtime stamps of the 256x256 GEMM (build with -DRAJNI_GEMM_STAMPS,
RAJNI_HIP_LIB=.../librajni_stamps.so).  Prints where a tile's time goes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import numpy as np, torch
from rajni_amd import ops, _native as nat
dev = "cuda"
for name, M, N, K, epi in [("qkv", 50432, 2304, 768, nat.EPI_BIAS), ("fc1", 50432, 3072, 768, nat.EPI_BIAS_GELU),
                           ("proj (256x128)", 50432, 768, 768, nat.EPI_BIAS_RESID), ("fc2 (256x128)", 50432, 768, 3072, nat.EPI_BIAS_RESID),
                           ("fc2 152 tokens (256x256)", 38912, 768, 3072, nat.EPI_BIAS_RESID)]:
    x = torch.randn(1, M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(N, device=dev)
    resid = torch.randn(1, M, N, device=dev) if epi == nat.EPI_BIAS_RESID else None
    wide = "256x128" not in name
    nblk = ((M + 255) // 256) * ((N + (255 if wide else 127)) // (256 if wide else 128))
    st = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    nat.lib().rajni_debug_force_gemm_tiling(4 if wide else 5)
    for _ in range(3):
        ops.linear(x, w, N, b, epi, resid=resid)
    nat.lib().rajni_debug_set_gemm_stamps(st.data_ptr())
    ops.linear(x, w, N, b, epi, resid=resid)
    torch.cuda.synchronize()
    nat.lib().rajni_debug_set_gemm_stamps(None)
    t = st.cpu().numpy().reshape(nblk, 4).astype(np.float64)
    pro, main, epi_t, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    span = t[:, 3].max() - t[:, 0].min()
    nk = K // 64
    print(f"{name}: blocks={nblk} nk={nk} | median cycles: prologue {np.median(pro):.0f}  main {np.median(main):.0f} "
          f"({np.median(main)/nk:.0f}/K-step; MFMA-bound = {2048 if wide else 1024})  epilogue {np.median(epi_t):.0f}  total {np.median(tot):.0f} "
          f"| kernel span {span:.0f} cyc | p10/p90 main {np.percentile(main,10):.0f}/{np.percentile(main,90):.0f}")
