#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of the persistent attention kernel (build -DRAJNI_ATTN_STAMPS)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import numpy as np, torch
from rajni_amd import ops, _native as nat
dev = "cuda"
B, H = 256, 12
for N in (197, 152, 87):
    qkv = torch.randn(B, N, 3 * H * 64, device=dev).to(torch.bfloat16)
    st = torch.zeros(B * H * 8, dtype=torch.int64, device=dev)
    for _ in range(3): ops.attention(qkv, None, H, 0.125)
    nat.lib().rajni_debug_set_gemm_stamps(st.data_ptr())
    ops.attention(qkv, None, H, 0.125)
    torch.cuda.synchronize()
    nat.lib().rajni_debug_set_gemm_stamps(None)
    t = st.cpu().numpy().reshape(-1, 8).astype(np.float64)
    t = t[(t[:, 1] > 0) & (t[:, 2] > 0) & (t[:, 0] > 0)]
    # wave 0 of the persistent kernel, per (image, head): [0] arrives at the per-item barrier, [3] released,
    # [4] prefetch (Q, indices, K/V DMA of the next item) issued, [1] S^T MFMAs done, [2] softmax + P.V done, [5] stores issued
    med = lambda x: float(np.median(x))
    print(f"N={N}: {len(t)} items | median cycles: barrier wait {med(t[:,3]-t[:,0]):.0f} | prefetch issue {med(t[:,4]-t[:,3]):.0f} | "
          f"S^T {med(t[:,1]-t[:,4]):.0f} | softmax+PV {med(t[:,2]-t[:,1]):.0f} | scale+stores {med(t[:,5]-t[:,2]):.0f} | "
          f"item total (release -> done) {med(t[:,5]-t[:,3]):.0f}")
