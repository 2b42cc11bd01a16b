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
    t = t[(t[:, 1] > 0) & (t[:, 2] > 0)]
    # the kernel records two s_memtime values per (image, head): after the S^T MFMAs and after the P.V MFMAs
    print(f"N={N}: items with stamps {len(t)} | median cycles softmax + P.V phase (S^T done -> O done): "
          f"{np.median(t[:, 2] - t[:, 1]):.0f}")
