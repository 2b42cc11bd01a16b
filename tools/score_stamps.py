#!/usr/bin/env python3
"""Phase anatomy of score_select_kernel from s_memtime stamps (needs a library built with -DRAJNI_SS_STAMPS):
K pass | softmax stats + A_cls | V pass | mean + norms | std + scores | rank + compact, as shares of a
workgroup's run time (median over workgroups)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat

for (B, N, H, keep) in [(256, 197, 12, 172), (256, 121, 12, 86), (64, 577, 16, 403)]:
    qkv = torch.randn(B, N, 3 * H * 64, device="cuda").to(torch.bfloat16)
    buf = torch.zeros(B * 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.score_select(qkv, H, keep)
    nat.lib().rajni_debug_set_gemm_stamps(buf.data_ptr())
    ops.score_select(qkv, H, keep)
    torch.cuda.synchronize()
    nat.lib().rajni_debug_set_gemm_stamps(None)
    st = buf.view(B, 16)[:, :7].double().cpu()
    d = st[:, 1:] - st[:, :-1]
    names = ["K pass", "softmax+A_cls", "V pass", "mean+norms", "std+scores", "rank+compact"]
    med = d.median(dim=0).values
    tot = float(med.sum())
    print(f"B={B} N={N} H={H}: " + "  ".join(f"{n} {100 * m / tot:.0f}%" for n, m in zip(names, med.tolist())))
