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
    full = buf.view(B, 16).double().cpu()
    r = full[:, [5, 7, 8, 9, 10, 6]]
    rd = (r[:, 1:] - r[:, :-1]).median(dim=0).values.tolist()
    print("   rank+compact detail (cycles): keys->LDS+sync %.0f | rank count loop %.0f | shuffles+ballot+sync %.0f | prefix+stores+sync %.0f | CLS slot %.0f" % tuple(rd))
    st = full[:, :7]
    d = st[:, 1:] - st[:, :-1]
    names = ["K pass", "softmax+A_cls", "V pass", "mean+norms", "std+scores", "rank+compact"]
    med = d.median(dim=0).values
    tot = float(med.sum())
    print(f"B={B} N={N} H={H}: " + "  ".join(f"{n} {100 * m / tot:.0f}% ({m:.0f})" for n, m in zip(names, med.tolist())) + f"  | total {tot:.0f} cycles; "
          f"workgroup start spread {float(st[:, 0].max() - st[:, 0].min()):.0f}, end spread {float(st[:, 6].max() - st[:, 6].min()):.0f}, "
          f"kernel span {float(st[:, 6].max() - st[:, 0].min()):.0f}")
