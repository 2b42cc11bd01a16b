"""LayerNorm kernel alone on the forward's row counts: microseconds and TB/s (fp32 stream in, bf16 out)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch
from rajni_amd import ops
for rows in (50432, 38912, 22272):
    x = torch.randn(rows, 768, device="cuda"); w = torch.ones(768, device="cuda"); b = torch.zeros(768, device="cuda")
    for _ in range(3): ops.layernorm(x, w, b, 1e-6)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.layernorm(x, w, b, 1e-6)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f"rows {rows}: {us:.1f} us  {rows * 768 * 6 / us / 1e6:.2f} TB/s", flush=True)
