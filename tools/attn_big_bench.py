"""attention timing beyond 256 tokens (ViT-L/16 @384 stages: the chunked online-softmax kernel), GPU box only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch
from rajni_amd import ops
def t(f):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
B, H = 64, 16
for N in (577, 404, 300):
    qkv = torch.randn(B, N, 3 * H * 64, device="cuda").to(torch.bfloat16)
    us = t(lambda: ops.attention(qkv, None, H, 0.125))
    print(f"N={N}: {us:.1f} us  {4.0 * N * N * H * 64 * B / us / 1e6:.0f} TFLOP/s", flush=True)
