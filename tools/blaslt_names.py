#!/usr/bin/env python3
"""Which hipBLASLt kernels torch.matmul picks on this box for a few bf16 shapes (run under rocprofv3
--kernel-trace --stats; the Tensile kernel names spell out macro tile, wave layout and prefetch depths)."""
import torch
dev = "cuda"
for M, N, K in [(8192, 8192, 8192), (4096, 4096, 4096), (50432, 768, 3072), (50432, 2304, 768)]:
    x = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    for _ in range(3):
        torch.matmul(x, w.t())
    torch.cuda.synchronize()
