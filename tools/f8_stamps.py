#!/usr/bin/env python3
"""Diagnostic: per-tile s_memtime stamps of the fp8 x fp8 256x128 persistent GEMM (a -DRAJNI_GEMM_STAMPS build loaded through
RAJNI_HIP_LIB): where a tile's cycles go - K loop, epilogue, the exposed fragment read - on the ViT-B shapes of configs[4].
    python rajni-vit_amd/build.py is NOT what builds it: see tools/README.md (build.build(out=..., extra=["-DRAJNI_GEMM_STAMPS"]))"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import numpy as np, torch
from rajni_amd import ops, _native as nat
dev = "cuda"


def codes(shape):
    b = torch.randint(0, 256, shape, dtype=torch.uint8, device=dev)
    b[(b & 0x7F) == 0x7F] = 0x38
    return b


for til, name, M, N, K, epi, f32 in [(2, "qkv 256x256", 50432, 2304, 768, nat.EPI_BIAS, False), (2, "fc1 256x256", 44288, 3072, 768, nat.EPI_BIAS_GELU, False),
                                     (2, "K=3072 bias 256x256", 44288, 768, 3072, nat.EPI_BIAS, False)] + [(1,) + t for t in [("qkv", 50432, 2304, 768, nat.EPI_BIAS, False), ("fc1", 44288, 3072, 768, nat.EPI_BIAS_GELU, False),
                                ("fc2 fp32 stream", 44288, 768, 3072, nat.EPI_BIAS_RESID, True),
                                ("fc2 bf16 stream", 44288, 768, 3072, nat.EPI_BIAS_RESID, False),
                                ("proj-shaped fp32 stream", 44288, 768, 768, nat.EPI_BIAS_RESID, True)]]:
    nat.lib().rajni_debug_force_f8_tiling(til)
    bn = 256 if til == 2 else 128
    xq, wq = codes((M, K)), codes(((N + 255) // 256 * 256, K))
    xs, ws = torch.rand(M, device=dev) / 64 + 0.01, torch.rand(N, device=dev) / 64 + 0.01
    b = torch.randn(N, device=dev)
    ys = torch.rand(M, device=dev) + 0.5 if epi == nat.EPI_BIAS_GELU else None
    resid = None
    if epi == nat.EPI_BIAS_RESID:
        resid = torch.randn(1, M, N, device=dev)
        resid = resid if f32 else resid.to(torch.bfloat16)
    ntile = ((M + 255) // 256) * ((N + bn - 1) // bn)
    st = torch.zeros(ntile * 4, dtype=torch.int64, device=dev)
    run = lambda: ops.linear(xq.view(1, M, K), wq, N, b, epi, resid=resid, w_scale=ws, x_scale=xs, y_scale=ys)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    nat.lib().rajni_debug_set_gemm_stamps(st.data_ptr())
    run()
    torch.cuda.synchronize()
    nat.lib().rajni_debug_set_gemm_stamps(None)
    t = st.cpu().numpy().reshape(ntile, 4).astype(np.float64)
    t = t[t[:, 3] > 0]                      # (a workgroup's LAST tile writes no stamps past the epilogue: `more` is false)
    main, epi_t, rd, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    nk = K // 128
    print(f"{name}: {M}x{N}x{K} {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TF | tiles {ntile} ({ntile / 256:.2f} rounds) nk={nk} | "
          f"median cycles: K loop {np.median(main):.0f} ({np.median(main) / nk:.0f}/step; MFMA-bound {8 * bn})  epilogue {np.median(epi_t):.0f}  "
          f"fragment read {np.median(rd):.0f}  tile {np.median(tot):.0f} | p10/p90 K loop {np.percentile(main, 10):.0f}/{np.percentile(main, 90):.0f} "
          f"epilogue {np.percentile(epi_t, 10):.0f}/{np.percentile(epi_t, 90):.0f}", flush=True)
