#!/usr/bin/env python3
"""Randomised differential test of the round-2 forms of rajni_linear against torch fp32 on the GPU:
  * fp8 x fp8 (x_scale / y_scale): random shapes (ragged row / column tiles, M from 1 up, K in multiples of 256),
    every epilogue, both tilings, gathered / in-place fp32 and bf16 residual streams.
python tools/fuzz_linear_r2.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import numpy as np
import torch
from rajni_amd import ops, _native as nat

dev = "cuda"


def codes(shape, gen):
    b = torch.randint(0, 256, shape, dtype=torch.uint8, device=dev, generator=gen)
    b[(b & 0x7F) == 0x7F] = 0x38
    return b


def deq(q, s):
    return q.view(torch.float8_e4m3fn).to(torch.float32) * s[:, None]


def run(cases=120, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=dev).manual_seed(seed)
    bad = 0
    for it in range(cases):
        M = int(rng.choice([rng.integers(1, 300), rng.integers(250, 1100), rng.integers(1000, 9000), rng.integers(9000, 40000)]))
        N = int(rng.choice([16 * rng.integers(1, 20), 16 * rng.integers(45, 50), 64 * rng.integers(3, 50), 768, 2304, 3072]))
        if True:            # ---- fp8 x fp8 (the LayerNorm-fold cases left with that code in round 3: DESIGN.md section 10)
            K = 256 * int(rng.choice([2, 3, 4, 6, 12, 16]))
            epi = int(rng.choice([nat.EPI_BIAS, nat.EPI_BIAS_GELU, nat.EPI_BIAS_RESID]))
            til = int(rng.choice([0, 1, 2]))
            nat.lib().rajni_debug_force_f8_tiling(til)
            xq, wq = codes((M, K), gen), codes(((N + 255) // 256 * 256, K), gen)
            xs = (torch.rand(M, device=dev, generator=gen) + 0.5) / (8 * K ** 0.5)
            ws = (torch.rand(N, device=dev, generator=gen) + 0.5) / 64
            b = torch.randn(N, device=dev, generator=gen) if rng.random() < 0.8 else None
            lin = deq(xq, xs) @ deq(wq[:N], ws).T
            if b is not None:
                lin = lin + b
            kw, tol, desc = {}, 1e-2, ""
            xin = xq.view(1, M, K)
            if epi == nat.EPI_BIAS_GELU:
                lin = torch.nn.functional.gelu(lin)
                ys = (lin.abs().amax(dim=1) * float(rng.uniform(1, 6)) / 448 + 1e-6).float()
                kw["y_scale"] = ys
            elif epi == nat.EPI_BIAS_RESID:
                f32 = bool(rng.random() < 0.7)
                gam = torch.randn(N, device=dev, generator=gen) if rng.random() < 0.5 else None
                if rng.random() < 0.5 and M >= 4:
                    Bn = int(rng.integers(1, min(M, 48) + 1)); Np = M // Bn; M2 = Bn * Np
                    xq, xs, lin, M = xq[:M2], xs[:M2], lin[:M2], M2
                    Nsrc = Np + int(rng.integers(0, 20))
                    resid = torch.randn(Bn, Nsrc, N, device=dev, generator=gen)
                    resid = resid if f32 else resid.to(torch.bfloat16)
                    idx = torch.stack([torch.randperm(Nsrc, device=dev)[:Np].sort().values for _ in range(Bn)]).to(torch.int32)
                    r = resid.float().gather(1, idx.long()[:, :, None].expand(-1, -1, N)).reshape(M, N)
                    kw = dict(resid=resid, r_idx=idx, gamma=gam); xin = xq.view(Bn, Np, K); desc = "gather"
                else:
                    resid = torch.randn(1, M, N, device=dev, generator=gen)
                    resid = resid if f32 else resid.to(torch.bfloat16)
                    r = resid.float().reshape(M, N)
                    kw = dict(resid=resid.clone(), gamma=gam); xin = xq.view(1, M, K); desc = "resid"
                lin = r + (gam * lin if gam is not None else lin)
                tol = 3e-4 if f32 else 1e-2
            y = ops.linear(xin, wq, N, b, epi, w_scale=ws, x_scale=xs, **kw).reshape(M, -1)[:, :N]
            if epi == nat.EPI_BIAS_GELU:      # e4m3 output: half an ulp is 2^-4 relative, 2^-10 * scale below the normal range
                yd = deq(y.contiguous(), ys)
                bound = torch.maximum(lin.abs() * 2.0 ** -4, ys[:, None] * 2.0 ** -10) * 1.02 + 3e-4 * lin.abs().max()
                ok = bool(((yd - lin).abs() <= bound).all())
                err, scale = float((yd - lin).abs().max()), float(lin.abs().max())
            else:
                err, scale = float((y.float() - lin).abs().max()), float(lin.abs().max()) + 1e-6
                ok = err <= tol * scale and bool(torch.isfinite(y.float()).all())
            what = f"f8xf8 M={M} N={N} K={K} epi={epi} tiling={til} {desc}"
            nat.lib().rajni_debug_force_f8_tiling(0)
        bad += not ok
        if (not ok or it % 20 == 0) and verbose:
            print(f"[{it}] {what}: err {err:.3g} / scale {scale:.3g} {'ok' if ok else 'FAIL'}", flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    failures = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{n} cases, {failures} failures")
    sys.exit(1 if failures else 0)
