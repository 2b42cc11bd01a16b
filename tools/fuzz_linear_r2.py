#!/usr/bin/env python3
"""Randomised differential test of the round-2 forms of rajni_linear against torch fp32 on the GPU:
  * fp8 x fp8 (x_scale / y_scale): random shapes (ragged row / column tiles, M from 1 up, K in multiples of 256),
    every epilogue, both tilings, gathered / in-place fp32 and bf16 residual streams;
  * the LayerNorm fold: producer (bf16 copy + block statistics, checked through rajni_ln_stats) and consumer
    (rstd * (x W'^T - mean * colsum) + b') on random streams, every tiling.
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
        if it % 3 != 2:     # ---- fp8 x fp8
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
        else:               # ---- LN fold: producer then consumer
            Cc = 64 * int(rng.choice([2, 3, 6, 12, 16]))
            K = 64 * int(rng.choice([2, 4, 12, 48]))
            til = int(rng.choice([0, 1, 4, 5]))
            nat.lib().rajni_debug_force_gemm_tiling(til)
            x = torch.randn(1, M, K, device=dev, generator=gen).to(torch.bfloat16)
            w = (torch.randn(Cc, K, device=dev, generator=gen) / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(Cc, device=dev, generator=gen)
            resid = torch.randn(1, M, Cc, device=dev, generator=gen) * 2 + torch.randn(1, M, 1, device=dev, generator=gen) * float(rng.choice([0.0, 1.0, 20.0]))
            copy = torch.empty(M, Cc, dtype=torch.bfloat16, device=dev)
            part = torch.empty(M, Cc // 64, 2, device=dev)
            y = ops.linear(x, ops.pack_weight(w), Cc, b, nat.EPI_BIAS_RESID, resid=resid, y_bf16_copy=copy, y_rowstat_partials=part).reshape(M, Cc)
            want = resid.reshape(M, Cc) + x.reshape(M, K).float() @ w.float().T + b
            st = ops.ln_stats(part, 1e-6)
            var = y.var(dim=1, unbiased=False)
            ok = (float((y - want).abs().max()) <= 3e-4 * float(want.abs().max()) and torch.equal(copy, y.to(torch.bfloat16))
                  and torch.allclose(st[:, 0], y.mean(dim=1), rtol=1e-4, atol=1e-4 * float(var.sqrt().max()))
                  and torch.allclose(st[:, 1], (var + 1e-6).rsqrt(), rtol=1e-4))
            # consumer on that stream
            lw, lb = torch.rand(Cc, device=dev, generator=gen) + 0.5, torch.randn(Cc, device=dev, generator=gen) * 0.1
            w2 = (torch.randn(N, Cc, device=dev, generator=gen) / Cc ** 0.5).to(torch.bfloat16)
            b2 = torch.randn(N, device=dev, generator=gen).to(torch.bfloat16).float()
            epi = int(rng.choice([nat.EPI_BIAS, nat.EPI_BIAS_GELU]))
            wf, bf, cs = ops.fold_layernorm(w2, b2, lw, lb, torch.bfloat16, dev)
            got = ops.linear(copy.view(1, M, Cc), wf, N, bf, epi, x_rowstats=st, w_colsum=cs).reshape(M, -1)[:, :N].float()
            ln = torch.nn.functional.layer_norm(y, (Cc,), lw.to(torch.bfloat16).float(), lb.to(torch.bfloat16).float(), 1e-6)
            ref = ln @ w2.float().T + b2
            if epi == nat.EPI_BIAS_GELU:
                ref = torch.nn.functional.gelu(ref)
            # tokens sit up to 10 std from zero here: the fold's noise law (sqrt(1 + r^2) x the kernel path's 3e-3)
            rmax = float((y.mean(dim=1).abs() * (var + 1e-6).rsqrt()).max())
            err, scale = float((got - ref).abs().max()), float(ref.abs().max()) + 1e-6
            ok = ok and err <= (6e-3 * (1 + rmax ** 2) ** 0.5 + 4e-3) * scale
            what = f"ln-fold M={M} C={Cc} K={K} N={N} epi={epi} tiling={til} max|mean|/std={rmax:.1f}"
            nat.lib().rajni_debug_force_gemm_tiling(0)
        bad += not ok
        if (not ok or it % 20 == 0) and verbose:
            print(f"[{it}] {what}: err {err:.3g} / scale {scale:.3g} {'ok' if ok else 'FAIL'}", flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    failures = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{n} cases, {failures} failures")
    sys.exit(1 if failures else 0)
