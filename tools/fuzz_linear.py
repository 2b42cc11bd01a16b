#!/usr/bin/env python3
"""Randomised differential test of rajni_linear against torch fp32 matmul on the GPU: random shapes around the
tiling thresholds (ragged row/column tiles, K from 64 to 4096), every epilogue, gathered / in-place residuals,
fp32 and bf16 residual streams, fp8 weights.  python tools/fuzz_linear.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import numpy as np
import torch
from rajni_amd import ops, _native as nat

def run(cases=200, seed=0, verbose=True):
    """returns the number of failing cases"""
    rng = np.random.default_rng(seed)
    dev = "cuda"
    bad = 0
    for it in range(cases):
        M = int(rng.choice([rng.integers(1, 300), rng.integers(250, 1100), rng.integers(1000, 9000), rng.integers(9000, 70000)]))
        N = int(rng.choice([8 * rng.integers(1, 40), 8 * rng.integers(90, 100), 8 * rng.integers(180, 200), 8 * rng.integers(280, 400), 768, 2304, 3072]))
        K = 64 * int(rng.choice([1, 2, 3, 4, 5, 6, 9, 12, 16, 33, 48, 64]))
        if M * (N + K) > 3.0e8:
            M = int(3.0e8 / (N + K))
        epi = int(rng.choice([nat.EPI_BIAS, nat.EPI_BIAS_GELU, nat.EPI_BIAS_RESID]))
        tiling = int(rng.choice([0, 0, 1, 4, 5]))           # 0 = the dispatcher's choice, else forced
        nat.lib().rajni_debug_force_gemm_tiling(tiling)
        nat.lib().rajni_debug_set_gemm_nblock_bytes(int(rng.choice([1600 * 1024, 1600 * 1024, 0, -1, -2, -3])))
        fp8 = bool(rng.random() < 0.3)
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * (1.0 / K ** 0.5)).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16).float() if rng.random() < 0.8 else None
        wp, wsc = ops.pack_weight_fp8(w, torch.bfloat16, dev) if fp8 else (ops.pack_weight(w), None)
        wref = (ops.dequantize_fp8(wp, wsc) if fp8 else w.float())
        lin = x.float() @ wref.T
        if b is not None:
            lin = lin + b
        kw, desc, tol = {}, "", 1e-2
        if epi == nat.EPI_BIAS_GELU:
            lin = torch.nn.functional.gelu(lin)
        if epi == nat.EPI_BIAS_RESID:
            f32 = bool(rng.random() < 0.7)
            gam = torch.randn(N, device=dev).to(torch.bfloat16).float() if rng.random() < 0.4 else None
            if rng.random() < 0.5 and M >= 4:      # gathered residual rows
                Bn = int(rng.integers(1, min(M, 64) + 1)); Np = M // Bn; Mg = Bn * Np
                x, lin = x[:Mg], lin[:Mg]; M = Mg
                Nsrc = Np + int(rng.integers(0, 30))
                resid = torch.randn(Bn, Nsrc, N, device=dev)
                idx = torch.stack([torch.randperm(Nsrc, device=dev)[:Np].sort().values for _ in range(Bn)]).to(torch.int32)
                r = torch.gather(resid, 1, idx.long()[:, :, None].expand(-1, -1, N)).reshape(M, N)
                resid_in = resid if f32 else resid.to(torch.bfloat16)
                if not f32: r = resid_in.float().reshape(Bn, Nsrc, N).gather(1, idx.long()[:, :, None].expand(-1, -1, N)).reshape(M, N)
                kw = dict(resid=resid_in, r_idx=idx, gamma=gam); xin = x.reshape(Bn, Np, K); desc = f"gather B={Bn} Np={Np}/{Nsrc}"
            else:
                resid = torch.randn(1, M, N, device=dev)
                resid_in = resid if f32 else resid.to(torch.bfloat16)
                r = resid_in.float().reshape(M, N)
                kw = dict(resid=resid_in.clone(), gamma=gam); xin = x.reshape(1, M, K); desc = "resid"
            lin = r + (gam * lin if gam is not None else lin)
            tol = 2e-4 if f32 else 1e-2
            desc += " f32stream" if f32 else " bf16stream"
        else:
            xin = x.reshape(1, M, K)
        y = ops.linear(xin, wp, N, b, epi, w_scale=wsc, **kw).reshape(M, -1)[:, :N].float()
        err = float((y - lin).abs().max()); scale = float(lin.abs().max()) + 1e-6
        ok = err <= tol * scale and bool(torch.isfinite(y).all())
        bad += not ok
        if (not ok or it % 25 == 0) and verbose:
            print(f"[{it}] M={M} N={N} K={K} epi={epi} tiling={tiling} fp8={fp8} {desc}: err {err:.3g} / scale {scale:.3g} {'ok' if ok else 'FAIL'}", flush=True)

    nat.lib().rajni_debug_force_gemm_tiling(0)
    nat.lib().rajni_debug_set_gemm_nblock_bytes(1600 * 1024)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    failures = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{n} cases, {failures} failures")
    sys.exit(1 if failures else 0)
