#!/usr/bin/env python3
"""Residual GEMM launches (fc2, proj; fp32 stream) of ViT-B at batch 256, per token count and tiling, for several builds of
the library on one box:  python tools/resid_ab.py librajni_a.so librajni_b.so ...   (names under rajni_amd/lib; GPU box only).
One child process per library (the library is loaded once per process), two interleaved passes, min over repetitions."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(256 * t, 768, k) for k in (3072, 768) for t in (197, 173, 152, 121, 87)]


def child():
    sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
    import torch
    from rajni_amd import ops, _native as nat
    out = []
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = ops.pack_weight((torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16))
        b = torch.randn(N, device="cuda")
        resid = torch.randn(1, M, N, device="cuda")
        for til in (5, 4):
            nat.lib().rajni_debug_force_gemm_tiling(til)
            best = 1e9
            for r in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.linear(x.view(1, M, K), w, N, b, nat.EPI_BIAS_RESID, resid=resid)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
            out.append(best)
    print(" ".join(f"{v:.1f}" for v in out))


def main():
    libs = sys.argv[1:]
    res = {l: [] for l in libs}
    for _ in range(2):
        for l in libs:
            env = dict(os.environ, RAJNI_HIP_LIB=os.path.join(ROOT, "rajni-vit_amd", "rajni_amd", "lib", l), RAJNI_RESID_AB_CHILD="1")
            r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=300)
            line = [x for x in r.stdout.strip().split("\n") if x and x[0].isdigit()][-1]
            res[l].append([float(v) for v in line.split()])
    print("shape (M N K) tiling: " + "  ".join(libs))
    i = 0
    for M, N, K in SHAPES:
        for til in ("256x128", "256x256"):
            print(f"{M:6d} {N} {K:4d} {til}: " + "  ".join(f"{min(p[i] for p in res[l]):7.1f}" for l in libs))
            i += 1


if __name__ == "__main__":
    child() if os.environ.get("RAJNI_RESID_AB_CHILD") else main()
