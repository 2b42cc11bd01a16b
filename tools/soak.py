#!/usr/bin/env python3
"""Race screen (guide: an early LDS read behind an LDS-DMA passes whenever the DMA happens to land first): every
GEMM class, attention kernel, score+select and the whole forward are run many times on the same inputs, with the
chip kept busy, and every result must equal the first one BIT FOR BIT.  python tools/soak.py [repeats] [noise]
With "noise" a side stream keeps streaming 1 GiB copies and small matmuls while the screened kernel runs, so that
memory latencies (and with them LDS-DMA landing times) wander."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
import rajni_amd
from rajni_amd import ops, _native as nat, timm_shaped as ts

dev = "cuda"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 200
NOISE = len(sys.argv) > 2 and sys.argv[2] == "noise"
bad = 0
side = torch.cuda.Stream()
na = torch.empty(1 << 28, dtype=torch.float32, device=dev); nb = torch.empty_like(na)
nm = torch.randn(2048, 2048, device=dev).to(torch.bfloat16)

def noise_burst():
    with torch.cuda.stream(side):
        nb.copy_(na)
        for _ in range(4):
            nm @ nm

def screen(label, fn, reps):
    global bad
    first = fn()
    first = [t.clone() for t in (first if isinstance(first, (tuple, list)) else (first,)) if t is not None]
    n_bad = 0
    for it in range(reps):
        if NOISE and it % 8 == 0:
            noise_burst()
        out = fn()
        out = [t for t in (out if isinstance(out, (tuple, list)) else (out,)) if t is not None]
        n_bad += not all(torch.equal(a, b) for a, b in zip(first, out))
    torch.cuda.synchronize()
    bad += n_bad
    print(f"{label:60s} {reps} repeats, {n_bad} differing", flush=True)

t0 = time.time()
for M, N, K, epi, f32 in [(50432, 2304, 768, nat.EPI_BIAS, False), (44288, 3072, 768, nat.EPI_BIAS_GELU, False),
                          (38912, 768, 3072, nat.EPI_BIAS_RESID, True), (50432, 768, 3072, nat.EPI_BIAS_RESID, True),
                          (30976, 768, 768, nat.EPI_BIAS_RESID, True), (22272, 768, 768, nat.EPI_BIAS_RESID, False),
                          (25856, 1024, 4096, nat.EPI_BIAS_RESID, True), (7744, 2304, 768, nat.EPI_BIAS, False),
                          (300, 1000, 768, nat.EPI_BIAS, False)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(N, device=dev)
    resid = None
    if epi == nat.EPI_BIAS_RESID:
        resid = torch.randn(1, M, N, device=dev) if f32 else torch.randn(1, M, N, device=dev).to(torch.bfloat16)
    for fp8 in (False, True):
        if fp8:
            w8, sc = ops.pack_weight_fp8((torch.randn(N, K, device=dev) * 0.05), torch.bfloat16, dev)
            fn = lambda: ops.linear(x.view(1, M, K), w8, N, b, epi, resid=resid, w_scale=sc)
        else:
            fn = lambda: ops.linear(x.view(1, M, K), w, N, b, epi, resid=resid)
        screen(f"linear {M}x{N}x{K} epi={epi} stream_f32={f32} fp8={fp8}", fn, R)
for B, N, Np, H, D in [(256, 197, 197, 12, 64), (256, 197, 173, 12, 64), (256, 121, 87, 12, 64), (64, 577, 404, 16, 64), (64, 257, 205, 16, 80)]:
    qkv = torch.randn(B, N, 3 * H * D, device=dev).to(torch.bfloat16)
    idx = None
    if Np != N:
        idx = torch.stack([torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), 1 + torch.randperm(N - 1, device=dev)[: Np - 1].sort().values]) for _ in range(B)]).to(torch.int32)
    screen(f"attention B={B} N={N} Np={Np} H={H} D={D}", lambda: ops.attention(qkv, idx, H, D ** -0.5), R)
    if N <= 300:
        screen(f"score_select B={B} N={N} H={H} D={D}", lambda: ops.score_select(qkv, H, Np - 1 if Np != N else N // 2), R)

# ---- round 2 kernels: fp8 x fp8 GEMMs (both tilings), fp8 LayerNorm
def e4m3_codes(shape):
    b = torch.randint(0, 256, shape, dtype=torch.uint8, device=dev)
    b[(b & 0x7F) == 0x7F] = 0x38
    return b
for M, N, K, epi in [(50432, 2304, 768, nat.EPI_BIAS), (44288, 3072, 768, nat.EPI_BIAS_GELU), (38912, 768, 3072, nat.EPI_BIAS_RESID),
                     (22272, 768, 3072, nat.EPI_BIAS_RESID), (300, 3072, 768, nat.EPI_BIAS_GELU)]:
    xq, wq = e4m3_codes((M, K)), e4m3_codes(((N + 255) // 256 * 256, K))
    xs, ws = torch.rand(M, device=dev) / 64 + 0.01, torch.rand(N, device=dev) / 64 + 0.01
    b = torch.randn(N, device=dev)
    ys = torch.rand(M, device=dev) + 0.5 if epi == nat.EPI_BIAS_GELU else None
    resid = torch.randn(1, M, N, device=dev) if epi == nat.EPI_BIAS_RESID else None
    for til in ((1, 2) if epi != nat.EPI_BIAS_RESID else (1,)):
        nat.lib().rajni_debug_force_f8_tiling(til)
        screen(f"linear f8xf8 {M}x{N}x{K} epi={epi} tiling={til}",
               lambda: ops.linear(xq.view(1, M, K), wq, N, b, epi, resid=resid, w_scale=ws, x_scale=xs, y_scale=ys), R)
    nat.lib().rajni_debug_force_f8_tiling(0)
xr = torch.randn(50432, 768, device=dev) * 3 + 1
lw, lb = torch.rand(768, device=dev) + 0.5, torch.randn(768, device=dev) * 0.1
screen("layernorm_fp8 50432x768 (+ hidden bound)", lambda: ops.layernorm_fp8(xr, lw, lb, 1e-6, hidden_bound=(0.6, 0.1)), R)
sched = {3: {"keep_ratio": 0.88, "update": True}, 4: {"keep_ratio": 0.88, "update": True}, 7: {"keep_ratio": 0.80, "update": True}, 8: {"keep_ratio": 0.72, "update": True}}
cfg = ts.CONFIGS["vit_base_patch16_224"]
m = rajni_amd.RAJNIViTWrapper(ts.create_model(cfg, seed=0).to(torch.bfloat16).cuda(), sched).eval()
imgs = torch.randn(256, 3, 224, 224, device=dev).to(torch.bfloat16)
screen("whole forward ViT-B/16 batch 256 README schedule", lambda: m(imgs), max(50, R // 2))
m.set_weight_format("fp8")
screen("whole forward, fp8 block weights", lambda: m(imgs), max(30, R // 4))
m.set_weight_format("fp8_mfma")
screen("whole forward, fp8_mfma (e4m3 activations on the fp8 pipe)", lambda: m(imgs), max(30, R // 4))
print(f"{'CLEAN' if bad == 0 else 'DIFFERENCES: %d' % bad}  ({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
