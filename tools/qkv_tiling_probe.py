"""QKV / FC1 (bf16-output GEMMs): 256x256 vs 256x128 tiling per token count at batch 256 and 64."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat
dev = "cuda"
for B in (256, 64):
  for (nm, N, K, epi) in (("qkv", 2304, 768, nat.EPI_BIAS), ("fc1", 3072, 768, nat.EPI_BIAS_GELU)):
    for tok in (197, 173, 152, 121, 87):
        M = B * tok
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = ops.pack_weight((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)); b = torch.randn(N, device=dev)
        res = {}
        for r in range(4):
            for til in (4, 5):
                nat.lib().rajni_debug_force_gemm_tiling(til)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): ops.linear(x.view(1, M, K), w, N, b, epi)
                e1.record(); torch.cuda.synchronize()
                if r: res.setdefault(til, []).append(e0.elapsed_time(e1) / 5 * 1e3)
        tw = math.ceil(M/256)*math.ceil(N/256)
        print(f"B={B} {nm}_{tok}: wide {min(res[4]):6.1f} ({tw/256:.2f} rounds)  mid {min(res[5]):6.1f} ({2*tw/256:.2f})  {'MID WINS' if min(res[5]) < min(res[4]) else ''}", flush=True)
nat.lib().rajni_debug_force_gemm_tiling(0)
