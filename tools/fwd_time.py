"""ms per 256-image ViT-B/16 forward (README schedule) with the library named by RAJNI_HIP_LIB; used by tools/ab_libs.sh."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch, rajni_amd
from rajni_amd import timm_shaped as ts
sched = {3: {"keep_ratio": 0.88, "update": True}, 4: {"keep_ratio": 0.88, "update": True}, 7: {"keep_ratio": 0.80, "update": True}, 8: {"keep_ratio": 0.72, "update": True}}
cfg = ts.CONFIGS["vit_base_patch16_224"]
m = rajni_amd.RAJNIViTWrapper(ts.create_model(cfg, seed=0).to(torch.bfloat16).cuda(), sched).eval()
if len(sys.argv) > 1 and sys.argv[1] in ("fp8", "fp8_mfma"): m.set_weight_format(sys.argv[1])
x = torch.randn(256, 3, 224, 224, device="cuda").to(torch.bfloat16)
for _ in range(5): m(x)
ts_ = []
for r in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize(); ts_.append((time.perf_counter() - t0) / 10 * 1e3)
print(f"{os.environ.get('RAJNI_HIP_LIB','default')[-20:]:22s} min {min(ts_):.3f} ms  med {sorted(ts_)[4]:.3f} ms", end="")
# per kernel class (HIP events on the launch stream), us per forward
from rajni_amd import _native as nat
nat.profile_reset(); nat.profile_enable(0x1FFFF)
for _ in range(5): m(x)
torch.cuda.synchronize(); nat.profile_enable(0)
pr = nat.profile_collect()
short = {"gemm_bf16_tn<bias>": "qkv", "gemm_bf16_tn<bias,gelu>": "fc1", "gemm_bf16_tn<bias,ls,resid>": "fc2", "gemm_bf16_tn<bias,ls,resid> K<=N": "proj",
         "layernorm_kernel": "ln", "attn_bf16_d64": "attn", "gemm_f8_tn<bias>": "qkv8", "gemm_f8_tn<bias,gelu,requant>": "fc1_8",
         "gemm_f8_tn<bias,ls,resid>": "fc2_8", "gemm_f8_tn<bias,ls,resid> K<=N": "proj8", "score_select_kernel<fused>": "score", "gemm_bf16_tn<patch>": "patch"}
print("  | us/forward: " + "  ".join(f"{short.get(k, k[:14])} {v['ms'] / 5 * 1e3:.0f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1]['ms'])))
