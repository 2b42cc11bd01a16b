#!/usr/bin/env python3
"""Diagnostic (GPU box): where does the fp8_mfma forward differ from the oracle with the same quantisation rule?
Prints logits distances between: device fp8_mfma, device fp8 (weights only), oracle(act_fp8), oracle(dequantised weights)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import rajni_amd
from oracle import rajni_oracle as orc
from rajni_amd import timm_shaped as ts, ops, _native as nat

DEV = "cuda"
sched = {1: {"keep_ratio": 0.75, "update": True}, 2: {"keep_ratio": 0.6, "update": False}}
cfg = ts.CONFIGS["vit_micro512_patch16_64"]
model = ts.create_model(cfg, seed=4, std=0.06, bias_std=0.02, round_bf16=True)
w = rajni_amd.RAJNIViTWrapper(model, sched).to(DEV).to(torch.bfloat16).eval()
imgs = ts.bf16_round_np(np.random.default_rng(9).standard_normal((8, 3, 64, 64), dtype=np.float32))
x = torch.from_numpy(imgs).to(DEV)
w.set_weight_format("fp8_mfma")
d8m = w(x).float().cpu().numpy()
forced = {i: t["keep_idx"].cpu().numpy() for i, t in w.get_last_trace().items()}
sd = ts.state_dict_numpy(model)
sd.update({k: v.cpu().numpy() for k, v in w.dequantized_state_dict().items()})
kw = dict(depth=cfg.depth, num_heads=cfg.num_heads, ln_eps=cfg.ln_eps, forced_keep=forced)
o8m, _ = orc.vit_forward(sd, imgs, sched, act_fp8=True, **kw)
o8, _ = orc.vit_forward(sd, imgs, sched, **kw)
w.set_weight_format("fp8")
w.force_keep_idx({i: torch.from_numpy(v).to(DEV) for i, v in forced.items()})
d8 = w(x).float().cpu().numpy()
sc = np.abs(o8).max()
f = lambda a, b: f"{np.abs(a - b).max() / sc:.4f}"
print("logit scale", sc)
print("device fp8_mfma vs oracle act_fp8     ", f(d8m, o8m))
print("device fp8_mfma vs oracle weights-only", f(d8m, o8))
print("device fp8      vs oracle weights-only", f(d8, o8))
print("oracle act_fp8  vs oracle weights-only", f(o8m, o8))
print("device fp8_mfma vs device fp8         ", f(d8m, d8))

# one linear on real activations: LN1 of a random stream -> qkv
C = cfg.embed_dim
rng = np.random.default_rng(0)
xs = rng.standard_normal((600, C)).astype(np.float32)
lnw, lnb = sd["blocks.0.norm1.weight"], sd["blocks.0.norm1.bias"]
q, s = ops.layernorm_fp8(torch.from_numpy(xs).to(DEV), torch.from_numpy(lnw).to(DEV), torch.from_numpy(lnb).to(DEV), cfg.ln_eps)
bw = w._weights["blocks"][0]
y = ops.linear(q, bw["qkv_w"], 3 * C, bw["qkv_b"], nat.EPI_BIAS, w_scale=bw["qkv_s"], x_scale=s).float().cpu().numpy()
xn = orc.layer_norm(xs.astype(np.float64), lnw.astype(np.float64), lnb.astype(np.float64), cfg.ln_eps)
xq = orc.quantize_rows_e4m3(xn, orc.row_scale_e4m3(xn))
want = xq @ sd["blocks.0.attn.qkv.weight"].astype(np.float64).T + sd["blocks.0.attn.qkv.bias"]
print("LN1->qkv on the fp8 pipe vs rule:", np.abs(y - want).max() / np.abs(want).max(), " vs unquantised:",
      np.abs(y - (xn @ sd["blocks.0.attn.qkv.weight"].astype(np.float64).T + sd["blocks.0.attn.qkv.bias"])).max() / np.abs(want).max())
