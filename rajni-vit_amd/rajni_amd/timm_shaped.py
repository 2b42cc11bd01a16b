"""A minimal timm-*shaped* Vision Transformer.

timm is not installed in the build/bench images, and the reference wrapper
(`/root/reference/rajni/wrapper/model.py:9-10,34-37,45-48,65-66`,
`attention.py:8-12`) only consumes an attribute contract, not timm itself:

  base : .patch_embed(x)->[B,N-1,C]  .cls_token [1,1,C]  .pos_embed [1,N,C]
         .pos_drop  .blocks  .norm  .head
  block: .norm1 .attn .norm2 .mlp  (+ optional .ls1 .ls2 .drop_path1 .drop_path2), blk(x)
  attn : .num_heads .scale .qkv (Linear C->3C laid out [3][H][D]) .proj .proj_drop

This module provides exactly that contract so tests, `bench.py` and the CLI have
a base model to wrap.  It is the *unpruned stock-PyTorch baseline* (the "4x"
denominator of BASELINE.json), not the product path: the product path is the HIP
forward behind `RAJNIViTWrapper`.

Weights are synthesised from a numpy PCG64 stream so that the same state dict
can be rebuilt bit-for-bit on the GPU box from `(config, seed)` alone (there is
no network for checkpoints).  Parameter names follow timm's state-dict naming
so a real timm checkpoint loads with `load_state_dict`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass(frozen=True)
class ViTConfig:
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    num_classes: int = 1000
    layer_scale: Optional[float] = None   # DeiT-3 style LayerScale init value
    no_embed_class: bool = False          # DeiT-3: pos_embed has no CLS row
    ln_eps: float = 1e-6

    @property
    def num_patches(self) -> int:
        return (self.img_size // self.patch_size) ** 2

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def hidden_dim(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)

    def to_dict(self):
        return asdict(self)


# The model names BASELINE.json's configs use.
CONFIGS: Dict[str, ViTConfig] = {
    "vit_tiny_patch16_224": ViTConfig(embed_dim=192, depth=12, num_heads=3),
    "vit_small_patch16_224": ViTConfig(embed_dim=384, depth=12, num_heads=6),
    "vit_base_patch16_224": ViTConfig(embed_dim=768, depth=12, num_heads=12),
    "vit_large_patch16_224": ViTConfig(embed_dim=1024, depth=24, num_heads=16),
    "vit_large_patch16_384": ViTConfig(img_size=384, embed_dim=1024, depth=24, num_heads=16),
    "deit3_base_patch16_224": ViTConfig(embed_dim=768, depth=12, num_heads=12,
                                        layer_scale=1e-6, no_embed_class=True),
    # head dim 80, patch 14: the general attention / importance kernels and the materialised patch columns
    "vit_huge_patch14_224": ViTConfig(patch_size=14, embed_dim=1280, depth=32, num_heads=16),
    # tiny head_dim-64 model for fast parity tests (not a timm name)
    "vit_micro_patch16_64": ViTConfig(img_size=64, embed_dim=128, depth=4, num_heads=2,
                                      num_classes=10),
    # the same with DeiT-3's LayerScale and no_embed_class pos-embed (N-1 rows): loader / B3 tests (not a timm name)
    "deit3_micro_patch16_64": ViTConfig(img_size=64, embed_dim=128, depth=4, num_heads=2, num_classes=10,
                                        layer_scale=1e-6, no_embed_class=True),
    # embed dim 512 / MLP width 2048 (multiples of 256, K >= 512: what the fp8 x fp8 kernel needs) at micro cost (not a timm name)
    "vit_micro512_patch16_64": ViTConfig(img_size=64, embed_dim=512, depth=4, num_heads=8, num_classes=10),
    # patch 14 (ViT-L/14, ViT-H/14, DINOv2): 3*14*14 = 588 input features, not whole 64-wide K steps
    "vit_micro_patch14_56": ViTConfig(img_size=56, patch_size=14, embed_dim=128, depth=4, num_heads=2,
                                      num_classes=10),
    # the same with head dim 80 (ViT-H's), for the general-head-dim kernels (not a timm name)
    "vit_micro_d80_patch16_64": ViTConfig(img_size=64, embed_dim=320, depth=4, num_heads=4,
                                          num_classes=10),
}


class PatchEmbed(nn.Module):
    def __init__(self, cfg: ViTConfig):
        super().__init__()
        self.img_size = (cfg.img_size, cfg.img_size)
        self.patch_size = (cfg.patch_size, cfg.patch_size)
        self.num_patches = cfg.num_patches
        self.proj = nn.Conv2d(cfg.in_chans, cfg.embed_dim, cfg.patch_size, cfg.patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.q_norm = nn.Identity()
        self.k_norm = nn.Identity()
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        x = F.scaled_dot_product_attention(q, k, v)
        return self.proj_drop(self.proj(x.transpose(1, 2).reshape(B, N, C)))


class LayerScale(nn.Module):
    def __init__(self, dim: int, init: float):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.drop1 = nn.Dropout(0.0)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop2 = nn.Dropout(0.0)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class Block(nn.Module):
    def __init__(self, cfg: ViTConfig):
        super().__init__()
        C = cfg.embed_dim
        self.norm1 = nn.LayerNorm(C, eps=cfg.ln_eps)
        self.attn = Attention(C, cfg.num_heads)
        self.ls1 = LayerScale(C, cfg.layer_scale) if cfg.layer_scale else nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2 = nn.LayerNorm(C, eps=cfg.ln_eps)
        self.mlp = Mlp(C, cfg.hidden_dim)
        self.ls2 = LayerScale(C, cfg.layer_scale) if cfg.layer_scale else nn.Identity()
        self.drop_path2 = nn.Identity()

    def forward(self, x):
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        x = x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))
        return x


class VisionTransformer(nn.Module):
    """timm-shaped ViT (class token, learned absolute pos-embed, pre-norm blocks)."""

    def __init__(self, cfg: ViTConfig):
        super().__init__()
        self.cfg = cfg
        self.num_classes = cfg.num_classes
        self.embed_dim = cfg.embed_dim
        self.num_prefix_tokens = 1
        self.no_embed_class = cfg.no_embed_class
        self.patch_embed = PatchEmbed(cfg)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, cfg.embed_dim))
        n_pos = cfg.num_patches if cfg.no_embed_class else cfg.num_patches + 1
        self.pos_embed = nn.Parameter(torch.zeros(1, n_pos, cfg.embed_dim))
        self.pos_drop = nn.Dropout(0.0)
        self.blocks = nn.Sequential(*[Block(cfg) for _ in range(cfg.depth)])
        self.norm = nn.LayerNorm(cfg.embed_dim, eps=cfg.ln_eps)
        self.fc_norm = nn.Identity()
        self.head_drop = nn.Dropout(0.0)
        self.head = nn.Linear(cfg.embed_dim, cfg.num_classes)

    def _pos_embed(self, x):
        cls = self.cls_token.expand(x.shape[0], -1, -1)
        if self.no_embed_class:
            x = torch.cat([cls, x + self.pos_embed], dim=1)
        else:
            x = torch.cat([cls, x], dim=1) + self.pos_embed
        return self.pos_drop(x)

    def forward_features(self, x):
        x = self._pos_embed(self.patch_embed(x))
        x = self.blocks(x)
        return self.norm(x)

    def forward(self, x):
        x = self.forward_features(x)
        return self.head(self.head_drop(self.fc_norm(x[:, 0])))


# ----------------------------------------------------------------------------------------------
# deterministic synthetic weights
# ----------------------------------------------------------------------------------------------

def synth_state_dict(cfg: ViTConfig, seed: int = 0, std: float = 0.02,
                     bias_std: float = 0.0) -> Dict[str, np.ndarray]:
    """timm-named float32 state dict drawn from numpy PCG64(seed).

    `std` is the normal std of every linear / conv / pos-embed weight (timm's init is
    trunc_normal(.02)); parity fixtures use a larger std so importance scores are well
    separated (with std .02 the keep-boundary gap is ~1e-7).  LayerNorm gains
    are drawn around 1 and biases around 0 so that LN/bias code paths are exercised.
    """
    rng = np.random.default_rng(seed)
    C, Hd, P = cfg.embed_dim, cfg.hidden_dim, cfg.patch_size

    def nrm(*shape, s=std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(s)).astype(np.float32)

    sd: Dict[str, np.ndarray] = {}
    sd["cls_token"] = nrm(1, 1, C)
    n_pos = cfg.num_patches if cfg.no_embed_class else cfg.num_patches + 1
    sd["pos_embed"] = nrm(1, n_pos, C)
    fan_in = cfg.in_chans * P * P
    sd["patch_embed.proj.weight"] = nrm(C, cfg.in_chans, P, P, s=1.0 / math.sqrt(fan_in))
    sd["patch_embed.proj.bias"] = nrm(C, s=bias_std) if bias_std else np.zeros(C, np.float32)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        for ln in ("norm1", "norm2"):
            sd[p + ln + ".weight"] = (1.0 + nrm(C, s=bias_std)).astype(np.float32)
            sd[p + ln + ".bias"] = nrm(C, s=bias_std) if bias_std else np.zeros(C, np.float32)
        sd[p + "attn.qkv.weight"] = nrm(3 * C, C)
        sd[p + "attn.qkv.bias"] = nrm(3 * C, s=bias_std) if bias_std else np.zeros(3 * C, np.float32)
        sd[p + "attn.proj.weight"] = nrm(C, C)
        sd[p + "attn.proj.bias"] = nrm(C, s=bias_std) if bias_std else np.zeros(C, np.float32)
        sd[p + "mlp.fc1.weight"] = nrm(Hd, C)
        sd[p + "mlp.fc1.bias"] = nrm(Hd, s=bias_std) if bias_std else np.zeros(Hd, np.float32)
        sd[p + "mlp.fc2.weight"] = nrm(C, Hd)
        sd[p + "mlp.fc2.bias"] = nrm(C, s=bias_std) if bias_std else np.zeros(C, np.float32)
        if cfg.layer_scale:
            # a trained DeiT-3 has gammas of order 0.1-1; keep the configured init but jitter it
            # so the LayerScale multiply is observable in parity tests.
            sd[p + "ls1.gamma"] = (np.float32(cfg.layer_scale) + np.abs(nrm(C, s=0.5))).astype(np.float32)
            sd[p + "ls2.gamma"] = (np.float32(cfg.layer_scale) + np.abs(nrm(C, s=0.5))).astype(np.float32)
    sd["norm.weight"] = (1.0 + nrm(C, s=bias_std)).astype(np.float32)
    sd["norm.bias"] = nrm(C, s=bias_std) if bias_std else np.zeros(C, np.float32)
    sd["head.weight"] = nrm(cfg.num_classes, C)
    sd["head.bias"] = nrm(cfg.num_classes, s=bias_std) if bias_std else np.zeros(cfg.num_classes, np.float32)
    return sd


def bf16_round_np(a: np.ndarray) -> np.ndarray:
    """Round a float32 array to the nearest bf16-representable float32 (RNE)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def create_model(name_or_cfg, seed: int = 0, std: float = 0.02, bias_std: float = 0.0,
                 round_bf16: bool = False) -> VisionTransformer:
    """Build a timm-shaped ViT with deterministic synthetic weights.

    `round_bf16=True` rounds every weight to a bf16-representable value while keeping the
    parameters in fp32, so that an fp32 oracle and a bf16 device model see identical weights.
    """
    cfg = CONFIGS[name_or_cfg] if isinstance(name_or_cfg, str) else name_or_cfg
    model = VisionTransformer(cfg)
    sd = synth_state_dict(cfg, seed=seed, std=std, bias_std=bias_std)
    if round_bf16:
        sd = {k: bf16_round_np(v) for k, v in sd.items()}
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return model.eval()


def state_dict_numpy(model: nn.Module) -> Dict[str, np.ndarray]:
    """float32 numpy copy of a (timm-shaped) model's state dict."""
    return {k: v.detach().to(torch.float32).cpu().numpy() for k, v in model.state_dict().items()}
