"""`RAJNIAttention` - pruned attention of one scheduled block, on the MI355X.

Mirrors the reference module (`rajni/wrapper/attention.py:5-60`): same constructor, same attributes
borrowed from the timm `Attention`, same `forward(x, prev_scores=None) -> (out, keep_idx,
next_scores)`.  Differences are in HOW: QKV and proj are hand-written MFMA GEMMs, score + top-k +
compaction is one kernel, and the gather of the kept rows (attention.py:42-43) is fused into the
attention kernel's tile loads, so no gathered copy of qkv exists.

`RAJNIViTWrapper.forward` does not call this module's forward (it runs the whole network as one
native plan); it is kept for API parity and for block-level use and tests.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import _native as nat
from .. import ops


class RAJNIAttention(nn.Module):
    def __init__(self, attn: nn.Module, keep_ratio: float, update: bool):
        super().__init__()
        # attribute contract of a timm Attention (attention.py:8-12)
        self.num_heads = attn.num_heads
        self.scale = attn.scale
        self.qkv = attn.qkv
        self.proj = attn.proj
        self.proj_drop = attn.proj_drop
        for extra in ("q_norm", "k_norm"):
            mod = getattr(attn, extra, None)
            if mod is not None and not isinstance(mod, nn.Identity):
                raise NotImplementedError(
                    f"RAJNIAttention: attn.{extra} is {type(mod).__name__}; the reference silently drops "
                    "it (SURVEY Q5) - refusing instead of computing something different")
        self.keep_ratio = keep_ratio
        self.update = update
        self._packed = None
        self._packed_key = None

    # ---- weights in the layout the kernels want (rebuilt when parameters change) -------------
    def _weights(self, device, dtype):
        params = [self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias]
        key = (str(device), dtype) + tuple((p.data_ptr(), p._version) for p in params if p is not None)
        if self._packed_key != key:
            self._packed = dict(
                qkv_w=ops.pack_weight(self.qkv.weight, dtype, device),
                qkv_b=ops.pack_vec(self.qkv.bias, dtype, device),
                proj_w=ops.pack_weight(self.proj.weight, dtype, device),
                proj_b=ops.pack_vec(self.proj.bias, dtype, device))
            self._packed_key = key
        return self._packed

    @torch.no_grad()
    def forward(self, x: torch.Tensor, prev_scores: Optional[torch.Tensor] = None):
        """x: [B, N, C] (already normed).  Returns out [B, Np, C], keep_idx [B, Np] int64,
        next_scores [B, Np]."""
        nat.require_device(x, "x")
        B, N, Cc = x.shape
        w = self._weights(x.device, x.dtype)
        qkv = ops.linear(x, w["qkv_w"], 3 * Cc, w["qkv_b"], nat.EPI_BIAS)            # attention.py:21-22
        keep = ops.keep_count(self.keep_ratio, N)                                     # attention.py:31-32
        if self.update or prev_scores is None:                                        # attention.py:25-28
            _, keep_idx, next_scores = ops.score_select(qkv, self.num_heads, keep, want_scores=False)
        else:
            keep_idx, next_scores = ops.select_topk(prev_scores.to(x.dtype), keep)    # attention.py:34-39,58
        out = ops.attention(qkv, keep_idx, self.num_heads, self.scale)                # attention.py:42-54
        out = ops.linear(out, w["proj_w"], Cc, w["proj_b"], nat.EPI_BIAS)             # attention.py:55
        if isinstance(self.proj_drop, nn.Dropout) and self.proj_drop.p > 0 and self.training:
            raise NotImplementedError("RAJNIAttention: proj_drop > 0 in training mode (inference path only)")
        return out, keep_idx.long(), next_scores                                      # attention.py:60
