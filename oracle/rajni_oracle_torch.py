"""CPU oracle, torch flavour.  TEST INFRASTRUCTURE ONLY - the same restatement as `rajni_oracle.py` (numpy, the
high-precision checker) written with torch CPU ops, because that is what the reference itself executes on a host:
ATen matmuls, `layer_norm`, exact-erf `gelu`, `softmax`.  It exists so that `bench.py`'s `cpu_baseline` times a
CPU path of the reference's own speed class (numpy's single-threaded elementwise passes made the numpy oracle ~6x
slower than the reference's CPU run on the same cores).  Shares no code with the reference and does not import it;
only `tests/` and `bench.py`'s baseline legs import this module: `cpu_baseline` (host cores) and
`pruned_torch_images_per_sec` (the same op graph on stock PyTorch-ROCm kernels on the GPU - what the reference's own
attention.py:17-60 / model.py:50-59 execute there; a denominator, never the product).

Pinned like the numpy oracle: `tests/test_oracle_golden.py` holds it to the reference-generated fixtures in
`tests/golden/` (logits, token counts, selections) and to the numpy oracle.

Each function cites the reference file:line it restates (paths relative to /root/reference/rajni).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def importance_scores(qkv: torch.Tensor, num_heads: int, eps: float = 1e-6) -> torch.Tensor:
    """A_cls * sigmoid(zscore(||Vbar - mean Vbar||))                        (importance.py:4-34)"""
    B, N, threeC = qkv.shape
    D = threeC // 3 // num_heads
    t = qkv.reshape(B, N, 3, num_heads, D)
    q_cls = t[:, 0, 0]                                        # importance.py:18
    k, v = t[:, :, 1], t[:, :, 2]
    logits = torch.einsum("bhd,bnhd->bhn", q_cls, k) / math.sqrt(D)          # importance.py:19
    a_cls = torch.softmax(logits, dim=-1).mean(dim=1)                         # importance.py:20-21
    vbar = v.mean(dim=2)                                                      # importance.py:24
    vbar = vbar - vbar.mean(dim=1, keepdim=True)                              # importance.py:25
    vnorm = vbar.norm(dim=-1)                                                 # importance.py:27
    z = (vnorm - vnorm.mean(dim=1, keepdim=True)) / (vnorm.std(dim=1, keepdim=True) + eps)   # importance.py:28-29
    return a_cls * torch.sigmoid(z)                                           # importance.py:31-34


def keep_count(keep_ratio: float, n_tokens: int) -> int:
    return max(1, int(keep_ratio * (n_tokens - 1)))                           # attention.py:31-32


def select_tokens(scores: torch.Tensor, keep: int, ref_ops: bool = False) -> torch.Tensor:
    """indices of the `keep` largest patch scores, ascending, +1, CLS prepended (attention.py:34-39) with the
    DEFINED tie rule of the numpy oracle: larger first, then lower index (stable descending sort; NaN sorts first).
    ref_ops=True issues the reference's own op pair instead (torch.topk + sort, attention.py:34-36; tie order
    unspecified) - used only where the op graph, not the tie rule, is what is being timed (bench.py's GPU baseline)."""
    if ref_ops:
        order = torch.topk(scores[:, 1:], keep, dim=1).indices
    else:
        order = torch.sort(scores[:, 1:], dim=1, descending=True, stable=True).indices[:, :keep]
    idx = torch.sort(order, dim=1).values + 1
    return torch.cat([torch.zeros((scores.shape[0], 1), dtype=idx.dtype, device=idx.device), idx], dim=1)


def _attention(qkv: torch.Tensor, num_heads: int) -> torch.Tensor:
    B, N, threeC = qkv.shape
    C = threeC // 3
    D = C // num_heads
    q, k, v = qkv.reshape(B, N, 3, num_heads, D).permute(2, 0, 3, 1, 4)       # attention.py:46-49
    attn = torch.softmax((q @ k.transpose(-2, -1)) * D ** -0.5, dim=-1)       # attention.py:51-52
    return (attn @ v).transpose(1, 2).reshape(B, N, C)                        # attention.py:54


def vit_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, schedule, *, depth: int, num_heads: int,
                ln_eps: float = 1e-6, forced_keep: Optional[Dict[int, torch.Tensor]] = None, return_trace: bool = False,
                ref_op_graph: bool = False):
    """RAJNIViTWrapper.forward (model.py:30-69) + RAJNIAttention.forward (attention.py:17-60) over a timm-named
    state dict of tensors (fp32 on the CPU for the checker; any device / dtype the ATen ops take for the baseline legs
    of bench.py).  returns logits, {"token_counts": [...]} (and the per-stage trace).
    ref_op_graph=True issues exactly the reference's ops where this restatement otherwise takes an equal-valued shortcut
    or a defined rule: torch.topk + sort for the selection (attention.py:34-36) and the final LayerNorm over all tokens
    (model.py:65) - for timing the reference's op graph (bench.py `pruned_torch_images_per_sec`), not for checking."""
    schedule = {int(k): {"keep_ratio": float(v["keep_ratio"]), "update": bool(v.get("update", True))}
                for k, v in (schedule or {}).items()}
    with torch.no_grad():
        w = sd["patch_embed.proj.weight"]
        x = F.conv2d(images, w, sd["patch_embed.proj.bias"], stride=w.shape[-1]).flatten(2).transpose(1, 2)   # model.py:34
        B, _, C = x.shape
        cls = sd["cls_token"].expand(B, -1, -1)
        pos = sd["pos_embed"]
        if pos.shape[1] == x.shape[1]:                                        # no_embed_class
            x = torch.cat([cls, x + pos], dim=1)
        else:
            x = torch.cat([cls, x], dim=1) + pos[:, : x.shape[1] + 1]         # model.py:35-37
        scores = None
        counts: List[int] = []
        trace = {}
        for i in range(depth):                                                # model.py:42
            counts.append(x.shape[1])                                         # model.py:43
            p = f"blocks.{i}."
            xn = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], ln_eps)
            qkv = F.linear(xn, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])          # attention.py:21-22
            if i in schedule:                                                 # model.py:50-59
                cfg = schedule[i]
                if cfg["update"] or scores is None:                           # attention.py:25-28
                    full = importance_scores(qkv, num_heads)
                else:
                    full = scores
                keep = keep_count(cfg["keep_ratio"], x.shape[1])
                idx = select_tokens(full, keep, ref_op_graph) if forced_keep is None or i not in forced_keep \
                    else torch.as_tensor(forced_keep[i], dtype=torch.int64, device=x.device)
                qkv = torch.gather(qkv, 1, idx.unsqueeze(-1).expand(-1, -1, qkv.shape[-1]))   # attention.py:42-43
                scores = torch.gather(full, 1, idx)                           # attention.py:58
                x = torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, C))   # model.py:55-56
                trace[i] = {"scores": full, "keep_idx": idx, "next_scores": scores}
            else:
                scores = None                                                 # model.py:61-63
            out = F.linear(_attention(qkv, num_heads), sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
            if p + "ls1.gamma" in sd:
                out = out * sd[p + "ls1.gamma"]
            x = x + out                                                       # model.py:58
            h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], ln_eps)
            h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])),
                         sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
            if p + "ls2.gamma" in sd:
                h = h * sd[p + "ls2.gamma"]
            x = x + h                                                         # model.py:59
        if ref_op_graph:      # the reference normalises every token and then takes row 0 (model.py:65-66): same values
            x = F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"], ln_eps)[:, 0]
        else:
            x = F.layer_norm(x[:, 0], (C,), sd["norm.weight"], sd["norm.bias"], ln_eps)   # model.py:65 (LN is per token)
        logits = F.linear(x, sd["head.weight"], sd["head.bias"])              # model.py:66
    stats = {"token_counts": counts}                                          # model.py:68
    return (logits, stats, trace) if return_trace else (logits, stats)
