"""CPU oracle for the RAJNI token-pruning forward path.  TEST INFRASTRUCTURE ONLY.

This is a numpy restatement of the reference algorithm (dRaniwal/RAJNI-ViT), written from the
behaviour documented in SURVEY.md section 3 - it shares no code with the reference and does not import
it.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`rajni-vit_amd/`) never does, and fails loudly without its HIP library.

Pinning: the reference holds no tests or golden vectors of its own, so this oracle is
pinned by outputs of the reference itself, run in the authoring container on CPU:
`tests/golden/make_golden.py` imports `/root/reference/rajni` and writes `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against those fixtures, and
`tests/test_oracle_vs_reference.py` checks it against the live import when `/root/reference` exists.
Parity vs *real timm* for the patch-embed / unpruned-block / head code is unpinned (timm is not in
the image); those parts follow timm's documented block semantics.

Each function cites the reference file:line it restates (paths relative to /root/reference/rajni).
All arithmetic is done in `dtype` (float64 by default: the oracle is the high-precision answer).
`rajni_oracle_torch.py` is the same restatement on torch CPU ops (fp32), held to the same fixtures: it is what
`bench.py` times as `cpu_baseline`, this module is what the parity tests check against.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

try:  # exact-erf GELU (timm `nn.GELU()` default)
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover - scipy is in both images
    _erf = np.vectorize(math.erf, otypes=[np.float64])

Schedule = Dict[int, Dict]


# ----------------------------------------------------------------------------------------------
# L0  importance score        wrapper/importance.py:4-34
# ----------------------------------------------------------------------------------------------

def importance_scores(qkv: np.ndarray, num_heads: int, eps: float = 1e-6,
                      dtype=np.float64) -> np.ndarray:
    """scores[b,n] = A_cls[b,n] * sigmoid(zscore_n(||Vbar[b,n] - mean_n Vbar||_2)).

    qkv is [B, N, 3C] with the 3C axis laid out [3][H][D]          (importance.py:14-15).
    """
    qkv = np.asarray(qkv, dtype=dtype)
    B, N, threeC = qkv.shape
    C = threeC // 3
    D = C // num_heads
    t = qkv.reshape(B, N, 3, num_heads, D)
    q_cls = t[:, 0, 0]                    # [B,H,D]   CLS query            (importance.py:18)
    k = t[:, :, 1]                        # [B,N,H,D]
    v = t[:, :, 2]                        # [B,N,H,D]
    # CLS->token attention, softmax over ALL N incl. CLS, 1/sqrt(D)         (importance.py:19-20)
    logits = np.einsum("bhd,bnhd->bhn", q_cls, k) / math.sqrt(D)
    logits = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(logits)
    attn = e / e.sum(axis=-1, keepdims=True)
    a_cls = attn.mean(axis=1)             # mean over heads                (importance.py:21)
    vbar = v.mean(axis=2)                 # [B,N,D] mean over heads        (importance.py:24)
    vbar = vbar - vbar.mean(axis=1, keepdims=True)   # centre over tokens  (importance.py:25)
    vnorm = np.sqrt((vbar * vbar).sum(axis=-1))      # [B,N]               (importance.py:27)
    mu = vnorm.mean(axis=1, keepdims=True)           #                     (importance.py:28)
    # torch.std default is the UNBIASED estimator (N-1); eps is added to std (importance.py:29)
    std = np.sqrt(((vnorm - mu) ** 2).sum(axis=1, keepdims=True) / (N - 1)) + eps
    z = (vnorm - mu) / std                           #                     (importance.py:31)
    return a_cls * (1.0 / (1.0 + np.exp(-z)))        #                     (importance.py:32-34)


# ----------------------------------------------------------------------------------------------
# L1  keep count, selection    wrapper/attention.py:31-39
# ----------------------------------------------------------------------------------------------

def keep_count(keep_ratio: float, n_tokens: int) -> int:
    """Python-double multiply then truncation; never below 1         (attention.py:31-32)."""
    return max(1, int(keep_ratio * (n_tokens - 1)))


def token_counts(n0: int, depth: int, schedule: Schedule) -> List[int]:
    """Token count at the ENTRY of every block (model.py:43,68).  Data independent."""
    out, n = [], n0
    for i in range(depth):
        out.append(n)
        if i in schedule:
            n = keep_count(schedule[i]["keep_ratio"], n) + 1
    return out


def select_tokens(scores: np.ndarray, keep: int) -> np.ndarray:
    """keep_idx [B, keep+1]: 0 (CLS) then the `keep` best patch tokens in ascending index order.

    Restates topk -> sort -> +1 -> prepend 0                             (attention.py:34-39).
    torch.topk leaves ties unspecified; the build DEFINES: larger score first, then lower index;
    NaN ranks above every number (torch.topk also treats NaN as largest).
    """
    s = np.asarray(scores)
    B, N = s.shape
    out = np.zeros((B, keep + 1), dtype=np.int64)
    for b in range(B):
        p = s[b, 1:].astype(np.float64)
        key = np.where(np.isnan(p), np.inf, p)
        # stable sort on -key keeps lower index first among equals
        order = np.argsort(-key, kind="stable")[:keep]
        out[b, 1:] = np.sort(order) + 1
    return out


def selection_is_valid_topk(scores: np.ndarray, keep_idx: np.ndarray, keep: int) -> bool:
    """True iff keep_idx is *a* correct answer of attention.py:34-39 for `scores`, whatever the
    tie-break: slot 0 is CLS, the rest strictly ascending in [1,N), and the multiset of selected
    patch scores equals the multiset of the `keep` largest patch scores."""
    s = np.asarray(scores, dtype=np.float64)
    ki = np.asarray(keep_idx)
    B, N = s.shape
    if ki.shape != (B, keep + 1):
        return False
    for b in range(B):
        row = ki[b]
        if row[0] != 0 or np.any(row[1:] < 1) or np.any(row[1:] >= N) or np.any(np.diff(row[1:]) <= 0):
            return False
        p = np.where(np.isnan(s[b, 1:]), np.inf, s[b, 1:])
        best = np.sort(p)[::-1][:keep]
        got = np.sort(p[row[1:] - 1])[::-1]
        if not np.array_equal(best, got):
            return False
    return True


# ----------------------------------------------------------------------------------------------
# building blocks (timm semantics; unpinned vs real timm, see module docstring)
# ----------------------------------------------------------------------------------------------

def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray]) -> np.ndarray:
    y = x @ w.T
    return y if b is None else y + b


def gelu(x: np.ndarray) -> np.ndarray:
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def softmax_attention(q: np.ndarray, k: np.ndarray, v: np.ndarray, scale: float) -> np.ndarray:
    """q,k,v [B,H,N,D] -> [B,N,H*D]; explicit softmax(q k^T * scale) v   (attention.py:51-54)."""
    s = np.einsum("bhqd,bhkd->bhqk", q, k) * scale
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(axis=-1, keepdims=True)
    o = np.einsum("bhqk,bhkd->bqhd", p, v)
    B, N, H, D = o.shape
    return o.reshape(B, N, H * D)


def split_heads(qkv: np.ndarray, num_heads: int):
    """[B,N,3C] -> q,k,v [B,H,N,D]                                      (attention.py:46-49)."""
    B, N, threeC = qkv.shape
    D = threeC // 3 // num_heads
    t = qkv.reshape(B, N, 3, num_heads, D).transpose(2, 0, 3, 1, 4)
    return t[0], t[1], t[2]


def gather_rows(x: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """x [B,N,...], idx [B,K] -> [B,K,...]            (attention.py:42-43,58; model.py:55-56)."""
    return np.take_along_axis(x, idx.reshape(idx.shape + (1,) * (x.ndim - 2)), axis=1)


# ----------------------------------------------------------------------------------------------
# L1  pruned attention          wrapper/attention.py:17-60
# ----------------------------------------------------------------------------------------------

def rajni_attention(x_norm: np.ndarray, sd: Dict[str, np.ndarray], prefix: str, num_heads: int,
                    keep_ratio: float, update: bool = True,
                    prev_scores: Optional[np.ndarray] = None,
                    forced_keep_idx: Optional[np.ndarray] = None, dtype=np.float64,
                    fp8_out_scale: Optional[float] = None):
    """returns (out [B,Np,C], keep_idx [B,Np] int64, next_scores [B,Np], scores [B,N]).

    `forced_keep_idx` replaces the selection (selection-conditional parity).
    `fp8_out_scale`: the build's opt-in fp8 format - the attention output passes through e4m3 with this one scale before
    proj (rajni_attention_fp8) where the launch qualifies (attention_out_is_fp8).
    """
    x_norm = np.asarray(x_norm, dtype=dtype)
    B, N, C = x_norm.shape
    W = lambda n: np.asarray(sd[prefix + n], dtype=dtype)
    qkv = linear(x_norm, W("qkv.weight"), W("qkv.bias"))              # attention.py:21-22
    if update or prev_scores is None:                                  # attention.py:25-28
        scores = importance_scores(qkv, num_heads, dtype=dtype)
    else:
        scores = np.asarray(prev_scores, dtype=dtype)
    keep = keep_count(keep_ratio, N)                                   # attention.py:31-32
    keep_idx = select_tokens(scores, keep) if forced_keep_idx is None \
        else np.asarray(forced_keep_idx, dtype=np.int64)               # attention.py:34-39
    qkv = gather_rows(qkv, keep_idx)                                   # attention.py:42-43
    q, k, v = split_heads(qkv, num_heads)                              # attention.py:46-49
    D = C // num_heads
    out = softmax_attention(q, k, v, D ** -0.5)                        # attention.py:51-54
    if fp8_out_scale is not None and attention_out_is_fp8(D, out.shape[1]):
        out = quantize_rows_e4m3(out, np.float32(fp8_out_scale)).astype(dtype)
    out = linear(out, W("proj.weight"), W("proj.bias"))                # attention.py:55-56
    next_scores = np.take_along_axis(scores, keep_idx, axis=1)         # attention.py:58
    return out, keep_idx, next_scores, scores


# ----------------------------------------------------------------------------------------------
# L2  wrapper forward           wrapper/model.py:30-69
# ----------------------------------------------------------------------------------------------

def normalise_schedule(schedule) -> Schedule:
    """int keys (the reference forgets to for JSON input: SURVEY B1); `update` defaults True
    (model.py:19)."""
    out = {}
    for k, v in (schedule or {}).items():
        out[int(k)] = {"keep_ratio": float(v["keep_ratio"]), "update": bool(v.get("update", True))}
    return out


def patch_embed(images: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """conv PxP stride P -> [B, num_patches, C]   (timm PatchEmbed; model.py:34)."""
    B, Cin, Hh, Ww = images.shape
    C, _, P, _ = w.shape
    gh, gw = Hh // P, Ww // P
    cols = images.reshape(B, Cin, gh, P, gw, P).transpose(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, Cin * P * P)
    return cols @ w.reshape(C, -1).T + b


# ----------------------------------------------------------------------------------------------
# fp8 activations (the build's own opt-in "fp8_mfma" format; the reference has no fp8 semantics - SURVEY 7
# "hard parts" - so this restates include/rajni_hip.h's rajni_layernorm_fp8 / rajni_linear_args.x_scale rule)
# ----------------------------------------------------------------------------------------------

def e4m3_rne(v: np.ndarray) -> np.ndarray:
    """Round to the nearest OCP e4m3 "fn" value (ties to even), saturating at +-448; returns float64 values."""
    v = np.asarray(v, dtype=np.float64)
    a = np.minimum(np.abs(v), 448.0)
    e = np.maximum(np.floor(np.log2(np.maximum(a, 2.0 ** -30))), -6.0)     # subnormals share the spacing of 2^-6
    q = 2.0 ** (e - 3.0)                                                   # 3 mantissa bits
    return np.sign(v) * np.minimum(np.round(a / q) * q, 448.0)


def quantize_rows_e4m3(x: np.ndarray, scale: np.ndarray) -> np.ndarray:
    """Rows of x [..., C] as the fp8 kernels see them: e4m3(x * (1/scale)) * scale, with the reciprocal and the
    product in fp32 like the device (so a host restatement reproduces the stored bytes)."""
    s32 = np.asarray(scale, dtype=np.float32)[..., None]
    inv = (np.float32(1.0) / s32).astype(np.float32)
    return e4m3_rne(np.asarray(x, dtype=np.float32) * inv) * s32.astype(np.float64)


def row_scale_e4m3(x: np.ndarray) -> np.ndarray:
    """max |row| / 448 in fp32 (1 for an all-zero row)."""
    amax = np.abs(np.asarray(x, dtype=np.float32)).max(axis=-1)
    return np.where(amax > 0, amax / np.float32(448.0), np.float32(1.0)).astype(np.float32)


def hidden_scale_bound(xn: np.ndarray, fc1_w: np.ndarray, fc1_b: np.ndarray) -> np.ndarray:
    """Per-row scale of the MLP hidden activations from the bound |gelu(xn W1^T + b1)| <= ||xn|| max_n ||W1[n]|| + max |b1|
    (with the 1.0625 margin of rajni_layernorm_fp8), in fp32."""
    wn = np.float32(np.sqrt((np.asarray(fc1_w, dtype=np.float64) ** 2).sum(axis=1)).max())
    bm = np.float32(np.abs(fc1_b).max())
    nrm = np.sqrt((np.asarray(xn, dtype=np.float32).astype(np.float64) ** 2).sum(axis=-1)).astype(np.float32)
    bound = np.float32(1.0625) * nrm * wn + bm
    return np.where(bound > 0, bound / np.float32(448.0), np.float32(1.0)).astype(np.float32)


def attention_out_is_fp8(head_dim: int, n_tokens: int) -> bool:
    """where an act_fp8 plan emits e4m3 attention rows: the launches the persistent head-dim-64 kernel serves
    (rajni_attention_fp8: D == 64, at most 224 kept tokens); elsewhere proj keeps bf16 activations"""
    return head_dim == 64 and n_tokens <= 224


def attention_out_scale(norm1_w: np.ndarray, norm1_b: np.ndarray, wv: np.ndarray, bv: Optional[np.ndarray]) -> np.float32:
    """The one scale of a block's e4m3 attention output (include/rajni_hip.h, rajni_attention_fp8): attention rows are convex
    combinations of V rows, |V[j,c]| <= ||ln1(x)[j]||_2 ||Wv[c]||_2 + |bv[c]|, ||ln1(x)[j]||_2 <= sqrt(C) max|gamma1| +
    ||beta1||_2;  (1.0625 * that * max_c ||Wv[c]||_2 + max|bv|) / 448 in fp32."""
    g, b = np.asarray(norm1_w, dtype=np.float32), np.asarray(norm1_b, dtype=np.float32)
    wv = np.asarray(wv, dtype=np.float32)
    wn = np.float32(np.sqrt((wv.astype(np.float64) ** 2).sum(axis=1)).max())
    ln = np.float32(np.sqrt(np.float32(wv.shape[1]))) * np.float32(np.abs(g).max()) + np.float32(np.sqrt((b.astype(np.float64) ** 2).sum()))
    bound = np.float32(1.0625) * ln * wn
    if bv is not None:
        bound = bound + np.float32(np.abs(np.asarray(bv, dtype=np.float32)).max())
    return np.float32(bound / np.float32(448.0))


def vit_forward(sd: Dict[str, np.ndarray], images: np.ndarray, schedule, *, depth: int,
                num_heads: int, ln_eps: float = 1e-6,
                forced_keep: Optional[Dict[int, np.ndarray]] = None, dtype=np.float64,
                return_trace: bool = False, act_fp8: bool = False):
    """RAJNIViTWrapper.forward restated (model.py:30-69) over a timm-named state dict.

    returns logits [B,num_classes], stats {"token_counts": [...]}, and (optionally) a per-block
    trace {block: {"scores","keep_idx","next_scores"}} for the scheduled blocks.
    A pos_embed with N-1 rows (timm `no_embed_class`) is added to the patch tokens only - the
    mathematically identical fix for SURVEY B3.
    `act_fp8`: the build's opt-in fp8 activations - norm1 / norm2 outputs and the MLP hidden activations pass
    through per-row e4m3 quantisation (see quantize_rows_e4m3) before qkv / fc1 / fc2, and the attention output through
    e4m3 with one scale per block (attention_out_scale) before proj where attention_out_is_fp8.
    """
    schedule = normalise_schedule(schedule)
    P = lambda n: np.asarray(sd[n], dtype=dtype)
    x = patch_embed(np.asarray(images, dtype=dtype), P("patch_embed.proj.weight"),
                    P("patch_embed.proj.bias"))                        # model.py:34
    B = x.shape[0]
    cls = np.broadcast_to(P("cls_token"), (B, 1, x.shape[-1]))
    pos = P("pos_embed")
    if pos.shape[1] == x.shape[1]:                                     # no_embed_class (B3)
        x = np.concatenate([cls, x + pos], axis=1)
    else:
        x = np.concatenate([cls, x], axis=1)                           # model.py:35-36
        x = x + pos[:, : x.shape[1]]                                   # model.py:37
    scores = None                                                      # model.py:39
    counts: List[int] = []
    trace: Dict[int, Dict[str, np.ndarray]] = {}
    for i in range(depth):                                             # model.py:42
        counts.append(x.shape[1])                                      # model.py:43
        p = f"blocks.{i}."
        ls1 = P(p + "ls1.gamma") if (p + "ls1.gamma") in sd else None  # model.py:45-48
        ls2 = P(p + "ls2.gamma") if (p + "ls2.gamma") in sd else None
        xn = layer_norm(x, P(p + "norm1.weight"), P(p + "norm1.bias"), ln_eps)
        osc = None
        if act_fp8:
            xn = quantize_rows_e4m3(xn, row_scale_e4m3(xn)).astype(dtype)
            Cc = x.shape[-1]
            osc = attention_out_scale(sd[p + "norm1.weight"], sd[p + "norm1.bias"], sd[p + "attn.qkv.weight"][2 * Cc:3 * Cc],
                                      sd[p + "attn.qkv.bias"][2 * Cc:3 * Cc] if (p + "attn.qkv.bias") in sd else None)
        if i in schedule:                                              # model.py:50-59
            cfg = schedule[i]
            out, keep_idx, scores, full = rajni_attention(
                xn, sd, p + "attn.", num_heads, cfg["keep_ratio"], cfg["update"], scores,
                None if forced_keep is None else forced_keep.get(i), dtype=dtype, fp8_out_scale=osc)
            trace[i] = {"scores": full, "keep_idx": keep_idx, "next_scores": scores}
            x = gather_rows(x, keep_idx)                               # model.py:55-56
        else:                                                          # model.py:61-63 (timm Block)
            qkv = linear(xn, P(p + "attn.qkv.weight"), P(p + "attn.qkv.bias"))
            q, k, v = split_heads(qkv, num_heads)
            out = softmax_attention(q, k, v, (x.shape[-1] // num_heads) ** -0.5)
            if osc is not None and attention_out_is_fp8(x.shape[-1] // num_heads, out.shape[1]):
                out = quantize_rows_e4m3(out, osc).astype(dtype)
            out = linear(out, P(p + "attn.proj.weight"), P(p + "attn.proj.bias"))
            scores = None
        x = x + (out if ls1 is None else out * ls1)                    # model.py:58
        h = layer_norm(x, P(p + "norm2.weight"), P(p + "norm2.bias"), ln_eps)
        if act_fp8:
            hs = hidden_scale_bound(h, P(p + "mlp.fc1.weight"), P(p + "mlp.fc1.bias"))
            h = quantize_rows_e4m3(h, row_scale_e4m3(h)).astype(dtype)
        h = gelu(linear(h, P(p + "mlp.fc1.weight"), P(p + "mlp.fc1.bias")))
        if act_fp8:
            h = quantize_rows_e4m3(h, hs).astype(dtype)
        h = linear(h, P(p + "mlp.fc2.weight"), P(p + "mlp.fc2.bias"))
        x = x + (h if ls2 is None else h * ls2)                        # model.py:59
    x = layer_norm(x[:, 0], P("norm.weight"), P("norm.bias"), ln_eps)  # model.py:65 (CLS row only:
    logits = linear(x, P("head.weight"), P("head.bias"))               #  LN is per token) model.py:66
    stats = {"token_counts": counts}                                   # model.py:68
    if return_trace:
        return logits, stats, trace
    return logits, stats


# ----------------------------------------------------------------------------------------------
# L3  evaluate_model bookkeeping      eval.py:6-75
# ----------------------------------------------------------------------------------------------

def evaluate_plan(n_loader_batches: int, warmup: int, max_batches: Optional[int]) -> Tuple[int, int]:
    """(warm-up forwards, timed forwards) evaluate_model executes for a loader of that length:
    warm-up restarts the iterator when it runs out (eval.py:19-26); the timed loop walks a fresh
    iterator and stops at max_batches (eval.py:44-46)."""
    timed = n_loader_batches if max_batches is None else min(n_loader_batches, max_batches)
    return warmup, timed


def top1_percent(logits_batches: Sequence[np.ndarray], label_batches: Sequence[np.ndarray]) -> float:
    """100 * correct / max(total, 1) with argmax over dim 1            (eval.py:61-64,73)."""
    correct = total = 0
    for lg, lb in zip(logits_batches, label_batches):
        correct += int((np.argmax(lg, axis=1) == np.asarray(lb)).sum())
        total += len(lb)
    return 100.0 * correct / max(total, 1)
