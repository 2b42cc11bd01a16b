"""Tensor-level entry points over the C ABI (one function per `rajni_*` symbol of
include/rajni_hip.h).  They only validate, allocate outputs with torch, and pass raw pointers plus
torch's current stream; all compute is in librajni_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _native as nat


def keep_count(keep_ratio: float, n_tokens: int) -> int:
    """`keep = max(1, int(keep_ratio * (N - 1)))` - Python-double semantics of the reference
    (rajni/wrapper/attention.py:31-32); data independent."""
    return max(1, int(keep_ratio * (n_tokens - 1)))


def _dt(t: torch.Tensor) -> int:
    return nat.dtype_code(t.dtype)


def pack_weight(w: torch.Tensor, dtype=torch.bfloat16, device=None, k_multiple: int = 1) -> torch.Tensor:
    """[N, K...] Linear/Conv weight -> contiguous [ceil256(N), K] in `dtype`, zero padded rows
    (the GEMM stages whole 128- or 256-row W tiles).  `k_multiple` = 64 also zero-pads the columns to
    whole K steps - the patch-embed weight of a patch size whose Cin*P*P is not one (3*14*14 = 588 -> 640)."""
    n = w.shape[0]
    w2 = w.detach().reshape(n, -1)
    npad = (n + 255) // 256 * 256
    k = w2.shape[1]
    kpad = (k + k_multiple - 1) // k_multiple * k_multiple
    out = torch.zeros((npad, kpad), dtype=dtype, device=device if device is not None else w.device)
    out[:n, :k].copy_(w2)
    return out


FP8_E4M3_MAX = 448.0   # largest finite e4m3 "fn" value (OCP fp8: no infinities, one NaN pattern)


def quantize_rows_fp8(w: torch.Tensor):
    """Per-output-row symmetric fp8 e4m3 quantisation of a [N, K] weight held in the model dtype:
    scale[n] = max|w[n,:]| / 448 (1 for an all-zero row), q = round-to-nearest-even(w / scale) in e4m3.
    Returns (q as torch.float8_e4m3fn [N, K], scale fp32 [N]).  Host-side; runs wherever `w` lives."""
    w32 = w.detach().to(torch.float32)
    amax = w32.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / FP8_E4M3_MAX, torch.ones_like(amax))
    q = (w32 / scale[:, None]).clamp_(-FP8_E4M3_MAX, FP8_E4M3_MAX).to(torch.float8_e4m3fn)
    return q, scale


def pack_weight_fp8(w: torch.Tensor, model_dtype=torch.bfloat16, device=None, k_multiple: int = 1):
    """[N, K...] Linear weight -> (uint8 [ceil256(N), K] of e4m3 bytes with zero padded rows, fp32 scale [N]).
    The weight is first rounded to `model_dtype` (what the reference model would hold), then quantised.
    `k_multiple` zero-pads the columns (e4m3 0x00 = +0) like pack_weight."""
    n = w.shape[0]
    w2 = w.detach().reshape(n, -1).to(model_dtype)
    k = w2.shape[1]
    kpad = (k + k_multiple - 1) // k_multiple * k_multiple
    if kpad % 16 != 0:
        raise ValueError("fp8 weights need the (padded) input dimension to be a multiple of 16")
    q, scale = quantize_rows_fp8(w2)
    dev = device if device is not None else w.device
    npad = (n + 255) // 256 * 256
    out = torch.zeros((npad, kpad), dtype=torch.uint8, device=dev)
    out[:n, :k].copy_(q.view(torch.uint8))
    return out, scale.to(dev).contiguous()


def dequantize_fp8(q_packed: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """fp32 [N, K] weight the fp8 kernels multiply by: e4m3 value x row scale (the oracle side of parity tests)."""
    n = scale.shape[0]
    return q_packed[:n].view(torch.float8_e4m3fn).to(torch.float32) * scale.to(torch.float32)[:, None]


def pack_vec(v: Optional[torch.Tensor], like_dtype=torch.bfloat16, device=None) -> Optional[torch.Tensor]:
    """bias / LayerNorm / LayerScale vector -> fp32 copy of the value the model dtype holds."""
    if v is None:
        return None
    return v.detach().to(like_dtype).to(torch.float32).to(device if device is not None else v.device).contiguous()


def importance(qkv: torch.Tensor, num_heads: int, eps: float = 1e-6) -> torch.Tensor:
    nat.require_device(qkv, "qkv")
    qkv = qkv.contiguous()
    B, N, threeC = qkv.shape
    D = threeC // 3 // num_heads
    out = torch.empty((B, N), dtype=qkv.dtype, device=qkv.device)
    with nat.device_guard(qkv.device):
        nat.check(nat.lib().rajni_importance(qkv.data_ptr(), out.data_ptr(), B, N, num_heads, D, eps, _dt(qkv),
                                             nat.stream_ptr(qkv.device)), "rajni_importance")
    return out


def select_topk(scores: torch.Tensor, keep: int) -> Tuple[torch.Tensor, torch.Tensor]:
    nat.require_device(scores, "scores")
    scores = scores.contiguous()
    B, N = scores.shape
    idx = torch.empty((B, keep + 1), dtype=torch.int32, device=scores.device)
    nxt = torch.empty((B, keep + 1), dtype=scores.dtype, device=scores.device)
    with nat.device_guard(scores.device):
        nat.check(nat.lib().rajni_select_topk(scores.data_ptr(), B, N, keep, idx.data_ptr(), nxt.data_ptr(),
                                              _dt(scores), nat.stream_ptr(scores.device)), "rajni_select_topk")
    return idx, nxt


def score_select(qkv: torch.Tensor, num_heads: int, keep: int, eps: float = 1e-6, want_scores: bool = True):
    nat.require_device(qkv, "qkv")
    qkv = qkv.contiguous()
    B, N, threeC = qkv.shape
    D = threeC // 3 // num_heads
    scores = torch.empty((B, N), dtype=qkv.dtype, device=qkv.device) if want_scores else None
    idx = torch.empty((B, keep + 1), dtype=torch.int32, device=qkv.device)
    nxt = torch.empty((B, keep + 1), dtype=qkv.dtype, device=qkv.device)
    with nat.device_guard(qkv.device):
        nat.check(nat.lib().rajni_score_select(qkv.data_ptr(), B, N, num_heads, D, eps, keep, nat.ptr(scores),
                                               idx.data_ptr(), nxt.data_ptr(), _dt(qkv),
                                               nat.stream_ptr(qkv.device)), "rajni_score_select")
    return scores, idx, nxt


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """src [B, N, E], idx [B, K] int32 -> [B, K, E]"""
    nat.require_device(src, "src")
    src = src.contiguous()
    idx = idx.to(torch.int32).contiguous()
    B, N, E = src.shape
    K = idx.shape[1]
    out = torch.empty((B, K, E), dtype=src.dtype, device=src.device)
    with nat.device_guard(src.device):
        nat.check(nat.lib().rajni_gather_rows(src.data_ptr(), idx.data_ptr(), out.data_ptr(), B, N, K, E, _dt(src),
                                              nat.stream_ptr(src.device)), "rajni_gather_rows")
    return out


def attention(qkv: torch.Tensor, keep_idx: Optional[torch.Tensor], num_heads: int, scale: float) -> torch.Tensor:
    """qkv [B, N, 3C]; keep_idx [B, Np] int32 or None -> [B, Np, C]"""
    nat.require_device(qkv, "qkv")
    qkv = qkv.contiguous()
    B, N, threeC = qkv.shape
    Cc = threeC // 3
    D = Cc // num_heads
    if keep_idx is not None:
        keep_idx = keep_idx.to(torch.int32).contiguous()
        Np = keep_idx.shape[1]
    else:
        Np = N
    out = torch.empty((B, Np, Cc), dtype=qkv.dtype, device=qkv.device)
    with nat.device_guard(qkv.device):
        nat.check(nat.lib().rajni_attention(qkv.data_ptr(), nat.ptr(keep_idx), out.data_ptr(), B, N, Np, num_heads, D,
                                            float(scale), _dt(qkv), nat.stream_ptr(qkv.device)), "rajni_attention")
    return out


def attention_out_scale(norm1_w: torch.Tensor, norm1_b: torch.Tensor, wv: torch.Tensor, bv: Optional[torch.Tensor]) -> float:
    """out_scale of `attention_fp8` from a block's parameters (the bound of rajni_attention_fp8, include/rajni_hip.h):
    (1.0625 * (sqrt(C) * max|gamma1| + ||beta1||_2) * max_c ||Wv[c]||_2 + max|bv|) / 448, evaluated in fp32.
    `wv` are the V rows of the qkv weight AS THE KERNELS HOLD THEM (dequantised e4m3), [C, C]."""
    g, b = norm1_w.detach().to(torch.float32), norm1_b.detach().to(torch.float32)
    w32 = wv.detach().to(torch.float32)
    c = torch.tensor(float(w32.shape[1]), dtype=torch.float32).sqrt()
    bound = torch.tensor(1.0625, dtype=torch.float32) * (c * g.abs().max().cpu() + b.norm().cpu()) * w32.norm(dim=1).max().cpu()
    if bv is not None:
        bound = bound + bv.detach().to(torch.float32).abs().max().cpu()
    return float(bound / torch.tensor(448.0, dtype=torch.float32))


def attention_fp8(qkv: torch.Tensor, keep_idx: Optional[torch.Tensor], num_heads: int, scale: float, out_scale: float):
    """attention with e4m3 output rows (opt-in fp8_mfma format): qkv bf16 [B, N, 3C], head dim 64, at most 224 kept tokens
    -> (bytes uint8 [B, Np, C] = e4m3_rne_sat(attn / out_scale), row scales fp32 [B * Np] = out_scale)"""
    nat.require_device(qkv, "qkv")
    qkv = qkv.contiguous()
    B, N, threeC = qkv.shape
    Cc = threeC // 3
    D = Cc // num_heads
    if keep_idx is not None:
        keep_idx = keep_idx.to(torch.int32).contiguous()
        Np = keep_idx.shape[1]
    else:
        Np = N
    out = torch.empty((B, Np, Cc), dtype=torch.uint8, device=qkv.device)
    rs = torch.empty(B * Np, dtype=torch.float32, device=qkv.device)
    with nat.device_guard(qkv.device):
        nat.check(nat.lib().rajni_attention_fp8(qkv.data_ptr(), nat.ptr(keep_idx), out.data_ptr(), float(out_scale), rs.data_ptr(),
                                                B, N, Np, num_heads, D, float(scale), nat.stream_ptr(qkv.device)),
                  "rajni_attention_fp8")
    return out, rs


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, rows: Optional[int] = None,
              row_stride: Optional[int] = None, out_dtype=torch.bfloat16) -> torch.Tensor:
    """LayerNorm over the last axis; w, b fp32.  x may be `out_dtype` or fp32 (the fp32 residual
    stream); the result is `out_dtype`.  With rows/row_stride reads a strided subset of rows."""
    nat.require_device(x, "x")
    x = x.contiguous()
    x_f32 = int(x.dtype == torch.float32 and out_dtype != torch.float32)
    Cc = x.shape[-1]
    if rows is None:
        rows = x.numel() // Cc
        row_stride = Cc
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    else:
        out = torch.empty((rows, Cc), dtype=out_dtype, device=x.device)
    with nat.device_guard(x.device):
        nat.check(nat.lib().rajni_layernorm(x.data_ptr(), row_stride, w.data_ptr(), b.data_ptr(), out.data_ptr(), rows,
                                            Cc, float(eps), nat.dtype_code(out_dtype), x_f32,
                                            nat.stream_ptr(x.device)), "rajni_layernorm")
    return out


def layernorm_fp8(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float,
                  hidden_bound: Optional[Tuple[float, float]] = None):
    """LayerNorm over the last axis with the output row quantised for the fp8 matrix pipe (rajni_layernorm_fp8):
    returns (q uint8 [..., C] of e4m3 bytes, scale fp32 [rows]) and, with `hidden_bound = (max row norm of the
    dequantised fc1 weight, max |fc1 bias|)`, also the per-row scale of the MLP hidden activations."""
    nat.require_device(x, "x")
    x = x.contiguous()
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise NotImplementedError("layernorm_fp8: x must be fp32 (the residual stream) or bf16")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    hs = torch.empty(rows, dtype=torch.float32, device=x.device) if hidden_bound is not None else None
    wn, bm = (float(hidden_bound[0]), float(hidden_bound[1])) if hidden_bound is not None else (0.0, 0.0)
    with nat.device_guard(x.device):
        nat.check(nat.lib().rajni_layernorm_fp8(x.data_ptr(), Cc, w.data_ptr(), b.data_ptr(), q.data_ptr(), scale.data_ptr(),
                                                nat.ptr(hs), wn, bm, rows, Cc, float(eps), int(x.dtype == torch.float32),
                                                nat.stream_ptr(x.device)), "rajni_layernorm_fp8")
    return (q, scale) if hs is None else (q, scale, hs)


def linear(x: torch.Tensor, w_packed: torch.Tensor, n_out: int, bias: Optional[torch.Tensor] = None,
           epilogue: int = nat.EPI_BIAS, gamma: Optional[torch.Tensor] = None,
           resid: Optional[torch.Tensor] = None, r_idx: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, w_scale: Optional[torch.Tensor] = None,
           x_scale: Optional[torch.Tensor] = None, y_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = epi(x @ W^T): x [..., K]; w_packed from pack_weight(); bias/gamma fp32 [n_out].
    With `w_scale` (fp32 [n_out]) w_packed is the uint8 e4m3 tensor of pack_weight_fp8() and x is bf16.
    resid [B, N_src, n_out] (+ r_idx [B, Np] int32 to gather its rows) for EPI_BIAS_RESID; an fp32
    resid selects the fp32 residual stream (the output is then fp32 too).
    With `x_scale` (fp32 [M]) x is the uint8 e4m3 tensor of layernorm_fp8() and the product runs on the fp8 matrix
    pipe (w_scale required); with EPI_BIAS_GELU `y_scale` (fp32 [M]) is required and the result is uint8 e4m3."""
    nat.require_device(x, "x")
    x = x.contiguous()
    K = x.shape[-1]
    M = x.numel() // K
    f8 = x_scale is not None
    if f8 and (x.dtype != torch.uint8 or w_scale is None):
        raise ValueError("x_scale needs the uint8 e4m3 activations of layernorm_fp8() and fp8 weights (w_scale)")
    act_dtype = torch.bfloat16 if f8 else x.dtype
    out8 = f8 and epilogue == nat.EPI_BIAS_GELU
    ld = (n_out + 15) // 16 * 16 if out8 else (n_out + 7) // 8 * 8
    stream_f32 = int(resid is not None and resid.dtype == torch.float32 and act_dtype != torch.float32)
    if out is None:
        out = torch.empty((M, ld), dtype=torch.uint8 if out8 else (torch.float32 if stream_f32 else act_dtype), device=x.device)
    a = nat.LinearArgs()
    a.x, a.lda, a.w, a.ldw = x.data_ptr(), K, w_packed.data_ptr(), w_packed.shape[1]
    a.bias, a.gamma, a.w_scale = nat.ptr(bias), nat.ptr(gamma), nat.ptr(w_scale)
    if w_scale is not None and w_packed.dtype != torch.uint8:
        raise ValueError("w_scale given but w_packed is not the uint8 tensor of pack_weight_fp8()")
    a.y, a.ldc = out.data_ptr(), out.shape[-1] if out.dim() == 2 else out.stride(-2)
    a.M, a.N, a.K, a.epilogue, a.dtype, a.stream_f32 = M, n_out, K, epilogue, nat.dtype_code(act_dtype), stream_f32
    a.x_scale, a.y_scale = nat.ptr(x_scale), nat.ptr(y_scale)
    if resid is not None:
        resid = resid.contiguous()
        a.resid, a.ldr = resid.data_ptr(), resid.shape[-1]
        if r_idx is not None:
            r_idx = r_idx.to(torch.int32).contiguous()
            a.r_idx, a.r_np, a.r_nsrc = r_idx.data_ptr(), r_idx.shape[1], resid.shape[1]
    with nat.device_guard(x.device):
        nat.check(nat.lib().rajni_linear(C.byref(a), nat.stream_ptr(x.device)), "rajni_linear")
    y = out[:, :n_out] if ld != n_out else out
    return y.reshape(*x.shape[:-1], n_out) if ld == n_out else y


def patch_embed(images: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, cls: torch.Tensor,
                pos: torch.Tensor, pos_has_cls: bool, patch: int, embed_dim: int,
                out_f32: bool = False) -> torch.Tensor:
    nat.require_device(images, "images")
    images = images.contiguous()
    B, Cin, S, _ = images.shape
    n = (S // patch) ** 2 + 1
    x = torch.empty((B, n, embed_dim), dtype=torch.float32 if out_f32 else images.dtype, device=images.device)
    kpad = (Cin * patch * patch + 63) // 64 * 64
    if w_packed.shape[1] != kpad:
        raise ValueError(f"patch_embed: weight must be packed with k_multiple=64 ([*, {kpad}]), got {tuple(w_packed.shape)}")
    nbytes = nat.lib().rajni_patch_embed_workspace_bytes(B, Cin, S, patch, _dt(images))   # 0: im2col fused into the loads
    ws = torch.empty(nbytes, dtype=torch.uint8, device=images.device) if nbytes else None
    with nat.device_guard(images.device):
        nat.check(nat.lib().rajni_patch_embed(images.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), cls.data_ptr(),
                                              pos.data_ptr(), int(pos_has_cls), x.data_ptr(), int(out_f32), B, Cin, S, patch,
                                              embed_dim, _dt(images), nat.ptr(ws), nbytes, nat.stream_ptr(images.device)),
                  "rajni_patch_embed")
    return x
