"""Head dims other than 64 (reference: attention.py:46-54 and importance.py:8-24 are written for any
C // num_heads; timm has 80 (ViT-H), 88 (ViT-g), 72 (SigLIP so400m), 32/48 (small heads), 128).  GPU box only.

The tuned kernels are D = 64; every other D % 8 == 0 up to 128 takes `attn_bf16_dgen` / `attn_f32_dgen`, the
plain importance passes of `score_select_kernel` and the general CLS kernel.  Same oracle, same tolerances as
the D = 64 tests in test_gpu_kernels.py / test_gpu_fuzz.py.  D = 64 itself is in the lists where the general
code is reachable for it (importance, CLS attention), so old and new paths are held to the same answers.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import rajni_amd
from oracle import rajni_oracle as orc
from rajni_amd import _native as nat, ops, timm_shaped as ts
from rajni_amd.timm_shaped import bf16_round_np

DEV = "cuda"
HEAD_DIMS = [8, 16, 32, 48, 72, 80, 88, 96, 128]


def dev_bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(torch.bfloat16)


def host(t):
    return t.float().cpu().numpy().astype(np.float64)


def close(got, want, rel, what):
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g}"


def pick_rows(rng, B, N, Np):
    return np.stack([np.concatenate([[0], 1 + np.sort(rng.choice(N - 1, Np - 1, replace=False))]) for _ in range(B)])


@pytest.mark.parametrize("D", HEAD_DIMS)
@pytest.mark.parametrize("B,N,Np,H", [(2, 197, 173, 3), (1, 300, 257, 2), (3, 17, 13, 2), (2, 64, 64, 1),
                                      (1, 65, 65, 2), (2, 40, 2, 2), (1, 130, 1, 1)])
def test_attention_any_head_dim_bf16(B, N, Np, H, D):
    rng = np.random.default_rng(N * 31 + Np + D)
    Cc = H * D
    qkv = bf16_round_np(rng.standard_normal((B, N, 3 * Cc), dtype=np.float32))
    if Np == N:
        idx_t, g = None, qkv
    else:
        idx = pick_rows(rng, B, N, Np)
        idx_t = torch.from_numpy(idx.astype(np.int32)).to(DEV)
        g = orc.gather_rows(qkv, idx.astype(np.int64))
    out = ops.attention(dev_bf16(qkv), idx_t, H, D ** -0.5)
    q, k, v = orc.split_heads(g.astype(np.float64), H)
    want = orc.softmax_attention(q, k, v, D ** -0.5)
    assert tuple(out.shape) == (B, Np, Cc)
    close(host(out), want, 1.5e-2, f"attention D={D}")


@pytest.mark.parametrize("D", [16, 48, 80, 128])
def test_attention_any_head_dim_online_softmax_spike(D):
    """one late key dominates a query row: the running-max rescale across 64-key chunks"""
    rng = np.random.default_rng(D)
    B, N, H = 1, 200, 1
    qkv = rng.standard_normal((B, N, 3 * D), dtype=np.float32) * 0.3
    qkv[0, 5, 0:D] = 4.0
    qkv[0, 170, D:2 * D] = 4.0
    qkv = bf16_round_np(qkv)
    out = ops.attention(dev_bf16(qkv), None, H, D ** -0.5)
    q, k, v = orc.split_heads(qkv.astype(np.float64), H)
    close(host(out), orc.softmax_attention(q, k, v, D ** -0.5), 1.5e-2, f"attention spike D={D}")


@pytest.mark.parametrize("D", [16, 40, 80, 104, 128])
@pytest.mark.parametrize("B,N,Np,H", [(2, 70, 33, 2), (1, 45, 45, 3)])
def test_attention_any_head_dim_fp32(B, N, Np, H, D):
    rng = np.random.default_rng(N + D)
    Cc = H * D
    qkv = rng.standard_normal((B, N, 3 * Cc), dtype=np.float32)
    if Np == N:
        idx_t, g = None, qkv
    else:
        idx = pick_rows(rng, B, N, Np)
        idx_t = torch.from_numpy(idx.astype(np.int32)).to(DEV)
        g = orc.gather_rows(qkv, idx.astype(np.int64))
    out = ops.attention(torch.from_numpy(qkv).to(DEV), idx_t, H, D ** -0.5)
    q, k, v = orc.split_heads(g.astype(np.float64), H)
    close(host(out), orc.softmax_attention(q, k, v, D ** -0.5), 2e-5, f"fp32 attention D={D}")


def test_attention_rejects_unsupported_head_dims():
    for D in (4, 20, 136):
        qkv = torch.zeros((1, 8, 3 * D), dtype=torch.bfloat16, device=DEV)
        with pytest.raises(NotImplementedError, match="head dim"):
            ops.attention(qkv, None, 1, 1.0)


@pytest.mark.parametrize("D", HEAD_DIMS + [64])
@pytest.mark.parametrize("B,N,H", [(3, 197, 5), (2, 33, 2), (1, 257, 16)])
def test_score_select_any_head_dim(B, N, H, D):
    rng = np.random.default_rng(N * H + D)
    qkv = bf16_round_np(rng.standard_normal((B, N, 3 * H * D), dtype=np.float32))
    keep = orc.keep_count(0.7, N)
    scores, idx, nxt = ops.score_select(dev_bf16(qkv), H, keep)
    s = host(scores)
    close(s, orc.importance_scores(qkv, H), 6e-3, f"scores D={D}")
    np.testing.assert_array_equal(idx.cpu().numpy(), orc.select_tokens(s, keep))
    np.testing.assert_array_equal(host(nxt), np.take_along_axis(s, idx.cpu().numpy().astype(np.int64), axis=1))
    assert torch.equal(ops.importance(dev_bf16(qkv), H), scores)
    # fp32 activations through the same passes
    q32 = rng.standard_normal((B, N, 3 * H * D), dtype=np.float32)
    close(host(ops.importance(torch.from_numpy(q32).to(DEV), H)), orc.importance_scores(q32, H), 2e-5, f"fp32 scores D={D}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("heads,D", [(4, 80), (2, 32), (1, 128), (4, 48), (2, 96), (8, 8), (8, 72), (8, 88)])
def test_forward_any_head_dim_vs_oracle(heads, D, dtype):
    """Whole forward of a small timm-shaped model whose head dim is not 64, pruning in two blocks (one of
    them carrying scores), against the oracle run with the device's own selections injected - the same
    statement as test_gpu_fuzz.py's forward fuzz.  embed_dim must be a multiple of 64 (GEMM K)."""
    C = heads * D
    assert C % 64 == 0
    cfg = ts.ViTConfig(img_size=64, embed_dim=C, depth=4, num_heads=heads, num_classes=10,
                       layer_scale=0.5 if D == 80 else None)
    sched = {1: {"keep_ratio": 0.75, "update": True}, 2: {"keep_ratio": 0.6, "update": False}}
    model = ts.create_model(cfg, seed=D, std=0.08, bias_std=0.02, round_bf16=True)
    sd = ts.state_dict_numpy(model)
    rng = np.random.default_rng(D)
    imgs = bf16_round_np(rng.standard_normal((3, 3, 64, 64), dtype=np.float32))
    wrapped = rajni_amd.RAJNIViTWrapper(model, sched).to(DEV).to(dtype).eval().trace_scores(True)
    logits = wrapped(torch.from_numpy(imgs).to(DEV).to(dtype)).float().cpu().numpy()
    forced = {}
    for i, d in wrapped.get_last_trace().items():
        idx = d["keep_idx"].cpu().numpy()
        np.testing.assert_array_equal(idx, orc.select_tokens(d["scores"].float().cpu().numpy().astype(np.float64), idx.shape[1] - 1))
        forced[i] = idx
    want, stats = orc.vit_forward(sd, imgs, sched, depth=cfg.depth, num_heads=heads, ln_eps=cfg.ln_eps,
                                  forced_keep=forced, dtype=np.float32)
    assert wrapped.get_last_stats() == stats
    close(logits, want, 1.5e-2 if dtype == torch.bfloat16 else 1e-3, f"forward heads={heads} D={D} {dtype}")
    # and the CLS-only last block (general CLS attention kernel) gives the same logits
    wrapped.set_last_block_cls_only(True)
    again = wrapped(torch.from_numpy(imgs).to(DEV).to(dtype)).float().cpu().numpy()
    close(again, logits, 8e-3 if dtype == torch.bfloat16 else 1e-5, "cls-only last block")


@pytest.mark.parametrize("fmt", ["bf16", "fp32", "fp8"])
@pytest.mark.parametrize("hidden", [336, 200, 1080])
def test_forward_mlp_width_not_whole_k_steps(hidden, fmt):
    """An MLP width that is not a multiple of 64 (timm so400m: 4304): the wrapper zero-pads fc1's rows / bias and
    fc2's columns to whole K steps at pack time; the logits are those of the unpadded model (oracle)."""
    C, heads = 128, 2
    cfg = ts.ViTConfig(img_size=64, embed_dim=C, depth=3, num_heads=heads, num_classes=10, mlp_ratio=hidden / C)
    assert cfg.hidden_dim == hidden and hidden % 64 != 0
    sched = {1: {"keep_ratio": 0.7, "update": True}}
    model = ts.create_model(cfg, seed=hidden, std=0.08, bias_std=0.02, round_bf16=True)
    dtype = torch.float32 if fmt == "fp32" else torch.bfloat16
    rng = np.random.default_rng(hidden)
    imgs = bf16_round_np(rng.standard_normal((2, 3, 64, 64), dtype=np.float32))
    wrapped = rajni_amd.RAJNIViTWrapper(model, sched).to(DEV).to(dtype).eval().trace_scores(True)
    if fmt == "fp8":
        wrapped.set_weight_format("fp8")
    logits = wrapped(torch.from_numpy(imgs).to(DEV).to(dtype)).float().cpu().numpy()
    sd = ts.state_dict_numpy(model)
    if fmt == "fp8":
        for k, v in wrapped.dequantized_state_dict().items():
            assert sd[k].shape == tuple(v.shape), k
            sd[k] = v.cpu().numpy()
    forced = {i: d["keep_idx"].cpu().numpy() for i, d in wrapped.get_last_trace().items()}
    want, stats = orc.vit_forward(sd, imgs, sched, depth=cfg.depth, num_heads=heads, ln_eps=cfg.ln_eps,
                                  forced_keep=forced, dtype=np.float32)
    assert wrapped.get_last_stats() == stats
    close(logits, want, 1e-3 if fmt == "fp32" else 1.5e-2, f"forward hidden={hidden} {fmt}")
