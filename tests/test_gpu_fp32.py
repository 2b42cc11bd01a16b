"""fp32 model path (`dtype = RAJNI_F32`): every tensor fp32, exact-fp32 MFMA GEMMs.  BASELINE.json's
fp32 bar is 1e-3; fp32 scores have no rounding ties, so here the device's keep_idx must equal the
REFERENCE's own keep_idx bit for bit (golden fixtures) and the free-running logits must match the
reference's fp32 logits.  GPU box only."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import rajni_amd
from oracle import rajni_oracle as orc
from rajni_amd import ops, _native as nat, timm_shaped as ts
from helpers import GOLDEN, load_case, case_images, pruned_blocks

DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def host(t):
    return t.double().cpu().numpy()


def close(got, want, rel, what=""):
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max()
    assert err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g} (rel {err / scale:.3g})"


@pytest.mark.parametrize("M,N,K", [(394, 2304, 768), (130, 768, 3072), (7, 1000, 768), (64, 10, 128), (1154, 576, 192)])
def test_linear_f32(M, N, K):
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * 0.05
    b = rng.standard_normal(N, dtype=np.float32)
    wp = ops.pack_weight(dev(w), torch.float32)
    y = ops.linear(dev(x), wp, N, dev(b), nat.EPI_BIAS)
    assert y.dtype == torch.float32
    close(host(y), x.astype(np.float64) @ w.astype(np.float64).T + b, 6e-6, "linear f32")  # fp32 accumulation over K <= 3072
    y = ops.linear(dev(x), wp, N, dev(b), nat.EPI_BIAS_GELU)
    close(host(y), orc.gelu(x.astype(np.float64) @ w.astype(np.float64).T + b), 6e-6, "linear+gelu f32")


def test_linear_resid_f32():
    rng = np.random.default_rng(9)
    B, Nsrc, Np, Cc, K = 3, 50, 37, 256, 192
    x = rng.standard_normal((B, Np, K), dtype=np.float32)
    w = rng.standard_normal((Cc, K), dtype=np.float32) * 0.1
    b = rng.standard_normal(Cc, dtype=np.float32) * 0.1
    gam = rng.standard_normal(Cc, dtype=np.float32)
    resid = rng.standard_normal((B, Nsrc, Cc), dtype=np.float32)
    idx = np.stack([np.sort(rng.choice(Nsrc, Np, replace=False)) for _ in range(B)]).astype(np.int32)
    y = ops.linear(dev(x), ops.pack_weight(dev(w), torch.float32), Cc, dev(b), nat.EPI_BIAS_RESID, gamma=dev(gam),
                   resid=dev(resid), r_idx=torch.from_numpy(idx).to(DEV))
    want = orc.gather_rows(resid.astype(np.float64), idx.astype(np.int64)) + gam * (x.astype(np.float64) @ w.astype(np.float64).T + b)
    close(host(y).reshape(want.shape), want, 2e-6, "linear+resid f32")


def test_importance_select_f32_match_reference_fixture():
    with open(os.path.join(GOLDEN, "importance_cases.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, "importance_cases.npz"))
    rng = np.random.default_rng(meta["seed"])
    for j, c in enumerate(meta["cases"]):
        qkv = ts.bf16_round_np(rng.standard_normal((c["B"], c["N"], 3 * c["H"] * c["D"]), dtype=np.float32) * c["scale"])
        got = ops.importance(dev(qkv), c["H"])
        assert got.dtype == torch.float32
        close(host(got), data[f"c{j}.scores"].astype(np.float64), 2e-5, f"importance f32 case {j} vs reference")
        keep = orc.keep_count(0.6, c["N"])
        idx, nxt = ops.select_topk(got, keep)
        np.testing.assert_array_equal(idx.cpu().numpy(), orc.select_tokens(host(got), keep))
        s2, idx2, nxt2 = ops.score_select(dev(qkv), c["H"], keep) if c["D"] in (32, 64, 128) else (None, None, None)
        assert torch.equal(idx2, idx) and torch.equal(s2, got) and torch.equal(nxt2, nxt)


@pytest.mark.parametrize("B,N,Np,H", [(2, 197, 173, 3), (1, 577, 404, 2), (3, 17, 13, 2), (2, 87, 87, 3), (1, 70, 2, 1)])
def test_attention_f32(B, N, Np, H):
    rng = np.random.default_rng(N + Np)
    qkv = rng.standard_normal((B, N, 3 * H * 64), dtype=np.float32)
    if Np == N:
        idx_t, g = None, qkv
    else:
        idx = np.stack([np.concatenate([[0], 1 + np.sort(rng.choice(N - 1, Np - 1, replace=False))]) for _ in range(B)])
        idx_t = torch.from_numpy(idx.astype(np.int32)).to(DEV)
        g = orc.gather_rows(qkv, idx.astype(np.int64))
    out = ops.attention(dev(qkv), idx_t, H, 64 ** -0.5)
    q, k, v = orc.split_heads(g.astype(np.float64), H)
    close(host(out), orc.softmax_attention(q, k, v, 64 ** -0.5), 5e-6, "attention f32")


def test_layernorm_and_patch_embed_f32():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((37, 768), dtype=np.float32) * 2 + 0.5
    w = (1 + 0.1 * rng.standard_normal(768)).astype(np.float32)
    b = (0.1 * rng.standard_normal(768)).astype(np.float32)
    y = ops.layernorm(dev(x), dev(w), dev(b), 1e-6, out_dtype=torch.float32)
    close(host(y), orc.layer_norm(x.astype(np.float64), w, b, 1e-6), 3e-6, "layernorm f32")
    for S, P, Cc, B in [(64, 16, 128, 3), (56, 14, 128, 3), (30, 10, 64, 2)]:   # fused loader / materialised columns
        img = rng.standard_normal((B, 3, S, S), dtype=np.float32)
        wc = rng.standard_normal((Cc, 3, P, P), dtype=np.float32) * 0.05
        bc = rng.standard_normal(Cc, dtype=np.float32) * 0.1
        cls = rng.standard_normal(Cc, dtype=np.float32)
        pos = rng.standard_normal(((S // P) ** 2 + 1, Cc), dtype=np.float32)
        xx = ops.patch_embed(dev(img), ops.pack_weight(dev(wc), torch.float32, k_multiple=64), dev(bc), dev(cls), dev(pos), True, P, Cc, out_f32=True)
        tok = orc.patch_embed(img.astype(np.float64), wc.astype(np.float64), bc.astype(np.float64))
        want = np.concatenate([np.broadcast_to(cls, (B, 1, Cc)), tok], axis=1) + pos[None]
        close(host(xx), want, 3e-6, f"patch embed f32 P={P}")


@pytest.mark.parametrize("name", ["micro_fp32", "tiny224_fp32", "base224_fp32", "deit3_fp32", "large384_fp32", "microd80_fp32", "microp14_fp32"])
def test_forward_f32_equals_reference(name):
    """Free-running fp32 forward vs the reference's fp32 run: same keep_idx (bit exact wherever the
    reference's own boundary gap exceeds fp32 noise), same token_counts, logits within 1e-3."""
    meta, data = load_case(name)
    cfg = ts.CONFIGS[meta["cfg_name"]]
    model = ts.create_model(cfg, seed=meta["seed"], std=meta["std"], bias_std=meta["bias_std"], round_bf16=True)
    wrapped = rajni_amd.RAJNIViTWrapper(model, meta["schedule"]).to(DEV).eval()      # fp32 model, README.md:33-34
    wrapped.trace_scores(True)
    images = torch.from_numpy(case_images(meta, data)).to(DEV)
    logits = wrapped(images)
    assert logits.dtype == torch.float32
    assert wrapped.get_last_stats()["token_counts"] == data["token_counts"].tolist()
    tr = wrapped.get_last_trace()
    all_equal = True
    for i in pruned_blocks(meta):
        got_idx = tr[i]["keep_idx"].cpu().numpy()
        ref_idx = data[f"blk{i}.keep_idx"]
        if meta["boundary_gap"][i] > 2e-7:
            np.testing.assert_array_equal(got_idx, ref_idx)
        else:  # the reference's own top-k gap is below fp32 resolution of the scores: allow that one swap
            assert np.mean(got_idx == ref_idx) > 0.98
        all_equal &= bool(np.array_equal(got_idx, ref_idx))
        if all_equal:
            close(host(tr[i]["scores"]), data[f"blk{i}.scores"].astype(np.float64), 5e-4, f"{name} blk{i} scores")
    ref = data["logits"].astype(np.float64)
    if all_equal:
        close(host(logits), ref, 1e-3, f"{name} logits")
        err = np.abs(host(logits) - ref).max() / np.abs(ref).max()
        assert err < 2e-4, err     # in practice two orders inside the bar
