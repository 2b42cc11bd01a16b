"""fp8 (e4m3) weight path - BASELINE.json configs[4], SURVEY 8(f)-4.  GPU box only (`-m gpu`).

Semantics under test: weights of qkv/proj/fc1/fc2 are stored as e4m3 bytes + one fp32 scale per output
row; the kernels compute  epi(x . (q * s))  with bf16 activations and fp32 accumulation, i.e. exactly
the model run with the DEQUANTISED weights.  The oracle therefore gets q*s (fp64) as its weights and the
tolerances are the ordinary bf16-activation ones - quantisation error itself is not part of the parity
budget (it is reported separately by `test_fp8_forward_quantisation_cost`).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import rajni_amd
from oracle import rajni_oracle as orc
from rajni_amd import _native as nat, ops, timm_shaped as ts
from rajni_amd.timm_shaped import bf16_round_np
from helpers import load_case, case_state_dict, case_images, pruned_blocks

DEV = "cuda"


def dev_bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(torch.bfloat16)


def host(t):
    return t.float().cpu().numpy().astype(np.float64)


def close(got, want, rel, what):
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g}"


@pytest.fixture(params=[1, 4, 5], ids=["small128x128", "wide256x256", "mid256x128"])
def tiling(request):
    nat.lib().rajni_debug_force_gemm_tiling(request.param)
    yield request.param
    nat.lib().rajni_debug_force_gemm_tiling(0)


def quant(w_np):
    """(packed uint8 [pad256(N),K] on device, scale fp32 [N] on device, dequantised fp64 [N,K] on host)"""
    wq, sc = ops.pack_weight_fp8(torch.from_numpy(w_np), torch.bfloat16, DEV)
    deq = ops.dequantize_fp8(wq.cpu(), sc.cpu()).numpy().astype(np.float64)
    return wq, sc, deq


def test_fp8_decode_every_code_exact(tiling):
    """X = I against a weight holding EVERY finite e4m3 code: the in-register fp8 -> bf16 conversion, the
    64-byte-row LDS swizzle and the output column mapping are all exact or this fails bit-wise."""
    N, K, M = 256, 256, 256
    codes = np.array([c for c in range(256) if c & 0x7F != 0x7F], dtype=np.uint8)   # 0x7f / 0xff are NaN
    rng = np.random.default_rng(3)
    q = codes[rng.integers(0, len(codes), size=(N, K))]
    q[0, :len(codes)] = codes                     # every code appears at least once
    scale = (2.0 ** rng.integers(-3, 4, size=N)).astype(np.float32)   # powers of two: products stay exact
    wq = torch.from_numpy(q).to(DEV)
    want = torch.from_numpy(q).view(torch.float8_e4m3fn).to(torch.float64).numpy() * scale[:, None].astype(np.float64)
    want = bf16_round_np(want.astype(np.float32)).astype(np.float64)   # output dtype is bf16 (e4m3 x 2^k is exact there)
    y = ops.linear(dev_bf16(np.eye(M, K, dtype=np.float32)), wq, N, None, nat.EPI_BIAS, w_scale=torch.from_numpy(scale).to(DEV))
    np.testing.assert_array_equal(host(y), want.T)


@pytest.mark.parametrize("M,N,K", [(394, 2304, 768), (256, 768, 768), (130, 3072, 768), (346, 768, 3072),
                                   (1, 1000, 192), (1100, 520, 256), (1300, 1536, 320)])
def test_fp8_linear_bias(M, N, K, tiling):
    rng = np.random.default_rng(M * 7 + N)
    x = bf16_round_np(rng.standard_normal((M, K), dtype=np.float32))
    w = bf16_round_np(rng.standard_normal((N, K), dtype=np.float32) * 0.05 * (1 + 3 * rng.random((N, 1), dtype=np.float32)))
    b = bf16_round_np(rng.standard_normal(N, dtype=np.float32))
    wq, sc, deq = quant(w)
    y = ops.linear(dev_bf16(x), wq, N, torch.from_numpy(b).to(DEV), nat.EPI_BIAS, w_scale=sc)
    want = x.astype(np.float64) @ deq.T + b
    assert tuple(y.shape) == (M, N)
    close(host(y), want, 1e-2, f"fp8 linear {M}x{N}x{K}")
    # and the quantiser itself: e4m3 has 3 mantissa bits -> relative step 2^-3, RNE error <= 2^-4 of the value
    assert np.abs(deq - w).max() <= 2.0 ** -4 * np.abs(w).max(axis=1).max() * 1.001


def test_fp8_linear_gelu(tiling):
    rng = np.random.default_rng(5)
    M, N, K = 300, 512, 256
    x = bf16_round_np(rng.standard_normal((M, K), dtype=np.float32))
    w = bf16_round_np(rng.standard_normal((N, K), dtype=np.float32) * 0.1)
    b = bf16_round_np(rng.standard_normal(N, dtype=np.float32) * 0.1)
    wq, sc, deq = quant(w)
    y = ops.linear(dev_bf16(x), wq, N, torch.from_numpy(b).to(DEV), nat.EPI_BIAS_GELU, w_scale=sc)
    close(host(y), orc.gelu(x.astype(np.float64) @ deq.T + b), 1e-2, "fp8 linear+gelu")


@pytest.mark.parametrize("stream_f32", [False, True])
@pytest.mark.parametrize("gather", [False, True])
def test_fp8_linear_resid_layerscale(gather, stream_f32, tiling):
    rng = np.random.default_rng(9)
    B, Nsrc, Np, Cc, K = 3, 50, 37, 256, 320
    x = bf16_round_np(rng.standard_normal((B, Np if gather else Nsrc, K), dtype=np.float32))
    w = bf16_round_np(rng.standard_normal((Cc, K), dtype=np.float32) * 0.1)
    b = bf16_round_np(rng.standard_normal(Cc, dtype=np.float32) * 0.1)
    gam = bf16_round_np(rng.standard_normal(Cc, dtype=np.float32))
    resid = bf16_round_np(rng.standard_normal((B, Nsrc, Cc), dtype=np.float32))
    idx = np.stack([np.sort(rng.choice(Nsrc, Np, replace=False)) for _ in range(B)]).astype(np.int32)
    if stream_f32:
        resid = resid + rng.standard_normal(resid.shape, dtype=np.float32) * 1e-3
    rdev = torch.from_numpy(resid).to(DEV) if stream_f32 else dev_bf16(resid)
    wq, sc, deq = quant(w)
    y = ops.linear(dev_bf16(x), wq, Cc, torch.from_numpy(b).to(DEV), nat.EPI_BIAS_RESID,
                   gamma=torch.from_numpy(gam).to(DEV), resid=rdev,
                   r_idx=torch.from_numpy(idx).to(DEV) if gather else None, w_scale=sc)
    lin = x.astype(np.float64) @ deq.T + b
    r = orc.gather_rows(resid.astype(np.float64), idx.astype(np.int64)) if gather else resid
    want = r + gam * lin
    assert y.dtype == (torch.float32 if stream_f32 else torch.bfloat16)
    close(host(y).reshape(want.shape), want, 1e-5 if stream_f32 else 1e-2, "fp8 linear+resid")


def test_fp8_rejects_fp32_activations():
    x = torch.zeros((4, 64), dtype=torch.float32, device=DEV)
    wq, sc = ops.pack_weight_fp8(torch.zeros((8, 64)), torch.bfloat16, DEV)
    with pytest.raises(NotImplementedError, match="fp8 weights need bf16"):
        ops.linear(x, wq, 8, None, nat.EPI_BIAS, w_scale=sc)


def _build(meta):
    cfg = ts.CONFIGS[meta["cfg_name"]]
    model = ts.create_model(cfg, seed=meta["seed"], std=meta["std"], bias_std=meta["bias_std"], round_bf16=True)
    wrapped = rajni_amd.RAJNIViTWrapper(model, meta["schedule"]).to(DEV).to(torch.bfloat16).eval()
    return cfg, wrapped


@pytest.mark.parametrize("name", ["micro_fp32", "base224_fp32", "deit3_fp32", "microd80_fp32", "microp14_fp32"])
def test_fp8_forward_vs_oracle_on_dequantised_weights(name):
    """Whole forward with fp8 block weights == the oracle (fp32 numpy restatement of the reference) run on the
    dequantised weights with the device's own selections injected: same 1e-2-of-logit-scale bar as bf16."""
    meta, data = load_case(name)
    cfg, wrapped = _build(meta)
    wrapped.set_weight_format("fp8").trace_scores(True)
    images_np = case_images(meta, data)
    logits = wrapped(torch.from_numpy(images_np).to(DEV)).float().cpu().numpy()
    assert wrapped.get_last_stats()["token_counts"] == data["token_counts"].tolist()
    tr = wrapped.get_last_trace()
    forced = {}
    for i in pruned_blocks(meta):
        s = tr[i]["scores"].float().cpu().numpy().astype(np.float64)
        idx = tr[i]["keep_idx"].cpu().numpy()
        np.testing.assert_array_equal(idx, orc.select_tokens(s, idx.shape[1] - 1))
        forced[i] = idx
    _, sd = case_state_dict(meta)
    for k, v in wrapped.dequantized_state_dict().items():
        assert sd[k].shape == tuple(v.shape)
        sd[k] = v.cpu().numpy()
    want, _ = orc.vit_forward(sd, images_np, meta["schedule"], depth=cfg.depth, num_heads=cfg.num_heads,
                              ln_eps=cfg.ln_eps, forced_keep=forced, dtype=np.float32)
    close(logits, want, 1e-2, f"{name} fp8 forward")
    assert (logits.argmax(1) == want.argmax(1)).all()


def test_fp8_forward_quantisation_cost():
    """What the fp8 weights themselves cost against the reference's un-quantised fixture (reported, loosely
    bounded): with the reference's selections injected the logits stay within 1.5e-1 of the logit scale
    (measured 7.9e-2 on the random-init ViT-B fixture; bf16 weights: 6.4e-3)."""
    meta, data = load_case("base224_fp32")
    cfg, wrapped = _build(meta)
    images = torch.from_numpy(case_images(meta, data)).to(DEV)
    wrapped.force_keep_idx({i: torch.from_numpy(data[f"blk{i}.keep_idx"]).to(DEV) for i in pruned_blocks(meta)})
    ref = data["logits"]
    scale = np.abs(ref).max()
    e_bf16 = np.abs(wrapped(images).float().cpu().numpy() - ref).max() / scale
    wrapped.set_weight_format("fp8")
    e_fp8 = np.abs(wrapped(images).float().cpu().numpy() - ref).max() / scale
    print(f"relative logit error vs reference fp32: bf16 weights {e_bf16:.4f}, fp8 weights {e_fp8:.4f}")
    assert e_bf16 <= 1e-2 and e_fp8 <= 1.5e-1
    wrapped.set_weight_format("model")
    assert np.abs(wrapped(images).float().cpu().numpy() - ref).max() / scale == e_bf16   # switching back is lossless
