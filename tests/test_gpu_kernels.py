"""Parity of every HIP kernel, called THROUGH THE C ABI (rajni_amd.ops -> librajni_hip.so), against
the CPU oracle on the same seeded inputs.  GPU box only (`-m gpu`).

Tolerances: integer/index results are bit exact.  bf16 results are compared with the fp64 oracle
evaluated on the same bf16-representable inputs; the only legitimate difference is fp32
accumulation order plus ONE bf16 rounding of the output (rel 2^-9 = 0.2 %), so `rtol=1e-2` of the
tensor's scale is the bar BASELINE.json states for bf16.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import rajni_oracle as orc
from rajni_amd import ops, _native as nat
from rajni_amd.timm_shaped import bf16_round_np
from helpers import GOLDEN

DEV = "cuda"


def dev_bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(torch.bfloat16)


def host(t):
    return t.float().cpu().numpy().astype(np.float64)


def close(got, want, rel=1e-2, what=""):
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max()
    assert err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g} (rel {err / scale:.3g})"


def test_device_is_gfx950():
    nat.check(nat.lib().rajni_device_check(), "rajni_device_check")


# ---------------------------------------------------------------------------------------------
# GEMM + epilogues
# ---------------------------------------------------------------------------------------------

@pytest.fixture(params=[1, 4, 5], ids=["small128x128", "wide256x256", "mid256x128"])
def tiling(request):
    """Every GEMM tiling must give the same answers: force each one (rajni_debug_force_gemm_tiling)."""
    nat.lib().rajni_debug_force_gemm_tiling(request.param)
    yield request.param
    nat.lib().rajni_debug_force_gemm_tiling(0)


@pytest.mark.parametrize("M,N,K", [(394, 2304, 768), (256, 768, 768), (130, 3072, 768), (346, 768, 3072),
                                   (7, 1000, 768), (64, 10, 128), (1, 192, 192), (1154, 576, 192), (2100, 384, 64),
                                   (513, 260, 128)])
def test_linear_bias(M, N, K, tiling):
    rng = np.random.default_rng(M * 7 + N)
    x = bf16_round_np(rng.standard_normal((M, K), dtype=np.float32))
    w = bf16_round_np(rng.standard_normal((N, K), dtype=np.float32) * 0.05)
    b = bf16_round_np(rng.standard_normal(N, dtype=np.float32))
    y = ops.linear(dev_bf16(x), ops.pack_weight(dev_bf16(w)), N, torch.from_numpy(b).to(DEV), nat.EPI_BIAS)
    want = x.astype(np.float64) @ w.astype(np.float64).T + b
    assert tuple(y.shape) == (M, N)
    close(host(y), want, what=f"linear {M}x{N}x{K}")


def test_linear_identity_asymmetric(tiling):
    """A = I against an ASYMMETRIC weight: catches a transposed / permuted output mapping exactly."""
    K = N = 256
    M = 256
    x = np.eye(M, K, dtype=np.float32)
    w = (np.arange(N * K, dtype=np.float32).reshape(N, K) % 251) - 125.0   # exact in bf16
    w = bf16_round_np(w)
    y = ops.linear(dev_bf16(x), ops.pack_weight(dev_bf16(w)), N, None, nat.EPI_BIAS)
    np.testing.assert_array_equal(host(y), w.T.astype(np.float64))


def test_linear_gelu(tiling):
    rng = np.random.default_rng(5)
    M, N, K = 300, 512, 256
    x = bf16_round_np(rng.standard_normal((M, K), dtype=np.float32))
    w = bf16_round_np(rng.standard_normal((N, K), dtype=np.float32) * 0.1)
    b = bf16_round_np(rng.standard_normal(N, dtype=np.float32) * 0.1)
    y = ops.linear(dev_bf16(x), ops.pack_weight(dev_bf16(w)), N, torch.from_numpy(b).to(DEV), nat.EPI_BIAS_GELU)
    want = orc.gelu(x.astype(np.float64) @ w.astype(np.float64).T + b)
    close(host(y), want, what="linear+gelu")


@pytest.mark.parametrize("stream_f32", [False, True])
@pytest.mark.parametrize("gather", [False, True])
def test_linear_resid_layerscale(gather, stream_f32, tiling):
    rng = np.random.default_rng(9)
    B, Nsrc, Np, Cc, K = 3, 50, 37, 256, 192
    x = bf16_round_np(rng.standard_normal((B, Np if gather else Nsrc, K), dtype=np.float32))
    w = bf16_round_np(rng.standard_normal((Cc, K), dtype=np.float32) * 0.1)
    b = bf16_round_np(rng.standard_normal(Cc, dtype=np.float32) * 0.1)
    gam = bf16_round_np(rng.standard_normal(Cc, dtype=np.float32))
    resid = bf16_round_np(rng.standard_normal((B, Nsrc, Cc), dtype=np.float32))
    idx = np.stack([np.sort(rng.choice(Nsrc, Np, replace=False)) for _ in range(B)]).astype(np.int32)
    if stream_f32:   # fp32 residual stream: residual values need not be bf16 representable
        resid = resid + rng.standard_normal(resid.shape, dtype=np.float32) * 1e-3
    rdev = torch.from_numpy(resid).to(DEV) if stream_f32 else dev_bf16(resid)
    y = ops.linear(dev_bf16(x), ops.pack_weight(dev_bf16(w)), Cc, torch.from_numpy(b).to(DEV), nat.EPI_BIAS_RESID,
                   gamma=torch.from_numpy(gam).to(DEV), resid=rdev,
                   r_idx=torch.from_numpy(idx).to(DEV) if gather else None)
    lin = x.astype(np.float64) @ w.astype(np.float64).T + b
    r = orc.gather_rows(resid.astype(np.float64), idx.astype(np.int64)) if gather else resid
    want = r + gam * lin
    assert y.dtype == (torch.float32 if stream_f32 else torch.bfloat16)
    close(host(y).reshape(want.shape), want, rel=1e-5 if stream_f32 else 1e-2, what="linear+resid")


@pytest.mark.parametrize("K", [576, 768, 2112, 3072])
@pytest.mark.parametrize("gather", [False, True])
def test_linear_resid_fp32_stream_many_tiles(gather, K, tiling):
    """The proj / fc2 shapes of the forward: fp32 residual stream, K long enough for the 256x128 tiling's
    deferred epilogue I/O (residual rows loaded during the K loop, outputs stored during the NEXT tile's
    loop), several tiles per persistent workgroup, ragged last row tile, in-place (y = resid) when not
    gathered - exactly how forward.hip calls it."""
    rng = np.random.default_rng(K + gather)
    B, Nsrc, Np, Cc = 150, 197, 173, 768
    rows = Np if gather else Nsrc
    x = torch.randn(B, rows, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(Cc, K, device=DEV) * 0.05).to(torch.bfloat16)
    b = torch.randn(Cc, device=DEV).to(torch.bfloat16).float()
    resid = torch.randn(B, Nsrc, Cc, device=DEV)
    idx = torch.from_numpy(np.stack([np.sort(rng.choice(Nsrc, Np, replace=False)) for _ in range(B)]).astype(np.int32)).to(DEV)
    lin = x.float().reshape(-1, K) @ w.float().T + b          # fp32 torch reference (TF32 off by default on ROCm)
    if gather:
        r = torch.gather(resid, 1, idx.long()[:, :, None].expand(-1, -1, Cc))
        y = ops.linear(x, ops.pack_weight(w), Cc, b, nat.EPI_BIAS_RESID, resid=resid, r_idx=idx)
    else:
        r = resid.clone()
        out = resid.reshape(-1, Cc)                           # in place, like fc2 in the forward
        y = ops.linear(x, ops.pack_weight(w), Cc, b, nat.EPI_BIAS_RESID, resid=resid, out=out)
    want = (r.reshape(-1, Cc).double() + lin.double()).cpu().numpy()
    got = y.reshape(-1, Cc).double().cpu().numpy()
    err = np.abs(got - want).max()
    assert err <= 2e-4 * np.abs(want).max(), f"max err {err:.3g}"


@pytest.mark.parametrize("nblk", [1, 2, 4, 5, 7])
def test_linear_tile_order_blocks_do_not_change_results(nblk, tiling):
    """The persistent tilings walk their tiles in N blocks (an XCD keeps its W slice in L2).  Any block size
    must give bit-identical results to the plain order: same tiles, same K order, different schedule.
    2816 columns = 11 (256-wide) or 22 (128-wide) column tiles, so the last block is ragged for every size."""
    M, N, K = 2900, 2816, 256
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(N, K, device=DEV) * 0.1).to(torch.bfloat16))
    b = torch.randn(N, device=DEV)
    nat.lib().rajni_debug_set_gemm_nblock_bytes(0)
    try:
        ref = ops.linear(x, w, N, b, nat.EPI_BIAS_GELU).clone()
        nat.lib().rajni_debug_set_gemm_nblock_bytes(-nblk)
        got = ops.linear(x, w, N, b, nat.EPI_BIAS_GELU)
        assert torch.equal(got, ref)
    finally:
        nat.lib().rajni_debug_set_gemm_nblock_bytes(1600 * 1024)
    want = orc.gelu(x.double().cpu().numpy() @ w[:N].double().cpu().numpy().T + b.double().cpu().numpy())
    close(host(got), want, what="blocked tile order")


# ---------------------------------------------------------------------------------------------
# LayerNorm, gather
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("x_f32", [False, True])
@pytest.mark.parametrize("rows,Cc", [(394, 768), (5, 192), (33, 1024), (2, 128),
                                     (4097, 768), (5001, 1024), (4100, 384), (4096, 1280)])   # >= 4096 rows: the two-rows-per-wave kernel (C <= 1024)
def test_layernorm(rows, Cc, x_f32):
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, Cc), dtype=np.float32) * 2 + 0.5
    if not x_f32:
        x = bf16_round_np(x)
    w = bf16_round_np(1 + 0.1 * rng.standard_normal(Cc, dtype=np.float32))
    b = bf16_round_np(0.1 * rng.standard_normal(Cc, dtype=np.float32))
    xd = torch.from_numpy(x).to(DEV) if x_f32 else dev_bf16(x)
    y = ops.layernorm(xd, torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV), 1e-6)
    assert y.dtype == torch.bfloat16
    close(host(y), orc.layer_norm(x.astype(np.float64), w, b, 1e-6), what="layernorm")


def test_layernorm_strided_cls_rows():
    rng = np.random.default_rng(1)
    B, N, Cc = 6, 11, 256
    x = bf16_round_np(rng.standard_normal((B, N, Cc), dtype=np.float32))
    w = np.ones(Cc, np.float32)
    b = np.zeros(Cc, np.float32)
    y = ops.layernorm(dev_bf16(x), torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV), 1e-6, rows=B,
                      row_stride=N * Cc)
    close(host(y), orc.layer_norm(x[:, 0].astype(np.float64), w, b, 1e-6), what="layernorm cls rows")


def test_gather_rows_bit_exact():
    rng = np.random.default_rng(2)
    B, N, K, E = 4, 197, 173, 2304
    src = dev_bf16(rng.standard_normal((B, N, E), dtype=np.float32))
    idx = np.stack([np.sort(rng.choice(N, K, replace=False)) for _ in range(B)]).astype(np.int32)
    got = ops.gather_rows(src, torch.from_numpy(idx).to(DEV))
    want = torch.gather(src, 1, torch.from_numpy(idx).long().to(DEV).unsqueeze(-1).expand(-1, -1, E))
    assert torch.equal(got, want)


# ---------------------------------------------------------------------------------------------
# importance + selection
# ---------------------------------------------------------------------------------------------

def test_importance_golden_cases():
    with open(os.path.join(GOLDEN, "importance_cases.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, "importance_cases.npz"))
    rng = np.random.default_rng(meta["seed"])
    for j, c in enumerate(meta["cases"]):
        qkv = bf16_round_np(rng.standard_normal((c["B"], c["N"], 3 * c["H"] * c["D"]), dtype=np.float32) * c["scale"])
        got = host(ops.importance(dev_bf16(qkv), c["H"]))
        ref = data[f"c{j}.scores"].astype(np.float64)     # the reference's own fp32 answer
        want = orc.importance_scores(qkv, c["H"])
        close(got, want, rel=6e-3, what=f"importance case {j} vs oracle")
        close(got, ref, rel=6e-3, what=f"importance case {j} vs reference fixture")
        # what the kernel returns is exactly bf16(fp32 score): compare pre-rounding via ulp bound
        assert np.all(np.abs(got - want) <= np.abs(want) * 2.0 ** -8 + 1e-12)


@pytest.mark.parametrize("N,keep", [(197, 172), (173, 151), (152, 120), (121, 86), (577, 403), (61, 1), (2, 1), (50, 49)])
def test_select_bit_exact(N, keep):
    rng = np.random.default_rng(N + keep)
    B = 5
    s = bf16_round_np(rng.random((B, N), dtype=np.float32) * 1e-2)      # bf16 -> many exact ties
    idx, nxt = ops.select_topk(dev_bf16(s), keep)
    want = orc.select_tokens(s, keep)
    np.testing.assert_array_equal(idx.cpu().numpy(), want)
    np.testing.assert_array_equal(host(nxt), np.take_along_axis(s.astype(np.float64), want, axis=1))


def test_select_degenerate_rows():
    B, N = 3, 64
    s = np.ones((B, N), np.float32)                     # all equal: lowest indices win
    idx, _ = ops.select_topk(dev_bf16(s), 10)
    np.testing.assert_array_equal(idx.cpu().numpy(), np.tile(np.arange(11), (B, 1)))
    s = np.linspace(0, 1, N, dtype=np.float32)[None].repeat(B, 0)
    s[:, 7] = np.nan                                    # NaN ranks first
    s = bf16_round_np(s)
    t = torch.from_numpy(s).to(DEV).to(torch.bfloat16)
    idx, nxt = ops.select_topk(t, 3)
    np.testing.assert_array_equal(idx.cpu().numpy(), orc.select_tokens(s, 3))
    assert 7 in idx[0].tolist()


@pytest.mark.parametrize("N", [2, 5, 64, 129, 197, 258, 577, 1030])
@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_select_special_values_and_slice_boundaries(N, dt):
    """The bf16 path ranks on packed 32-bit keys (sortable16(score) << 16 | 0xFFFF - index): signs, +-0, +-inf, NaN, denormals
    and runs of ties - placed so that they straddle the slice and 16-byte chunk boundaries of the count loop for every
    lanes-per-token setting (N = 2 ... 1030: 8, 4, 2 and 1 lanes per token; several rank-loop iterations) - must select
    exactly what the defined rule selects (larger first, then lower index; NaN = +inf; -0 == +0), every keep count class."""
    rng = np.random.default_rng(N)
    B = 6
    vals = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-38, -1e-38, 3.0e-3, 3.0e-3, 2.0 ** -7, -(2.0 ** -7), 0.5, 0.5],
                    dtype=np.float32)
    s = rng.choice(vals, size=(B, N)).astype(np.float32)
    s[1] = np.where(rng.random(N) < 0.5, np.float32(0.0), np.float32(-0.0))     # only zeros of both signs: index order decides
    s[2] = rng.standard_normal(N).astype(np.float32)                              # plain signed values
    s[2, :: 3] = s[2, 0]                                                          # ... with a long run of ties
    if dt == "bf16":
        s = bf16_round_np(s)
        t = torch.from_numpy(s).to(DEV).to(torch.bfloat16)
    else:
        t = torch.from_numpy(s).to(DEV)
    for keep in sorted({1, max(1, (N - 1) // 2), max(1, N - 2), N - 1}):
        idx, nxt = ops.select_topk(t, keep)
        want = orc.select_tokens(s, keep)
        np.testing.assert_array_equal(idx.cpu().numpy(), want, err_msg=f"N={N} keep={keep} {dt}")
        got_next = nxt.float().cpu().numpy()
        exp_next = np.take_along_axis(s, want, axis=1)
        assert np.array_equal(got_next, exp_next, equal_nan=True)


@pytest.mark.parametrize("B,N,H", [(4, 197, 12), (2, 577, 16), (3, 17, 2), (2, 87, 3)])
def test_score_select_fused(B, N, H):
    rng = np.random.default_rng(N * H)
    qkv = bf16_round_np(rng.standard_normal((B, N, 3 * H * 64), dtype=np.float32))
    keep = orc.keep_count(0.7, N)
    scores, idx, nxt = ops.score_select(dev_bf16(qkv), H, keep)
    s = host(scores)
    close(s, orc.importance_scores(qkv, H), rel=6e-3, what="fused scores")
    # the fused selection is exactly the defined rule applied to the scores it returned
    np.testing.assert_array_equal(idx.cpu().numpy(), orc.select_tokens(s, keep))
    np.testing.assert_array_equal(host(nxt), np.take_along_axis(s, idx.cpu().numpy().astype(np.int64), axis=1))
    # and identical to the two-step path
    s2 = ops.importance(dev_bf16(qkv), H)
    assert torch.equal(s2, scores)


@pytest.mark.parametrize("B,N,H,D,dt", [(5, 197, 12, 64, "bf16"), (3, 404, 16, 64, "bf16"), (2, 173, 6, 64, "f32"),
                                         (4, 87, 4, 32, "bf16"), (2, 152, 2, 128, "bf16"), (3, 61, 4, 80, "bf16")])
def test_score_select_one_pass_equals_two_pass(B, N, H, D, dt):
    """The kernel reads K and V in ONE pass when logits and vbar both fit in LDS, in two passes otherwise (vbar
    reusing the logits' region; N = 577 x 16 heads).  Every score is the same fixed-order fp32 sum in both
    layouts: scores, selections and carried scores must be BIT-identical."""
    rng = np.random.default_rng(N + H)
    qkv = rng.standard_normal((B, N, 3 * H * D), dtype=np.float32)
    t = dev_bf16(bf16_round_np(qkv)) if dt == "bf16" else torch.from_numpy(qkv).to(DEV)
    keep = orc.keep_count(0.8, N)
    try:
        nat.lib().rajni_debug_force_score_two_pass(1)
        s2, i2, n2 = ops.score_select(t, H, keep)
        only2 = ops.importance(t, H)
    finally:
        nat.lib().rajni_debug_force_score_two_pass(0)
    s1, i1, n1 = ops.score_select(t, H, keep)
    assert torch.equal(s1, s2) and torch.equal(i1, i2) and torch.equal(n1, n2)
    assert torch.equal(ops.importance(t, H), only2) and torch.equal(only2, s1)


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------

@pytest.fixture(params=[0, 1, 2], ids=["persistent", "online_chunked", "full_row"])
def attn_mode(request):
    """Both attention kernels (np <= 256 may use either) must agree with the oracle."""
    nat.lib().rajni_debug_force_attention(request.param)
    yield request.param
    nat.lib().rajni_debug_force_attention(0)


@pytest.mark.parametrize("B,N,Np,H", [(2, 197, 173, 12), (1, 577, 404, 16), (3, 17, 13, 2), (2, 87, 87, 3),
                                      (1, 130, 129, 1), (2, 40, 2, 2), (1, 300, 257, 2), (2, 256, 256, 2),
                                      (2, 260, 225, 1), (1, 152, 121, 3), (1, 40, 33, 2)])
def test_attention_packed(B, N, Np, H, attn_mode):
    rng = np.random.default_rng(N * 31 + Np)
    Cc = H * 64
    qkv = bf16_round_np(rng.standard_normal((B, N, 3 * Cc), dtype=np.float32))
    if Np == N:
        idx, idx_t = None, None
        g = qkv
    else:
        idx = np.stack([np.concatenate([[0], 1 + np.sort(rng.choice(N - 1, Np - 1, replace=False))]) for _ in range(B)])
        idx_t = torch.from_numpy(idx.astype(np.int32)).to(DEV)
        g = orc.gather_rows(qkv, idx.astype(np.int64))
    out = ops.attention(dev_bf16(qkv), idx_t, H, 64 ** -0.5)
    q, k, v = orc.split_heads(g.astype(np.float64), H)
    want = orc.softmax_attention(q, k, v, 64 ** -0.5)
    assert tuple(out.shape) == (B, Np, Cc)
    close(host(out), want, rel=1.5e-2, what="attention")


def test_attention_online_softmax_spike(attn_mode):
    """Force the running-max rescale: one late key dominates (guide rule 26)."""
    rng = np.random.default_rng(3)
    B, N, H = 1, 200, 1
    qkv = rng.standard_normal((B, N, 192), dtype=np.float32) * 0.3
    qkv[0, 5, 0:64] = 4.0          # query row 5
    qkv[0, 170, 64:128] = 4.0      # key 170 (in the second LDS chunk) aligns with it
    qkv = bf16_round_np(qkv)
    out = ops.attention(dev_bf16(qkv), None, H, 64 ** -0.5)
    q, k, v = orc.split_heads(qkv.astype(np.float64), H)
    close(host(out), orc.softmax_attention(q, k, v, 64 ** -0.5), rel=1.5e-2, what="attention spike")


# ---------------------------------------------------------------------------------------------
# patch embed
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("out_f32", [False, True])
@pytest.mark.parametrize("S,P,Cc,B,has_cls", [(64, 16, 128, 3, True), (224, 16, 192, 2, True), (64, 16, 128, 2, False),
                                              (32, 8, 64, 5, True),
                                              # not a power of two / K not whole K steps: materialised columns
                                              (56, 14, 128, 3, True), (224, 14, 320, 2, False), (70, 10, 64, 2, True),
                                              (28, 7, 64, 9, True), (36, 4, 128, 2, True), (96, 32, 192, 2, True),
                                              (60, 12, 64, 1, True)])
def test_patch_embed(S, P, Cc, B, has_cls, out_f32, tiling):
    rng = np.random.default_rng(S + Cc)
    img = bf16_round_np(rng.standard_normal((B, 3, S, S), dtype=np.float32))
    w = bf16_round_np(rng.standard_normal((Cc, 3, P, P), dtype=np.float32) * 0.05)
    b = bf16_round_np(rng.standard_normal(Cc, dtype=np.float32) * 0.1)
    cls = bf16_round_np(rng.standard_normal(Cc, dtype=np.float32))
    npatch = (S // P) ** 2
    pos = bf16_round_np(rng.standard_normal((npatch + int(has_cls), Cc), dtype=np.float32))
    x = ops.patch_embed(dev_bf16(img), ops.pack_weight(dev_bf16(w), k_multiple=64), torch.from_numpy(b).to(DEV), dev_bf16(cls),
                        dev_bf16(pos), has_cls, P, Cc, out_f32=out_f32)
    tok = orc.patch_embed(img.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    if has_cls:
        want = np.concatenate([np.broadcast_to(cls, (B, 1, Cc)), tok], axis=1) + pos[None]
    else:
        want = np.concatenate([np.broadcast_to(cls, (B, 1, Cc)), tok + pos[None]], axis=1)
    close(host(x), want, rel=1e-5 if out_f32 else 1e-2, what="patch embed")


@pytest.mark.parametrize("Cin,S,P", [(1, 64, 16), (4, 32, 8), (1, 56, 14), (4, 28, 7), (2, 64, 32)])
def test_patch_embed_other_channel_counts(Cin, S, P):
    """in_chans other than 3 (grayscale, RGBA / multispectral): both the fused loader and the column path"""
    rng = np.random.default_rng(Cin * 100 + P)
    B, Cc = 3, 128
    img = bf16_round_np(rng.standard_normal((B, Cin, S, S), dtype=np.float32))
    w = bf16_round_np(rng.standard_normal((Cc, Cin, P, P), dtype=np.float32) * 0.05)
    b = bf16_round_np(rng.standard_normal(Cc, dtype=np.float32) * 0.1)
    cls = bf16_round_np(rng.standard_normal(Cc, dtype=np.float32))
    pos = bf16_round_np(rng.standard_normal(((S // P) ** 2 + 1, Cc), dtype=np.float32))
    x = ops.patch_embed(dev_bf16(img), ops.pack_weight(dev_bf16(w), k_multiple=64), torch.from_numpy(b).to(DEV), dev_bf16(cls),
                        dev_bf16(pos), True, P, Cc, out_f32=True)
    tok = orc.patch_embed(img.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    want = np.concatenate([np.broadcast_to(cls, (B, 1, Cc)), tok], axis=1) + pos[None]
    close(host(x), want, rel=1e-5, what=f"patch embed Cin={Cin} P={P}")
