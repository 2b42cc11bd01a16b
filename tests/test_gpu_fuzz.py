"""Randomised differential test of rajni_linear (tools/fuzz_linear.py) against torch fp32 matmul: random
shapes around the tiling thresholds, every epilogue, gathered / in-place residuals, both residual-stream
precisions, fp8 weights, forced tilings and tile orders.  A fixed seed keeps the run reproducible."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_linear_fuzz_against_torch():
    spec = importlib.util.spec_from_file_location("fuzz_linear", os.path.join(ROOT, "tools", "fuzz_linear.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(cases=250, seed=2026, verbose=False) == 0


def test_linear_fuzz_round2_forms_against_torch():
    """fp8 x fp8 launches (both tilings, every epilogue, ragged shapes, M from 1 up; tools/fuzz_linear_r2.py)."""
    spec = importlib.util.spec_from_file_location("fuzz_linear_r2", os.path.join(ROOT, "tools", "fuzz_linear_r2.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(cases=150, seed=2026, verbose=False) == 0


def test_attention_and_selection_fuzz_against_torch():
    spec = importlib.util.spec_from_file_location("fuzz_attention", os.path.join(ROOT, "tools", "fuzz_attention.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(cases=80, seed=2026, verbose=False) == 0


def test_forward_fuzz_random_models_and_schedules_against_oracle():
    """Random small timm-shaped models (width, depth, heads, LayerScale, no_embed_class, image size), random
    pruning schedules (ratios, update flags, consecutive stages) and batch sizes: the whole HIP forward against
    the oracle (the numpy restatement of the reference) run with the device's selections injected, plus the
    exact selection rule and the token counts."""
    import numpy as np
    import torch
    import rajni_amd
    from oracle import rajni_oracle as orc
    from rajni_amd import timm_shaped as ts

    rng = np.random.default_rng(77)
    for it in range(12):
        heads = int(rng.choice([2, 3, 4, 6]))
        depth = int(rng.integers(2, 7))
        cfg = ts.ViTConfig(img_size=int(rng.choice([32, 64, 96])), embed_dim=64 * heads, depth=depth, num_heads=heads,
                           num_classes=int(rng.choice([10, 100, 1000])),
                           layer_scale=(0.5 if rng.random() < 0.4 else None), no_embed_class=bool(rng.random() < 0.4))
        n0 = (cfg.img_size // 16) ** 2 + 1
        sched, n = {}, n0
        for blk in sorted(rng.choice(depth, size=int(rng.integers(0, depth + 1)), replace=False).tolist()):
            r = float(rng.choice([0.3, 0.5, 0.72, 0.88, 1.0]))
            if n <= 2:
                break
            sched[int(blk)] = {"keep_ratio": r, "update": bool(rng.random() < 0.7)}
            n = orc.keep_count(r, n) + 1
        seed = int(rng.integers(0, 1000))
        model = ts.create_model(cfg, seed=seed, std=0.08, bias_std=0.02, round_bf16=True)
        sd = ts.state_dict_numpy(model)
        B = int(rng.integers(1, 10))
        imgs = ts.bf16_round_np(rng.standard_normal((B, 3, cfg.img_size, cfg.img_size), dtype=np.float32))
        wrapped = rajni_amd.RAJNIViTWrapper(model, sched).to("cuda").to(torch.bfloat16).eval().trace_scores(True)
        logits = wrapped(torch.from_numpy(imgs).to("cuda")).float().cpu().numpy()
        forced = {}
        for i, d in wrapped.get_last_trace().items():
            idx = d["keep_idx"].cpu().numpy()
            np.testing.assert_array_equal(idx, orc.select_tokens(d["scores"].float().cpu().numpy().astype(np.float64), idx.shape[1] - 1))
            forced[i] = idx
        want, stats = orc.vit_forward(sd, imgs, sched, depth=cfg.depth, num_heads=cfg.num_heads, ln_eps=cfg.ln_eps,
                                      forced_keep=forced, dtype=np.float32)
        assert wrapped.get_last_stats() == stats, (it, sched)
        err, scale = np.abs(logits - want).max(), np.abs(want).max()
        assert err <= 1.5e-2 * scale, f"case {it}: cfg={cfg} sched={sched} B={B}: |dlogit| {err:.4g} vs scale {scale:.4g}"
