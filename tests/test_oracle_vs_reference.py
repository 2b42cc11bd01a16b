"""The oracles against the LIVE reference (authoring container only: skipped where /root/reference is absent,
i.e. on the GPU box).  The committed fixtures in tests/golden/ pin the oracles on fixed seeds; this file re-runs the
comparison on fresh random inputs every time, for shapes the fixtures do not hold (other head dims, patch 14, LayerScale,
`update=False` stages, odd keep ratios).  The reference is imported read-only under an alias, exactly as
tests/golden/make_golden.py does; nothing of it is copied.
"""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from oracle import rajni_oracle as orc
from oracle import rajni_oracle_torch as ort
from rajni_amd import timm_shaped as ts

REF_DIR = "/root/reference/rajni"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_DIR), reason="the reference is only mounted in the authoring container")


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("rajni_ref", os.path.join(REF_DIR, "__init__.py"),
                                                  submodule_search_locations=[REF_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rajni_ref"] = mod
    spec.loader.exec_module(mod)
    import rajni_ref.wrapper as w
    return mod, w


@pytest.mark.parametrize("B,N,H,D", [(2, 197, 12, 64), (3, 50, 4, 80), (1, 257, 16, 80), (2, 17, 2, 32), (2, 33, 3, 128), (4, 2, 1, 8)])
def test_importance_on_fresh_inputs(ref, B, N, H, D):
    _, w = ref
    rng = np.random.default_rng(B * 1000 + N + D)
    qkv = rng.standard_normal((B, N, 3 * H * D), dtype=np.float32) * rng.choice([0.3, 1.0, 3.0])
    want = w.compute_importance(torch.from_numpy(qkv), H).numpy()          # importance.py:4-34
    np.testing.assert_allclose(orc.importance_scores(qkv, H), want, rtol=3e-5, atol=1e-8)
    np.testing.assert_allclose(ort.importance_scores(torch.from_numpy(qkv), H).numpy(), want, rtol=3e-5, atol=1e-8)


CASES = [
    # cfg kwargs, schedule, batch
    (dict(img_size=64, embed_dim=128, depth=5, num_heads=2, num_classes=10),
     {1: {"keep_ratio": 0.75}, 2: {"keep_ratio": 0.61, "update": False}, 4: {"keep_ratio": 0.33}}, 3),
    (dict(img_size=56, patch_size=14, embed_dim=320, depth=4, num_heads=4, num_classes=7, layer_scale=0.3),
     {0: {"keep_ratio": 0.9}, 3: {"keep_ratio": 0.5}}, 2),
    (dict(img_size=64, embed_dim=192, depth=3, num_heads=3, num_classes=5, mlp_ratio=2.625),
     {1: {"keep_ratio": 0.01}}, 2),                    # keep = max(1, int(...)) = 1 patch token
    (dict(img_size=96, embed_dim=128, depth=4, num_heads=4, num_classes=10),
     {}, 2),                                           # nothing scheduled: the unpruned path
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_forward_on_fresh_models(ref, case):
    """RAJNIViTWrapper.forward of the reference on a freshly seeded timm-shaped model == both oracles: token counts
    exactly; logits to fp32 round-off when the selections coincide (they do unless two scores tie in fp32)."""
    mod, w = ref
    kw, schedule, batch = CASES[case]
    cfg = ts.ViTConfig(**kw)
    seed = 4242 + case
    model = ts.create_model(cfg, seed=seed, std=0.08, bias_std=0.02)
    sd = ts.state_dict_numpy(model)
    images = np.random.default_rng(seed).standard_normal((batch, 3, cfg.img_size, cfg.img_size), dtype=np.float32)

    # the reference mutates the model it wraps (model.py:16-21): give it its own copy
    ref_model = ts.create_model(cfg, seed=seed, std=0.08, bias_std=0.02)
    trace = {}
    orig = w.RAJNIAttention.forward

    def recording(self, x, prev_scores=None):
        out, keep_idx, nxt = orig(self, x, prev_scores)
        trace[self._blk] = keep_idx.numpy().copy()
        return out, keep_idx, nxt

    wrapped = mod.RAJNIViTWrapper(ref_model, {int(k): dict(v) for k, v in schedule.items()}).eval()
    for i, blk in enumerate(wrapped.blocks):
        if getattr(blk, "has_pruner", False):
            blk.attn._blk = i
    w.RAJNIAttention.forward = recording
    try:
        with torch.no_grad():
            want = wrapped(torch.from_numpy(images)).numpy()
    finally:
        w.RAJNIAttention.forward = orig
    counts = wrapped.get_last_stats()["token_counts"]

    got, stats, tr = orc.vit_forward(sd, images, schedule, depth=cfg.depth, num_heads=cfg.num_heads, ln_eps=cfg.ln_eps,
                                     dtype=np.float64, return_trace=True)
    assert stats["token_counts"] == counts == orc.token_counts(cfg.num_patches + 1, cfg.depth, schedule)
    sd_t = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items()}
    got_t, stats_t, tr_t = ort.vit_forward(sd_t, torch.from_numpy(images), schedule, depth=cfg.depth, num_heads=cfg.num_heads,
                                           ln_eps=cfg.ln_eps, return_trace=True)
    assert stats_t["token_counts"] == counts
    same = all(np.array_equal(tr[i]["keep_idx"], trace[i]) for i in trace)
    same_t = all(np.array_equal(tr_t[i]["keep_idx"].numpy(), trace[i]) for i in trace)
    assert sorted(trace) == sorted(int(k) for k in schedule)
    assert same and same_t, "selections differ from the reference's (an exact fp32 tie would be needed)"
    scale = max(1.0, np.abs(want).max())
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(got_t.numpy(), want, rtol=0, atol=2e-4 * scale)


def test_rajni_attention_forward_contract(ref):
    """RAJNIAttention.forward(x, prev_scores) -> (out, keep_idx int64 ascending with CLS first, next_scores), with
    and without carried scores (attention.py:17-60) == oracle.rajni_attention on the same weights."""
    _, w = ref
    cfg = ts.ViTConfig(img_size=64, embed_dim=128, depth=1, num_heads=2, num_classes=3)
    model = ts.create_model(cfg, seed=9, std=0.1, bias_std=0.05)
    sd = ts.state_dict_numpy(model)
    x = np.random.default_rng(1).standard_normal((3, 17, 128), dtype=np.float32)
    for update, prev in ((True, None), (False, np.random.default_rng(2).random((3, 17), dtype=np.float32)), (False, None)):
        attn = w.RAJNIAttention(ts.create_model(cfg, seed=9, std=0.1, bias_std=0.05).blocks[0].attn, 0.6, update)
        with torch.no_grad():
            out, idx, nxt = attn(torch.from_numpy(x), None if prev is None else torch.from_numpy(prev))
        o, i, n, _ = orc.rajni_attention(x, sd, "blocks.0.attn.", 2, 0.6, update, prev)
        assert idx.dtype == torch.int64 and (idx[:, 0] == 0).all() and (idx[:, 1:] > idx[:, :-1]).all()
        np.testing.assert_array_equal(i, idx.numpy())
        np.testing.assert_allclose(o, out.numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(n, nxt.numpy(), rtol=2e-5, atol=1e-8)
