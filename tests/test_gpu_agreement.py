"""Parity statements with resolution.  GPU box only (`-m gpu`).

The tolerance reading used everywhere in this suite: north_star's "1e-2 bf16" is taken RELATIVE to max |logit| of the
reference's fp32 run (DESIGN.md section 2 says why: the reference's own bf16 CPU forward is 1e-2..2e-2 of the logit
scale away from its fp32 forward, so an absolute 1e-2 on logits of magnitude 4-5 is below what bf16 storage of the
reference's own activations allows).  Every test here PRINTS the absolute figure next to the relative one.
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import rajni_amd
from oracle import rajni_oracle as orc
from rajni_amd import timm_shaped as ts
from helpers import load_case, case_images, pruned_blocks

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(meta):
    cfg = ts.CONFIGS[meta["cfg_name"]]
    model = ts.create_model(cfg, seed=meta["seed"], std=meta["std"], bias_std=meta["bias_std"], round_bf16=True)
    return cfg, rajni_amd.RAJNIViTWrapper(model, meta["schedule"]).to(DEV).to(torch.bfloat16).eval()


def test_against_the_reference_bf16_run():
    """tests/golden/base224_bf16 is literally "the reference PyTorch path in bf16" (its CPU bf16 forward of the
    base224 fixture: same weights, same images as base224_fp32).  With ITS selections injected, the build must be no
    further from it than the reference's bf16 run is from the reference's own fp32 run - and, being an fp32-stream
    forward, it must sit closer to the fp32 run than the reference's bf16 run does."""
    meta16, d16 = load_case("base224_bf16")
    meta32, d32 = load_case("base224_fp32")
    assert (meta16["seed"], meta16["image_seed"], meta16["cfg_name"]) == (meta32["seed"], meta32["image_seed"], meta32["cfg_name"])
    ref16, ref32 = d16["logits"], d32["logits"]
    scale = float(np.abs(ref32).max())
    d_ref = float(np.abs(ref16 - ref32).max())            # what bf16 costs the REFERENCE (selections included)
    _, wrapped = _build(meta16)
    images = torch.from_numpy(case_images(meta16, d16)).to(DEV)
    wrapped.force_keep_idx({i: torch.from_numpy(d16[f"blk{i}.keep_idx"]).to(DEV) for i in pruned_blocks(meta16)})
    got16 = wrapped(images).float().cpu().numpy()
    assert wrapped.get_last_stats()["token_counts"] == d16["token_counts"].tolist()
    d_build = float(np.abs(got16 - ref16).max())
    wrapped.force_keep_idx({i: torch.from_numpy(d32[f"blk{i}.keep_idx"]).to(DEV) for i in pruned_blocks(meta32)})
    got32 = wrapped(images).float().cpu().numpy()
    d_build32 = float(np.abs(got32 - ref32).max())
    print(f"\nbase224: |ref_bf16 - ref_fp32| = {d_ref:.4f} abs = {d_ref / scale:.4f} rel;  "
          f"|build - ref_bf16| (its selections) = {d_build:.4f} abs = {d_build / scale:.4f} rel;  "
          f"|build - ref_fp32| (its selections) = {d_build32:.4f} abs = {d_build32 / scale:.4f} rel;  logit scale {scale:.3f}")
    assert d_build <= d_ref, (d_build, d_ref)
    assert d_build32 <= d_ref and d_build32 <= 1e-2 * scale, (d_build32, d_ref, scale)
    assert (got16.argmax(1) == ref16.argmax(1)).all()


def _bench():
    spec = importlib.util.spec_from_file_location("rajni_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_top1_agreement_on_256_reference_images():
    """The number bench.py reports as `reference_agreement` (north_star: "<= 0.1 top-1 delta vs the reference
    wrapper"; the reference counts argmax hits, eval.py:61-64): 256 images through the reference on CPU (fp32), the
    same 256 through the build in bf16.
      * selections injected: logits within 1e-2 of the logit scale on all 256 images, and at least as many argmax
        agreements as the reference's own bf16 run achieves;
      * free-running (the device ranks its own bf16 scores - SURVEY Q7: one swapped boundary token moves logits by
        ~5e-2, so this is a statistical statement): top-1 agreement within 0.1 ABSOLUTE of the injected one, i.e. the
        north_star delta, and no worse than the reference's own bf16 run by more than 2 images in 256."""
    res = _bench().reference_agreement(torch.device(DEV))
    assert "error" not in res, res
    print("\nreference_agreement:", res)
    n = res["images"]
    inj, free, own = res["injected_selections"], res["free_running"], res["reference_own_bf16_run"]
    assert n == 256 and res["token_counts_equal"]
    assert inj["max_abs_dlogit_over_logit_scale"] <= 1e-2
    assert inj["top1_agree"] >= own["top1_agree"]
    assert inj["max_abs_dlogit"] <= own["max_abs_dlogit"]
    # free-running vs injected: a statement in IMAGES, not a 0.1 fraction (north_star's "<= 0.1 top-1 delta" read as 0.1
    # percentage points of accuracy is unmeasurable without trained weights and labels: DESIGN.md section 2, "unpinned")
    assert free["top1_agree"] >= own["top1_agree"] - 2
    assert inj["top1_agree"] - free["top1_agree"] <= 8


def test_top1_agreement_floor_of_the_fp8_mfma_format():
    """The opt-in fp8_mfma format (BASELINE configs[4]) has no reference semantics, so its end-to-end price is held as
    an agreement FLOOR on the same 256 reference images: a wiring bug in the e4m3 path (wrong row scale, wrong hidden
    bound) costs tens of images, the quantisation itself ~40-45 (measured r02: 213 injected / 216 free-running of 256; r03
    with the attention output as a fourth e4m3 point: 209 / 212; the bf16 default: 252 / 247)."""
    res = _bench().reference_agreement(torch.device(DEV), weight_format="fp8_mfma")
    assert "error" not in res, res
    print("\nreference_agreement fp8_mfma:", res["injected_selections"], res["free_running"])
    assert res["token_counts_equal"]
    assert res["injected_selections"]["top1_agree"] >= 205 and res["free_running"]["top1_agree"] >= 205
    assert res["injected_selections"]["max_abs_dlogit_over_logit_scale"] <= 0.2


@pytest.mark.parametrize("cfg_name,fmt", [("vit_micro_patch16_64", "safetensors"), ("deit3_micro_patch16_64", "pt")])
def test_weights_loader_logits_on_device(tmp_path, cfg_name, fmt):
    """f3: a timm-named state dict (incl. ls{1,2}.gamma and an N-1-row pos_embed) written to disk, loaded through
    `run.create_base(--weights)` (the offline stand-in for run.py:89-92's pretrained download), wrapped and run on the
    GPU gives the logits of the oracle on those weights, and bit-identical logits to a model that was handed the same
    tensors directly."""
    from rajni_amd import run
    cfg = ts.CONFIGS[cfg_name]
    sd_np = {k: ts.bf16_round_np(v) for k, v in ts.synth_state_dict(cfg, seed=21, std=0.08, bias_std=0.02).items()}
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    if fmt == "safetensors":
        from safetensors.torch import save_file
        path = str(tmp_path / "w.safetensors")
        save_file(sd, path)
    else:
        path = str(tmp_path / "w.pt")
        torch.save(sd, path)
    model, _ = run.create_base(run.get_args(["--model", cfg_name, "--weights", path, "--seed", "99"]))
    sched = {1: {"keep_ratio": 0.75, "update": True}, 2: {"keep_ratio": 0.6, "update": False}}
    w = rajni_amd.RAJNIViTWrapper(model, sched).to(DEV).to(torch.bfloat16).eval()
    imgs = ts.bf16_round_np(np.random.default_rng(5).standard_normal((3, 3, 64, 64), dtype=np.float32))
    got = w(torch.from_numpy(imgs).to(DEV)).float().cpu().numpy()
    direct = ts.create_model(cfg, seed=0)
    direct.load_state_dict(sd)
    w2 = rajni_amd.RAJNIViTWrapper(direct, sched).to(DEV).to(torch.bfloat16).eval()
    assert np.array_equal(got, w2(torch.from_numpy(imgs).to(DEV)).float().cpu().numpy())
    forced = {i: t["keep_idx"].cpu().numpy() for i, t in w.get_last_trace().items()}
    want, stats = orc.vit_forward(sd_np, imgs, sched, depth=cfg.depth, num_heads=cfg.num_heads, ln_eps=cfg.ln_eps,
                                  forced_keep=forced)
    assert stats == w.get_last_stats()
    assert np.abs(got - want).max() <= 1e-2 * np.abs(want).max()
