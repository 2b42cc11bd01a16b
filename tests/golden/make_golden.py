#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE on CPU.

Run in the authoring container only (needs /root/reference; the GPU box has neither it nor this
need):        PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference package is imported under the alias `rajni_ref` straight from /root/reference
(read-only; nothing of it is copied).  The base model is the build's timm-shaped ViT with weights
synthesised from (config, seed) - see rajni_amd/timm_shaped.py - so fixtures hold only inputs that
cannot be regenerated, the reference's outputs, and the recipe (config name, seed, std) to rebuild
the weights on any machine.

All weights and images are rounded to bf16-representable fp32 values before the reference sees
them, so the same fixture serves the fp32 oracle and the bf16 device path ("same inputs").
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
sys.dont_write_bytecode = True

from rajni_amd import timm_shaped as ts  # noqa: E402

REF_DIR = "/root/reference/rajni"


def load_reference():
    spec = importlib.util.spec_from_file_location(
        "rajni_ref", os.path.join(REF_DIR, "__init__.py"), submodule_search_locations=[REF_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rajni_ref"] = mod
    spec.loader.exec_module(mod)
    import rajni_ref.wrapper as w  # noqa
    return mod, w


README_SCHEDULE = {3: {"keep_ratio": 0.88, "update": True}, 4: {"keep_ratio": 0.88, "update": True},
                   7: {"keep_ratio": 0.80, "update": True}, 8: {"keep_ratio": 0.72, "update": True}}
AGGRESSIVE_L384 = {4: {"keep_ratio": 0.7}, 12: {"keep_ratio": 0.5}, 20: {"keep_ratio": 0.3}}
MICRO_SCHEDULE = {1: {"keep_ratio": 0.75, "update": True}, 2: {"keep_ratio": 0.6, "update": False},
                  3: {"keep_ratio": 0.5, "update": True}}


def synth_images(cfg, batch, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, cfg.in_chans, cfg.img_size, cfg.img_size), dtype=np.float32)
    return ts.bf16_round_np(x)


def boundary_gap(scores, keep):
    """min over the batch of s_(keep) - s_(keep+1) among patch scores."""
    p = np.sort(scores[:, 1:].astype(np.float64), axis=1)[:, ::-1]
    if keep >= p.shape[1]:
        return float("inf")
    return float(np.min(p[:, keep - 1] - p[:, keep]))


def run_case(ref, refw, name, cfg_name, schedule, batch, seed, std, bias_std, dtype=torch.float32):
    cfg = ts.CONFIGS[cfg_name]
    model = ts.create_model(cfg, seed=seed, std=std, bias_std=bias_std, round_bf16=True)
    if cfg.no_embed_class:
        # the reference adds pos_embed[:, :N] to all N tokens; give it the
        # mathematically identical padded table (zero row for CLS).
        padded = torch.cat([torch.zeros(1, 1, cfg.embed_dim), model.pos_embed.data], dim=1)
        model.pos_embed = torch.nn.Parameter(padded)
        model.no_embed_class = False
    images = synth_images(cfg, batch, seed + 1000)

    trace = {}
    orig_forward = refw.RAJNIAttention.forward

    wrapped = ref.RAJNIViTWrapper(model, {int(k): dict(v) for k, v in schedule.items()})
    wrapped = wrapped.to(dtype)
    for i, blk in enumerate(wrapped.blocks):
        if blk.has_pruner:
            blk.attn._blk_index = i
    # capture the full-N scores each stage ranks: recompute them from the captured qkv
    def qkv_hook(mod, inp, out):
        mod._last_qkv = out
    hooks = []
    for i, blk in enumerate(wrapped.blocks):
        if blk.has_pruner:
            hooks.append(blk.attn.qkv.register_forward_hook(qkv_hook))

    def rec_forward2(self, x, prev_scores=None):
        out, keep_idx, nxt = orig_forward(self, x, prev_scores)
        if self.update or prev_scores is None:
            sc = refw.compute_importance(self.qkv._last_qkv, self.num_heads)
        else:
            sc = prev_scores
        trace[self._blk_index] = dict(out=out.float().numpy().copy(), keep_idx=keep_idx.numpy().copy(),
                                      next_scores=nxt.float().numpy().copy(),
                                      scores=sc.float().numpy().copy())
        return out, keep_idx, nxt

    refw.RAJNIAttention.forward = rec_forward2
    try:
        with torch.no_grad():
            logits = wrapped(torch.from_numpy(images).to(dtype))
    finally:
        refw.RAJNIAttention.forward = orig_forward
        for h in hooks:
            h.remove()
    stats = wrapped.get_last_stats()

    out = {"logits": logits.float().numpy(), "token_counts": np.asarray(stats["token_counts"], np.int64)}
    gaps = {}
    for i, t in trace.items():
        for k, v in t.items():
            # `out` of the big models is bulky; keep scores/indices only for them
            if k == "out" and v.size > 200_000:
                continue
            out[f"blk{i}.{k}"] = v
        keep = t["keep_idx"].shape[1] - 1
        gaps[int(i)] = boundary_gap(t["scores"], keep)
    meta = dict(name=name, cfg_name=cfg_name, schedule={str(k): v for k, v in schedule.items()},
                batch=batch, seed=seed, image_seed=seed + 1000, std=std, bias_std=bias_std,
                dtype=str(dtype).replace("torch.", ""), boundary_gap=gaps,
                generator="tests/golden/make_golden.py: reference RAJNIViTWrapper on CPU, torch "
                          + torch.__version__)
    if images.nbytes <= 400_000:
        out["images"] = images
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(f"{name}: counts={stats['token_counts']} gaps={gaps} |logit|max={np.abs(out['logits']).max():.3f}")


def importance_cases(refw):
    """compute_importance (importance.py:4-34) on raw qkv tensors."""
    rng = np.random.default_rng(7)
    out, metas = {}, []
    for j, (B, N, H, D, scale) in enumerate([(2, 17, 2, 64, 1.0), (3, 197, 3, 64, 0.5), (1, 577, 16, 64, 0.3),
                                             (2, 2, 2, 64, 1.0), (2, 87, 12, 64, 2.0), (1, 61, 4, 32, 1.0)]):
        qkv = ts.bf16_round_np((rng.standard_normal((B, N, 3 * H * D), dtype=np.float32) * scale))
        sc = refw.compute_importance(torch.from_numpy(qkv), H).numpy()
        out[f"c{j}.qkv"] = qkv if qkv.nbytes < 300_000 else np.zeros(0, np.float32)
        out[f"c{j}.scores"] = sc
        metas.append(dict(B=B, N=N, H=H, D=D, scale=scale, stored_qkv=bool(qkv.nbytes < 300_000)))
    np.savez_compressed(os.path.join(HERE, "importance_cases.npz"), **out)
    with open(os.path.join(HERE, "importance_cases.json"), "w") as f:
        json.dump(dict(seed=7, cases=metas,
                       note="qkv = bf16_round(PCG64(7).standard_normal * scale), cases drawn in order"), f, indent=1)
    print("importance_cases written")


def selection_cases():
    """torch.topk + sort + prepend (attention.py:34-39) on tie-heavy / degenerate score rows.
    The reference's tie order is unspecified, so tests check multiset validity for these."""
    rng = np.random.default_rng(11)
    rows = {
        "distinct": rng.permutation(64).astype(np.float32)[None, :] / 64.0,
        "ties": rng.integers(0, 5, size=(3, 197)).astype(np.float32),
        "all_equal": np.ones((2, 50), np.float32),
        "with_nan": np.where(rng.random((2, 40)) < 0.1, np.nan, rng.random((2, 40))).astype(np.float32),
        "bf16_like": ts.bf16_round_np(rng.random((4, 197), dtype=np.float32) * 1e-2),
    }
    out = {}
    for name, s in rows.items():
        N = s.shape[1]
        for ratio in (0.88, 0.5, 0.0, 1.0):
            keep = max(1, int(ratio * (N - 1)))
            t = torch.from_numpy(s)
            _, idx = torch.topk(t[:, 1:], keep, dim=1)
            idx = torch.sort(idx, dim=1).values
            keep_idx = torch.cat([torch.zeros((s.shape[0], 1), dtype=torch.long), idx + 1], dim=1)
            out[f"{name}.r{ratio}.keep_idx"] = keep_idx.numpy()
        out[f"{name}.scores"] = s
    np.savez_compressed(os.path.join(HERE, "selection_cases.npz"), **out)
    print("selection_cases written")


def evaluate_cases(ref):
    """evaluate_model bookkeeping (eval.py:6-75): accuracy and number of forwards executed."""
    class Counting(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.calls = 0
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            self.calls += 1
            return x  # "images" are already logits [B, classes]

    rng = np.random.default_rng(3)
    cases = []
    for n_batches, bsz, warmup, max_batches in [(4, 8, 2, None), (3, 5, 5, 2), (6, 4, 0, 10), (2, 16, 7, 1)]:
        logits = [rng.standard_normal((bsz, 10)).astype(np.float32) for _ in range(n_batches)]
        labels = [rng.integers(0, 10, size=bsz) for _ in range(n_batches)]
        # make about half of them right
        for lg, lb in zip(logits, labels):
            for r in range(0, bsz, 2):
                lg[r, lb[r]] = 10.0
        loader = [(torch.from_numpy(a), torch.from_numpy(b)) for a, b in zip(logits, labels)]
        m = Counting()
        acc, thr = ref.evaluate_model(m, loader, device="cpu", max_batches=max_batches, warmup=warmup)
        cases.append(dict(n_batches=n_batches, batch=bsz, warmup=warmup, max_batches=max_batches,
                          acc=acc, forwards=m.calls, seed=3,
                          logits=[a.tolist() for a in logits], labels=[b.tolist() for b in labels]))
    with open(os.path.join(HERE, "evaluate_cases.json"), "w") as f:
        json.dump(cases, f)
    print("evaluate_cases written:", [(c["acc"], c["forwards"]) for c in cases])


def agreement_case(ref, refw, name="base224_agree256", cfg_name="vit_base_patch16_224", schedule=README_SCHEDULE,
                   batch=256, chunk=32, seed=2, std=0.04, bias_std=0.02):
    """A fixture with RESOLUTION for "top-1 delta vs the reference wrapper" (eval.py:61-64 is an argmax count):
    the reference's fp32 CPU forward of `batch` images of the BASELINE model dims and schedule - logits (fp32) and the
    keep_idx of every stage (int16) - plus the reference's OWN bf16 CPU forward of the same images as the yardstick
    (its argmax and its per-image max |dlogit| against the fp32 run).  Images are regenerated from the seed."""
    cfg = ts.CONFIGS[cfg_name]
    images = synth_images(cfg, batch, seed + 3000)
    sched = {int(k): dict(v) for k, v in schedule.items()}
    keep = {}
    orig_forward = refw.RAJNIAttention.forward

    def rec_forward(self, x, prev_scores=None):
        out, keep_idx, nxt = orig_forward(self, x, prev_scores)
        keep.setdefault(self._blk_index, []).append(keep_idx.numpy().astype(np.int16))
        return out, keep_idx, nxt

    def run(dtype, record):
        model = ts.create_model(cfg, seed=seed, std=std, bias_std=bias_std, round_bf16=True)
        wrapped = ref.RAJNIViTWrapper(model, {k: dict(v) for k, v in sched.items()}).to(dtype)
        for i, blk in enumerate(wrapped.blocks):
            if blk.has_pruner:
                blk.attn._blk_index = i
        if record:
            refw.RAJNIAttention.forward = rec_forward
        outs = []
        try:
            with torch.no_grad():
                for c in range(0, batch, chunk):
                    outs.append(wrapped(torch.from_numpy(images[c:c + chunk]).to(dtype)).float().numpy())
                    print(f"  {name} {dtype}: {c + chunk}/{batch}", flush=True)
        finally:
            refw.RAJNIAttention.forward = orig_forward
        return np.concatenate(outs), wrapped.get_last_stats()

    logits, stats = run(torch.float32, True)
    logits_bf16, _ = run(torch.bfloat16, False)
    out = {"logits": logits, "token_counts": np.asarray(stats["token_counts"], np.int64),
           "ref_bf16_argmax": logits_bf16.argmax(1).astype(np.int16),
           "ref_bf16_max_abs_dlogit": np.abs(logits_bf16 - logits).max(1).astype(np.float32)}
    for i, parts in keep.items():
        out[f"blk{i}.keep_idx"] = np.concatenate(parts)
    srt = np.sort(logits, axis=1)
    meta = dict(name=name, cfg_name=cfg_name, schedule={str(k): v for k, v in schedule.items()}, batch=batch,
                seed=seed, image_seed=seed + 3000, std=std, bias_std=bias_std, dtype="float32", boundary_gap={},
                ref_bf16_top1_agree=int((logits_bf16.argmax(1) == logits.argmax(1)).sum()),
                ref_bf16_max_abs_dlogit=float(np.abs(logits_bf16 - logits).max()),
                logit_scale=float(np.abs(logits).max()),
                median_top2_margin=float(np.median(srt[:, -1] - srt[:, -2])),
                generator="tests/golden/make_golden.py agreement_case: reference RAJNIViTWrapper on CPU (fp32 run + its "
                          "own bf16 run), torch " + torch.__version__)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(f"{name}: counts={stats['token_counts']} ref bf16 vs fp32: top-1 {meta['ref_bf16_top1_agree']}/{batch}, "
          f"max|dlogit| {meta['ref_bf16_max_abs_dlogit']:.4f} (scale {meta['logit_scale']:.3f}), "
          f"median top-2 margin {meta['median_top2_margin']:.4f}")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref, refw = load_reference()
    only = sys.argv[1:]          # forward case names: regenerate just those
    if not only:
        importance_cases(refw)
        selection_cases()
        evaluate_cases(ref)
    cases = [
        ("micro_fp32", "vit_micro_patch16_64", MICRO_SCHEDULE, 4, 0, 0.08, 0.02, torch.float32),
        ("tiny224_fp32", "vit_tiny_patch16_224", README_SCHEDULE, 2, 1, 0.06, 0.02, torch.float32),
        ("base224_fp32", "vit_base_patch16_224", README_SCHEDULE, 2, 2, 0.04, 0.02, torch.float32),
        ("base224_bf16", "vit_base_patch16_224", README_SCHEDULE, 2, 2, 0.04, 0.02, torch.bfloat16),
        ("deit3_fp32", "deit3_base_patch16_224", README_SCHEDULE, 2, 3, 0.04, 0.02, torch.float32),
        ("large384_fp32", "vit_large_patch16_384", AGGRESSIVE_L384, 1, 4, 0.03, 0.02, torch.float32),
        ("microd80_fp32", "vit_micro_d80_patch16_64", MICRO_SCHEDULE, 3, 5, 0.08, 0.02, torch.float32),
        ("microp14_fp32", "vit_micro_patch14_56", MICRO_SCHEDULE, 3, 6, 0.08, 0.02, torch.float32),
    ]
    for c in cases:
        if only and c[0] not in only:
            continue
        run_case(ref, refw, *c)
    if not only or "base224_agree256" in only:
        agreement_case(ref, refw)


if __name__ == "__main__":
    main()
