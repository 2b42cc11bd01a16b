"""CPU-only tests (`-m "not gpu"`): the C-ABI library loads and exports every symbol the header
declares, the host-side mirror of the reference interface behaves like the reference (names,
argument meaning, error behaviour), and the product path refuses to run without a device instead of
falling back."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

import rajni_amd
from oracle import rajni_oracle as orc
from rajni_amd import _native as nat, ops, timm_shaped as ts
from rajni_amd.wrapper.model import normalise_schedule, plan_token_counts
from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_public_names_match_reference_surface():
    # rajni/__init__.py:1-2 and rajni/wrapper/__init__.py:1-3
    assert {"RAJNIViTWrapper", "evaluate_model"} <= set(rajni_amd.__all__)
    from rajni_amd.wrapper import RAJNIViTWrapper, RAJNIAttention, compute_importance  # noqa: F401
    import inspect
    assert list(inspect.signature(rajni_amd.RAJNIViTWrapper.__init__).parameters)[1:] == ["base_model", "pruning_schedule"]
    assert list(inspect.signature(rajni_amd.evaluate_model).parameters) == ["model", "dataloader", "device", "max_batches", "warmup"]
    sig = inspect.signature(rajni_amd.evaluate_model)
    assert sig.parameters["device"].default == "cuda" and sig.parameters["warmup"].default == 5
    assert list(inspect.signature(RAJNIAttention.__init__).parameters)[1:] == ["attn", "keep_ratio", "update"]
    assert list(inspect.signature(compute_importance).parameters) == ["qkv", "num_heads", "eps"]


def test_library_exports_every_declared_symbol():
    """Every `rajni_*(` prototype in include/rajni_hip.h (the boundary) and include/rajni_hip_debug.h (test hooks,
    kept out of the boundary header) resolves in librajni_hip.so (no compute)."""
    with open(os.path.join(ROOT, "include", "rajni_hip.h")) as f:
        header = f.read()
    assert "rajni_debug_" not in header, "debug hooks belong in rajni_hip_debug.h"
    with open(os.path.join(ROOT, "include", "rajni_hip_debug.h")) as f:
        header += f.read()
    declared = set(re.findall(r"\b(rajni_[a-z0-9_]+)\s*\(", header))
    declared -= {"rajni_stream_t"}
    assert len(declared) >= 15
    lib = nat.load_library()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in the header but not exported"
    assert set(nat.EXPORTED_SYMBOLS) == declared
    assert lib.rajni_abi_version() == nat.ABI_VERSION == 8
    assert lib.rajni_profile_class_name(0).decode().startswith("gemm")


def test_struct_layouts_match_header_sizes():
    """ctypes mirrors of the ABI structs: field order/size must match what the C side compiled.
    The workspace query walks the plan struct on the host, so a layout slip shows here."""
    import ctypes as C
    plan = nat.VitPlan()
    plan.dtype, plan.B, plan.in_chans, plan.img_size, plan.patch_size = nat.RAJNI_BF16, 2, 3, 64, 16
    plan.C, plan.H, plan.D, plan.depth, plan.hidden, plan.num_classes = 128, 2, 64, 4, 512, 10
    n0 = 17
    want = 0
    a256 = lambda v: (v + 255) // 256 * 256
    rows = 2 * n0
    for nbytes in (rows * 128 * 4, rows * 128 * 4, rows * 128 * 2, rows * 384 * 2, rows * 128 * 2, rows * 512 * 2,
                   2 * 128 * 2, rows * 2):
        want += a256(nbytes)
    assert nat.lib().rajni_vit_workspace_bytes(C.byref(plan)) == want
    plan.resid_bf16 = 1
    assert nat.lib().rajni_vit_workspace_bytes(C.byref(plan)) == want - 2 * (a256(rows * 128 * 4) - a256(rows * 128 * 2))
    plan.resid_bf16 = 0
    plan.act_fp8 = 1        # two per-row scale vectors
    assert nat.lib().rajni_vit_workspace_bytes(C.byref(plan)) == want + 2 * a256(rows * 4)
    plan.act_fp8 = 0        # (the LAST int of the struct, ABI 8: the whole layout lines up)


def test_no_cpu_fallback():
    m = ts.create_model("vit_micro_patch16_64")
    w = rajni_amd.RAJNIViTWrapper(m, {1: {"keep_ratio": 0.5}})
    assert w.get_last_stats() is None                      # model.py:25 before the first forward
    with pytest.raises(nat.NativeError, match="no CPU fallback"):
        w(torch.randn(1, 3, 64, 64))
    with pytest.raises(nat.NativeError):
        rajni_amd.compute_importance(torch.randn(1, 5, 3 * 128), 2)
    with pytest.raises(nat.NativeError):
        w.blocks[1].attn(torch.randn(1, 17, 128))
    with pytest.raises(nat.NativeError, match="not found"):
        nat._lib, keep = None, nat._lib
        try:
            nat.load_library("/nonexistent/librajni_hip.so")
        finally:
            nat._lib = keep


def test_wrapper_surgery_matches_reference_semantics():
    """model.py:12-23: scheduled blocks get RAJNIAttention sharing the timm attention's parameters,
    every block gets has_pruner, the parameter set is unchanged."""
    m = ts.create_model("vit_micro_patch16_64")
    n_params = len(list(m.parameters()))
    qkv_before = m.blocks[2].attn.qkv
    w = rajni_amd.RAJNIViTWrapper(m, {2: {"keep_ratio": 0.7, "update": False}, 3: {"keep_ratio": 0.5}})
    assert [b.has_pruner for b in w.blocks] == [False, False, True, True]
    assert isinstance(w.blocks[2].attn, rajni_amd.RAJNIAttention) and w.blocks[2].attn.qkv is qkv_before
    assert w.blocks[2].attn.update is False and w.blocks[3].attn.update is True          # model.py:19
    assert w.blocks[2].attn.keep_ratio == 0.7 and w.blocks[2].attn.num_heads == 2
    assert len(list(w.parameters())) == n_params
    assert w.m is m and w.blocks is m.blocks
    with pytest.raises(KeyError):                                                         # model.py:18
        rajni_amd.RAJNIViTWrapper(ts.create_model("vit_micro_patch16_64"), {1: {"update": True}})


def test_schedule_json_string_keys_are_normalised():
    """the reference tests `i in schedule` with int i against JSON string keys and never
    prunes; the build normalises.  schedule.json's content is restated here as data."""
    sched = json.loads('{"3": {"keep_ratio": 0.95, "update": false}, "4": {"keep_ratio": 0.95, "update": true},'
                       ' "5": {"keep_ratio": 0.85, "update": true}, "6": {"keep_ratio": 0.85, "update": true},'
                       ' "7": {"keep_ratio": 0.95, "update": true}}')
    norm = normalise_schedule(sched)
    assert sorted(norm) == [3, 4, 5, 6, 7] and norm[3]["update"] is False
    # SURVEY Q1 value for schedule.json with int keys
    assert plan_token_counts(197, 12, norm) == [197, 197, 197, 197, 187, 177, 150, 127, 120, 120, 120, 120]
    assert plan_token_counts(197, 12, norm) == orc.token_counts(197, 12, orc.normalise_schedule(sched))


@settings(max_examples=200, deadline=None)
@given(st.integers(2, 1500), st.floats(0.0, 1.0, allow_nan=False))
def test_keep_count_is_python_double_truncation(n, ratio):
    assert ops.keep_count(ratio, n) == max(1, int(ratio * (n - 1))) == orc.keep_count(ratio, n)
    assert 1 <= ops.keep_count(ratio, n) <= max(1, n - 1)


@settings(max_examples=50, deadline=None)
@given(st.integers(2, 600), st.integers(1, 24),
       st.dictionaries(st.integers(0, 23), st.floats(0.05, 1.0, allow_nan=False), max_size=6))
def test_token_counts_plan_matches_oracle(n0, depth, ratios):
    sched = {k: {"keep_ratio": v} for k, v in ratios.items()}
    assert plan_token_counts(n0, depth, normalise_schedule(sched)) == orc.token_counts(n0, depth, orc.normalise_schedule(sched))


def test_readme_schedule_counts():
    sched = {3: {"keep_ratio": 0.88}, 4: {"keep_ratio": 0.88}, 7: {"keep_ratio": 0.8}, 8: {"keep_ratio": 0.72}}
    assert plan_token_counts(197, 12, normalise_schedule(sched)) == [197, 197, 197, 197, 173, 152, 152, 152, 121, 87, 87, 87]
    agg = {4: {"keep_ratio": 0.7}, 12: {"keep_ratio": 0.5}, 20: {"keep_ratio": 0.3}}
    assert plan_token_counts(577, 24, normalise_schedule(agg)) == [577] * 5 + [404] * 8 + [202] * 8 + [61] * 3


class _Counting(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.calls = 0
        self.w = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        self.calls += 1
        return x


def test_evaluate_model_matches_reference_fixture():
    """eval.py:6-75 bookkeeping, against outcomes captured from the reference's evaluate_model."""
    with open(os.path.join(GOLDEN, "evaluate_cases.json")) as f:
        cases = json.load(f)
    for c in cases:
        loader = [(torch.tensor(a, dtype=torch.float32), torch.tensor(b)) for a, b in zip(c["logits"], c["labels"])]
        m = _Counting()
        acc, thr = rajni_amd.evaluate_model(m, loader, device="cpu", max_batches=c["max_batches"], warmup=c["warmup"])
        assert acc == pytest.approx(c["acc"], abs=1e-9)
        assert m.calls == c["forwards"]
        assert thr > 0


def test_evaluate_model_accepts_generators_without_len():
    def gen():
        for _ in range(3):
            yield torch.eye(4), torch.arange(4)

    class Loader:
        def __iter__(self):
            return gen()

    acc, thr = rajni_amd.evaluate_model(_Counting(), Loader(), device="cpu", max_batches=None, warmup=1)
    assert acc == 100.0


def test_oracle_is_not_imported_by_the_product():
    """The product package must never import oracle/ (prompt section 3)."""
    pkg = os.path.join(ROOT, "rajni-vit_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn
                assert "rajni_oracle" not in src, fn


def test_cli_flags_match_reference_and_schedule_loader(tmp_path):
    """rajni/run.py:17-43 flag set; JSON schedule keys become ints (B1)."""
    from rajni_amd import run
    a = run.get_args(["--batch_size", "32", "--model", "vit_tiny_patch16_224", "--schedule", "s.json", "--compare_base",
                      "--max_batches", "3", "--warmup", "2", "--device", "cuda", "--num_workers", "4", "--data_path", "/x"])
    assert (a.batch_size, a.model, a.schedule, a.compare_base, a.max_batches, a.warmup, a.device, a.num_workers,
            a.data_path) == (32, "vit_tiny_patch16_224", "s.json", True, 3, 2, "cuda", 4, "/x")
    d = run.get_args([])
    assert (d.batch_size, d.warmup, d.model, d.device, d.max_batches) == (256, 5, "vit_base_patch16_224", "cuda", None)
    f = tmp_path / "schedule.json"
    f.write_text('{"3": {"keep_ratio": 0.95, "update": false}, "4": {"keep_ratio": 0.95}}')
    assert run.load_schedule(str(f)) == {3: {"keep_ratio": 0.95, "update": False}, 4: {"keep_ratio": 0.95}}
    assert run.load_schedule(None) == run.README_SCHEDULE


def _timm_named_micro_state_dict(cfg_name, seed):
    """A timm-NAMED fp32 state dict with non-default values everywhere (incl. DeiT-3's ls{1,2}.gamma and an
    N-1-row pos_embed when the config is no_embed_class)."""
    cfg = ts.CONFIGS[cfg_name]
    sd = ts.synth_state_dict(cfg, seed=seed, std=0.08, bias_std=0.02)
    return cfg, {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


@pytest.mark.parametrize("cfg_name", ["vit_micro_patch16_64", "deit3_micro_patch16_64"])
@pytest.mark.parametrize("fmt", ["safetensors", "pt", "pt_wrapped"])
def test_weights_loader_round_trip(tmp_path, cfg_name, fmt):
    """`--weights FILE` (the offline replacement of the reference's `timm.create_model(pretrained=True)`,
    rajni/run.py:89-92): a timm-named state dict written as .safetensors / .pt (also wrapped as a training
    checkpoint {"state_dict": ...}) loads through run.create_base into a model whose state dict is bit-identical -
    including LayerScale gammas and the N-1-row pos_embed of a no_embed_class model."""
    from rajni_amd import run
    cfg, sd = _timm_named_micro_state_dict(cfg_name, seed=11)
    if cfg.layer_scale:
        assert any(k.endswith("ls1.gamma") for k in sd) and sd["pos_embed"].shape[1] == cfg.num_patches
    if fmt == "safetensors":
        from safetensors.torch import save_file
        path = tmp_path / "w.safetensors"
        save_file(sd, str(path))
    else:
        path = tmp_path / "w.pt"
        torch.save({"state_dict": sd, "epoch": 3} if fmt == "pt_wrapped" else sd, str(path))
    args = run.get_args(["--model", cfg_name, "--weights", str(path), "--seed", "5"])
    model, source = run.create_base(args)
    assert os.path.basename(str(path)) in source and "WARNING" not in source
    got = model.state_dict()
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    # a file that does not fit the model is an error, not a partial load (strict=True)
    bad = dict(sd)
    bad.pop("head.bias")
    torch.save(bad, str(tmp_path / "bad.pt"))
    with pytest.raises(RuntimeError, match="head.bias"):
        run.create_base(run.get_args(["--model", cfg_name, "--weights", str(tmp_path / "bad.pt")]))
    # without --weights / --pretrained the report says the weights are random
    _, src2 = run.create_base(run.get_args(["--model", cfg_name]))
    assert "WARNING: random weights" in src2


def test_shard_sampler_covers_every_sample_once():
    from rajni_amd.run import ShardSampler
    for n, world in [(10, 4), (7, 2), (3, 8), (0, 2)]:
        seen = sorted(i for r in range(world) for i in ShardSampler(n, r, world))
        assert seen == list(range(n))
        assert sum(len(ShardSampler(n, r, world)) for r in range(world)) == n


def test_param_cache_sees_replaced_parameters_and_modules():
    """RAJNIViTWrapper caches the walk over the base model's parameters and validates the cache on EVERY forward
    (a swapped head / re-assigned Parameter used to be invisible for up to 63 forwards)."""
    m = ts.create_model("vit_micro_patch16_64")
    w = rajni_amd.RAJNIViTWrapper(m, {1: {"keep_ratio": 0.5}})
    a = w._all_params()
    assert w._all_params() is a                                   # unchanged model: the cached list
    m.head.bias = torch.nn.Parameter(m.head.bias.detach() + 1)     # re-assigned Parameter object
    b = w._all_params()
    assert b is not a and any(p is m.head.bias for p in b)
    m.head = torch.nn.Linear(m.embed_dim, 7)                       # replaced module
    c = w._all_params()
    assert c is not b and any(p is m.head.weight for p in c)
    m.blocks[0].mlp.fc2 = torch.nn.Linear(m.blocks[0].mlp.fc2.in_features, m.embed_dim)
    d = w._all_params()
    assert d is not c and any(p is m.blocks[0].mlp.fc2.weight for p in d)
    assert w._all_params() is d


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rajni_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_without_launcher_spawns_n_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` with no launcher (WORLD_SIZE unset) must START N ranks - never run one rank and
    label it N - before touching a GPU, relay rank 0's line and return the launcher's status."""
    import subprocess
    bench = _load_bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    monkeypatch.setattr(bench, "worker", lambda args: pytest.fail("the parent must not run the benchmark itself"))
    calls = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            calls["cmd"], calls["env"] = cmd, env
            self.stdout = iter(["rank chatter\n", json.dumps({"metric": "images/sec", "n_gpus": 4, "ranks_seen": 4}) + "\n"])

        def wait(self):
            return 0

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    rc = bench.main(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert rc == 0
    cmd = calls["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["ranks_seen"] == 4          # ONE JSON line, relayed
    assert not torch.cuda.is_initialized()                                   # the parent stayed off the GPU


def test_bench_refuses_fewer_devices_than_ranks(monkeypatch, capsys):
    """Fewer visible devices than --gpus: exit non-zero, never fall back to a smaller job."""
    import subprocess
    bench = _load_bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RAJNI_BENCH_ONE_DEVICE", raising=False)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: pytest.fail("must not launch"))
    assert bench.main(["--gpus", "8"]) == 3
    assert "needs 8 visible" in capsys.readouterr().err
    # a launcher-provided world size that disagrees with --gpus is refused too
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="refusing"):
        bench.main(["--gpus", "8"])


def test_fp8_row_quantiser_properties():
    """ops.quantize_rows_fp8 / pack_weight_fp8 (host side of the fp8-weight path): per-row
    scale maps the row maximum onto +-448, zero rows are harmless, RNE error bound, padded layout."""
    rng = np.random.default_rng(0)
    w = rng.standard_normal((37, 64)).astype(np.float32) * rng.uniform(0.01, 5.0, size=(37, 1)).astype(np.float32)
    w[5] = 0.0
    wt = torch.from_numpy(w).to(torch.bfloat16)
    q, scale = ops.quantize_rows_fp8(wt)
    assert q.dtype == torch.float8_e4m3fn and scale.dtype == torch.float32 and tuple(scale.shape) == (37,)
    qf = q.to(torch.float32)
    assert torch.isfinite(qf).all()
    rows = [i for i in range(37) if i != 5]
    assert (qf[rows].abs().amax(dim=1) == ops.FP8_E4M3_MAX).all()
    assert scale[5] == 1.0 and (qf[5] == 0).all()
    deq = qf * scale[:, None]
    ref = wt.to(torch.float32)
    # normal range: relative error <= 2^-4; below the smallest normal (2^-6 * scale) absolute error <= 2^-10 * scale
    bound = torch.maximum(ref.abs() * 2.0 ** -4, scale[:, None] * 2.0 ** -10) * 1.0001
    assert (deq - ref).abs().le(bound).all()
    packed, s2 = ops.pack_weight_fp8(torch.from_numpy(w), torch.bfloat16, "cpu")
    assert packed.dtype == torch.uint8 and tuple(packed.shape) == (256, 64) and torch.equal(s2, scale)
    assert (packed[37:] == 0).all() and torch.equal(packed[:37], q.view(torch.uint8))
    assert torch.equal(ops.dequantize_fp8(packed, s2), deq)
    with pytest.raises(ValueError):
        ops.pack_weight_fp8(torch.zeros(4, 24), torch.bfloat16, "cpu")


def test_graft_entry_build_runs():
    """The driver's build check: compiles (or finds up to date) the gfx950 library, imports the package and
    resolves every exported symbol - must stay in step with the ABI version."""
    import importlib
    ge = importlib.import_module("__graft_entry__")
    ge.build()


def test_bench_drops_stale_or_foreign_pmc_profiles(tmp_path, monkeypatch):
    """`roofline.traffic` / `mfma_busy_frac` come from the newest committed PMC profile - only when that profile says it
    was taken on THIS tree's kernel sources; fp8 profiles never stand in for the bf16 command."""
    import json
    bench = _load_bench()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from srchash import csrc_fingerprint
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.latest_profile("mfma_pmc.json") == (None, None)
    (prof / "r03_a_mfma_pmc.json").write_text(json.dumps({"by_bench_class": {}, "csrc_fingerprint": "0" * 16}))
    assert bench.latest_profile("mfma_pmc.json") == (None, None)                      # other sources: dropped
    (prof / "r03_b_mfma_pmc.json").write_text(json.dumps({"by_bench_class": {"k": 1}}))
    assert bench.latest_profile("mfma_pmc.json") == (None, None)                      # no fingerprint at all: dropped
    good = {"by_bench_class": {"k": 2}, "csrc_fingerprint": csrc_fingerprint()}
    (prof / "r03_c_mfma_pmc.json").write_text(json.dumps(good))
    (prof / "r03_d_fp8_mfma_pmc.json").write_text(json.dumps({**good, "by_bench_class": {"f8": 1}}))
    path, rec = bench.latest_profile("mfma_pmc.json")
    assert path.endswith("r03_c_mfma_pmc.json") and rec["by_bench_class"] == {"k": 2}  # the fp8 file sorts later but is skipped


def test_bench_roofline_picks_the_dominant_gemm_and_prices_proj_against_hbm():
    bench = _load_bench()
    prof = {"gemm_bf16_tn<bias>": dict(launches=10, ms=1.0, flops=1e12, bytes=1e9),
            "gemm_bf16_tn<bias,ls,resid> K<=N": dict(launches=10, ms=3.0, flops=1e12, bytes=9e9)}
    name, r = bench.gemm_roofline(prof)
    assert name.endswith("K<=N") and r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert r["achieved"] == pytest.approx(9e9 / 10 / 0.3e-3 / 1e9, rel=1e-3) and r["frac"] == pytest.approx(r["achieved"] / 8000.0, abs=1e-4)
    prof["gemm_f8_tn<bias,gelu,requant>"] = dict(launches=10, ms=5.0, flops=5e12, bytes=1e9)
    name, r = bench.gemm_roofline(prof)
    assert name.startswith("gemm_f8") and r["bound"] == "mfma" and r["peak"] == 5000.0
    assert bench.gemm_roofline({}) == (None, None)


def test_pmc_tools_coverage_rules(tmp_path):
    """tools/pmc_common.py: exact FC1 quantities of a bench.py line, and the rescale / refuse rule of a counter pass."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_common as pc
    line = {"config": {"dims": {"C": 768, "hidden": 3072, "batch_per_gpu": 256, "classes": 1000},
                       "token_counts": [197, 197, 197, 197, 173, 152, 152, 152, 121, 87, 87, 87], "csrc_fingerprint": "abc"}}
    f = tmp_path / "b.json"
    f.write_text("noise\n" + json.dumps(line) + "\n")
    e = pc.expected_fc1(str(f))
    rows = 256 * (197 * 3 + 173 + 152 * 3 + 121 + 87 * 4) / 12        # tokens AFTER each block's selection
    assert e["flops"] == pytest.approx(2 * rows * 3072 * 768)
    qkv = 256 * (197 * 4 + 173 + 152 * 3 + 121 + 87 * 3) * 2304 * 2          # QKV runs on the tokens ENTERING each block
    assert e["qkv_out_bytes"] == pytest.approx(qkv / 12) and e["qkv_out_bytes_with_head"] == pytest.approx((qkv + 256 * 1000 * 2) / 13)
    assert e["csrc_fingerprint_of_run"] == "abc"
    assert pc.judge(1.0, "x")[0] == 1.0 and pc.judge(0.985, "x")[0] == 1.0
    assert pc.judge(0.75, "x")[0] == pytest.approx(1 / 0.75) and "rescaled" in pc.judge(0.75, "x")[1]
    assert pc.judge(None, "x")[0] == 1.0
    for bad in (0.3, 1.2):
        with pytest.raises(SystemExit):
            pc.judge(bad, "x")
    assert pc.bench_class(pc.clean("void (anonymous namespace)::wide::gemm_bf16_tn_stream<2, 0, true, 4, 2, 4, 3, false, 1>(GemmParams)")) \
        == "gemm_bf16_tn<bias,ls,resid> K<=N"
    assert pc.bench_class(pc.clean("void wide::gemm_bf16_tn_stream<2, 0, true, 4, 2, 4, 3, false, 0>(GemmParams)")) == "gemm_bf16_tn<bias,ls,resid>"
    assert pc.bench_class("void f8w::gemm_f8_tn_wide<4>(GemmParams)") == "gemm_f8_tn<bias,gelu,requant>"
    assert pc.bench_class("void layernorm_kernel<float>(...)") is None
