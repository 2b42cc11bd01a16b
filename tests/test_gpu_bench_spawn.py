"""`python bench.py --gpus 2` with NO launcher on the GPU box (VERDICT r1 top item): the parent must start two ranks
itself, each rank runs the HIP forward and `evaluate_model`'s all-reduce joins them.  A lease has ONE GPU, so the two
ranks share cuda:0 (RAJNI_BENCH_ONE_DEVICE=1) and meet over gloo (RCCL refuses two ranks on one device: "Duplicate
GPU detected") - the rank bookkeeping, the spawn and the JSON contract are what is under test; with 2+ devices visible
the same command takes the RCCL path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=600)


def test_bench_two_ranks_without_a_launcher():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "64", "--no-cpu-baseline", "--no-torch-baseline"],
             {"RAJNI_BENCH_ONE_DEVICE": "1", "RAJNI_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0, relayed by the parent
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["steps"] == 4 and d["warmup"] == 2
    assert "all_reduce" in d["config"]["collective"]
    # whole-job throughput = images of BOTH ranks over the slower rank's seconds: ms_per_step is per global step
    assert abs(d["ms_per_step"] - 2 * 64 / d["value"] * 1e3) < 1e-2


def test_bench_refuses_when_devices_are_missing():
    """More ranks than visible devices (and no one-device rehearsal switch): exit 3 and no result line - never a
    smaller run in the bigger run's slot."""
    import torch
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 1), "--steps", "2"], {"RAJNI_BENCH_ONE_DEVICE": "0"})
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert r.stdout.strip() == "" and f"needs {n + 1} visible" in r.stderr
