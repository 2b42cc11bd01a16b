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


def test_attention_and_selection_fuzz_against_torch():
    spec = importlib.util.spec_from_file_location("fuzz_attention", os.path.join(ROOT, "tools", "fuzz_attention.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(cases=80, seed=2026, verbose=False) == 0
