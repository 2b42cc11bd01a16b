"""Static check of the generated gfx950 ISA (no GPU needed, ~15 s): the persistent GEMM kernels must not
contain a DMA-draining wait or a scratch access inside their K loop, nor spill in any instantiation the
dispatcher launches.  Both regressions are silent (results stay correct) and cost 10-15 %."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not installed")
def test_persistent_gemm_isa_has_no_dma_drain_or_spill():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("dispatched  ok") >= 12, r.stdout
