import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rajni-vit_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no librajni_hip.so (built artefacts are not in the history): build it once, the same way
    `__graft_entry__.build()` does (hipcc cross-compiles gfx950 without a GPU, ~30 s).  A build that is present is
    left alone - rebuilding after source edits is `python rajni-vit_amd/build.py`."""
    lib = os.path.join(ROOT, "rajni-vit_amd", "rajni_amd", "lib", "librajni_hip.so")
    if os.path.exists(lib):
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("rajni_build", os.path.join(ROOT, "rajni-vit_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
