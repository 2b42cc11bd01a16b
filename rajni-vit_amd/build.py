#!/usr/bin/env python3
"""Build librajni_hip.so (gfx950 only) in-tree with hipcc.  No cmake, no torch extension: the
library is a plain C-ABI shared object (include/rajni_hip.h) loaded through ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "rajni_amd", "lib")
OUT = os.path.join(OUT_DIR, "librajni_hip.so")
SOURCES = ["capi.hip", "gemm.hip", "rowops.hip", "score_select.hip", "attention.hip", "forward.hip"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", h)
                                                               for h in ("rajni_hip.h", "rajni_hip_debug.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra=(), out=None):
    """`out`/`extra` build an experimental variant (e.g. -DRAJNI_GEMM_X_AUX=2) next to the default."""
    global OUT
    if out is not None:
        OUT = out
        force = True
    if not force and not needs_build():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(OUT_DIR, (os.path.basename(OUT) + "." if out is not None else "") + src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
               "-c", os.path.join(CSRC, src), "-o", obj, *extra]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
