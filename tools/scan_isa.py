#!/usr/bin/env python3
"""Compile csrc/gemm.hip for gfx950 (device pass only, ~15 s, no GPU needed) and check two properties of
the persistent GEMM kernels that cost 10-15 % each when they break:
  * no `s_waitcnt vmcnt(0)` and no scratch access inside the inner K loop (either one drains the LDS-DMA
    queue in every K step);
  * no register spills inside that loop, and at most 64 bytes per lane outside it, in the instantiations the dispatcher
    actually launches.
Prints one line per kernel; exit status 1 on a violation.  Used by tests/test_isa_invariants.py."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rajni-vit_amd", "csrc", "gemm.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SPILL_BYTES_OUTSIDE_LOOP = 64


def kernel_id(mangled):
    """gemm_bf16_tn_stream<EPI, ALOAD, SF32, WM, WN, MI, NS, W8, TAG> -> tuple of ints"""
    m = re.search(r"gemm_bf16_tn_streamI(.*?)EEv", mangled)
    if m is None:          # the fp8 x fp8 kernels: <EPI, SF32, TAG> / <EPI>; every instantiation is dispatched
        m = re.search(r"gemm_f8_tn_(?:stream|wide)I(.*?)EEv", mangled)
        return ("f8",) + tuple(int(x[2:]) for x in re.findall(r"L[ib]\d+", m.group(1)))
    return tuple(int(x[2:]) for x in re.findall(r"L[ib]\d+", m.group(1)))


def dispatched(k):
    """instantiations launch_gemm picks without a test hook: everything except the 256x256 tiling with the
    fused patch loader (a patch embed wider than 1536 channels has K = 3*14*14, not a multiple of 64)."""
    if k[0] == "f8":
        return True
    epi, aload, sf32, wm, wn, mi, ns, w8, tag = k
    return not ((wm, wn, mi, ns) == (2, 4, 8, 2) and aload == 1)


def scan(asm_path):
    rows, name = [], None
    for line in open(asm_path):
        m = re.match(r"^(_ZN\S*gemm_(?:bf16_tn_stream|f8_tn_stream|f8_tn_wide)\S*):", line)
        if m:
            name, inner, drains, scratch, spills = m.group(1), False, 0, 0, None
            continue
        if name is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:", line):
            inner = "Depth=2" in line
        if "Inner Loop Header" in line:
            inner = True
        if inner and re.search(r"s_waitcnt vmcnt\(0\)\s*$", line):
            drains += 1
        if inner and "scratch_" in line:
            scratch += 1
        m = re.search(r"; ScratchSize: (\d+)", line)
        if m:
            spills = int(m.group(1))
            rows.append((kernel_id(name), drains, scratch, spills))
            name = None
    return rows


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "gemm.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", SRC, "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        rows = scan(out)
    bad = 0
    for k, drains, scratch, spills in sorted(rows, key=lambda r: tuple(map(str, r[0]))):
        prod = dispatched(k)
        # spills outside the K loop: up to SPILL_BYTES_OUTSIDE_LOOP per lane are tolerated (the 256 x 256 RESID instantiation -
        # 128 accumulators - keeps 36 bytes of its EDGE-tile epilogue's operands in scratch: a handful of scratch accesses
        # per tile, none in the K loop); anything inside the loop is a violation
        ok = (drains == 0 and scratch == 0 and spills <= SPILL_BYTES_OUTSIDE_LOOP) or not prod
        bad += not ok
        print(f"stream<{','.join(map(str, k))}>  in-loop vmcnt(0): {drains}  in-loop scratch: {scratch}  "
              f"scratch bytes/lane: {spills}  {'dispatched' if prod else 'test hook only'}  {'ok' if ok else 'VIOLATION'}")
    if not rows:
        print("no persistent GEMM kernels found in the ISA")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
