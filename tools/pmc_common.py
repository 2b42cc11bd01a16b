"""Shared by tools/pmc_traffic.py and tools/pmc_mfma.py: kernel-name -> bench.py class, and the EXACT per-launch
quantities of a bench.py run (flops, output bytes per GEMM class) that the PMC passes are checked against.

Why: rocprofv3 counter passes on this pool do not always cover the whole chip (the same 110.7 MB fc2 output read
95.0 / 84.1 / 110.7 MB in three profile sets: 0.86 / 0.76 / 1.0 of the XCDs answered).  Every pass therefore carries a
`coverage` = measured / exact for a quantity that is known exactly (output bytes of the QKV launches through
WRITE_SIZE / TCC_EA0_WRREQ_64B; 2 M N K of the FC1 launches through SQ_INSTS_VALU_MFMA_MOPS_*), the tool rescales by it
when it is below 0.98 and refuses below 0.5 or above 1.05, and the JSON says which sources it was taken on."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from srchash import csrc_fingerprint  # noqa: E402

BENCH_CLASS = {0: "gemm_bf16_tn<bias>", 1: "gemm_bf16_tn<bias,gelu>", 2: "gemm_bf16_tn<bias,ls,resid>", 3: "gemm_bf16_tn<patch>"}
RESID_SQ = "gemm_bf16_tn<bias,ls,resid> K<=N"   # the projection: same kernel as fc2, an instantiation of its own (TAG = 1)
F8_RESID_SQ = "gemm_f8_tn<bias,ls,resid> K<=N"    # proj on e4m3 attention output (TAG = 1)
F8_CLASS = {0: "gemm_f8_tn<bias>", 4: "gemm_f8_tn<bias,gelu,requant>", 2: "gemm_f8_tn<bias,ls,resid>"}
CHECK_CLASSES = ("gemm_bf16_tn<bias,gelu>", "gemm_f8_tn<bias,gelu,requant>")   # FC1 (MFMA count): one shape family, nothing else in the class
WRITE_CHECK_CLASSES = ("gemm_bf16_tn<bias>", "gemm_f8_tn<bias>")              # QKV (+ the head in the bf16 class): output bytes
# (FC1's hidden activations leave as NON-TEMPORAL stores since round 3: they reach the fabric partly as 32-byte requests and
#  WRITE_SIZE reads 1.35 x their bytes - a fact about that store flavour, kept out of the coverage check)


def clean(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if re.search(r"gemm_bf16_tn_stream<2,.*, 1>\(", name) or re.search(r"gemm_f8_tn_stream<2, (?:true|false), 1>\(", name):
        name += " [proj]"
    return name


def bench_class(name):
    """bench.py kernel class of a (cleaned) kernel name, or None"""
    m = re.search(r"gemm_bf16_tn_(?:stream|128x128)<(\d)", name)
    if m:
        return RESID_SQ if name.endswith("[proj]") else BENCH_CLASS[int(m.group(1))]
    m8 = re.search(r"gemm_f8_tn_(?:stream|wide)<(\d)", name)
    if m8:
        return F8_RESID_SQ if name.endswith("[proj]") else F8_CLASS[int(m8.group(1))]
    return None


def expected_fc1(bench_json_path):
    """Exact per-launch AVERAGES of the run that wrote this bench.py line: `flops` of an FC1 launch (B x Np_i rows, Np_i =
    tokens after block i's selection) and `qkv_out_bytes` of a launch of the QKV class (bf16 [B x N_i, 3C]; in the bf16
    format the class also holds the classifier head: `qkv_out_bytes_with_head`)."""
    with open(bench_json_path) as f:
        line = [ln for ln in f.read().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    dims, counts = d["config"]["dims"], d["config"]["token_counts"]
    B, C, hid = dims["batch_per_gpu"], dims["C"], dims["hidden"]
    after = counts[1:] + [counts[-1]]          # tokens leaving block i = tokens entering block i + 1 (the last block never prunes here)
    rows = sum(B * n for n in after) / len(after)
    qkv = [B * n * 3 * C * 2.0 for n in counts]
    head = B * ((dims.get("classes", 1000) + 7) // 8 * 8) * 2.0
    return {"flops": 2.0 * rows * hid * C, "qkv_out_bytes": sum(qkv) / len(qkv),
            "qkv_out_bytes_with_head": (sum(qkv) + head) / (len(qkv) + 1),
            "csrc_fingerprint_of_run": d["config"].get("csrc_fingerprint")}


def judge(cov, what):
    """coverage -> (scale factor to apply, note); refuses an implausible pass"""
    if cov is None:
        return 1.0, f"{what}: coverage unknown (no bench JSON or no FC1 launches in the pass)"
    if cov < 0.5 or cov > 1.05:
        raise SystemExit(f"{what}: coverage {cov:.3f} is implausible - refusing to write a profile from this pass")
    if cov < 0.98:
        return 1.0 / cov, f"{what}: coverage {cov:.3f} < 0.98 - every value of this pass rescaled by 1 / coverage"
    return 1.0, f"{what}: coverage {cov:.3f}"


def provenance():
    return {"csrc_fingerprint": csrc_fingerprint(), "git_commit": os.environ.get("RAJNI_GIT_HEAD")}
