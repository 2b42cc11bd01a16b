#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 PMC pass (rocpd sqlite output):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE \
              --kernel-trace -d out_m -o m -- python3 bench.py --steps 5 --warmup 2 ...
    python tools/pmc_mfma.py out_m/m_results.db > profiles/rNN_x_mfma_pmc.json

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * CUs * 4): busy cycles of the matrix pipes summed over
every SIMD of the chip, over the SIMD-cycles the dispatch lasted (GRBM_GUI_ACTIVE is reported summed over the 8 XCDs;
4 SIMDs per CU; MI355X_MICROARCH.md: the counter counts cycles, 16 per v_mfma_f32_16x16x32_bf16).  It is the
fraction of the clock-for-clock MFMA roof; `achieved / peak` of bench.py's roofline additionally carries the clock
the chip held (1.9-2.1 GHz under these GEMMs against the 2.4 GHz of the datasheet peak).
`mops_bf16_per_launch` x 512 should equal the launch's 2 M N K (a MOPS unit = 512 flop) - printed as a cross-check."""
import json
import re
import sqlite3
import sys

CUS = 256
BENCH_CLASS = {0: "gemm_bf16_tn<bias>", 1: "gemm_bf16_tn<bias,gelu>", 2: "gemm_bf16_tn<bias,ls,resid>", 3: "gemm_bf16_tn<patch>"}
RESID_SQ = "gemm_bf16_tn<bias,ls,resid> K<=N"
COUNTERS = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F8",
            "GRBM_GUI_ACTIVE")


def clean(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if re.search(r"gemm_bf16_tn_stream<2,.*, 1, (?:true|false)>\(", name):
        name += " [proj]"
    return name


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection").fetchall()
    disp = {}
    for did, kname, cname, v in rows:
        d = disp.setdefault(did, {"name": clean(kname)})
        d[cname] = d.get(cname, 0.0) + v
    by_kernel, by_class = {}, {}
    for d in disp.values():
        name = d["name"]
        if name.startswith("void at::") or "rocclr" in name:
            continue
        k = by_kernel.setdefault(name, {"launches": 0, **{c: 0.0 for c in COUNTERS}})
        k["launches"] += 1
        for c in COUNTERS:
            k[c] += d.get(c, 0.0)
        m = re.search(r"gemm_bf16_tn_(?:stream|128x128)<(\d)", name)
        m8 = re.search(r"gemm_f8_tn_(?:stream|wide)<(\d)", name)
        if m or m8:
            if m8:
                cls = {0: "gemm_f8_tn<bias>", 4: "gemm_f8_tn<bias,gelu,requant>", 2: "gemm_f8_tn<bias,ls,resid>"}[int(m8.group(1))]
            else:
                cls = RESID_SQ if name.endswith("[proj]") else BENCH_CLASS[int(m.group(1))]
            c2 = by_class.setdefault(cls, {"launches": 0, **{c: 0.0 for c in COUNTERS}})
            c2["launches"] += 1
            for c in COUNTERS:
                c2[c] += d.get(c, 0.0)

    def summarise(v):
        simd_cycles = v["GRBM_GUI_ACTIVE"] / 8.0 * CUS * 4.0
        out = {"launches": v["launches"],
               "mfma_busy_frac": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4) if simd_cycles else None,
               "mfma_busy_cycles_per_launch": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["launches"]),
               "busy_cu_cycles_per_launch": round(v["SQ_BUSY_CU_CYCLES"] / v["launches"]),
               "gui_active_per_launch_sum_over_xcds": round(v["GRBM_GUI_ACTIVE"] / v["launches"]),
               "mops_bf16_per_launch": round(v["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / v["launches"])}
        if v["SQ_INSTS_VALU_MFMA_MOPS_F8"]:
            out["mops_f8_per_launch"] = round(v["SQ_INSTS_VALU_MFMA_MOPS_F8"] / v["launches"])
        return out

    json.dump({"note": __doc__.split("\n\n")[-1].replace("\n", " "),
               "by_bench_class": {k: summarise(v) for k, v in by_class.items()},
               "by_kernel": {k: summarise(v) for k, v in by_kernel.items() if v["SQ_VALU_MFMA_BUSY_CYCLES"] > 0}},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
