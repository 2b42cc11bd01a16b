#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 PMC pass (rocpd sqlite output):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE \
              --kernel-trace -d out_m -o m -- python3 bench.py --steps 5 --warmup 2 ... > m.json
    python tools/pmc_mfma.py out_m/m_results.db [m.json] > profiles/rNN_x_mfma_pmc.json

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * CUs * 4): busy cycles of the matrix pipes summed over
every SIMD of the chip, over the SIMD-cycles the dispatch lasted (GRBM_GUI_ACTIVE is reported summed over the 8 XCDs;
4 SIMDs per CU; MI355X_MICROARCH.md: the counter counts cycles, 16 per v_mfma_f32_16x16x32_bf16).  It is the
fraction of the clock-for-clock MFMA roof; `achieved / peak` of bench.py's roofline additionally carries the clock
the chip held (1.9-2.1 GHz under these GEMMs against the 2.4 GHz of the datasheet peak).

Self-check (tools/pmc_common.py): with the bench.py line of the pass given, `coverage.sq` = SQ_INSTS_VALU_MFMA_MOPS_* x
512 / (2 M N K) of the FC1 launches (a MOPS unit = 512 flop; exact), and `coverage.grbm_clock_GHz` = GRBM_GUI_ACTIVE / 8 /
kernel duration of the same launches - the shader clock the pass implies (1.9-2.4 GHz when all 8 XCDs answered,
proportionally less otherwise).  The SQ numerator is rescaled by 1 / coverage.sq when that is below 0.98; the GRBM
denominator is rescaled by 8 / k only when the implied clock says k < 8 XCDs answered (implied clock < 1.7 GHz)."""
import json
import sqlite3
import sys

from pmc_common import CHECK_CLASSES, bench_class, clean, expected_fc1, judge, provenance

CUS = 256
COUNTERS = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F8",
            "GRBM_GUI_ACTIVE")


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection").fetchall()
    exp = expected_fc1(sys.argv[2]) if len(sys.argv) > 2 else None
    durations = {}
    try:      # --kernel-trace rows of the same pass: mean duration (ns) per kernel name
        for name, dur in cur.execute("select name, duration from kernels").fetchall():
            a = durations.setdefault(clean(name), [0, 0.0])
            a[0] += 1; a[1] += dur
    except sqlite3.Error:
        pass
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
        cls = bench_class(name)
        if cls:
            c2 = by_class.setdefault(cls, {"launches": 0, "ns": 0.0, "ns_n": 0, **{c: 0.0 for c in COUNTERS}})
            c2["launches"] += 1
            for c in COUNTERS:
                c2[c] += d.get(c, 0.0)
    for name, (n, ns) in durations.items():
        cls = bench_class(name)
        if cls in by_class:
            by_class[cls]["ns"] += ns; by_class[cls]["ns_n"] += n

    # ---- coverage of the pass, from the FC1 class
    cov_sq = clock = None
    for cls in CHECK_CLASSES:
        v = by_class.get(cls)
        if v and exp and cov_sq is None:
            mops = (v["SQ_INSTS_VALU_MFMA_MOPS_BF16"] + v["SQ_INSTS_VALU_MFMA_MOPS_F8"]) / v["launches"]
            if mops > 0:
                cov_sq = mops * 512.0 / exp["flops"]
        if v and v["ns_n"] and clock is None:
            clock = (v["GRBM_GUI_ACTIVE"] / v["launches"] / 8.0) / (v["ns"] / v["ns_n"])     # cycles per ns = GHz
    scale_sq, note_sq = judge(cov_sq, "SQ counters (MOPS x 512 of the FC1 launches vs 2 M N K)")
    scale_grbm, note_grbm = 1.0, f"GRBM_GUI_ACTIVE / 8 / duration of the FC1 launches = {clock:.2f} GHz" if clock else "GRBM: no kernel durations in the pass"
    if clock is not None and clock < 1.7:
        k = max(1, min(8, round(clock / 2.05 * 8)))
        scale_grbm = 8.0 / k
        note_grbm += f" - implies {k} of 8 XCDs answered: GRBM_GUI_ACTIVE rescaled by 8 / {k}"

    def summarise(v):
        simd_cycles = v["GRBM_GUI_ACTIVE"] * scale_grbm / 8.0 * CUS * 4.0
        out = {"launches": v["launches"],
               "mfma_busy_frac": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] * scale_sq / simd_cycles, 4) if simd_cycles else None,
               "mfma_busy_cycles_per_launch": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] * scale_sq / v["launches"]),
               "busy_cu_cycles_per_launch": round(v["SQ_BUSY_CU_CYCLES"] * scale_sq / v["launches"]),
               "gui_active_per_launch_sum_over_xcds": round(v["GRBM_GUI_ACTIVE"] * scale_grbm / v["launches"]),
               "mops_bf16_per_launch": round(v["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * scale_sq / v["launches"])}
        if v["SQ_INSTS_VALU_MFMA_MOPS_F8"]:
            out["mops_f8_per_launch"] = round(v["SQ_INSTS_VALU_MFMA_MOPS_F8"] * scale_sq / v["launches"])
        return out

    json.dump({"note": __doc__.split("\n\n")[2].replace("\n", " "),
               **provenance(),
               "csrc_fingerprint_of_profiled_run": exp["csrc_fingerprint_of_run"] if exp else None,
               "coverage": {"sq": round(cov_sq, 4) if cov_sq else None, "grbm_clock_GHz": round(clock, 3) if clock else None,
                            "notes": [note_sq, note_grbm]},
               "by_bench_class": {k: summarise(v) for k, v in by_class.items()},
               "by_kernel": {k: summarise(v) for k, v in by_kernel.items() if v["SQ_VALU_MFMA_BUSY_CYCLES"] > 0}},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
