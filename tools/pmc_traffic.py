#!/usr/bin/env python3
"""HBM-side traffic per kernel launch from two rocprofv3 PMC passes (rocpd sqlite output):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out_f -o f -- python bench.py --steps 5 --warmup 2 ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out_w -o w -- python bench.py --steps 5 --warmup 2 ...
    python tools/pmc_traffic.py out_f/f_results.db out_w/w_results.db > profiles/rNN_hbm_traffic_pmc.json

Counters are in KiB; FETCH_SIZE is doubled on gfx950 (it tallies 128-byte requests at 64 bytes:
MI355X_MICROARCH.md, HBM / rocprofv3 section).  These are L2 <-> fabric bytes: Infinity Cache hits included."""
import json
import re
import sqlite3
import sys

BENCH_CLASS = {  # GEMM template arguments <EPI, ...> -> bench.py kernel class
    0: "gemm_bf16_tn<bias>", 1: "gemm_bf16_tn<bias,gelu>", 2: "gemm_bf16_tn<bias,ls,resid>", 3: "gemm_bf16_tn<patch>"}
RESID_SQ = "gemm_bf16_tn<bias,ls,resid> K<=N"   # the projection: same kernel as fc2, every other residual launch


def per_kernel(db_path, counter):
    """{kernel name: (launches, sum KiB)}; the K <= N residual launches (the projection) run an instantiation of
    their own (template argument TAG = 1) and are listed as '<name> [proj]'."""
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select kernel_name, value from counters_collection where counter_name = ? order by dispatch_id",
                       (counter,)).fetchall()
    out = {}
    for name, v in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        if re.search(r"gemm_bf16_tn_stream<2,.*, 1, (?:true|false)>\(", name):
            name += " [proj]"
        c, t = out.get(name, (0, 0.0))
        out[name] = (c + 1, t + v)
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    by_kernel, by_class = {}, {}
    for name, (n, kib) in fetch.items():
        if name.startswith("void at::") or "rocclr" in name:
            continue
        wn, wkib = write.get(name, (n, 0.0))
        f_mb, w_mb = 2.0 * kib * 1024 / n / 1e6, wkib * 1024 / wn / 1e6
        by_kernel[name] = {"launches": n, "fetch_MB_corrected": round(f_mb, 1), "write_MB": round(w_mb, 1)}
        m = re.search(r"gemm_bf16_tn_(?:stream|128x128)<(\d)", name)
        m8 = re.search(r"gemm_f8_tn_(?:stream|wide)<(\d)", name)
        if m or m8:
            if m8:
                cls = {0: "gemm_f8_tn<bias>", 4: "gemm_f8_tn<bias,gelu,requant>", 2: "gemm_f8_tn<bias,ls,resid>"}[int(m8.group(1))]
            else:
                cls = RESID_SQ if name.endswith("[proj]") else BENCH_CLASS[int(m.group(1))]
            c = by_class.setdefault(cls, {"launches": 0, "f": 0.0, "w": 0.0})
            c["launches"] += n; c["f"] += f_mb * n; c["w"] += w_mb * n
    out = {"note": __doc__.split("\n\n")[-1].replace("\n", " "),
           "by_bench_class": {k: {"launches": v["launches"], "fetch_MB_per_launch": round(v["f"] / v["launches"], 1),
                                  "write_MB_per_launch": round(v["w"] / v["launches"], 1),
                                  "hbm_bytes_per_launch": int((v["f"] + v["w"]) / v["launches"] * 1e6)}
                              for k, v in by_class.items()},
           "by_kernel": by_kernel}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
