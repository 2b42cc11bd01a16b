#!/usr/bin/env python3
"""HBM-side traffic per kernel launch from two rocprofv3 PMC passes (rocpd sqlite output):

    rocprofv3 --pmc FETCH_SIZE TCC_EA0_WRREQ_64B_sum --kernel-trace -d out_f -o f -- python3 bench.py --steps 5 --warmup 2 ... > f.json
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out_w -o w -- python3 bench.py --steps 5 --warmup 2 ... > w.json
    python tools/pmc_traffic.py out_f/f_results.db out_w/w_results.db [f.json] > profiles/rNN_hbm_traffic_pmc.json

Counters are in KiB; FETCH_SIZE is doubled on gfx950 (it tallies 128-byte requests at 64 bytes:
MI355X_MICROARCH.md, HBM / rocprofv3 section).  These are L2 <-> fabric bytes: Infinity Cache hits included.

Self-check (tools/pmc_common.py): with the bench.py line of a pass given, `coverage` = measured / exact output bytes of
the QKV launches - through WRITE_SIZE in the write pass and through TCC_EA0_WRREQ_64B_sum (64-byte write requests: what
16-byte-per-lane streaming stores become) in the fetch pass; a pass below 0.98 is rescaled, below 0.5 refused."""
import json
import sqlite3
import sys

from pmc_common import WRITE_CHECK_CLASSES, bench_class, clean, expected_fc1, judge, provenance


def per_kernel(db_path, counter):
    """{kernel name: (launches, sum)} of one counter"""
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select kernel_name, value from counters_collection where counter_name = ? order by dispatch_id",
                       (counter,)).fetchall()
    out = {}
    for name, v in rows:
        name = clean(name)
        c, t = out.get(name, (0, 0.0))
        out[name] = (c + 1, t + v)
    return out


def class_mean(per, cls_names, scale):
    n = tot = 0.0
    for name, (c, t) in per.items():
        if bench_class(name) in cls_names:
            n += c; tot += t
    return tot * scale / n if n else None


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    fetch_wr64 = per_kernel(sys.argv[1], "TCC_EA0_WRREQ_64B_sum")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    exp = expected_fc1(sys.argv[3]) if len(sys.argv) > 3 else None
    cov_w = cov_f = None
    if exp:
        for cls, key in zip(WRITE_CHECK_CLASSES, ("qkv_out_bytes_with_head", "qkv_out_bytes")):
            w = class_mean(write, (cls,), 1024.0)
            if w is not None and cov_w is None:
                cov_w = w / exp[key]
            f = class_mean(fetch_wr64, (cls,), 64.0)
            if f is not None and cov_f is None:
                cov_f = f / exp[key]
    scale_w, note_w = judge(cov_w, "write pass (WRITE_SIZE of the QKV-class launches vs rows x 3C x 2 bytes)")
    scale_f, note_f = judge(cov_f, "fetch pass (TCC_EA0_WRREQ_64B_sum x 64 of the QKV-class launches vs the same)")
    by_kernel, by_class = {}, {}
    for name, (n, kib) in fetch.items():
        if name.startswith("void at::") or "rocclr" in name:
            continue
        wn, wkib = write.get(name, (n, 0.0))
        f_mb, w_mb = 2.0 * kib * 1024 / n / 1e6 * scale_f, wkib * 1024 / wn / 1e6 * scale_w
        by_kernel[name] = {"launches": n, "fetch_MB_corrected": round(f_mb, 1), "write_MB": round(w_mb, 1)}
        cls = bench_class(name)
        if cls:
            c = by_class.setdefault(cls, {"launches": 0, "f": 0.0, "w": 0.0})
            c["launches"] += n; c["f"] += f_mb * n; c["w"] += w_mb * n
    out = {"note": __doc__.split("\n\n")[2].replace("\n", " "),
           **provenance(),
           "csrc_fingerprint_of_profiled_run": exp["csrc_fingerprint_of_run"] if exp else None,
           "coverage": {"write_pass": round(cov_w, 4) if cov_w else None, "fetch_pass": round(cov_f, 4) if cov_f else None,
                        "notes": [note_w, note_f]},
           "by_bench_class": {k: {"launches": v["launches"], "fetch_MB_per_launch": round(v["f"] / v["launches"], 1),
                                  "write_MB_per_launch": round(v["w"] / v["launches"], 1),
                                  "hbm_bytes_per_launch": int((v["f"] + v["w"]) / v["launches"] * 1e6)}
                              for k, v in by_class.items()},
           "by_kernel": by_kernel}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
