#!/usr/bin/env python3
"""LDS bank-conflict share per kernel from one rocprofv3 PMC pass (rocpd sqlite output):
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace -d out -o l -- python3 bench.py --steps 5 --warmup 2 ...
    python tools/pmc_lds.py out/l_results.db
conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent re-issuing conflicting accesses over the cycles it was busy)."""
import sqlite3
import sys

from pmc_common import bench_class, clean

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection").fetchall()
disp = {}
for did, kname, cname, v in rows:
    d = disp.setdefault(did, {"name": clean(kname)})
    d[cname] = d.get(cname, 0.0) + v
agg = {}
for d in disp.values():
    name = bench_class(d["name"]) or d["name"].split("(")[0][:70]
    if name.startswith("void at::") or "rocclr" in name:
        continue
    a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
    a[0] += 1
    a[1] += d.get("SQ_LDS_BANK_CONFLICT", 0.0); a[2] += d.get("SQ_LDS_IDX_ACTIVE", 0.0); a[3] += d.get("SQ_ACTIVE_INST_LDS", 0.0)
for name, (n, bc, act, inst) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{name:60s} launches {n:4d}  bank-conflict cycles / LDS-active cycles = {bc / act if act else 0:.3f}   (LDS-active {act / n:.3g} per launch)")
