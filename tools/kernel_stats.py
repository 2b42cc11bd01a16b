#!/usr/bin/env python3
"""Per-kernel summary (CSV) of a rocprofv3 --kernel-trace run from its rocpd sqlite output:
    rocprofv3 --kernel-trace --stats -d out -o run -- python3 bench.py ...
    python tools/kernel_stats.py out/run_results.db > profiles/rNN_kernel_stats.csv"""
import csv
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, duration from kernels").fetchall()
agg = {}
for name, dur in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if re.search(r"gemm_bf16_tn_stream<2,.*, 1>\(", name):     # TAG = 1: the K <= N residual launches (projection)
        name += " [K<=N: bench class gemm_bf16_tn<bias,ls,resid> K<=N]"
    elif re.search(r"gemm_bf16_tn_stream<2,", name):
        name += " [K>N: bench class gemm_bf16_tn<bias,ls,resid>]"
    elif re.search(r"gemm_f8_tn_stream<2, (?:true|false), 1>\(", name):   # proj on e4m3 attention output
        name += " [K<=N: bench class gemm_f8_tn<bias,ls,resid> K<=N]"
    a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
total = sum(a[1] for a in agg.values())
w = csv.writer(sys.stdout)
w.writerow(["name", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
for name, (n, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([name, n, round(t / 1e3, 1), round(t / n / 1e3, 2), round(lo / 1e3, 2), round(hi / 1e3, 2), round(100 * t / total, 2)])
