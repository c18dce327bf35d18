#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) result: per-kernel dispatch statistics (the `--stats` view)
and, when the run collected PMC counters, per-kernel counter sums and per-dispatch averages.

    python tools/rocpd_summary.py gpurun_out/prof/stats_results.db > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('v2x::', '').replace('void ', '')
    return name[:70]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = list(c.execute("select %s, start, end, dispatch_id from kernels" % name_col))
    stats = {}
    for n, s, e, _ in rows:
        st = stats.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
    tot = sum(v[1] for v in stats.values()) or 1
    print("# %s" % path)
    print("%-72s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %8d %12.1f %10.2f %10.2f %10.2f %6.1f%%" % (k, v[0], v[1] / 1e3, v[1] / v[0] / 1e3, v[2] / 1e3, v[3] / 1e3,
                                                               100.0 * v[1] / tot))
    try:
        pm = list(c.execute("select kernel_name, counter_name, value from counters_collection"))
    except sqlite3.Error:
        pm = []
    if pm:
        agg = {}
        for n, cn, val in pm:
            a = agg.setdefault((short(n), cn), [0, 0.0])
            a[0] += 1; a[1] += val
        print("\n%-72s %-14s %8s %16s %16s" % ("kernel", "counter", "calls", "sum", "per_dispatch"))
        for (k, cn), (cnt, s) in sorted(agg.items()):
            print("%-72s %-14s %8d %16.1f %16.1f" % (k, cn, cnt, s, s / cnt))


if __name__ == "__main__":
    main(sys.argv[1])
