#!/usr/bin/env python3
"""Mean PMC counter values per kernel from rocprofv3 --pmc output (csv or rocpd sqlite):
   tools/pmc_summary.py <dir> [kernel-substring]"""
import collections, csv, glob, sqlite3, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), 0))
for path in glob.glob(sys.argv[1] + "/**/*results.db", recursive=True):
    con = sqlite3.connect(path)
    for name, ctr, val, dur in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        rows[name][ctr].append((float(val), dur))
key = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in rows.items():
    if key not in k:
        continue
    print(k[:100])
    for c, v in sorted(cs.items()):
        v = v[len(v) // 4:]            # drop warm-up launches
        print(f"   {c:32s} n={len(v):4d} mean={sum(x for x, _ in v) / len(v):.4g}  dur_us={sum(d for _, d in v) / len(v) / 1e3:.1f}")
