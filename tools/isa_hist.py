#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S output:  tools/isa_hist.py file.s <mangled-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(key.split()[-1]) or (l.startswith("_Z") and key in l and ":" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
cnt = collections.Counter()
for l in lines[start + 1:end + 1]:
    l = l.strip()
    if not l or l[0] in ".;/" or l.endswith(":"):
        continue
    cnt[l.split()[0]] += 1
groups = collections.Counter()
for op, c in cnt.items():
    if op.startswith("v_pk_"): groups["v_pk_*"] += c
    elif op.startswith("v_") and "f32" in op: groups["v_*_f32 scalar"] += c
    elif op.startswith("v_"): groups["v_* int/mov/other"] += c
    elif op.startswith("ds_"): groups["ds_*"] += c
    elif op.startswith("s_"): groups["s_*"] += c
    elif op.startswith(("global_", "buffer_", "flat_")): groups["vmem"] += c
    else: groups["other"] += c
print("total", sum(cnt.values()), dict(groups))
print(cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 20))
