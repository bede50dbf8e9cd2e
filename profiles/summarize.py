#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/<dir>/*/...) into the small summaries committed under profiles/.

  python profiles/summarize.py stats   <kernel_stats.csv>                      -> prints a compact table
  python profiles/summarize.py traffic <fetch_counters.csv> <write_counters.csv> <out.json>

Traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
SEPARATE --pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE counts half the bytes of a coalesced
streaming read, so the read side is doubled.  Values are per launch (mean over the profiled launches).
"""
import collections
import csv
import json
import sys


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(name):
    for key in ("mel_power_rp_kernel", "mel_power_kernel", "core_fused_kernel", "emotion_kernel", "mel_log_kernel", "ema_scan", "smooth_kernel"):
        if key in name:
            return key
    return None


def main():
    if sys.argv[1] == "stats":
        for r in csv.DictReader(open(sys.argv[2])):
            print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
        return
    fetch = mean_by_kernel(sys.argv[2], "FETCH_SIZE")
    write = mean_by_kernel(sys.argv[3], "WRITE_SIZE")
    out = {}
    for name, f in fetch.items():
        k = short(name)
        if not k:
            continue
        w = write.get(name, 0.0)
        rec = out.setdefault(k, {"variants": {}})
        rec["variants"][name[:80]] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
                                      "hbm_bytes_per_launch": int((2.0 * f + w) * 1024)}
    for k, rec in out.items():   # headline = the variant with the most traffic (the one in the bench step)
        rec["hbm_bytes_per_launch"] = max(v["hbm_bytes_per_launch"] for v in rec["variants"].values())
    out["_method"] = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH + WRITE) * 1024"
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out.items() if k != "_method"}))


if __name__ == "__main__":
    main()
