#!/bin/bash
# Per-launch durations of the training step (rocprofv3 kernel trace): bash tools/trace_train.sh <batch> -> gpurun_out/trace_train_<batch>.txt
set -e
B=${1:-8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/tt_$B
KM_BENCH_NO_SPINUP=1 timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt_$B -- python3 bench.py --workload c3 --batch $B --steps 30 --warmup 5 --cpu-seconds 0 > gpurun_out/tt_$B.log 2>&1
python3 - "$B" <<'PY'
import csv, glob, sys, collections
B = sys.argv[1]
f = glob.glob(f"gpurun_out/tt_{B}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:60] for r in rows]
# find the period: launches per step = distance between consecutive front-end launches
idx = [i for i, n in enumerate(names) if "mel_power" in n]
per = idx[-1] - idx[-2]
steps = [(idx[k], idx[k + 1]) for k in range(len(idx) - 12, len(idx) - 2)]
acc = collections.OrderedDict()
gaps = []
for a, b in steps:
    for j in range(a, b):
        d = (int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) / 1e3
        acc.setdefault((j - a, names[j]), []).append(d)
    gaps.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
out = open(f"gpurun_out/trace_train_{B}.txt", "w")
tot = 0
for (pos, n), v in acc.items():
    m = sum(v) / len(v); tot += m
    print(f"{pos:3d} {n:62s} {m:8.1f} us", file=out)
print(f"launches per step {per}; sum of kernel time {tot:.1f} us; step period {sum(gaps)/len(gaps):.1f} us UNDER THE PROFILER (rocprofv3 adds host time per launch: "
      f"a short step becomes host-bound; the un-profiled period is bench.py's ms_per_step, profiles/r04_workloads.json)", file=out)
print(open(f"gpurun_out/trace_train_{B}.txt").read())
PY
