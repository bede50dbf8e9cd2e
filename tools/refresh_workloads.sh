#!/bin/bash
# Bench lines of every workload -> gpurun_out/wl/*.json (profiles/r03_workloads.json is assembled from them)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wl
python bench.py > gpurun_out/wl/c2.json 2>/dev/null
python bench.py --workload c3 > gpurun_out/wl/c3.json 2>/dev/null
python bench.py --workload c3 --batch 64 --cpu-seconds 0 > gpurun_out/wl/c3_b64.json 2>/dev/null
python bench.py --workload c3 --graph --cpu-seconds 0 > gpurun_out/wl/c3_graph.json 2>/dev/null
python bench.py --workload c4 > gpurun_out/wl/c4.json 2>/dev/null
python bench.py --workload c4 --heads 16 --cpu-seconds 0 > gpurun_out/wl/c4_h16.json 2>/dev/null
python bench.py --workload c5 > gpurun_out/wl/c5.json 2>/dev/null
python bench.py --workload c5 --batch 1024 > gpurun_out/wl/c5_1024.json 2>/dev/null
for f in gpurun_out/wl/*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"], d.get("ms_per_step_no_spinup"))
PY
done
