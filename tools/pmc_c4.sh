#!/bin/bash
# SQ counters of the C4 kernels: bash tools/pmc_c4.sh [heads] -> gpurun_out/pmc_c4_<heads>.txt
set -e
H=${1:-8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcc4_$H
rm -rf $O; mkdir -p $O
KM_BENCH_OPTIONS=no_core_merge=1 KM_BENCH_NO_SPINUP=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --output-format csv -d $O/a -- python3 bench.py --workload c4 --heads $H --steps 20 --warmup 5 --cpu-seconds 0 > $O/a.log 2>&1
python3 - "$O" "$H" <<'PY'
import csv, glob, sys, collections
O, H = sys.argv[1], sys.argv[2]
f = glob.glob(f"{O}/a/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = open(f"gpurun_out/pmc_c4_{H}.txt", "w")
print("kernel                                        gui_cyc/xcd mfma_busy/gui cu_busy/gui | wave: issuing waiting stalled | valu/cu_busy lds/cu_busy", file=out)
for k, cs in acc.items():
    m = {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in cs.items()}
    if "GRBM_GUI_ACTIVE" not in m or m["GRBM_GUI_ACTIVE"] < 8 * 20000: continue
    gui = m["GRBM_GUI_ACTIVE"] / 8; cub = m["SQ_BUSY_CU_CYCLES"]; wave = max(m["SQ_WAVE_CYCLES"], 1)
    print(f"{k:46s} {gui:10.0f} {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / gui:12.3f} {cub / 256 / gui:11.3f} |      {m['SQ_ACTIVE_INST_ANY'] / wave:7.3f} {m['SQ_WAIT_ANY'] / wave:7.3f} {1 - (m['SQ_ACTIVE_INST_ANY'] + m['SQ_WAIT_ANY']) / wave:7.3f} | {m['SQ_ACTIVE_INST_VALU'] / cub:11.3f} {m['SQ_LDS_IDX_ACTIVE'] / cub:10.3f}", file=out)
out.close()
print(open(f"gpurun_out/pmc_c4_{H}.txt").read())
PY
