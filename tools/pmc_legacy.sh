#!/bin/bash
# SQ counters of the legacy model's kernels: bash tools/pmc_legacy.sh -> gpurun_out/pmc_legacy.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/pmcl
rm -rf $O; mkdir -p $O
WARM=20 ITERS=30 timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --output-format csv -d $O/a -- python3 tools/bench_legacy.py > $O/a.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(f"{O}/a/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = open("gpurun_out/pmc_legacy.txt", "w")
print("kernel                                        gui_cyc/xcd mfma_busy/gui cu_busy/gui | wave: issuing waiting stalled | valu/cu_busy lds/cu_busy", file=out)
for k, cs in acc.items():
    m = {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in cs.items()}
    if "GRBM_GUI_ACTIVE" not in m or m["GRBM_GUI_ACTIVE"] < 8 * 5000: continue
    gui = m["GRBM_GUI_ACTIVE"] / 8; cub = m["SQ_BUSY_CU_CYCLES"]; wave = max(m["SQ_WAVE_CYCLES"], 1)
    print(f"{k:46s} {gui:10.0f} {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / gui:12.3f} {cub / 256 / gui:11.3f} |      {m['SQ_ACTIVE_INST_ANY'] / wave:7.3f} {m['SQ_WAIT_ANY'] / wave:7.3f} {1 - (m['SQ_ACTIVE_INST_ANY'] + m['SQ_WAIT_ANY']) / wave:7.3f} | {m['SQ_ACTIVE_INST_VALU'] / cub:11.3f} {m['SQ_LDS_IDX_ACTIVE'] / cub:10.3f}", file=out)
out.close()
print(open("gpurun_out/pmc_legacy.txt").read())
PY
