#!/bin/bash
# SQ counters per launch of the training step (two rocprofv3 --pmc passes): bash tools/pmc_train.sh <windows> -> gpurun_out/pmc_train_<windows>.txt
set -e
B=${1:-64}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmct_$B
rm -rf $O; mkdir -p $O
KM_BENCH_NO_SPINUP=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 bench.py --workload c3 --batch $B --steps 12 --warmup 4 --cpu-seconds 0 > $O/a.log 2>&1
KM_BENCH_NO_SPINUP=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/b -- python3 bench.py --workload c3 --batch $B --steps 12 --warmup 4 --cpu-seconds 0 > $O/b.log 2>&1
python3 - "$O" "$B" <<'PY'
import csv, glob, sys, collections
O, B = sys.argv[1], sys.argv[2]
def load(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        per.setdefault(k, {"name": r["Kernel_Name"].split("(")[0][:40]})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]
out = open(f"gpurun_out/pmc_train_{B}.txt", "w")
res = {}
for tag in ("a", "b"):
    rows = load(f"{O}/{tag}")
    idx = [i for i, r in enumerate(rows) if "mel_power" in r["name"]]
    per = idx[-1] - idx[-2]
    acc = collections.OrderedDict()
    for a in idx[-7:-1]:
        for j in range(a, a + per):
            for c, v in rows[j].items():
                if c != "name":
                    acc.setdefault((j - a, rows[j]["name"]), collections.defaultdict(list))[c].append(v)
    for k, cs in acc.items():
        res.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
print("pos kernel                                    gui_cyc/xcd  mfma_busy/gui  cu_busy/gui  wave: issuing waiting stalled | valu/cu_busy lds/cu_busy bankconf/cu_busy", file=out)
for (pos, name), r in res.items():
    gui = r["GRBM_GUI_ACTIVE"] / 8
    cub = r["SQ_BUSY_CU_CYCLES"]
    wave = max(r["SQ_WAVE_CYCLES"], 1)
    print(f"{pos:3d} {name:42s} {gui:10.0f} {r['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / gui:13.3f} {cub / 256 / gui:12.3f}"
          f"        {r['SQ_ACTIVE_INST_ANY'] / wave:6.3f} {r['SQ_WAIT_ANY'] / wave:7.3f} {1 - (r['SQ_ACTIVE_INST_ANY'] + r['SQ_WAIT_ANY']) / wave:7.3f} |"
          f" {r['SQ_ACTIVE_INST_VALU'] / max(cub, 1):11.3f} {r['SQ_LDS_IDX_ACTIVE'] / max(cub, 1):10.3f} {r['SQ_LDS_BANK_CONFLICT'] / max(cub, 1):10.3f}", file=out)
out.close()
print(open(f"gpurun_out/pmc_train_{B}.txt").read())
PY
