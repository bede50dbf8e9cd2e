# SQ counters of the fused KoeMorphModel kernels (256 windows x 30 frames): bash tools/pmc_koemorph.sh -> gpurun_out/kmm_pmc/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/kmm_pmc
ONLY=0 WARM=100 ITERS=100 timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/kmm_pmc/a -- python3 tools/bench_koemorph.py > gpurun_out/kmm_pmc/a.log 2>&1
echo pass A done
python3 - <<'PY' > gpurun_out/kmm_pmc/summary.txt
import collections, csv, glob
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob("gpurun_out/kmm_pmc/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for path in glob.glob("gpurun_out/kmm_pmc/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, cs in rows.items():
    if "kmmf" not in k:
        continue
    m = {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in cs.items()}
    d = dur.get(k, [0]); us = sum(d[len(d) // 4:]) / max(1, len(d[len(d) // 4:])) / 1e3
    gui = m["GRBM_GUI_ACTIVE"] / 8
    print(k[:60])
    print(f"  duration_us {us:.1f}  gui_active_cycles/xcd {gui:.0f}  => clock {gui / us / 1e3:.3f} GHz (under the profiler)")
    print(f"  mfma_busy/simd {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.0f}  util_of_gui {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / gui:.3f}  util_of_cu_busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (m['SQ_BUSY_CU_CYCLES'] / 256):.3f}")
    print(f"  mfma_flops {m['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512:.4g}  issuing {m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}  waiting {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}  wait_inst {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
PY
cat gpurun_out/kmm_pmc/summary.txt
