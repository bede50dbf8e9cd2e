#!/bin/bash
# Per-kernel time, executed vector work and wait fractions of the GPU eGeMAPS front end (64 windows x 20 s), with the roofs each
# kernel is priced against: bash tools/prof_egemaps.sh -> gpurun_out/egemaps/summary.txt  (copied to profiles/r04_egemaps_roofline.txt)
#   bytes   algorithmic HBM bytes of the kernel per call / its time / 8 TB/s
#   valu    executed vector instructions x 64 lanes / its time / the chip's vector issue rate (256 CUs x 4 SIMDs x 16 lanes per
#           cycle at the measured clock): the fraction of the fp32 vector pipe the kernel keeps busy
#   wait    share of wave cycles spent in s_waitcnt / barrier (SQ_WAIT_ANY / SQ_WAVE_CYCLES)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/egemaps
rm -rf gpurun_out/egemaps/a
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d gpurun_out/egemaps/a -- python3 tools/bench_egemaps.py 64 10 > gpurun_out/egemaps/a.log 2>&1
python3 - <<'PY' > gpurun_out/egemaps/summary.txt
import collections, csv, glob
NW, SECONDS, SR = 64, 20.0, 16000
NF = int(SECONDS * 100)                      # 10 ms frames per window
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob("gpurun_out/egemaps/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for path in glob.glob("gpurun_out/egemaps/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
# algorithmic HBM bytes per call: what the kernel must read and write once (36 floats per frame record, 88 outputs per window)
samples = NW * SECONDS * SR * 4
recs = NW * NF * 36 * 4
alg = {"egm_peak_kernel": samples, "egm_frame_kernel": samples + recs, "egm_viterbi_kernel": 2 * NW * NF * 8 * 4,
       "egm_voiced_kernel": samples + recs, "egm_functional_kernel": recs + NW * 88 * 4}
print(f"eGeMAPSv02 functionals, {NW} windows x {SECONDS:.0f} s ({NF} frames each); parity unpinned (openSMILE absent) -- see DESIGN 3.12")
print(f"{'kernel':24s} {'us':>9s} {'HBM frac':>9s} {'VALU frac':>10s} {'LDS busy':>9s} {'wait':>6s}   bound")
tot = 0.0
for k in sorted(rows, key=lambda k: -sum(dur.get(k, [0]))):
    short = next((n for n in alg if n in k), None)
    if not short:
        continue
    cs = rows[k]
    m = {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in cs.items()}
    d = dur[k]; us = sum(d[len(d) // 4:]) / len(d[len(d) // 4:]) / 1e3
    tot += us
    ghz = m["GRBM_GUI_ACTIVE"] / 8 / us / 1e3
    hbm = alg[short] / (us * 1e-6) / 8e12
    valu = m["SQ_INSTS_VALU"] * 64 / (us * 1e-6) / (256 * 4 * 16 * ghz * 1e9)
    lds = m["SQ_ACTIVE_INST_LDS"] / 256 / (m["SQ_BUSY_CU_CYCLES"] / 256) if m.get("SQ_BUSY_CU_CYCLES") else 0.0
    wait = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    bound = "latency: dependent chains (waves wait, no pipe near its roof)" if max(hbm, valu) < 0.3 else ("vector pipe" if valu > hbm else "HBM")
    print(f"{short:24s} {us:9.1f} {hbm:9.4f} {valu:10.4f} {lds:9.3f} {wait:6.3f}   {bound}")
print(f"sum of kernels {tot / 1e3:.2f} ms per call of {NW} windows => {NW / (tot * 1e-6):.0f} windows/s")
PY
cat gpurun_out/egemaps/summary.txt
