#!/bin/bash
# Per-kernel rocprofv3 stats of the legacy SimplifiedKoeMorphModel forward (256 windows from audio): bash tools/prof_legacy.sh -> gpurun_out/legacy/
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/legacy
WARM=100 ITERS=200 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/legacy/ks -o lg -- python3 tools/bench_legacy.py > gpurun_out/legacy/bench_prof.json 2> gpurun_out/legacy/prof.err
f=$(find gpurun_out/legacy/ks -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 profiles/summarize.py stats "$f" > gpurun_out/legacy/kernel_stats.txt
cat gpurun_out/legacy/bench_prof.json
