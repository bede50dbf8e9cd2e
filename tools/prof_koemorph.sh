#!/bin/bash
# Per-kernel rocprofv3 stats of the legacy KoeMorphModel forward (256 windows x 30 frames): bash tools/prof_koemorph.sh -> gpurun_out/kmm/
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/kmm
ONLY=0 WARM=300 ITERS=300 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kmm/ks -o kmm -- python3 tools/bench_koemorph.py > gpurun_out/kmm/bench_prof.json 2> gpurun_out/kmm/prof.err
f=$(find gpurun_out/kmm/ks -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 profiles/summarize.py stats "$f" > gpurun_out/kmm/kernel_stats.txt
WARM=300 ITERS=300 python3 tools/bench_koemorph.py > gpurun_out/kmm/bench.json 2> gpurun_out/kmm/bench.err
cat gpurun_out/kmm/bench.json
