#!/bin/bash
# Per-kernel rocprofv3 stats of the C5 stream tick at 1024 streams: bash tools/prof_c5.sh -> gpurun_out/c5/
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/c5
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5/ks -o c5 -- python3 bench.py --workload c5 --batch 1024 --steps 300 --cpu-seconds 0 > gpurun_out/c5/bench_prof.json 2> gpurun_out/c5/prof.err
f=$(find gpurun_out/c5/ks -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 profiles/summarize.py stats "$f" > gpurun_out/c5/kernel_stats.txt
python3 bench.py --workload c5 --batch 1024 --cpu-seconds 0 > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.err
cat gpurun_out/c5/bench.json
