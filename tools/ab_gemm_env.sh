#!/bin/bash
# same build, environment switches of launch_gemm: time the workloads that run on the generic GEMMs
for e in "$@"; do
  echo "env: $e"
  for i in 1 2; do
    env $e ONLY=0 python tools/bench_koemorph.py 2>/dev/null | cut -c1-128
    env $e ONLY=1 python tools/bench_koemorph.py 2>/dev/null | cut -c1-128
    env $e python tools/bench_train.py --batch 8 2>/dev/null | cut -c150-215
    env $e python tools/bench_train.py --batch 64 2>/dev/null | cut -c150-215
    env $e B=256 python tools/bench_c4.py 2>/dev/null | cut -c80-130
  done
done
