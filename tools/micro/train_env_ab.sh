#!/bin/bash
# A/B of run-time options of the training step through bench.py --workload c3 (options are read from KM_<NAME> at km_create):
#   VARS="KM_TRAIN_BM32_BELOW=192 KM_TRAIN_BM32_BELOW=512" BATCH=64 bash tools/micro/train_env_ab.sh
cd "$(dirname "$0")/../.."
for v in $VARS; do
    echo -n "$v: "
    env $v timeout -k 5 200 python3 bench.py --workload c3 --batch ${BATCH:-8} --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
