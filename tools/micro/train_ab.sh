#!/bin/bash
# Same-run A/B of whole-library variants through bench.py --workload c3:  LIBS="a b a b" BATCH=64 bash tools/micro/train_ab.sh
cd "$(dirname "$0")/../.."
for l in $LIBS; do
    echo -n "$l: "
    KM_LIBRARY=tools/micro/bin/libkm_$l.so timeout -k 5 200 python3 bench.py --workload c3 --batch ${BATCH:-8} --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
