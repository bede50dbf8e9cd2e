#!/bin/bash
# Same-run A/B of whole-library variants through tools/bench_koemorph.py (256 windows x 30 frames):
#   LIBS="base relu base relu" bash tools/micro/kmmf_ab.sh      (names of tools/micro/bin/libkm_<name>.so)
cd "$(dirname "$0")/../.."
for l in $LIBS; do
    echo -n "$l: "
    KM_LIBRARY=tools/micro/bin/libkm_$l.so ONLY=0 WARM=300 ITERS=300 timeout -k 5 120 python3 tools/bench_koemorph.py | python3 -c "import sys, json; print(json.loads(sys.stdin.read())['ms_per_forward'])"
done
