#!/bin/bash
# build the library with each flag set on the GPU box and time the training step at 8 and 64 windows
for flags in "$@"; do
  echo "== flags: $flags"
  KM_EXTRA_FLAGS="$flags" python -m koemorph_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  for v in 8 64; do
    echo -n "  b$v: "
    KM_ALLOW_STALE=1 python bench.py --workload c3 --batch $v --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])"
  done
done
