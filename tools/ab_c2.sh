#!/bin/bash
# build the library with each flag set on the GPU box and time the C2 headline step (rotating inputs and cached)
for flags in "$@"; do
  echo "== flags: $flags"
  KM_EXTRA_FLAGS="$flags" python -m koemorph_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  KM_ALLOW_STALE=1 python bench.py --cpu-seconds 0 --no-split 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_cached_input'), d['value'])"
done
