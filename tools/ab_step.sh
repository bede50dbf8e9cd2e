#!/bin/bash
# build the library with each flag set on the GPU box and time the C2 step (bench.py: value, ms/step, per-kernel ms)
for flags in "$@"; do
  KM_EXTRA_FLAGS="$flags" python -m koemorph_amd.build --force > /dev/null 2>&1 && echo "flags: $flags" && \
  for i in 1 2; do python bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d.get('kernel_ms'))"; done
done
