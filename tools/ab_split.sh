#!/bin/bash
# same build: the C2 step with phases 2+3 in fp32 (default) and as split-bf16 products (KM_CORE_SPLIT = 3 / 6 terms)
for t in 0 6 3 0; do
  echo "KM_CORE_SPLIT=$t"
  KM_CORE_SPLIT=$t python bench.py --cpu-windows 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d.get('kernel_ms'))"
done
