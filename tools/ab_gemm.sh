#!/bin/bash
# build the library with each flag set on the GPU box and time the products that run on the NT GEMM
for flags in "$@"; do
  KM_EXTRA_FLAGS="$flags" python -m koemorph_amd.build --force > /dev/null 2>&1 && echo "flags: $flags" && \
  for i in 1 2; do ONLY=0 python tools/bench_koemorph.py 2>/dev/null | cut -c1-130; B=256 python tools/bench_c4.py 2>/dev/null | tail -1 | cut -c1-130; done
done
