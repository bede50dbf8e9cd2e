#!/bin/bash
# build the library with each flag set on the GPU box and time the front-end kernel
for flags in "$@"; do
  KM_EXTRA_FLAGS="$flags" python -m koemorph_amd.build --force > /dev/null 2>&1 && KM_EXTRA_FLAGS="$flags" python tools/bench_mel.py 2>/dev/null
done
