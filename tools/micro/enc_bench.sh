#!/bin/bash
# builds (here or on the GPU box) and runs the encoder timing harness for each combination of compiled-out parts
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/bin
for v in ${VARIANTS:-0 1 2 4 8 6 14 9}; do
    out=tools/micro/bin/enc_bench_$v
    [ -x $out ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I koemorph_amd/csrc -I include -DKM_ENC_SKIP=$v tools/micro/enc_bench.hip -o $out
done
if [ -z "$BUILD_ONLY" ]; then for v in ${VARIANTS:-0 1 2 4 8 6 14 9}; do tools/micro/bin/enc_bench_$v; done; fi
