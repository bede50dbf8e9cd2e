#!/bin/bash
# builds (here or on the GPU box) and runs the attention-chain timing harness; VARIANTS = "<KM_SC_SKIP>:<KM_VR_SKIP> ..."
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/bin
for v in ${VARIANTS:-0:0}; do
    sc=${v%%:*}; vr=${v##*:}; out=tools/micro/bin/attn_bench_${sc}_${vr}
    [ -x $out ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I koemorph_amd/csrc -I include -DKM_SC_SKIP=$sc -DKM_VR_SKIP=$vr tools/micro/attn_bench.hip -o $out
done
if [ -z "$BUILD_ONLY" ]; then for v in ${VARIANTS:-0:0}; do sc=${v%%:*}; vr=${v##*:}; for h in ${HEADS:-8 16}; do tools/micro/bin/attn_bench_${sc}_${vr} $h; done; done; fi
