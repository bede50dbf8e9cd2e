#!/bin/bash
# A/B builds of the whole library with the FFT exchanges of mel_power_rp_kernel through LDS or in registers
# (-DKM_MEL_XCHG=0..3, km_mel.hip) into bin/libkm_xchg<v>.so:  KM_LIBRARY=tools/micro/bin/libkm_xchg<v>.so python bench.py ...
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/bin/obj
CS=koemorph_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$CS -Wno-unused-function"
OTHERS="km_host.cpp km_wire.cpp km_core.hip km_generic.hip km_koemorph.hip km_kmmf.hip km_train.hip km_trainp.hip km_egemaps.hip km_data.hip km_api.hip"
for f in $OTHERS; do
    o=tools/micro/bin/obj/${f%.*}.o
    newest=$(ls -t $CS/$f $CS/*.h include/*.h | head -1)
    [ $o -nt $newest ] || /opt/rocm/bin/hipcc $FLAGS -c $CS/$f -o $o &
done
wait
for v in ${VARIANTS:-0 1 2 3}; do
    /opt/rocm/bin/hipcc $FLAGS -DKM_MEL_XCHG=$v ${EXTRA} -c $CS/km_mel.hip -o tools/micro/bin/obj/km_mel_x$v.o &
done
wait
OBJS=""
for f in $OTHERS; do OBJS="$OBJS tools/micro/bin/obj/${f%.*}.o"; done
for v in ${VARIANTS:-0 1 2 3}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/micro/bin/obj/km_mel_x$v.o -o tools/micro/bin/libkm_xchg$v.so
done
ls -la tools/micro/bin/libkm_xchg*.so
