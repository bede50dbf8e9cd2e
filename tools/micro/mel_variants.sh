#!/bin/bash
# A/B builds of the whole library with parts of mel_power_rp_kernel compiled out (KM_MEL_SKIP bits, km_mel.hip), to see
# what each part of the front end costs inside the real step:  KM_LIBRARY=tools/micro/bin/libkm_mel<bits>.so python bench.py ...
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/bin/obj
CS=koemorph_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$CS -Wno-unused-function"
for f in km_host.cpp km_wire.cpp km_core.hip km_generic.hip km_koemorph.hip km_kmmf.hip km_train.hip km_trainp.hip km_egemaps.hip km_data.hip km_api.hip; do
    o=tools/micro/bin/obj/${f%.*}.o
    newest=$(ls -t $CS/$f $CS/*.h include/*.h | head -1)
    [ $o -nt $newest ] || /opt/rocm/bin/hipcc $FLAGS -c $CS/$f -o $o &
done
wait
for v in ${VARIANTS:-0 1 2 4 8}; do
    /opt/rocm/bin/hipcc $FLAGS -DKM_MEL_SKIP=$v -c $CS/km_mel.hip -o tools/micro/bin/obj/km_mel_$v.o &
done
wait
for v in ${VARIANTS:-0 1 2 4 8}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/micro/bin/obj/km_host.o tools/micro/bin/obj/km_wire.o tools/micro/bin/obj/km_core.o \
        tools/micro/bin/obj/km_mel_$v.o tools/micro/bin/obj/km_generic.o tools/micro/bin/obj/km_koemorph.o tools/micro/bin/obj/km_kmmf.o tools/micro/bin/obj/km_train.o \
        tools/micro/bin/obj/km_trainp.o tools/micro/bin/obj/km_egemaps.o tools/micro/bin/obj/km_data.o tools/micro/bin/obj/km_api.o -o tools/micro/bin/libkm_mel$v.so
done
ls -la tools/micro/bin/*.so
