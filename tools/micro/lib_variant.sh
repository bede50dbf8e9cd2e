#!/bin/bash
# One A/B build of the whole library with extra flags for ONE source:  bash tools/micro/lib_variant.sh <name> <source> <flags...>
# -> tools/micro/bin/libkm_<name>.so  (objects of the other sources are reused from tools/micro/bin/obj; load with KM_LIBRARY)
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift; shift
mkdir -p tools/micro/bin/obj
CS=koemorph_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$CS -Wno-unused-function"
ALL="km_host.cpp km_wire.cpp km_core.hip km_mel.hip km_generic.hip km_koemorph.hip km_kmmf.hip km_train.hip km_trainp.hip km_egemaps.hip km_data.hip km_api.hip"
OBJS=""
for f in $ALL; do
    [ $f = $src ] && continue
    o=tools/micro/bin/obj/${f%.*}.o
    newest=$(ls -t $CS/$f $CS/*.h include/*.h | head -1)
    [ $o -nt $newest ] || /opt/rocm/bin/hipcc $FLAGS -c $CS/$f -o $o &
    OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/$src -o tools/micro/bin/obj/${src%.*}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/micro/bin/obj/${src%.*}_$name.o -o tools/micro/bin/libkm_$name.so
echo tools/micro/bin/libkm_$name.so
