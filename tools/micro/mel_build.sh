#!/bin/bash
# One A/B build of the whole library with extra flags for km_mel.hip:  bash tools/micro/mel_build.sh <name> <flags...>
# -> tools/micro/bin/libkm_<name>.so  (objects of the other sources are reused from tools/micro/bin/obj)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
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
/opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/km_mel.hip -o tools/micro/bin/obj/km_mel_$name.o
OBJS=""
for f in $OTHERS; do OBJS="$OBJS tools/micro/bin/obj/${f%.*}.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/micro/bin/obj/km_mel_$name.o -o tools/micro/bin/libkm_$name.so
echo tools/micro/bin/libkm_$name.so
