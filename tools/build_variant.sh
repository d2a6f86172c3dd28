#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>  -> variants/librodio_hip_<name>.so (diagnostic builds)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p variants/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function"
objs=""
for f in rh_runtime rh_elementwise rh_resample rh_recurrence rh_limit rh_agc rh_biquad_scan rh_stream rh_uniform rh_widemix rh_formats rh_wav rh_comm rh_pipeline rh_pipeline_plan rh_pipeline_stream rh_pipeline_sblk; do
  if [ "${name#sblk}" != "$name" ] && [ $f != rh_pipeline_sblk ] && [ -f rodio_amd/build/$f.o ]; then cp rodio_amd/build/$f.o variants/obj_$name/$f.o; objs="$objs variants/obj_$name/$f.o"; continue; fi
  if [ "${f#rh_pipeline}" != "$f" ] || [ $f = rh_limit ] || [ $f = rh_agc ] || [ ! -f variants/obj_$name/$f.o ] || [ rodio_amd/csrc/$f.hip -nt variants/obj_$name/$f.o ] || [ include/rodio_hip.h -nt variants/obj_$name/$f.o ]; then /opt/rocm/bin/hipcc $FLAGS "$@" -c rodio_amd/csrc/$f.hip -o variants/obj_$name/$f.o & fi
  objs="$objs variants/obj_$name/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/librodio_hip_$name.so $objs -ldl
echo variants/librodio_hip_$name.so
