#!/bin/bash
# usage: tools/build_scan_variant.sh <name> <extra hipcc flags...> -> variants/librodio_hip_<name>.so
# Only the scan kernels (rh_limit, rh_biquad_scan, rh_agc) are rebuilt with the flags; everything else is linked from rodio_amd/build/ (run rodio_amd/build.py first).
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p variants/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function"
objs=""
for f in rh_runtime rh_elementwise rh_resample rh_recurrence rh_stream rh_uniform rh_widemix rh_formats rh_wav rh_comm rh_pipeline rh_pipeline_plan rh_pipeline_stream rh_pipeline_sblk; do objs="$objs rodio_amd/build/$f.o"; done
for f in rh_limit rh_biquad_scan rh_agc; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c rodio_amd/csrc/$f.hip -o variants/obj_$name/$f.o &
  objs="$objs variants/obj_$name/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/librodio_hip_$name.so $objs -ldl
echo variants/librodio_hip_$name.so
