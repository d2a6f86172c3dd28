# usage: bash tools/sweep_libs2.sh <lib names...>   (variants/librodio_hip_<name>.so; "main" = the shipped library)
cd $GRAFT_REPO_ROOT
export RH_BENCH_NO_PMC=1
for name in "$@"; do
  if [ $name = main ]; then unset RODIO_HIP_LIB; else export RODIO_HIP_LIB=$PWD/variants/librodio_hip_$name.so; fi
  for cfg in limit biquad; do
    for shape in "64 1048576" "2048 32768"; do
      set -- $shape
      out=$(python bench.py --config $cfg --sources $1 --frames $2 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('call_ms', round(d['ms_per_step'],4), 'ev_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))")
      echo "$name $cfg streams=$1 frames=$2 : $out"
    done
  done
done
