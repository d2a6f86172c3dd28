cd $GRAFT_REPO_ROOT
export RH_BENCH_NO_PMC=1
for name in "$@"; do
  export RODIO_HIP_LIB=$PWD/variants/librodio_hip_$name.so
  for shape in "64 1048576" "2048 32768"; do
    set -- $shape
    out=$(python bench.py --config limit --sources $1 --frames $2 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('call_ms', round(d['ms_per_step'],4), 'ev_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))")
    echo "$name limit streams=$1 frames=$2 $EXTRA: $out"
  done
done
