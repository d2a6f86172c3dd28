cd $GRAFT_REPO_ROOT
export RH_BENCH_NO_PMC=1
for init in 0 1; do
  for cfg in limit biquad; do
    for shape in "64 1048576" "2048 32768"; do
      set -- $shape
      out=$(RH_LIMIT_INIT=$init python bench.py --config $cfg --sources $1 --frames $2 --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('call_ms', round(d['ms_per_step'],4), 'ev_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4), 'parity', d.get('parity',{}).get('max_abs_err'), d.get('parity',{}).get('ok'))")
      echo "RH_LIMIT_INIT=$init $cfg streams=$1 frames=$2 : $out"
    done
  done
done
for c in 3 5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print([(k['kernel'], round(k['kernel_ms'],4), round(k['frac'],3)) for k in d['config']['kernels']])"; done
