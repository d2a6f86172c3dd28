cd $GRAFT_REPO_ROOT
export RH_BENCH_NO_PMC=1
for nio in 1 0; do
  for shape in "64 1048576" "2048 32768" "256 262144"; do
    set -- $shape
    out=$(RH_LIMIT_NIO=$nio python bench.py --config limit --sources $1 --frames $2 --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('call_ms', round(d['ms_per_step'],4), 'ev_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4), 'parity', d.get('parity',{}).get('max_abs_err'), d.get('parity',{}).get('ok'))")
    echo "RH_LIMIT_NIO=$nio limit streams=$1 frames=$2 : $out"
  done
done
