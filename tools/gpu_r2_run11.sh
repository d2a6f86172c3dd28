cd $GRAFT_REPO_ROOT
for e in "" "RH_CVT_ROWS=1"; do
echo "== env '$e'"
env $e RH_BENCH_NO_PMC=1 python bench.py --config 5 --steps 30 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in d['config']['kernels']: print(k['kernel'], round(k['kernel_ms'],4), round(k['frac'],4))"
done
RH_CVT_ROWS=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "convert or i16 or sample_type or formats" 2>&1 | tail -n 3
