export RH_BENCH_NO_PMC=1
for u in 8 4 12 14 18 8 14 18; do
  echo "RH_RS_PIPE=$u: $(RH_RS_PIPE=$u python bench.py --config 3 --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))")"
done
RH_RS_PIPE=14 timeout 300 python -m pytest tests -m gpu -x -q -k "reverb" 2>&1 | grep -E "passed|failed" | tail -2
python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6
