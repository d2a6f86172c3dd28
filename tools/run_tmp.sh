export RH_BENCH_NO_PMC=1
for u in 24 26 28 34 24 26 28 34; do
  echo "RH_RS_PIPE=$u: $(RH_RS_PIPE=$u python bench.py --config 3 --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))")"
done
