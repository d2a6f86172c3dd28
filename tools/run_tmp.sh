export RH_BENCH_NO_PMC=1
python -m pytest tests/test_gpu_multi.py tests/test_gpu_mix_first.py tests/test_gpu_effects.py tests/test_host_mirror.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do
python bench.py --shared-device --no-cpu-baseline --no-per-source 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tickets', round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), d['config']['geometry']['tiles_by'])"
python bench.py --no-cpu-baseline --no-per-source 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('index', round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), d['config']['geometry']['tiles_by'])"
done
