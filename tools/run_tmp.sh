export RH_BENCH_NO_PMC=1
python bench.py --config biquad --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print([(k['kernel'], round(k['kernel_ms'],4), round(k['ms_per_step'],4)) for k in d['config']['kernels']])"
python bench.py --config biquad --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print([(k['kernel'], round(k['kernel_ms'],4), round(k['ms_per_step'],4)) for k in d['config']['kernels']])"
RH_PROF_KERNEL=k_b tools/kt_cmd.sh bq30 python bench.py --config biquad --steps 30 --no-cpu-baseline 2>&1 | cut -c1-200 | head -20
