export RH_BENCH_NO_PMC=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_mix_first.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
run() { python bench.py --config ragged --steps 20 --warmup 3 --no-autotune --no-cpu-baseline --frames-per-lane $1 --ring-stages 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('R $1 resident ${RH_RAG_RESIDENT:-all} two=${RH_RAG_TWO_KERNELS:-0}', round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), d['config']['geometry']['frames_per_lane'], d.get('parity',{}).get('max_abs_err'))"; }
for w in 2 3 ""; do export RH_RAG_RESIDENT=$w; [ -z "$w" ] && unset RH_RAG_RESIDENT; run 18; done
RH_RAG_TWO_KERNELS=1 RH_RAG_RESIDENT=2 run 18
for w in 3 4 5 ""; do export RH_RAG_RESIDENT=$w; [ -z "$w" ] && unset RH_RAG_RESIDENT; run 12; run 14; done
for w in 3 4 6; do export RH_RAG_RESIDENT=$w; run 9; run 10; done
unset RH_RAG_RESIDENT
python bench.py --config ragged --steps 20 --warmup 3 --no-autotune 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('default', round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), d['parity'])"
