export RH_BENCH_NO_PMC=1
timeout 600 python -m pytest tests/test_gpu_effects.py tests/test_gpu_parity.py tests/test_host_mirror.py -m gpu -x -q -k "agc or chain" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
for a in "" 2048; do
  if [ -z "$a" ]; then extra=""; else extra="--sources 2048 --frames 32768"; fi
  timeout 300 python bench.py --config agc $extra --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('fused $a', round(d['roofline']['kernel_ms'],4), d.get('parity'))"
done
timeout 300 python bench.py --config agc --sources 256 --frames 262144 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('fused 256x256Ki', round(d['roofline']['kernel_ms'],4))"
RH_AGC_SEGMENTS=1 timeout 300 python bench.py --config agc --sources 256 --frames 262144 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('segments 256x256Ki', round(d['roofline']['kernel_ms'],4))"
