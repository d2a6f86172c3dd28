#!/bin/bash
# The limiter chain and a filter chain, 100 runs each, while a second process keeps the GPU saturated (the condition that exposed
# round 1's ordering bugs): every run must give the same bits.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
RH_BENCH_NO_PMC=1 timeout 600 python bench.py --steps 2000000 --no-cpu-baseline --no-autotune > /dev/null 2>&1 &
BG=$!
sleep 20
{
echo "== background: bench.py --steps 2000000 (pid $BG alive: $(kill -0 $BG 2>/dev/null && echo yes || echo no))"
for cfg in "777 limit" "265 limit" "777 low_pass:300 limit"; do
  echo "== block $cfg"
  python tools/stress_chain.py 100 $cfg 2>&1 | tail -n 3
done
echo "== background still alive: $(kill -0 $BG 2>/dev/null && echo yes || echo no)"
} > gpurun_out/r02_stress_shared_gpu.txt 2>&1
kill $BG 2>/dev/null; wait $BG 2>/dev/null
cat gpurun_out/r02_stress_shared_gpu.txt
