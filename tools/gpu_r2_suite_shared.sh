#!/bin/bash
# The whole GPU suite while a second process saturates the device with the headline benchmark: ordering bugs that hide on an idle
# GPU show under exactly this condition.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
RH_BENCH_NO_PMC=1 timeout 900 python bench.py --steps 3000000 --no-cpu-baseline --no-autotune > /dev/null 2>&1 &
BG=$!
sleep 25
{
echo "== background: bench.py --steps 3000000 alive: $(kill -0 $BG 2>/dev/null && echo yes || echo no)"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -n 15
echo "== background still alive: $(kill -0 $BG 2>/dev/null && echo yes || echo no)"
} > gpurun_out/r02_suite_shared_gpu.txt 2>&1
kill $BG 2>/dev/null; wait $BG 2>/dev/null
cat gpurun_out/r02_suite_shared_gpu.txt
