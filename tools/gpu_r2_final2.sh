#!/bin/bash
# closing check of the final tree: the GPU suite; GpuMixer jobs with non-stereo sources, repeated, under a saturating background load
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -n 3 > gpurun_out/r02_final_gputests.txt
cat gpurun_out/r02_final_gputests.txt
RH_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 3000000 --no-cpu-baseline --no-autotune > /dev/null 2>&1 &
BG=$!
sleep 20
{
echo "== background bench alive: $(kill -0 $BG 2>/dev/null && echo yes || echo no)"
for m in any adapters; do echo "== stress_mixany $m 40"; python tools/stress_mixany.py $m 40 2>&1 | tail -n 3; done
echo "== background bench still alive: $(kill -0 $BG 2>/dev/null && echo yes || echo no)"
} > gpurun_out/r02_final_mixany_shared.txt 2>&1
kill $BG 2>/dev/null; wait $BG 2>/dev/null
cat gpurun_out/r02_final_mixany_shared.txt
