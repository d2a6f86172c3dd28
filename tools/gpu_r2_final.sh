#!/bin/bash
# Round-2 closing evidence on one MI355X: the GPU suite, smoke(), the driver's bench line, the scan kernels' profiles, the chain stress
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -n 3 > gpurun_out/r02_final_gputests.txt
cat gpurun_out/r02_final_gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
python bench.py 2>/dev/null | tail -n 1 > gpurun_out/r02_bench_cfg2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_cfg2.json"))
print("cfg2", d["value"], d["ms_per_step"], d["roofline"], d["cpu_baseline"]["value"], d.get("parity"))
PY
bash tools/gpu_r2_prof2.sh 2>&1 | grep -v "^#\|^SQ_\|^##" | tail -n 16
python tools/stress_chain.py 100 777 limit 2>&1 | tail -n 2 > gpurun_out/r02_final_stress.txt; cat gpurun_out/r02_final_stress.txt
