cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -n 6
RH_BENCH_NO_PMC=1 python bench.py --config limit --steps 30 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('limit', d['roofline']['kernel_ms'], round(d['roofline']['frac'],4))"
