#!/bin/bash
# rocprofv3 evidence for the two scan kernels after the last changes (limiter: packed arithmetic + one poll point; rh_biquad mode 1: its own kernel)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
RH_PROF_KERNEL=k_limit_scan bash tools/pmc_cmd.sh limit python bench.py --config limit --steps 10 --no-cpu-baseline > /dev/null 2>&1
RH_PROF_KERNEL=k_limit_scan bash tools/pmc_cmd.sh limit2048 python bench.py --config limit --sources 2048 --frames 32768 --steps 10 --no-cpu-baseline > /dev/null 2>&1
RH_PROF_KERNEL=k_biquad_scan bash tools/pmc_cmd.sh biquad python bench.py --config biquad --steps 10 --no-cpu-baseline > /dev/null 2>&1
for c in limit limit2048 biquad; do echo "=== $c"; grep -v "at::native\|rocclr" gpurun_out/prof/$c/summary.txt | cut -c1-200 | head -34; done
RH_BENCH_NO_PMC=1 python bench.py --config limit --steps 30 2>/dev/null | tail -n 1 > gpurun_out/r02_bench_limit.json
RH_BENCH_NO_PMC=1 python bench.py --config limit --sources 2048 --frames 32768 --steps 30 --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/r02_bench_limit_2048.json
RH_BENCH_NO_PMC=1 python bench.py --config biquad --steps 30 2>/dev/null | tail -n 1 > gpurun_out/r02_bench_biquad.json
RH_BENCH_NO_PMC=1 python bench.py --config biquad --sources 2048 --frames 32768 --steps 30 --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/r02_bench_biquad_2048.json
python - <<'PY'
import json
for f in ("limit", "limit_2048", "biquad", "biquad_2048"):
    d = json.load(open(f"gpurun_out/r02_bench_{f}.json"))
    print(f, [(k["kernel"], round(k["kernel_ms"], 4), round(k["frac"], 4)) for k in d["config"]["kernels"]], d.get("cpu_baseline", {}).get("value"))
PY
