#!/bin/bash
# rocprofv3 evidence for the side kernels (kernel trace + SQ counters + FETCH/WRITE_SIZE), one directory per config
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
RH_PROF_KERNEL=k_reverb_spatial bash tools/pmc_cmd.sh cfg3 python bench.py --config 3 --steps 10 > /dev/null 2>&1
RH_PROF_KERNEL=k_ bash tools/pmc_cmd.sh cfg5 python bench.py --config 5 --steps 10 > /dev/null 2>&1
RH_PROF_KERNEL=k_limit_scan bash tools/pmc_cmd.sh limit python bench.py --config limit --steps 10 > /dev/null 2>&1
RH_PROF_KERNEL=k_rlm bash tools/pmc_cmd.sh ragged python bench.py --config ragged --steps 10 --no-cpu-baseline > /dev/null 2>&1
RH_PROF_KERNEL=k_rlm bash tools/pmc_cmd.sh biquad python bench.py --config biquad --steps 10 > /dev/null 2>&1
RH_PROF_KERNEL=k_agc bash tools/pmc_cmd.sh agc python bench.py --config agc --sources 2048 --frames 32768 --steps 20 > /dev/null 2>&1
RH_PROF_KERNEL=k_rlm RH_BENCH_NO_PMC=1 bash tools/pmc_cmd.sh cfg2 python bench.py --steps 10 --no-cpu-baseline > /dev/null 2>&1
for c in cfg3 cfg5 limit ragged biquad agc cfg2; do echo "=== $c"; grep -v "^ *[0-9]* .*at::native\|rocclr" gpurun_out/prof/$c/summary.txt | cut -c1-200 | head -40; done
