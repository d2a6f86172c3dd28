#!/bin/bash
# usage: tools/pmc_run.sh <tag> [bench.py args...]   (run on the GPU box; writes gpurun_out/prof/<tag>/)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $out/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $out/pmc1 -o pmc1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $out/pmc2 -o pmc2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/pmc2.log 2>&1
python tools/prof_summary.py $out/kt/kt_results.db $out/pmc1/pmc1_results.db $out/pmc2/pmc2_results.db | cut -c1-200 > $out/summary.txt
grep -E "k_rlm|SQ_|grid=" $out/summary.txt
