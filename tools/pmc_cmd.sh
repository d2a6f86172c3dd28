#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <command...>   (GPU box) kernel trace + two SQ counter passes of an arbitrary command -> gpurun_out/prof/<tag>/summary.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- "$@" > $out/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $out/pmc1 -o pmc1 -- "$@" > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $out/pmc2 -o pmc2 -- "$@" > $out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch -- "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o write -- "$@" > $out/write.log 2>&1
python tools/prof_summary.py $out/kt/kt_results.db $out/pmc1/pmc1_results.db $out/pmc2/pmc2_results.db $out/fetch/fetch_results.db $out/write/write_results.db | cut -c1-220 > $out/summary.txt
cat $out/summary.txt
