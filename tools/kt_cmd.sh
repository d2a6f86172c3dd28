#!/bin/bash
# usage: kt.sh <tag> <command...>: kernel trace only
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- "$@" > $out/kt.log 2>&1
python tools/prof_summary.py $out/kt/kt_results.db | cut -c1-200 > $out/summary.txt
cat $out/summary.txt
