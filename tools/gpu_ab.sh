#!/bin/bash
# usage: tools/gpu_ab.sh "<ENV=..>" ...   one bench line of the limiter and of the biquad scan per environment (A/B tests of launch knobs)
cd $GRAFT_REPO_ROOT
for e in "$@"; do
for cfg in "limit 64 1048576" "limit 2048 32768" "biquad 64 1048576" "biquad 2048 32768"; do set -- $cfg
env $e RH_BENCH_NO_PMC=1 python bench.py --config $1 --sources $2 --frames $3 --steps 30 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['config']['kernels'][0]; print('$e', '$1', '$2 x $3', round(k['kernel_ms'],4), round(k['frac'],4))"
done; done
