#!/bin/bash
# mid-size batches: which tile geometry should the host pick?  (limiter and biquad mode 1; auto = the library's choice)
cd $GRAFT_REPO_ROOT
GEOMS=${GEOMS:-"auto|16 8|8 4|8 1"}
for shape in "64 16384" "64 65536" "256 8192" "8 1048576" "1024 4096" "4 131072" "1 32768" "300 2048"; do set -- $shape
IFS='|' read -ra GS <<< "$GEOMS"
for g in "${GS[@]}"; do
  if [ "$g" = "auto" ]; then e="X=1"; else set -- $shape $g; e="RH_LIMIT_R=$3 RH_LIMIT_NW=$4 RH_BIQUAD_R=$3 RH_BIQUAD_NW=$4"; set -- $shape; fi
  for cfg in limit biquad; do
  env $e RH_BENCH_NO_PMC=1 python bench.py --config $cfg --sources $1 --frames $2 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); k=d['config']['kernels'][0]; print('$cfg $1 x $2 geometry [$g]:', round(k['kernel_ms']*1000,1), 'us', round(k['frac'],3))
except Exception: print('$cfg $1 x $2 geometry [$g]: FAILED')"
  done
done; done
