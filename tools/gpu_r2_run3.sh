#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O; rm -f $O/*.log
timeout 600 python -m pytest tests/test_gpu_limit.py -q -m gpu -x -s 2>&1 | tail -8 > $O/limit_tests.log
for cfg in "8 8 2" "8 16 1" "16 8 1" "16 4 1" "16 4 2"; do set -- $cfg
  echo "R=$1 NW=$2 wgs/cu=$3" >> $O/limit_bench.log
  RH_LIMIT_R=$1 RH_LIMIT_NW=$2 RH_LIMIT_WGS=$3 timeout 120 python tools/bench_effects.py 64 1048576 limit 2>&1 | grep streams >> $O/limit_bench.log
  RH_LIMIT_R=$1 RH_LIMIT_NW=$2 RH_LIMIT_WGS=$3 timeout 120 python tools/bench_effects.py 2048 32768 limit 2>&1 | grep streams >> $O/limit_bench.log
done
RH_LIMIT_R=16 RH_LIMIT_NW=4 timeout 600 python -m pytest tests/test_gpu_limit.py -q -m gpu -x 2>&1 | tail -3 >> $O/limit_tests.log
for f in limit_tests limit_bench; do echo "== $f"; tail -n 40 $O/$f.log; done
