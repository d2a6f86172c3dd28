#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_host_mirror.py -q -m gpu -x 2>&1 | tail -30 > $O/tests.log
(timeout 100 python bench.py --steps 400000 --no-autotune --no-cpu-baseline > $O/bg.log 2>&1 &)
sleep 20
timeout 70 python tools/stress_late.py 0 5 > $O/late_a.log 2>&1 &
timeout 70 python tools/stress_late.py -1 5 > $O/late_b.log 2>&1 &
timeout 70 python tools/stress_mixany.py any 16 > $O/mixany.log 2>&1
wait
for f in tests late_a late_b mixany; do echo "== $f"; tail -n 14 $O/$f.log | cut -c1-300; done
