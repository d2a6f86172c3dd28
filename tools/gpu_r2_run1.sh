#!/bin/bash
# GPU run 1 of round 2: (A) the null-stream race reproducer against the round-1 library and the fixed one,
# (B) full-size parity tests, (C) baseline timings of the sequential recurrences.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
echo "== race reproducer, round-1 library" > $O/race.log
RODIO_HIP_LIB=$PWD/variants/r1/librodio_hip.so timeout 400 python -m pytest tests/test_gpu_race.py -q -m gpu -x 2>&1 | tail -15 >> $O/race.log
echo "== race reproducer, fixed library" >> $O/race.log
timeout 400 python -m pytest tests/test_gpu_race.py -q -m gpu 2>&1 | tail -15 >> $O/race.log
echo "== C++ GpuMixer stress under load, fixed library" >> $O/race.log
(timeout 150 python bench.py --steps 300000 --no-autotune --no-cpu-baseline > $O/bg.log 2>&1 &)
sleep 25
timeout 120 python tools/stress_late.py 0 6 > $O/late_a.log 2>&1 &
timeout 120 python tools/stress_late.py 0 6 > $O/late_b.log 2>&1 &
timeout 120 python tools/stress_mixany.py any 24 > $O/mixany.log 2>&1
wait
tail -3 $O/late_a.log $O/late_b.log $O/mixany.log >> $O/race.log
echo "== full-size parity" > $O/full.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s 2>&1 | tail -25 >> $O/full.log
echo "== effects baseline" > $O/effects.log
timeout 300 python tools/bench_effects.py 64 1048576 >> $O/effects.log 2>&1
timeout 200 python tools/bench_effects.py 2048 32768 >> $O/effects.log 2>&1
cat $O/race.log $O/full.log $O/effects.log
