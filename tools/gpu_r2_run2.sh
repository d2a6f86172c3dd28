#!/bin/bash
# GPU run 2: (A) time-parallel limiter: parity + timing, (B) C++ GpuMixer stress under load (new rh_mix_sum), plain and poisoned
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_limit.py -q -m gpu -x -s 2>&1 | tail -25 > $O/limit_tests.log
for R in 8 16; do for W in 8 16; do
  echo "R=$R waves=$W" >> $O/limit_bench.log
  RH_LIMIT_R=$R RH_LIMIT_WAVES=$W timeout 120 python tools/bench_effects.py 64 1048576 limit >> $O/limit_bench.log 2>&1
done; done
RH_LIMIT_R=16 timeout 120 python tools/bench_effects.py 2048 32768 limit >> $O/limit_bench.log 2>&1
(timeout 170 python bench.py --steps 400000 --no-autotune --no-cpu-baseline > $O/bg.log 2>&1 &)
sleep 25
timeout 70 python tools/stress_late.py 0 6 > $O/late_a.log 2>&1 &
timeout 70 python tools/stress_late.py 0 6 > $O/late_b.log 2>&1 &
timeout 70 python tools/stress_mixany.py any 20 > $O/mixany.log 2>&1
wait
export RODIO_HIP_DEBUG_POISON=1
timeout 70 python tools/stress_late.py 0 6 > $O/late_pa.log 2>&1 &
timeout 70 python tools/stress_late.py 0 6 > $O/late_pb.log 2>&1 &
timeout 70 python tools/stress_mixany.py any 20 > $O/mixany_p.log 2>&1
wait
for f in limit_tests limit_bench late_a late_b mixany late_pa late_pb mixany_p; do echo "== $f"; tail -n 12 $O/$f.log; done
