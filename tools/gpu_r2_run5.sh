#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O; rm -f $O/*.log
timeout 600 python -m pytest tests/test_gpu_effects.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -12 > $O/tests.log
timeout 300 python tools/bench_effects.py 64 1048576 agc biquad_mode0 2>&1 | grep streams > $O/effects.log
timeout 300 python tools/bench_effects.py 2048 32768 agc biquad_mode0 limit biquad_mode1 2>&1 | grep streams >> $O/effects.log
for f in tests.log effects.log; do echo "== $f"; tail -n 16 $O/$f | cut -c1-400; done
