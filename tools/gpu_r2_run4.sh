#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O; rm -f $O/*.log
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/tests.log
timeout 200 python tools/bench_effects.py 64 1048576 biquad_mode1 limit 2>&1 | grep streams > $O/effects.log
RH_NO_TICKET_SHARDS=1 timeout 200 python tools/bench_effects.py 64 1048576 biquad_mode1 2>&1 | grep streams >> $O/effects.log
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
export RH_PROF_KERNEL=k_rlm
bash tools/pmc_cmd.sh biquad1 python tools/bench_effects.py 64 1048576 biquad_mode1 > /dev/null 2>&1
head -8 gpurun_out/prof/biquad1/summary.txt | cut -c1-160 > $O/biquad1_kt.txt
for f in tests.log effects.log bench.json biquad1_kt.txt; do echo "== $f"; tail -n 16 $O/$f | cut -c1-400; done
