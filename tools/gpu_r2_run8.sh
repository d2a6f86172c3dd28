#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_effects.py tests/test_gpu_parity.py -q -m gpu -x -k "biquad or blt or chain" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25 > $O/tests.log
for cfg in "8 8 2" "8 8 1" "16 8 1" "8 4 4" "8 4 2"; do set -- $cfg
  echo "R=$1 NW=$2 wgs=$3" >> $O/bench.log
  RH_BIQUAD_R=$1 RH_BIQUAD_NW=$2 RH_BIQUAD_WGS=$3 timeout 120 python bench.py --config biquad --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['kernels'][0]['kernel_ms'], d['config']['kernels'][0]['frac'])" >> $O/bench.log
done
RH_BIQUAD_R=8 timeout 120 python bench.py --config biquad --sources 2048 --frames 32768 --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=2048', d['config']['kernels'][0]['kernel_ms'], d['config']['kernels'][0]['frac'])" >> $O/bench.log
for f in tests bench; do echo "== $f"; tail -n 30 $O/$f.log | cut -c1-300; done
