#!/bin/bash
# usage: tools/sweep_libs.sh "lib1 lib2 ..." "R NS" ...   -- kernel_ms of each library variant per geometry (GPU box)
libs=$1; shift
for a in "$@"; do set -- $a; for l in $libs; do
RODIO_HIP_LIB=$PWD/variants/librodio_hip_$l.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --frames-per-lane $1 --ring-stages $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); g=j['config']['geometry']
print('$l R',g['frames_per_lane'],'NS',g['ring_stages'],'ms',round(j['roofline']['kernel_ms'],3),'frac',round(j['roofline']['frac'],3),'late',round(g['late_carries_per_launch']))"
done; done
