#!/bin/bash
# usage: tools/phases.sh "R NS [extra bench args]" ...   (GPU box; needs variants/librodio_hip_prof.so)
for a in "$@"; do set -- $a; R=$1; NS=$2; shift 2
RODIO_HIP_LIB=$PWD/variants/librodio_hip_prof.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames-per-lane $R --ring-stages $NS "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); g=j['config']['geometry']
ph=g.get('phase_cycles'); tot=sum(ph[:6]) if ph else 1
print('R',g['frames_per_lane'],'NS',g['ring_stages'],'$*','ms',round(j['roofline']['kernel_ms'],3),'lds',g['lds_bytes'],'phases% gran/stage/wait/carry/run/scan',[round(100*x/tot,1) for x in ph[:6]],'ticks',round(tot),'late',round(g['late_carries_per_launch']),'empty_polls',round(g.get('empty_polls_per_launch',0)))"
done
