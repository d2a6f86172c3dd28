#!/bin/bash
# HBM traffic of the fused kernel from the TCC counters (GPU box):  tools/pmc_traffic.sh <tag> [bench args]
# FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2), and on gfx950 FETCH_SIZE
# under-reports wide coalesced reads (MI355X_MICROARCH.md, HBM): a third pass measures the factor on
# tools/ubench/stream_ring, which moves a known byte count with the same LDS-DMA access pattern.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/cal -o cal -- tools/ubench/stream_ring mimic > $out/cal.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d $out/raw -o raw -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/raw.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d $out/rawcal -o rawcal -- tools/ubench/stream_ring mimic > $out/rawcal.log 2>&1
python tools/prof_summary.py $out/fetch/fetch_results.db $out/raw/raw_results.db | grep -E "TCC|FETCH"; python - <<EOF2
import sqlite3
con=sqlite3.connect('$out/rawcal/rawcal_results.db')
for r in con.execute('select counter_name, avg(value), count(*) from counters_collection group by counter_name'): print('cal', r)
EOF2
python tools/traffic_json.py $out "$@" | tee $out/traffic.json
