#!/bin/bash
# Round 5 GPU calls, in parts (one gpurun call each): bash tools/r05_call.sh <part>.  Everything lands in gpurun_out/r05/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
E=gpurun_out/r05
gaps() {  # gaps <tag> <kernel> <bench args...>: timeline of back-to-back launches
    tag=$1; kern=$2; shift 2
    rm -rf $E/gaps_$tag; mkdir -p $E/gaps_$tag
    (cd /tmp && RH_BENCH_NO_PMC=1 rocprofv3 --kernel-trace --hip-trace -d $GRAFT_REPO_ROOT/$E/gaps_$tag -o t -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $GRAFT_REPO_ROOT/$E/gaps_$tag.log 2>&1)
    db=$(find $E/gaps_$tag -name 't_results.db' | head -1)
    RH_PROF_KERNEL=$kern python tools/launch_gaps.py "$db" > $E/gaps_$tag.txt 2>&1
    rm -rf $E/gaps_$tag
    tail -30 $E/gaps_$tag.txt
}
case "$1" in
1)
    python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 > $E/gputests_1.txt; tail -5 $E/gputests_1.txt
    python bench.py > $E/bench_cfg2.json 2> $E/bench_cfg2.err; tail -c 1500 $E/bench_cfg2.json; tail -3 $E/bench_cfg2.err
    python bench.py --collective native --no-per-source --no-unscaled > $E/bench_cfg2_native1.json 2> $E/bench_cfg2_native1.err; tail -c 600 $E/bench_cfg2_native1.json; tail -3 $E/bench_cfg2_native1.err
    gaps limit k_limit_scan --config limit --steps 13 --warmup 2
    gaps biquad k_biquad_scan --config biquad --steps 13 --warmup 2
    ;;
2)
    python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $E/gputests_2.txt; tail -12 $E/gputests_2.txt
    for c in limit biquad; do RH_BENCH_NO_PMC=1 python bench.py --config $c --no-cpu-baseline > $E/bench_${c}_a.json 2>/dev/null; python bench.py --config $c > $E/bench_$c.json 2>/dev/null; python bench.py --config $c --sources 2048 --frames 32768 > $E/bench_${c}_2048.json 2>/dev/null; done
    python bench.py --config stream > $E/bench_stream.json 2> $E/bench_stream.err; tail -3 $E/bench_stream.err
    RH_BENCH_NO_PMC=1 python bench.py --config stream --block 16384 --no-cpu-baseline > $E/bench_stream_16k.json 2>/dev/null
    python bench.py > $E/bench_cfg2.json 2> $E/bench_cfg2.err
    for f in bench_limit_a bench_limit bench_limit_2048 bench_biquad_a bench_biquad bench_biquad_2048 bench_stream bench_stream_16k bench_cfg2; do python - "$E/$f.json" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
    r=d['roofline']; print(sys.argv[1].split('/')[-1], 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'parity', (d.get('parity') or {}).get('ok'), (d.get('parity') or {}).get('max_abs_err'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
    done
    ;;
3)
    python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $E/gputests_3.txt; tail -8 $E/gputests_3.txt
    python bench.py --config stream > $E/bench_stream.json 2> $E/bench_stream.err; tail -3 $E/bench_stream.err
    RH_BENCH_NO_PMC=1 python bench.py --config stream --block 16384 --no-cpu-baseline > $E/bench_stream_16k.json 2>/dev/null
    RH_BENCH_NO_PMC=1 python bench.py --config stream --block 262144 --no-cpu-baseline > $E/bench_stream_256k.json 2>/dev/null
    export RH_PROF_KERNEL=k_limit_scan
    bash tools/pmc_cmd.sh r05_limit python bench.py --config limit --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r05_limit/summary.txt $E/r05_limit_kernel_trace_pmc.txt
    export RH_PROF_KERNEL=k_biquad_scan
    bash tools/pmc_cmd.sh r05_biquad python bench.py --config biquad --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r05_biquad/summary.txt $E/r05_biquad_kernel_trace_pmc.txt
    for f in bench_stream bench_stream_16k bench_stream_256k; do python - "$E/$f.json" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
    r=d['roofline']; print(sys.argv[1].split('/')[-1], 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'parity', (d.get('parity') or {}))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
    done
    grep -A12 "counters" $E/r05_limit_kernel_trace_pmc.txt | head -40
    ;;
4)
    python -m pytest tests/test_gpu_mix_first.py tests/test_gpu_multi.py tests/test_gpu_mono.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $E/gputests_4.txt; tail -8 $E/gputests_4.txt
    python bench.py --config stream > $E/bench_stream.json 2> $E/bench_stream.err; tail -3 $E/bench_stream.err
    RH_BENCH_NO_PMC=1 python bench.py --config stream --block 16384 --no-cpu-baseline > $E/bench_stream_16k.json 2>/dev/null
    RH_BENCH_NO_PMC=1 python bench.py --config stream --block 262144 --no-cpu-baseline > $E/bench_stream_256k.json 2>/dev/null
    RH_PROF_KERNEL=k_ bash tools/kt_cmd.sh r05_stream python bench.py --config stream --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r05_stream/summary.txt $E/r05_stream_kernel_trace.txt; head -12 $E/r05_stream_kernel_trace.txt | cut -c1-180
    for f in bench_stream bench_stream_16k bench_stream_256k; do python - "$E/$f.json" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
    r=d['roofline']; print(sys.argv[1].split('/')[-1], 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'parity', (d.get('parity') or {}).get('max_abs_err'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
    done
    ;;
5)
    q() { python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
    {
    for g in 1 2 4 8; do echo "stream 64Ki RH_MIX_GROUPS=$g: $(RH_MIX_GROUPS=$g RH_BENCH_NO_PMC=1 python bench.py --config stream --steps 10 --no-cpu-baseline 2>/dev/null | q)"; done
    for g in 2 4 8; do echo "stream 16Ki RH_MIX_GROUPS=$g: $(RH_MIX_GROUPS=$g RH_BENCH_NO_PMC=1 python bench.py --config stream --block 16384 --steps 5 --no-cpu-baseline 2>/dev/null | q)"; done
    } > $E/r05_stream_groups.txt 2>&1; cat $E/r05_stream_groups.txt
    {
    for shape in "64 1048576" "2048 32768"; do set -- $shape
      for v in "16 8 0" "8 8 0" "8 16 0" "16 4 0" "8 4 0" "8 8 1" "8 4 2" "8 4 4"; do set -- $shape $v
        env="RH_LIMIT_R=$3 RH_LIMIT_NW=$4"; [ "$5" != "0" ] && env="$env RH_LIMIT_WGS=$5"
        echo "limit streams=$1 frames=$2 R=$3 NW=$4 WGS=$5: $(env $env RH_BENCH_NO_PMC=1 python bench.py --config limit --sources $1 --frames $2 --steps 30 --no-cpu-baseline 2>/dev/null | q)"
      done
    done
    } > $E/r05_limit_geometry.txt 2>&1; cat $E/r05_limit_geometry.txt
    ;;
6)
    { echo "## RH_LIMIT_PROFILE build (shader cycles per phase, summed over the tiles), k_limit_scan<2,16,8>"; for shape in "64 1048576" "2048 32768"; do echo "# streams frames = $shape"; RODIO_HIP_LIB=rodio_amd/build/librodio_hip_lprof.so python tools/prof_limit.py $shape 2>/dev/null | tail -1; done; } > $E/r05_limit_phases.txt 2>&1; cat $E/r05_limit_phases.txt
    ;;
7)
    q() { python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), (d.get('parity') or {}).get('max_abs_err'))"; }
    { for rep in 1 2; do for shape in "64 1048576" "2048 32768"; do set -- $shape; for w in 1 2 0; do echo "limit streams=$1 frames=$2 RH_SCAN_DMA_TOP=$w: $(RH_SCAN_DMA_TOP=$w RH_BENCH_NO_PMC=1 python bench.py --config limit --sources $1 --frames $2 --steps 30 2>/dev/null | q)"; done; done; done; } > $E/r05_limit_dma_where.txt 2>&1; cat $E/r05_limit_dma_where.txt
    ;;
8)  # the round's evidence: the suite, the driver's line, its trace + counters, the side configurations, the pull path
    python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $E/r05_final_gputests.txt; tail -4 $E/r05_final_gputests.txt
    python bench.py > $E/r05_bench_cfg2.json 2> $E/r05_bench_cfg2.err; tail -c 400 $E/r05_bench_cfg2.json
    RH_PROF_KERNEL=k_rlm bash tools/pmc_cmd.sh r05_cfg2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-autotune --no-per-source > /dev/null 2>&1
    cp gpurun_out/prof/r05_cfg2/summary.txt $E/r05_cfg2_kernel_trace_pmc.txt
    for c in 3 5 ragged agc 2span 2mono; do python bench.py --config $c > $E/r05_bench_$c.json 2>/dev/null; done
    python bench.py --collective native --no-per-source > $E/r05_bench_cfg2_native_one_rank.json 2>/dev/null
    RH_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 > $E/r05_bench_gpus2_one_device.json 2> $E/r05_bench_gpus2_one_device.err
    {
        echo "## tests/cpp/host_mirror_test bench <sources> <frames> <block_frames> <host_threads>"
        tests/cpp/host_mirror_test bench 256 4194304 65536 16
        tests/cpp/host_mirror_test bench 256 4194304 65536 16
        echo "## RH_TEST_SOURCE=buffer (SamplesBuffer: spans of 32768 samples, converted span by span)"
        RH_TEST_SOURCE=buffer tests/cpp/host_mirror_test bench 256 4194304 32768 16
    } > $E/r05_pull_path.txt 2>&1
    for f in r05_bench_cfg2 r05_bench_3 r05_bench_5 r05_bench_ragged r05_bench_agc r05_bench_2span r05_bench_2mono r05_bench_cfg2_native_one_rank r05_bench_gpus2_one_device; do python - "$E/$f.json" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
    r=d['roofline']; print(sys.argv[1].split('/')[-1], 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'parity', (d.get('parity') or {}).get('ok'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
    done
    grep -o '"frac": [0-9.]*' $E/r05_pull_path.txt | head
    ;;
9)
    python -m pytest tests/test_gpu_mix_first.py tests/test_host_mirror.py -q -m gpu -p no:cacheprovider -k "filter or class" 2>&1 | tail -4
    q() { python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; c=d['roofline']['per_class']; print('headline', round(d['roofline']['kernel_ms'],4), 'per_class', round(c['call_ms'],4), round(c['frac'],4), c.get('parity',{}).get('max_abs_err'))"; }
    for rep in 1 2; do echo "side by side: $(python bench.py --no-per-source --no-unscaled 2>/dev/null | q)"; echo "RH_CLASSES_SERIAL=1: $(RH_CLASSES_SERIAL=1 python bench.py --no-per-source --no-unscaled 2>/dev/null | q)"; done | tee $E/r05_per_class.txt
    RH_PROF_KERNEL=k_rlm bash tools/pmc_cmd.sh r05_cfg2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-autotune --no-per-source --no-per-class > /dev/null 2>&1
    cp gpurun_out/prof/r05_cfg2/summary.txt $E/r05_cfg2_kernel_trace_pmc.txt; head -8 $E/r05_cfg2_kernel_trace_pmc.txt | cut -c1-150
    ;;
10)  # the AGC's chain waves with nothing but their chains (k_agc_fused0) against round 4's kernel
    python -m pytest tests/test_gpu_effects.py -q -m gpu -p no:cacheprovider -k agc 2>&1 | tail -4
    q() { python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), (d.get('parity') or {}).get('max_abs_err'))"; }
    { for rep in 1; do for shape in "64 1048576" "2048 32768"; do set -- $shape; for w in 0 1; do v=""; [ $w = 1 ] && v="RH_AGC_FUSED_R4=1"; echo "agc streams=$1 frames=$2 $v: $(env $v RH_BENCH_NO_PMC=1 python bench.py --config agc --sources $1 --frames $2 --steps 5 2>/dev/null | q)"; done; done; done; } > $E/r05_agc_fused0.txt 2>&1; cat $E/r05_agc_fused0.txt
    ;;
11)  # k_agc_fused0: which stage sets a chunk's time?  diagnostics builds (wrong results) beside the product
    python -m pytest tests/test_gpu_effects.py -q -m gpu -p no:cacheprovider -k agc 2>&1 | tail -3
    q() { python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), (d.get('parity') or {}).get('max_abs_err'))"; }
    { for shape in "64 1048576" "2048 32768"; do set -- $shape; for v in "" twodiv diagD diagG nosq; do l=""; [ -n "$v" ] && l="RODIO_HIP_LIB=variants/librodio_hip_agc_$v.so"; echo "agc streams=$1 frames=$2 $v: $(env $l RH_BENCH_NO_PMC=1 python bench.py --config agc --sources $1 --frames $2 --steps 5 2>/dev/null | q)"; done; done; } > $E/r05_agc_stages.txt 2>&1; cat $E/r05_agc_stages.txt
    ;;
12)  # k_agc_fused0 by role (RH_AGC_PROFILE builds), and how far repeated runs of one build lie apart
    q() { python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(round(d['ms_per_step'],4), (d.get('parity') or {}).get('max_abs_err'))"; }
    {
        for v in prof prof2 profD; do echo "## variants/librodio_hip_agc_$v.so, 64 streams x 1 Mi frames"; RODIO_HIP_LIB=variants/librodio_hip_agc_$v.so RH_BENCH_NO_PMC=1 python bench.py --config agc --sources 64 --frames 1048576 --steps 2 --warmup 1 2>&1 | grep "agc role" | tail -6; done
        for rep in 1 2 3; do for v in "" twodiv diagD; do l=""; [ -n "$v" ] && l="RODIO_HIP_LIB=variants/librodio_hip_agc_$v.so"; echo "agc 64 x 1 Mi $v: $(env $l RH_BENCH_NO_PMC=1 python bench.py --config agc --sources 64 --frames 1048576 --steps 5 2>/dev/null | q)"; done; done
    } > $E/r05_agc_roles.txt 2>&1; cat $E/r05_agc_roles.txt
    ;;
13)  # k_agc_fused0: wave priorities and which roles share the chains' SIMDs
    q() { python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(round(d['ms_per_step'],4), (d.get('parity') or {}).get('max_abs_err'))"; }
    { for shape in "64 1048576" "2048 32768"; do set -- $shape; for v in "" $VARIANTS; do l=""; [ -n "$v" ] && l="RODIO_HIP_LIB=variants/librodio_hip_agc_$v.so"; echo "agc streams=$1 frames=$2 $v: $(env $l RH_BENCH_NO_PMC=1 python bench.py --config agc --sources $1 --frames $2 --steps 5 2>/dev/null | q)"; done; done; } > $E/r05_agc_placement_$TAG.txt 2>&1; cat $E/r05_agc_placement_$TAG.txt
    ;;
14)  # the AGC's evidence: tests, both shapes through bench.py, kernel trace + counters, time per role
    python -m pytest tests/test_gpu_effects.py -q -m gpu -p no:cacheprovider -k agc 2>&1 | tail -2
    python bench.py --config agc > $E/r05_bench_agc.json 2>/dev/null; python bench.py --config agc --sources 2048 --frames 32768 > $E/r05_bench_agc_2048.json 2>/dev/null
    RH_AGC_FUSED_R4=1 RH_BENCH_NO_PMC=1 python bench.py --config agc > $E/r05_bench_agc_round4_kernel.json 2>/dev/null
    RH_PROF_KERNEL=k_agc bash tools/pmc_cmd.sh r05_agc python bench.py --config agc --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r05_agc/summary.txt $E/r05_agc_64x1Mi_kernel_trace_pmc.txt
    { echo "## RH_AGC_PROFILE build, 64 streams x 1 Mi frames, workgroup 0: s_memtime ticks between leaving a barrier and reaching the next, per wave, summed over the steps"; RODIO_HIP_LIB=variants/librodio_hip_agc_prof.so RH_BENCH_NO_PMC=1 python bench.py --config agc --steps 2 --warmup 1 2>&1 | grep "agc hw" | tail -12 | sort; } > $E/r05_agc_roles_final.txt
    for f in r05_bench_agc r05_bench_agc_2048 r05_bench_agc_round4_kernel; do python - "$E/$f.json" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
r=d['roofline']; print(sys.argv[1].split('/')[-1], 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'parity', (d.get('parity') or {}).get('max_abs_err'))
PY
    done
    cat $E/r05_agc_roles_final.txt; head -12 $E/r05_agc_64x1Mi_kernel_trace_pmc.txt | cut -c1-160
    ;;
esac
