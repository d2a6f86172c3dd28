#!/bin/bash
# Round 4 evidence, in parts (one gpurun call each): bash tools/evidence_r04.sh <part>.  Everything lands in gpurun_out/ev/ and is
# copied to profiles/ by hand.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ev
export TMPDIR=/tmp
E=gpurun_out/ev
case "$1" in
1)  # the headline: the driver's line, its kernel trace + counters, the N > 1 line (two ranks time-sharing the one device), spans, mono
    python bench.py > $E/r04_bench_cfg2.json 2> $E/r04_bench_cfg2.err
    bash tools/pmc_cmd.sh r04_cfg2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-autotune --no-per-source > /dev/null 2>&1
    cp gpurun_out/prof/r04_cfg2/summary.txt $E/r04_cfg2_kernel_trace_pmc.txt
    RH_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 > $E/r04_bench_gpus2_one_device.json 2> $E/r04_bench_gpus2_one_device.err
    python bench.py --shared-device --no-cpu-baseline --no-per-source > $E/r04_bench_cfg2_tiles_by_ticket.json 2>/dev/null
    python bench.py --config 2span > $E/r04_bench_2span.json 2>/dev/null
    python bench.py --config 2mono > $E/r04_bench_2mono.json 2>/dev/null
    ;;
2)  # the side configs, each with its kernel trace + counters
    for c in 3 5 ragged limit agc biquad; do python bench.py --config $c > $E/r04_bench_$c.json 2>/dev/null; done
    for c in limit agc biquad; do python bench.py --config $c --sources 2048 --frames 32768 > $E/r04_bench_${c}_2048.json 2>/dev/null; done
    for c in limit biquad 3 5 ragged; do
        case $c in limit) export RH_PROF_KERNEL=k_limit_scan;; biquad) export RH_PROF_KERNEL=k_biquad_scan;; 3) export RH_PROF_KERNEL=reverb_spatial;; 5) export RH_PROF_KERNEL=k_int_to_f32;; *) export RH_PROF_KERNEL=k_rlm;; esac
        bash tools/pmc_cmd.sh r04_$c python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
        cp gpurun_out/prof/r04_$c/summary.txt $E/r04_${c}_kernel_trace_pmc.txt
    done
    export RH_PROF_KERNEL=k_agc
    bash tools/pmc_cmd.sh r04_agc python bench.py --config agc --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r04_agc/summary.txt $E/r04_agc_64x1Mi_kernel_trace_pmc.txt
    ;;
3)  # the pull path, the microbenchmarks, the filter contract, the suite
    {
        echo "## tools/h2d_bw.py"; python tools/h2d_bw.py
        echo "## tests/cpp/host_mirror_test bench <sources> <frames> <block_frames> <host_threads>   (prepare() on the control thread, then the consumer's reads)"
        tests/cpp/host_mirror_test bench 256 1048576 16384 16
        tests/cpp/host_mirror_test bench 256 4194304 32768 16
        tests/cpp/host_mirror_test bench 256 4194304 65536 16
        tests/cpp/host_mirror_test bench 256 4194304 65536 16
        echo "## RH_BENCH_NO_PREPARE=1: the consumer's first read starts the stream (round 3's behaviour)"
        RH_BENCH_NO_PREPARE=1 tests/cpp/host_mirror_test bench 256 4194304 65536 16
        echo "## RH_TEST_SOURCE=buffer (SamplesBuffer: spans of 32768 samples, converted span by span)"
        RH_TEST_SOURCE=buffer tests/cpp/host_mirror_test bench 256 4194304 32768 16
    } > $E/r04_pull_path.txt 2>&1
    tools/ubench/write_bw > $E/r04_write_bw.txt 2>&1
    python tools/filter_contract.py > $E/r04_filter_contract.txt 2>/dev/null
    for nio in 0 1; do for shape in "64 1048576" "2048 32768"; do set -- $shape; echo "RH_LIMIT_NIO=$nio streams=$1 frames=$2: $(RH_LIMIT_NIO=$nio RH_BENCH_NO_PMC=1 python bench.py --config limit --sources $1 --frames $2 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))")"; done; done > $E/r04_limit_io_waves.txt 2>&1
    python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $E/r04_final_gputests.txt
    ;;
4)  # what changed after parts 1-3: the ragged pair in one kernel, the AGC in one kernel, config 3's pipelined walk
    for u in 0 2 4 6 8; do echo "RH_RS_PIPE=$u: $(RH_RS_PIPE=$u RH_BENCH_NO_PMC=1 python bench.py --config 3 --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))")"; done > $E/r04_cfg3_pipe.txt 2>&1
    for c in ragged agc 3; do python bench.py --config $c > $E/r04_bench_$c.json 2>/dev/null; done
    python bench.py --config agc --sources 2048 --frames 32768 > $E/r04_bench_agc_2048.json 2>/dev/null
    RH_AGC_SEGMENTS=1 RH_BENCH_NO_PMC=1 python bench.py --config agc --no-cpu-baseline > $E/r04_bench_agc_segments.json 2>/dev/null
    RH_RAG_TWO_KERNELS=1 RH_BENCH_NO_PMC=1 python bench.py --config ragged --no-cpu-baseline > $E/r04_bench_ragged_two_kernels.json 2>/dev/null
    for c in ragged 3; do
        case $c in 3) export RH_PROF_KERNEL=reverb_spatial;; *) export RH_PROF_KERNEL=k_rlm;; esac
        bash tools/pmc_cmd.sh r04_$c python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
        cp gpurun_out/prof/r04_$c/summary.txt $E/r04_${c}_kernel_trace_pmc.txt
    done
    export RH_PROF_KERNEL=k_agc
    bash tools/pmc_cmd.sh r04_agc python bench.py --config agc --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r04_agc/summary.txt $E/r04_agc_64x1Mi_kernel_trace_pmc.txt
    bash tools/pmc_cmd.sh r04_agc2048 python bench.py --config agc --sources 2048 --frames 32768 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r04_agc2048/summary.txt $E/r04_agc_2048x32Ki_kernel_trace_pmc.txt
    tools/ubench/stream_ring o > $E/r04_stream_ring_occupancy.txt 2>&1
    ;;
5)  # the final library: the headline again (k_rlm_chunk's DMA runs and per-XCD tickets), the ragged batch, the suite
    python bench.py > $E/r04_bench_cfg2.json 2> $E/r04_bench_cfg2.err
    bash tools/pmc_cmd.sh r04_cfg2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-autotune --no-per-source > /dev/null 2>&1
    cp gpurun_out/prof/r04_cfg2/summary.txt $E/r04_cfg2_kernel_trace_pmc.txt
    RH_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 > $E/r04_bench_gpus2_one_device.json 2> $E/r04_bench_gpus2_one_device.err
    python bench.py --shared-device --no-cpu-baseline --no-per-source > $E/r04_bench_cfg2_tiles_by_ticket.json 2>/dev/null
    python bench.py --config 2span > $E/r04_bench_2span.json 2>/dev/null
    python bench.py --config 2mono > $E/r04_bench_2mono.json 2>/dev/null
    python bench.py --config ragged > $E/r04_bench_ragged.json 2>/dev/null
    export RH_PROF_KERNEL=k_rlm
    bash tools/pmc_cmd.sh r04_ragged python bench.py --config ragged --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r04_ragged/summary.txt $E/r04_ragged_kernel_trace_pmc.txt
    python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $E/r04_final_gputests.txt
    ;;
6)  # the side configurations again: final library, bench.py timing a region of back-to-back launches with one pair of events
    for c in 3 5 limit biquad agc; do python bench.py --config $c > $E/r04_bench_$c.json 2>/dev/null; done
    for c in limit agc biquad; do python bench.py --config $c --sources 2048 --frames 32768 > $E/r04_bench_${c}_2048.json 2>/dev/null; done
    export RH_PROF_KERNEL=reverb_spatial
    bash tools/pmc_cmd.sh r04_3 python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r04_3/summary.txt $E/r04_3_kernel_trace_pmc.txt
    ;;
esac
ls -la $E | tail -40
