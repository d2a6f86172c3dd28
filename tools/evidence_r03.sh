#!/bin/bash
# Round 3 evidence, in parts (one gpurun call each): bash tools/evidence_r03.sh <part>.  Everything lands in gpurun_out/ev/ and is
# copied to profiles/ by hand.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ev
export TMPDIR=/tmp
E=gpurun_out/ev
case "$1" in
1)  # the headline
    python bench.py > $E/r03_bench_cfg2.json 2> $E/r03_bench_cfg2.err
    bash tools/pmc_cmd.sh r03_cfg2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-autotune > /dev/null 2>&1
    cp gpurun_out/prof/r03_cfg2/summary.txt $E/r03_cfg2_kernel_trace_pmc.txt
    python bench.py --freq 1000 > $E/r03_bench_cfg2_lp1000.json 2>/dev/null
    RH_NO_CHUNK=1 python bench.py --no-cpu-baseline > $E/r03_bench_cfg2_two_launches.json 2>/dev/null
    RH_NO_MIX_FIRST=1 python bench.py --no-cpu-baseline > $E/r03_bench_cfg2_per_source.json 2>/dev/null
    ;;
2)  # the side configs
    for c in 2span 2mono 3 5 ragged limit agc biquad; do python bench.py --config $c > $E/r03_bench_$c.json 2>/dev/null; done
    for c in limit agc biquad; do python bench.py --config $c --sources 2048 --frames 32768 > $E/r03_bench_${c}_2048.json 2>/dev/null; done
    ;;
3)  # AGC profiles, the pull path, the microbenchmarks, the suite
    bash tools/pmc_cmd.sh r03_agc python bench.py --config agc --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r03_agc/summary.txt $E/r03_agc_64x1Mi_kernel_trace_pmc.txt
    bash tools/pmc_cmd.sh r03_agc2048 python bench.py --config agc --sources 2048 --frames 32768 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    cp gpurun_out/prof/r03_agc2048/summary.txt $E/r03_agc_2048x32Ki_kernel_trace_pmc.txt
    {
        echo "## tools/h2d_bw.py"; python tools/h2d_bw.py
        echo "## tests/cpp/host_mirror_test bench <sources> <frames> <block_frames> <host_threads>"
        tests/cpp/host_mirror_test bench 256 1048576 16384 1
        tests/cpp/host_mirror_test bench 256 1048576 16384 16
        tests/cpp/host_mirror_test bench 256 4194304 32768 16
        tests/cpp/host_mirror_test bench 256 4194304 65536 16
        echo "## RH_TEST_SOURCE=buffer (SamplesBuffer: spans of 32768 samples, converted span by span)"
        RH_TEST_SOURCE=buffer tests/cpp/host_mirror_test bench 256 4194304 32768 16
    } > $E/r03_pull_path.txt 2>&1
    tools/ubench/stream_ring t > $E/r03_ubench_stream_ring.txt 2>&1
    python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $E/r03_final_gputests.txt
    ;;
esac
ls -la $E | tail -30
