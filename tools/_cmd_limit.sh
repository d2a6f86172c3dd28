cd /root/repo
mkdir -p gpurun_out/lim2
timeout 900 python -m pytest tests/test_gpu_limit.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/lim2/pytest.txt
for v in "" limit_nolb; do
  for shape in "" "--sources 2048 --frames 32768" "--sources 512 --frames 131072"; do
    tag="${v:-shipped}$(echo $shape | tr -d ' -')"
    if [ -n "$v" ]; then export RODIO_HIP_LIB=/root/repo/variants/librodio_hip_$v.so; else unset RODIO_HIP_LIB; fi
    timeout 300 python bench.py --config limit $shape --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/lim2/bench_$tag.json
  done
done
echo done
