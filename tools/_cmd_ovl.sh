cd /root/repo
mkdir -p gpurun_out/ovl
timeout 900 python -m pytest tests/test_gpu_mix_first.py -m gpu -x -q -k "one_launch or stream" 2>&1 | tail -8 > gpurun_out/ovl/pytest.txt
for b in 65536 16384 262144; do
  timeout 300 python bench.py --config stream --block $b --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ovl/stream_$b.json
  timeout 300 python bench.py --config stream --block $b --overlap --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ovl/stream_${b}_overlap.json
done
echo done
