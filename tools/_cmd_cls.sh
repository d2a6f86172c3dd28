cd /root/repo
mkdir -p gpurun_out/cls12
timeout 1200 python -m pytest tests/test_gpu_mix_first.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/cls12/pytest.txt
echo done
