cd /root/repo
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/full/pytest.txt
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/full/bench_default.json
for shape in "" "--sources 2048 --frames 32768"; do
  tag="limit$(echo $shape | tr -d ' -')"
  timeout 300 python bench.py --config limit $shape 2>&1 | tail -1 > gpurun_out/full/bench_$tag.json
done
echo done
