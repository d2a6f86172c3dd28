cd /root/repo
mkdir -p gpurun_out/fin
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/fin/smoke.txt 2>&1
bash tools/pmc_cmd.sh r06_cfg2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-per-class --no-side > /dev/null 2>&1
bash tools/pmc_cmd.sh r06_limit python bench.py --config limit --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
bash tools/pmc_cmd.sh r06_limit_2048 python bench.py --config limit --sources 2048 --frames 32768 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
bash tools/kt_cmd.sh r06_per_class python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side > /dev/null 2>&1
bash tools/kt_cmd.sh r06_stream_overlap python bench.py --config stream --overlap --no-cpu-baseline > /dev/null 2>&1
echo done
