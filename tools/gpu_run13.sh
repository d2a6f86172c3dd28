cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_limit.py tests/test_gpu_effects.py tests/test_host_mirror.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -n 3
GEOMS="auto" bash tools/gpu_sweep_mid.sh 2>&1 | tee gpurun_out/r02_scan_geometry_auto_after.txt
