cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for cfg in "265 limit" "265 limit" "777 limit" "512 limit" "100 limit"; do
  echo "== block $cfg"
  python tools/stress_chain.py 150 $cfg 2>&1 | tail -n 6
done
} > gpurun_out/r02_limit_flake_after.txt 2>&1
cat gpurun_out/r02_limit_flake_after.txt
