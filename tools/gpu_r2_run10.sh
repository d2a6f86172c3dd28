cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for g in "16 4 2" "8 8 2" "8 16 1"; do set -- $g
for cfg in "64 1048576" "2048 32768"; do set -- $g $cfg
RH_LIMIT_R=$1 RH_LIMIT_NW=$2 RH_LIMIT_WGS=$3 RH_BENCH_NO_PMC=1 python bench.py --config limit --sources $4 --frames $5 --steps 30 --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/limit_x.json
python - <<PY
import json
d = json.load(open("gpurun_out/limit_x.json")); print("R=$1 NW=$2 wgs=$3: $4 x $5", round(d["roofline"]["kernel_ms"],4), round(d["roofline"]["frac"],4))
PY
done; done
