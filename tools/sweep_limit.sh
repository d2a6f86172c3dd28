cd $GRAFT_REPO_ROOT
export RH_BENCH_NO_PMC=1
for cfg in "16 8 1" "16 4 2" "8 8 2" "8 4 2" "8 4 4"; do
  set -- $cfg
  for shape in "64 1048576" "2048 32768"; do
    set -- $cfg; R=$1; NW=$2; W=$3; set -- $shape
    out=$(RH_LIMIT_R=$R RH_LIMIT_NW=$NW RH_LIMIT_WGS=$W python bench.py --config limit --sources $1 --frames $2 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))")
    echo "R=$R NW=$NW wgs=$W streams=$1 frames=$2 : $out"
  done
done
