#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_multi.py -q -m gpu -x 2>&1 | tail -12 > $O/tests.log
timeout 600 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for c in 2span ragged 3 5 limit biquad; do timeout 300 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 300 python bench.py --config agc > $O/bench_agc.json 2> $O/bench_agc.err
timeout 300 python bench.py --config limit --sources 2048 --frames 32768 > $O/bench_limit_2048.json 2> $O/bench_limit_2048.err
timeout 300 python bench.py --config agc --sources 2048 --frames 32768 > $O/bench_agc_2048.json 2> $O/bench_agc_2048.err
timeout 300 python bench.py --config biquad --sources 2048 --frames 32768 > $O/bench_biquad_2048.json 2> $O/bench_biquad_2048.err
echo "== tests"; tail -5 $O/tests.log
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(d["metric"], "| value %.4g" % d["value"], "| ms/step %.4f" % d["ms_per_step"], "| roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k in ("achieved","frac","traffic","kernel_ms")})
    for k in ("parity","cpu_baseline"):
        if k in d: print("  ",k, d[k])
    if "kernels" in d["config"]:
        for r in d["config"]["kernels"]: print("   ", r["kernel"], "kernel_ms %.4f frac %.4f" % (r["kernel_ms"], r["frac"]))
    if "geometry" in d["config"]: print("   geo", d["config"]["kernel"], d["config"]["geometry"].get("frames_per_lane"), d["config"]["geometry"].get("ring_stages"), d["roofline"].get("traffic_detail"))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
done
