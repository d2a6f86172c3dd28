cd /root/repo
mkdir -p gpurun_out/cls3
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-side --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', 'headline', r['frac'], 'per_source', r.get('per_source',{}).get('frac'), 'per_class', r['per_class']['call_ms'], r['per_class']['frac'])" >> gpurun_out/cls3/out.txt 2>&1; }
run multi_ticket A=1
run multi_direct RH_X_CLS_DIRECT=1
run each_chunk RH_CLASSES_ONE_BY_ONE=1
run each_nochunk RH_CLASSES_ONE_BY_ONE=1 RH_NO_CHUNK=1
run each_nomixfirst RH_CLASSES_ONE_BY_ONE=1 RH_NO_MIX_FIRST=1
run each_half RH_CLASSES_ONE_BY_ONE=1 RH_CHUNK_HALF=1
run multi_half RH_CHUNK_HALF=1
run multi_half_direct RH_CHUNK_HALF=1 RH_X_CLS_DIRECT=1
echo done
