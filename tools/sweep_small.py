"""Quick geometry sweep -- python tools/sweep_small.py [R,NS ...] [-- bench args]   (GPU box)"""
import json, subprocess, sys
args = sys.argv[1:]
extra = args[args.index("--") + 1:] if "--" in args else []
geos = [tuple(int(v) for v in a.split(",")) for a in (args[:args.index("--")] if "--" in args else args)] or [(6, 3), (8, 2), (8, 3), (9, 2), (9, 3), (12, 3)]
rows = []
for R, NS in geos:
    cmd = [sys.executable, "bench.py", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--frames-per-lane", str(R), "--ring-stages", str(NS)] + extra
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        g = j["config"]["geometry"]
        rows.append((j["roofline"]["kernel_ms"], R, NS))
        print(f"R={R:2d} NS={NS} kernel_ms={j['roofline']['kernel_ms']:.3f} frac={j['roofline']['frac']:.3f} tiles={g['n_tiles']} resident/CU={g['resident_waves_per_cu']} "
              f"lds={g['lds_bytes']} J={g['lookback_tiles']} late={g['late_carries_per_launch']:.0f}", flush=True)
    except Exception as e:
        print(f"R={R:2d} NS={NS} failed {type(e).__name__}: {out.stderr[-300:] if 'out' in dir() else ''}", flush=True)
rows.sort()
print("best:", rows[:4])
