"""Quick geometry sweep (subset) -- python tools/sweep_small.py [bench args]"""
import json, subprocess, sys
extra = sys.argv[1:]
rows = []
for R, T in [(4, 256), (6, 256), (6, 384), (8, 192), (8, 256), (8, 384), (12, 128), (12, 192), (12, 256), (12, 384), (12, 448), (16, 256), (16, 384)]:
    cmd = [sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--frames-per-lane", str(R), "--threads", str(T)] + extra
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        rows.append((j["roofline"]["kernel_ms"], R, T))
        print(f"R={R:2d} T={T:3d} kernel_ms={j['roofline']['kernel_ms']:.3f} frac={j['roofline']['frac']:.3f}", flush=True)
    except Exception as e:
        print(f"R={R:2d} T={T:3d} failed {type(e).__name__}: {out.stderr[-300:] if 'out' in dir() else ''}", flush=True)
rows.sort()
print("best:", rows[:4])
