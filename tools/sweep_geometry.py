"""Times the fused kernel over launch geometries (frames per lane x threads) on the GPU box.
    python tools/sweep_geometry.py [--span N]   -> one line per geometry, best last"""
import json
import subprocess
import sys

extra = sys.argv[1:]
rows = []
for R in (4, 6, 8, 12, 16):
    for T in (128, 192, 256, 320, 384, 448, 512):
        cmd = [sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
               "--frames-per-lane", str(R), "--ring-stages", str(T)] + extra
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
            j = json.loads(line)
            rows.append((j["roofline"]["kernel_ms"], R, T, j["roofline"]["frac"]))
            print(f"R={R:2d} T={T:3d} kernel_ms={j['roofline']['kernel_ms']:.3f} frac={j['roofline']['frac']:.3f}", flush=True)
        except Exception as e:  # unsupported geometry, timeout ...
            print(f"R={R:2d} T={T:3d} failed: {type(e).__name__} {str(out.stderr)[-200:] if 'out' in dir() else ''}", flush=True)
rows.sort()
print("best:", rows[:5])
