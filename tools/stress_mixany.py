"""Race hunt: the same GpuMixer job many times under background load; outputs must be identical."""
import os, subprocess, sys, tempfile
import numpy as np
EXE = "tests/cpp/host_mirror_test"
def rnd(seed, n, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, n) * scale).astype(np.float32)
mode = sys.argv[1] if len(sys.argv) > 1 else "any"
if mode == "any":
    spec = [(2, 44100, 1.0, 30000), (1, 44100, 0.7, 25000), (2, 48000, 0.9, 20000), (6, 22050, 0.5, 9000), (2, 96000, 0.8, 50000),
            (2, 192000, 0.6, 70000), (2, 44100, 1.1, 12345), (1, 8000, 0.4, 4000)]
elif mode == "stereo_rates":   # no GpuSource adapters in front: only generations
    spec = [(2, 44100, 1.0, 30000), (2, 48000, 0.9, 20000), (2, 22050, 0.5, 9000), (2, 96000, 0.8, 50000), (2, 44100, 1.1, 12345)]
elif mode == "one_rate":       # a single generation
    spec = [(2, 44100, 1.0, 30000), (2, 44100, 0.9, 20000), (2, 44100, 0.5, 9000)]
elif mode == "adapters":       # one generation, all through adapters
    spec = [(1, 44100, 0.7, 25000), (6, 44100, 0.5, 9000), (1, 44100, 0.4, 4000)]
d = tempfile.mkdtemp()
for i, (ch, rate, g, n) in enumerate(spec):
    rnd(3400 + i, ch * n, 0.1).tofile(f"{d}/src_{i}.f32")
open(f"{d}/spec.txt", "w").write("".join(f"{ch} {rate} {g}\n" for ch, rate, g, _ in spec))
ref = None
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for it in range(N):
    r = subprocess.run([EXE, "mixany", d, str(len(spec)), "48000", "-1", "0", "8192", "4"], capture_output=True, text=True)
    if r.returncode:
        print("run failed", r.stderr[:200]); bad += 1; continue
    got = np.fromfile(f"{d}/out.f32", dtype=np.float32)
    if ref is None:
        ref = got
    elif len(got) != len(ref) or not np.array_equal(got, ref):
        bad += 1
        if len(got) == len(ref):
            idx = np.nonzero(got != ref)[0]
            print(f"  run {it}: {len(idx)} samples differ, first {idx[0]} last {idx[-1]} (frames {idx[0]//2}..{idx[-1]//2}) max {np.max(np.abs(got-ref)):.3g}")
            sys.path.insert(0, os.getcwd())
            from oracle import rodio_oracle as O
            a, b = idx[0], idx[-1] + 1
            dd = (got - ref)[a:b].astype(np.float64)
            for i, (ch, rate, g, n) in enumerate(spec):
                u = O.UniformSourceIterator(O.TestSource(rnd(3400 + i, ch * n, 0.1), ch, rate).amplify(float(np.float32(g))), 2, 48000).collect()
                seg = np.zeros(b - a); m = min(b, len(u)) - a
                if m > 0: seg[:m] = u[a:a + m]
                if np.any(seg):
                    coef = float(np.dot(dd, seg) / np.dot(seg, seg))
                    resid = float(np.max(np.abs(dd - coef * seg)))
                    print(f"     source {i} ({ch}ch {rate}): diff ~ {coef:+.3f} x its stream, residual {resid:.3g}")
        else:
            print(f"  run {it}: length {len(got)} vs {len(ref)}")
print(mode, "mismatching runs:", bad, "of", N)
