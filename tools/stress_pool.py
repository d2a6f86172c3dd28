"""Race hunt for the pull path of round 3 (pull threads, prefetch ahead of the wait, copy stream): the same GpuMixer job many
times, large blocks (the pool and the pitched copy engage), plain / NaN-poisoned staging, spanned and continuous sources;
every run must give the bits of the first, and the first is compared with the oracle.
    python tools/stress_pool.py [runs]"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import rodio_oracle as O
EXE = "tests/cpp/host_mirror_test"
def rnd(seed, n, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, n) * scale).astype(np.float32)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for kind in ("test", "mixed"):
    spec = [(2, 44100, 0.5 + 0.05 * i, 150000 + 1111 * i) if kind == "test" else (2 if i % 4 else 1, (44100, 48000, 32000)[i % 3], 0.5 + 0.05 * i, 150000 + 1111 * i) for i in range(16)]
    d = tempfile.mkdtemp()
    xs = [rnd(9100 + i, ch * n, 0.08) for i, (ch, _, _, n) in enumerate(spec)]
    for i, x in enumerate(xs):
        x.tofile(f"{d}/src_{i}.f32")
    open(f"{d}/spec.txt", "w").write("".join(f"{ch} {rate} {g}\n" for ch, rate, g, _ in spec))
    m = O.Mixer(2, 48000)
    for i, (ch, rate, g, _) in enumerate(spec):
        k = kind if kind != "mixed" else ["test", "buffer", "spans:4096"][i % 3]
        src = O.SamplesBuffer(ch, rate, xs[i]) if k == "buffer" else O.SpanSource(xs[i], ch, rate, 4096) if k.startswith("spans") else O.TestSource(xs[i], ch, rate)
        m.add(O.UniformSourceIterator(src.amplify(float(np.float32(g))), 2, 48000))
    ref = m.collect()
    first = None
    for it in range(N):
        env = dict(os.environ, RH_TEST_SOURCE=kind)
        if it % 2:
            env["RODIO_HIP_DEBUG_POISON"] = "1"
        r = subprocess.run([EXE, "mixany", d, str(len(spec)), "48000", "-1", "0", "32768", "0"], capture_output=True, text=True, env=env)
        if r.returncode:
            print(kind, "run failed", r.stderr[:300]); bad += 1; continue
        got = np.fromfile(f"{d}/out.f32", dtype=np.float32)
        if first is None:
            first = got
            ok = len(got) == len(ref) and np.array_equal(got, ref)
            print(kind, "first run == oracle (bit for bit):", ok, r.stderr.strip().split("\n")[-1])
            bad += 0 if ok else 1
        elif len(got) != len(first) or not np.array_equal(got, first):
            bad += 1
            print(kind, f"run {it}: differs from the first run")
print(f"{2 * N} runs, {bad} bad")
sys.exit(1 if bad else 0)
