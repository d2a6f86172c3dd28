"""debug: each source alone through GpuMixer (mixany) with a filter, error vs the oracle"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rodio_oracle as O
EXE = "tests/cpp/host_mirror_test"
def rnd(seed, n, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, n) * scale).astype(np.float32)
spec = [(2, 44100, 1.0, 30000), (1, 44100, 0.7, 25000), (2, 48000, 0.9, 20000), (6, 22050, 0.5, 9000), (2, 96000, 0.8, 50000),
        (2, 192000, 0.6, 70000), (2, 44100, 1.1, 12345), (1, 8000, 0.4, 4000)]
freq = int(sys.argv[1]) if len(sys.argv) > 1 else 300
R = sys.argv[2] if len(sys.argv) > 2 else "4"
for i, (ch, rate, g, n) in enumerate(spec):
    d = tempfile.mkdtemp()
    x = rnd(3400 + i, ch * n, 0.1)
    x.tofile(f"{d}/src_0.f32")
    open(f"{d}/spec.txt", "w").write(f"{ch} {rate} {g}\n")
    r = subprocess.run([EXE, "mixany", d, "1", "48000", "0", str(freq), "8192", R], capture_output=True, text=True)
    got = np.fromfile(f"{d}/out.f32", dtype=np.float32)
    ref = O.UniformSourceIterator(O.TestSource(x, ch, rate).amplify(float(np.float32(g))), 2, 48000).low_pass(freq).collect()
    print(i, ch, rate, len(got), len(ref), float(np.max(np.abs(got - ref))) if len(got) == len(ref) else "LEN", r.stderr.strip()[:100])
