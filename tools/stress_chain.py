"""Flake hunt: one GpuSource chain job many times; every output must be identical.  python tools/stress_chain.py <runs> <block> op [op ...]"""
import hashlib, subprocess, sys, tempfile
import numpy as np
EXE = "tests/cpp/host_mirror_test"
runs, block, ops = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
d = tempfile.mkdtemp()
x = (np.random.default_rng(3204).uniform(-1, 1, 2 * 30000) * 0.9).astype(np.float32)
x.tofile(f"{d}/src_0.f32")
seen = {}
for i in range(runs):
    r = subprocess.run([EXE, "chain", d, "2", "48000", block] + ops, capture_output=True, text=True)
    if r.returncode:
        print("run", i, "failed:", r.stderr[:200]); continue
    got = np.fromfile(f"{d}/out.f32", dtype=np.float32)
    h = hashlib.md5(got.tobytes()).hexdigest()
    if h not in seen:
        seen[h] = (i, got)
        if len(seen) > 1:
            ref = list(seen.values())[0][1]
            if len(ref) == len(got):
                idx = np.nonzero(ref != got)[0]
                print(f"run {i}: differs from run 0 in {len(idx)} samples, first {idx[0]} last {idx[-1]}, max {np.nanmax(np.abs(ref - got)):.3g}, nan {int(np.isnan(got).sum())}")
            else:
                print(f"run {i}: length {len(got)} vs {len(ref)}")
print("distinct outputs:", len(seen), "of", runs)
