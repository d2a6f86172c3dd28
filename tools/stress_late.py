"""Race hunt under multi-process contention: `late` GpuMixer jobs, 4 at a time; every output must equal the first."""
import os, subprocess, sys, tempfile
import numpy as np
EXE = "tests/cpp/host_mirror_test"
def rnd(seed, n, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, n) * scale).astype(np.float32)
ns = [60000, 45000, 30011, 52000, 20000]
gains = np.array([1.0, 0.5, 0.8, 1.1, 0.6], dtype=np.float32)
filt = sys.argv[1] if len(sys.argv) > 1 else "-1"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dirs = []
for k in range(4):
    d = tempfile.mkdtemp()
    for i, n in enumerate(ns):
        rnd(3300 + i, 2 * n, 0.15).tofile(f"{d}/src_{i}.f32")
    gains.tofile(f"{d}/gains.f32")
    dirs.append(d)
ref = None
bad = 0
for it in range(rounds):
    procs = [subprocess.Popen([EXE, "late", d, "3", "2", "44100", "48000", filt, "200", "8192", "4", "10"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for d in dirs]
    for d, pr in zip(dirs, procs):
        out, err = pr.communicate()
        if pr.returncode:
            print(f"  round {it}: exit {pr.returncode}: {err.strip()[:120]}"); bad += 1; continue
        got = np.fromfile(f"{d}/out.f32", dtype=np.float32)
        join = int(open(f"{d}/join.txt").read())
        if ref is None:
            ref, ref_join = got, join
        elif join != ref_join:
            print(f"  round {it}: join {join} vs {ref_join} (timing-dependent by design), len {len(got)} vs {len(ref)}")
        elif len(got) != len(ref) or not np.array_equal(got, ref):
            bad += 1
            if len(got) == len(ref):
                idx = np.nonzero(got != ref)[0]
                z = int(np.count_nonzero(got[idx] == 0.0))
                nn = int(np.count_nonzero(np.isnan(got[idx])))
                print(f"  round {it}: {len(idx)} samples differ, frames {idx[0]//2}..{idx[-1]//2} of {len(ref)//2}, zeros among them {z}, NaNs {nn}, max {np.nanmax(np.abs(got-ref)):.3g}")
            else:
                print(f"  round {it}: length {len(got)} vs {len(ref)}")
print("late filt", filt, "bad runs:", bad, "of", rounds * 4)
