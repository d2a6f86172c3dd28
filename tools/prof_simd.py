"""Where did the dispatcher put the single-wave workgroups?  python tools/prof_simd.py dump.bin  (RH_PHASE_PROFILE run)"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
xcc = (a[:, 6] >> np.uint64(32)).astype(np.int64) & 0xf
hw = (a[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
wave = hw & 0xf; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = (((xcc * 8 + se) * 2 + sh) * 16 + cu)
ids = sorted(set(cuid))
per_cu = np.array([(cuid == c).sum() for c in ids])
print("tiles", len(a), "CUs used", len(ids), "waves/CU histogram", np.bincount(per_cu))
pat = {}
for c in ids:
    m = cuid == c
    key = tuple(sorted(np.bincount(simd[m], minlength=4).tolist(), reverse=True))
    pat[key] = pat.get(key, 0) + 1
print("per-CU SIMD load patterns (sorted loads: count of CUs):", dict(sorted(pat.items(), key=lambda kv: -kv[1])))
tot = a[:, :6].astype(np.float64).sum(axis=1)
maxload = np.array([np.bincount(simd[cuid == c], minlength=4).max() for c in cuid])
for k in sorted(set(maxload)):
    print(f"waves on a CU whose busiest SIMD holds {k}: {int((maxload == k).sum())}, mean busy ticks {tot[maxload == k].mean():.0f}")
