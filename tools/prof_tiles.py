"""Per-tile view of an RH_PHASE_PROFILE run: python tools/prof_tiles.py dump.bin   (u64 [tiles][8])"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
n = len(a)
ph = a[:, :6].astype(np.float64)
tot = ph.sum(axis=1)
xcc = (a[:, 6] >> np.uint64(32)).astype(np.int64) & 0xf
hw = (a[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
# HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
start = a[:, 7].astype(np.float64); start -= start.min()
print("tiles", n, "distinct CUs", len(set(cuid)), "waves/CU histogram", np.bincount(np.bincount(cuid)))
print("start skew ticks: max", start.max(), "median", np.median(start))
names = ["gran", "stage", "wait", "carry", "run", "scan"]
print("mean ticks", {k: round(v) for k, v in zip(names, ph.mean(axis=0))}, "total", round(tot.mean()))
work = tot - ph[:, 3]
print("non-carry ticks: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (work.min(), *np.percentile(work, [10, 50, 90]), work.max()))
print("carry wait: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (ph[:, 3].min(), *np.percentile(ph[:, 3], [10, 50, 90]), ph[:, 3].max()))
order = np.argsort(ph[:, 3])
print("tiles with least carry wait (pace setters):", order[:24].tolist())
print("  their xcc:", xcc[order[:24]].tolist())
print("  waves on their CU:", [int((cuid == cuid[t]).sum()) for t in order[:24]])
wcu = np.array([int((cuid == c).sum()) for c in cuid])
for k in sorted(set(wcu)):
    m = wcu == k
    print(f"CUs with {k} waves: {m.sum()} tiles, non-carry ticks mean {work[m].mean():.0f}, carry {ph[m, 3].mean():.0f}, run {ph[m,4].mean():.0f} wait {ph[m,2].mean():.0f}")
for x in range(8):
    m = xcc == x
    print(f"xcc {x}: {m.sum()} tiles, non-carry {work[m].mean():.0f} carry {ph[m,3].mean():.0f}")
blk = 64
print("by tile block of", blk, ": carry-wait mean", [int(ph[i:i + blk, 3].mean()) for i in range(0, n, blk * 4)])
print("by tile block of", blk, ": non-carry mean", [int(work[i:i + blk].mean()) for i in range(0, n, blk * 4)])
