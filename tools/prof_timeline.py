"""Timeline of an RH_PHASE_PROFILE run, by blocks of tiles: python tools/prof_timeline.py dump.bin [block]   (u64 [tiles][8]:
six phase sums, {xcc, hw_id}, start tick).  For batches whose tiles carry unequal loads (ragged): when does a block of tiles
start, when does it end, and how much of its time is spent waiting for its ring."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ph = a[:, :6].astype(np.float64)
start = a[:, 7].astype(np.float64)
xcc = (a[:, 6] >> np.uint64(32)).astype(np.int64) & 0xf
for x in set(xcc.tolist()):  # every XCD counts its own cycles: starts are comparable inside an XCD only
    start[xcc == x] -= start[xcc == x].min()
end = start + ph.sum(axis=1)
span = end.max()
print(f"tiles {len(a)}; span {span:.0f} ticks; tiles started within {start.max():.0f} ticks")
print("block: start (mean) | end (mean, max) | duration | wait / stage / sum / rest, % of the duration")
for i in range(0, len(a), blk):
    s, e, p = start[i:i + blk], end[i:i + blk], ph[i:i + blk]
    d = (e - s).mean()
    rest = p[:, [0, 3, 5]].sum(axis=1).mean()
    print(f"{i:5d}: {s.mean():8.0f} | {e.mean():8.0f} {e.max():8.0f} | {d:8.0f} | {100 * p[:, 2].mean() / d:5.1f} {100 * p[:, 1].mean() / d:5.1f} {100 * p[:, 4].mean() / d:5.1f} {100 * rest / d:5.1f}")
