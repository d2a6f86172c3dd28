"""rocprofv3 TCC counter databases -> traffic.json (see tools/pmc_traffic.sh).

    python tools/traffic_json.py <dir with fetch/ write/ cal/> [bench args]
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  The calibration run moves
known_bytes with global_load_lds_dwordx4; factor = known_bytes / (FETCH_SIZE*1024) (2.0 on gfx950
per MI355X_MICROARCH.md) corrects the kernel's read side.  WRITE_SIZE is uncalibrated."""
import glob, json, re, sqlite3, sys

d = sys.argv[1]
args = sys.argv[2:]


def mean_counter(sub, name, like):
    dbs = glob.glob(f"{d}/{sub}/**/*_results.db", recursive=True) + glob.glob(f"{d}/{sub}/*_results.db")
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select avg(value), count(*) from counters_collection where counter_name=? and kernel_name like ?", (name, like)).fetchone()
    return rows[0], rows[1]


fetch_kib, n1 = mean_counter("fetch", "FETCH_SIZE", "%k_rlm%")
write_kib, n2 = mean_counter("write", "WRITE_SIZE", "%k_rlm%")
cal_kib, n3 = mean_counter("cal", "FETCH_SIZE", "%")
known = None
for line in open(f"{d}/cal.log"):
    m = re.match(r"known_bytes_per_launch (\d+)", line)
    if m:
        known = int(m.group(1))
factor = known / (cal_kib * 1024.0) if known and cal_kib else None


def opt(name, default):
    return int(args[args.index(name) + 1]) if name in args else default


res = {
    "sources": opt("--sources", 256), "frames": opt("--frames", 1 << 20), "span": opt("--span", 0),
    "FETCH_SIZE_KiB_per_launch_raw": fetch_kib, "WRITE_SIZE_KiB_per_launch_raw": write_kib,
    "fetch_calibration": {"known_bytes": known, "FETCH_SIZE_KiB_raw": cal_kib, "factor": factor,
                          "how": "tools/ubench/stream_ring mimic: the kernel's chunk pattern with a known byte count (informative; see profiles/README.md)"},
    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
    # streaming read (16 B/lane, global_load and buffer_load ... lds alike): double it.  WRITE_SIZE as is.
    "fetch_correction": 2.0,
    "hbm_bytes_per_launch": (fetch_kib * 1024.0 * 2.0 + write_kib * 1024.0) if fetch_kib and write_kib is not None else None,
    "dispatches_averaged": [n1, n2, n3],
    "geometry": {"frames_per_lane": opt("--frames-per-lane", 0), "ring_stages": opt("--ring-stages", 0)},
}
print(json.dumps(res, indent=1))
