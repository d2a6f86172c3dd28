"""Timeline of back-to-back launches from a rocprofv3 run (rocpd sqlite): for the kernels matching RH_PROF_KERNEL, the gap between the
end of one dispatch and the start of the next, and -- when the run also traced the HIP API (--hip-trace) -- the host time inside the
launch calls.  VERDICT r4 next #4: where do the ~20 us between the traced kernel time and the time per call go?

    rocprofv3 --kernel-trace --hip-trace -d <dir> -o t -- python bench.py --config limit --steps 13 --no-cpu-baseline
    python tools/launch_gaps.py <dir>/**/t_results.db
"""
import os
import sqlite3
import sys

PAT = "%" + os.environ.get("RH_PROF_KERNEL", "k_limit_scan") + "%"


def main(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view') order by name")]
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("## kernels view columns:", ", ".join(cols))
    rows = cur.execute("select name, start, end, duration from kernels order by start").fetchall()
    print(f"## {len(rows)} dispatches; detail for {PAT}")
    like = PAT.strip("%")
    prev_end = None
    sel = []
    for i, (name, st, en, dur) in enumerate(rows):
        if like in name:
            gap = (st - prev_end) if prev_end is not None else None
            before = rows[i - 1][0][:40] if i else "-"
            sel.append((st, en, dur, gap, before))
        prev_end = en
    t0 = sel[0][0] if sel else 0
    print(f"{'start_us':>12} {'dur_us':>10} {'gap_before_us':>14}  previous dispatch")
    for st, en, dur, gap, before in sel[-16:]:
        print(f"{(st - t0) / 1e3:12.1f} {dur / 1e3:10.1f} {'' if gap is None else f'{gap / 1e3:14.1f}'}  {before}")
    if len(sel) > 3:
        steady = sel[len(sel) // 3:]
        per = (steady[-1][0] - steady[0][0]) / (len(steady) - 1)
        dur = sum(x[2] for x in steady) / len(steady)
        print(f"## steady state over the last {len(steady)} launches: start-to-start {per / 1e3:.1f} us, kernel {dur / 1e3:.1f} us, everything between two kernels {(per - dur) / 1e3:.1f} us")
    # host side, if traced
    for cand in ("regions", "hip_api", "api"):
        if cand in names or any(n.startswith(cand) for n in names):
            pass
    api = [n for n in names if "region" in n.lower() or "api" in n.lower()]
    print("## tables/views that may hold the HIP API trace:", ", ".join(api) or "(none)")
    for tbl in ("regions",):
        if tbl in names:
            c2 = [r[1] for r in cur.execute(f"pragma table_info({tbl})")]
            print(f"## {tbl} columns:", ", ".join(c2))
            try:
                q = f"select name, count(*), avg(end - start), max(end - start) from {tbl} group by name order by sum(end - start) desc limit 12"
                for name, n, avg, mx in cur.execute(q):
                    print(f"{n:8d} calls  avg {avg / 1e3:9.1f} us  max {mx / 1e3:9.1f} us  {name}")
            except sqlite3.Error as e:  # the schema differs between rocprofv3 versions: say so, keep the kernel part
                print("   (could not summarise:", e, ")")


if __name__ == "__main__":
    main(sys.argv[1])
