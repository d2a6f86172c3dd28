"""Turns rocprofv3 result databases (rocpd sqlite, the default output of ROCm 7.2's rocprofv3)
into a small text summary that can be committed under profiles/.

    python tools/prof_summary.py <kernel-trace.db> [<pmc.db> ...] > profiles/<name>.txt
"""
import os
import sqlite3
import sys

PAT = "%" + os.environ.get("RH_PROF_KERNEL", "k_rlm") + "%"  # which kernels get the per-dispatch detail


def kernel_trace(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    print(f"## kernel trace: {db}")
    print(f"{'calls':>6} {'total_ms':>12} {'avg_ms':>12} {'pct':>7}  kernel")  # top_kernels holds microseconds
    for name, calls, total, avg, pct in cur.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"{calls:6d} {total / 1e3:12.3f} {avg / 1e3:12.3f} {pct:7.2f}  {short}")
    rows = cur.execute(
        "select name,grid_x,workgroup_x,lds_size,vgpr_count,accum_vgpr_count,sgpr_count,scratch_size,"
        "min(duration),avg(duration),max(duration),count(*) "
        "from kernels where name like ? group by name,grid_x,workgroup_x", (PAT,)).fetchall()
    for r in rows:
        print(f"# {r[0]}\n#   grid={r[1]} wg={r[2]} lds={r[3]} vgpr={r[4]} agpr={r[5]} sgpr={r[6]} scratch={r[7]} "
              f"dur_us min/avg/max = {r[8] / 1e3:.1f}/{r[9] / 1e3:.1f}/{r[10] / 1e3:.1f} over {r[11]} dispatches")


def pmc(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    print(f"## counters: {db}  (per-dispatch mean over the dispatches of {PAT})")
    for name, val, n in cur.execute(
            "select counter_name, avg(value), count(*) from counters_collection "
            "where kernel_name like ? group by counter_name", (PAT,)):
        print(f"{name:28s} {val:20.1f}   (n={n})")


if __name__ == "__main__":
    kernel_trace(sys.argv[1])
    for d in sys.argv[2:]:
        pmc(d)
