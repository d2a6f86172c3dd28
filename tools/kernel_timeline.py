"""Start / end of every dispatch of the kernels that match RH_PROF_KERNEL in a rocprofv3 kernel-trace database, in microseconds from the first
one: which kernels ran side by side.   python tools/kernel_timeline.py <kt_results.db> [max rows]"""
import os
import sqlite3
import sys

pat = "%" + os.environ.get("RH_PROF_KERNEL", "k_rlm") + "%"
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, stream_id, queue_id from kernels where name like ? order by start", (pat,)).fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = rows[0][1] if rows else 0
for name, a, b, sid, qid in rows[:n]:
    print(f"{(a - t0) / 1e3:10.1f} .. {(b - t0) / 1e3:10.1f} us  ({(b - a) / 1e3:7.1f})  stream {sid} queue {qid}  {name[:60]}")
