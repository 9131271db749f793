"""Every kernel / device copy of a traced process as one timeline (rocprofv3 --kernel-trace results database): start relative to the
first dispatch, duration, the idle gap before it -- the view of a COLD process (the ComputeMatches executable: one matching call).
Kernels shorter than `min_us` that follow another without a gap are folded into one line per run.
Usage: python tools/process_timeline.py <results.db> [min gap us to print, default 200]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
end = t0
busy = 0.0
print("# %d dispatches; first at 0" % len(rows))
print("# start ms | duration us | idle gap before us | kernel")
folded = 0
for name, s, e in rows:
    gap = (s - end) / 1e3
    dur = (e - s) / 1e3
    if e > end:
        busy += (e - max(s, end)) / 1e3
    short = name.split("(")[0][-60:]
    if gap >= min_gap or dur >= 500.0:
        if folded:
            print("#      ... %d short dispatches" % folded)
            folded = 0
        print("%10.3f %12.1f %12.1f   %s" % ((s - t0) / 1e6, dur, max(gap, 0.0), short))
    else:
        folded += 1
    end = max(end, e)
if folded:
    print("#      ... %d short dispatches" % folded)
print("# first dispatch to last completion %.3f ms; device busy %.3f ms" % ((end - t0) / 1e6, busy / 1e3))
