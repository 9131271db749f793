"""One step of the bench job as a timeline: every kernel / device copy between two sweep-1 launches of a rocprofv3
--kernel-trace results database, with its duration and the gap before it.  Usage: python tools/step_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "sweep_kernel<1>" in r[0] and "i8" not in r[0]]
i0, i1 = idx[-2], idx[-1]
prev_end, tot_gap, tot = None, 0.0, 0.0
print("# kernel (or runtime copy / fill kernel)            duration us   gap before us")
for r in rows[i0:i1]:
    name = r[0].split("(")[0]
    name = name[-44:]
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0.0
    tot_gap += max(gap, 0.0)
    tot += (r[2] - r[1]) / 1e3
    print("%-46s %12.1f %12.1f" % (name, (r[2] - r[1]) / 1e3, gap))
    prev_end = r[2]
print("# sweep-1 launch to sweep-1 launch: %.3f ms; kernels %.3f ms; gaps %.3f ms (the two large ones at the end are the "
      "device-to-host copy of the lists on the SDMA engine and the host's setup of the next call)" % (
          (rows[i1][1] - rows[i0][1]) / 1e6, tot / 1e3, tot_gap / 1e3))
