"""One step of the bench job as a timeline: every kernel / device copy of one step (`launches` consecutive sweep-1
launches: the step is cut into that many sub-batches on two streams) of a rocprofv3 --kernel-trace results database,
with start / end relative to the step's first sweep 1, duration, and what else was running when it started.
Usage: python tools/step_timeline.py <results.db> [sweep-1 launches per step, default 4]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# sweep 1 of the main (float) job: sweep_kernel<1> on the fp16 route, sweep_i8_kernel<1> on route Q (followed by sweep_kernel<4>
# when the twins are coarse).  Run the bench with --u8-images 0: the byte job's sweeps have the same name.
idx = [i for i, r in enumerate(rows) if "sweep_kernel<1>" in r[0] and "i8" not in r[0]]
if len(idx) <= per_step:
    idx = [i for i, r in enumerate(rows) if "sweep_i8_kernel<1>" in r[0] and any("sweep_kernel<4>" in rows[j][0] for j in range(i, min(i + 150, len(rows))))]
if len(idx) <= per_step:
    idx = [i for i, r in enumerate(rows) if "sweep_i8_kernel<1>" in r[0]]
i0, i1 = idx[-1 - per_step], idx[-1]
t0 = rows[i0][1]
print("# kernel (or runtime copy / fill kernel)             start ms     end ms   duration us   running at its start")
busy_until = []
tot = 0.0
union_end, union = t0, 0.0
for r in rows[i0:i1]:
    name = r[0].split("(")[0][-44:]
    live = [n for n, e in busy_until if e > r[1]]
    print("%-46s %10.3f %10.3f %12.1f   %s" % (name, (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e3,
                                              ", ".join(sorted(set(x.split("<")[0].split("::")[-1] + ("<" + x.split("<")[1][:2] if "<" in x else "") for x in live))) or "-"))
    busy_until.append((name, r[2]))
    tot += (r[2] - r[1]) / 1e3
    s = max(r[1], union_end)
    if r[2] > s:
        union += (r[2] - s) / 1e3
        union_end = r[2]
step = (rows[i1][1] - t0) / 1e6
print("# first sweep-1 launch of the step to the first of the next: %.3f ms; sum of kernel durations %.3f ms; time with at "
      "least one kernel running %.3f ms (overlap %.3f ms; idle %.3f ms)" % (step, tot / 1e3, union / 1e3, (tot - union) / 1e3, step - union / 1e3))
