"""A multi-sub-batch job as a timeline of its matrix-pipe kernels (sweep 1 / sweep 2 of every sub-batch) from a rocprofv3
--kernel-trace results database: where the wall clock of a long job goes BETWEEN the sweeps.

    python tools/subbatch_timeline.py <results.db> [first sweep-1 launch to print, default: middle] [launches to print, default 6]

Prints (1) the whole job: span, sum of sweep-1 / sweep-2 durations, time with at least one sweep running, time with NO sweep
running ("matrix-idle") and which kernels ran during those gaps; (2) the listed sub-batches kernel by kernel."""
import collections
import sqlite3
import sys


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "msfm::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    if n.startswith("_ZN4msfm"):
        n = n[8:].lstrip("0123456789")[:24]
    return n[-40:]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = [(short(n), s, e) for n, s, e in rows]
    is_sweep = lambda n: n.startswith("sweep_i8_kernel") or n.startswith("sweep_kernel")
    is_s1 = lambda n: n in ("sweep_i8_kernel<1>", "sweep_kernel<1>")
    s1 = [i for i, r in enumerate(rows) if is_s1(r[0])]
    if len(s1) < 3:
        print("# fewer than 3 sweep-1 launches in the trace")
        return
    # the LAST run of consecutive sweep-1 launches closer than 0.5 s to one another = the job of interest
    job = [s1[-1]]
    for i in reversed(s1[:-1]):
        if rows[job[-1]][1] - rows[i][2] > 0.5e9:
            break
        job.append(i)
    job = job[::-1]
    t_begin = rows[job[0]][1]
    sweeps = [r for r in rows[job[0]:] if is_sweep(r[0])]
    t_end = max(r[2] for r in sweeps)
    span = (t_end - t_begin) / 1e6
    tot = collections.Counter()
    for n, s, e in rows[job[0]:]:
        if s < t_end:
            tot[n] += (min(e, t_end) - s) / 1e6
    # union of the sweeps' spans
    ivs = sorted((s, e) for n, s, e in sweeps)
    union, cur_s, cur_e, gaps = 0.0, ivs[0][0], ivs[0][1], []
    for s, e in ivs[1:]:
        if s > cur_e:
            union += (cur_e - cur_s) / 1e6
            gaps.append((cur_e, s))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += (cur_e - cur_s) / 1e6
    in_gap = collections.Counter()
    for n, s, e in rows[job[0]:]:
        if is_sweep(n):
            continue
        for gs, ge in gaps:
            o = min(e, ge) - max(s, gs)
            if o > 0:
                in_gap[n] += o / 1e6
    n_s1 = len(job)
    print("# job: %d sweep-1 launches, span %.2f ms (first sweep-1 start to last sweep end)" % (n_s1, span))
    print("#   sum of sweep-1 spans %.2f ms, of sweep-2 spans %.2f ms; at least one sweep running %.2f ms; NO sweep running %.2f ms in %d gaps (%.3f ms per sub-batch)"
          % (sum(v for k, v in tot.items() if is_s1(k)), sum(v for k, v in tot.items() if is_sweep(k) and not is_s1(k)), union,
             span - union, len(gaps), (span - union) / n_s1))
    print("#   kernels running during the no-sweep gaps (ms, summed): " + ", ".join("%s %.2f" % kv for kv in in_gap.most_common(8)))
    print("#   per kernel over the job (ms, sum of spans): " + ", ".join("%s %.1f" % kv for kv in tot.most_common(14)))
    first = int(sys.argv[2]) if len(sys.argv) > 2 else n_s1 // 2
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    first = max(0, min(first, n_s1 - 1))
    last = min(n_s1 - 1, first + count)
    t0 = rows[job[first]][1]
    print("# sub-batches %d .. %d of the job, times relative to the first one's sweep 1; small runtime fill / copy kernels summed per run" % (first, last - 1))
    print("# kernel                                      start ms     end ms      dur us   gap to the previous sweep's end (us)")
    prev_sweep_end = None
    pend_n, pend_c, pend_d, pend_s = None, 0, 0.0, 0
    def flush():
        nonlocal pend_n, pend_c, pend_d
        if pend_n:
            print("%-40s %10.3f %10s %11.1f   (x%d)" % (pend_n, (pend_s - t0) / 1e6, "", pend_d, pend_c))
        pend_n, pend_c, pend_d = None, 0, 0.0
    for n, s, e in rows[job[first]:job[last]]:
        if n.startswith("__amd_rocclr"):
            if pend_n != n:
                flush()
                pend_n, pend_s = n, s
            pend_c += 1
            pend_d += (e - s) / 1e3
            continue
        flush()
        gap = ""
        if is_sweep(n):
            if prev_sweep_end is not None:
                gap = "%.1f" % ((s - prev_sweep_end) / 1e3)
            prev_sweep_end = e if prev_sweep_end is None else max(prev_sweep_end, e)
        print("%-40s %10.3f %10.3f %11.1f   %s" % (n, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, gap))
    flush()


if __name__ == "__main__":
    main()
