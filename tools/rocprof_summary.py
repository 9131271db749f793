"""Turn a rocprofv3 results database (rocpd sqlite, --kernel-trace --stats) into the text summary
committed under profiles/.  Usage: python tools/rocprof_summary.py <results.db> [title]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    print("# rocprofv3 --kernel-trace --stats summary:", title)
    print("# columns: calls, total_s, avg_ms, pct, vgpr, agpr, sgpr, lds_bytes, scratch, grid, wg, name")
    meta = {}
    for r in db.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
                        "max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name"):
        meta[r[0]] = r[1:]
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage "
                                                    "from top_kernels order by total_duration desc"):
        m = meta.get(name, (None,) * 7)
        print("%6d %12.3f %12.3f %7.3f%%  v=%s a=%s s=%s lds=%s scr=%s grid=%s wg=%s  %s" % (
            calls, total / 1e6, avg / 1e3, pct, m[0], m[1], m[2], m[3], m[4], m[5], m[6], name))


if __name__ == "__main__":
    main()
