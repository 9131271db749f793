"""Turn a rocprofv3 results database (rocpd sqlite, --kernel-trace --stats) into the text summary
committed under profiles/.  Usage: python tools/rocprof_summary.py <results.db> [title]"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    print("# rocprofv3 --kernel-trace --stats summary:", title)
    print("# columns: calls, total_s, avg_ms, pct, isa=<vgpr>/<vgpr spills> (the code object's own metadata, "
          "tools/kernel_resources.py), pkt_v / a / s = the dispatch record's register fields as rocprofv3 stores them "
          "(allocation-granule units, NOT the ISA's register count), lds_bytes, scratch, grid, wg, name")
    try:
        import kernel_resources
        isa = kernel_resources.resources()
    except Exception as e:  # the summary is still useful without the library next to it
        print("# (no code-object metadata: %s)" % e)
        isa = {}
    by_demangled = {}
    if isa:
        for n, d in zip(sorted(isa), kernel_resources.demangle(sorted(isa))):
            by_demangled[d.split("(")[0]] = isa[n]
            by_demangled[n] = isa[n]
    meta = {}
    for r in db.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
                        "max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name"):
        meta[r[0]] = r[1:]
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage "
                                                    "from top_kernels order by total_duration desc"):
        m = meta.get(name, (None,) * 7)
        r = by_demangled.get(name.split("(")[0])
        isa_s = "%d/%d" % (r.get("vgpr_count", -1), r.get("vgpr_spill_count", 0)) if r else "-"
        print("%6d %12.3f %12.3f %7.3f%%  isa=%s pkt_v=%s a=%s s=%s lds=%s scr=%s grid=%s wg=%s  %s" % (
            calls, total / 1e6, avg / 1e3, pct, isa_s, m[0], m[1], m[2], m[3], m[4], m[5], m[6], name))


if __name__ == "__main__":
    main()
