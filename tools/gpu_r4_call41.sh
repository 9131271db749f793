#!/bin/bash
# Round 4, call 41: kernel trace of the bench job with one outlier image (mixed sub-batches): where the first sweeps' time goes
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rm -rf /tmp/mx
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/mx -o run -- python tools/mixed_store_ab.py 128 > /tmp/mx.log 2>&1; echo "rc=$?"
python - $(find /tmp/mx -name '*.db' | head -1) <<'PY' | tee $OUT/r4_mixed_trace.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
sw = [(n.split("(")[0][-40:], (e - s) / 1e6) for n, s, e in rows if "sweep" in n]
print("# the last 16 sweep launches (name, ms):")
for n, d in sw[-16:]:
    print("%-42s %.3f" % (n, d))
PY
