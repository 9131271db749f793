#!/bin/bash
# Round 4, call 27: the timing experiments of call 26 again, one process per build (tree, HALF_BARRIER, NO_BARRIER, FOLD_HALF, NO_DIGITS; the
# series twice), every launch of sweep_i8_kernel<1> listed per build from its own kernel trace: the full sweeps are the largest entries
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
for X in HALF_BARRIER NO_BARRIER FOLD_HALF NO_DIGITS; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_EXPERIMENT_$X -shared -o /tmp/libmsfm_$X.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error" &
done
wait
cp $ROOT/monocularsfm_amd/csrc/libmsfm_match.so /tmp/libmsfm_tree.so
cd /tmp && export TMPDIR=/tmp && cd $ROOT
: > $OUT/r4_s1exp2.txt
for job in "" "--u8"; do
 for rep in 1 2; do
  for X in tree HALF_BARRIER NO_BARRIER FOLD_HALF NO_DIGITS; do
    rm -rf /tmp/exp
    MSFM_LIBRARY=/tmp/libmsfm_$X.so timeout 200 rocprofv3 --kernel-trace -d /tmp/exp -o run -- python tools/s1_launches.py $job > /tmp/exp.log 2>&1 || echo "rc=$? $X"
    DB=$(find /tmp/exp -name '*.db' | head -1)
    python - "$DB" "$X" "$job" <<'PY' | tee -a $OUT/r4_s1exp2.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
s1 = [(e - s) / 1e6 for n, s, e in rows if "sweep_i8_kernel<1>" in n or "sweep_i8_kernelILi1" in n]
big = sorted(d for d in s1 if d > 0.6 * max(s1))
print("%-4s %-13s full sweeps: median %.3f ms of %d | all launches: %s" % (sys.argv[3] or "f32", sys.argv[2], big[len(big) // 2], len(big), " ".join("%.3f" % d for d in s1)))
PY
  done
 done
done
