#!/bin/bash
# Round 4, call 26: timing experiments on sweep 1 (wrong results; every launch listed from a kernel trace, contexts alternated): every second
# workgroup barrier dropped (HALF_BARRIER), all of them (NO_BARRIER), half of the epilogue's v_max3 (FOLD_HALF: 8 + 8 instead of 16 + 16)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
LIBS="tree"
for X in HALF_BARRIER NO_BARRIER FOLD_HALF; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_EXPERIMENT_$X -shared -o /tmp/libmsfm_$X.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error" &
  LIBS="$LIBS $X=/tmp/libmsfm_$X.so"
done
wait
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for job in "" "--u8"; do
  rm -rf /tmp/exp
  timeout 400 rocprofv3 --kernel-trace -d /tmp/exp -o run -- python tools/ab_multi.py $job --p1 --nocheck --images 40 --rounds 5 $LIBS > $OUT/r4_s1exp${job}.log 2>&1; echo "rc=$?"
  DB=$(find /tmp/exp -name '*.db' | head -1)
  python - "$DB" <<'PY' | tee $OUT/r4_s1exp${job}.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
s1 = [(e - s) / 1e6 for n, s, e in rows if "sweep_i8_kernel<1>" in n or "sweep_i8_kernelILi1" in n]
print("# %d launches of sweep_i8_kernel<1>; ms in launch order (contexts alternate: tree, HALF_BARRIER, NO_BARRIER, FOLD_HALF; a flooded call re-runs a smaller sweep):" % len(s1))
print(" ".join("%.3f" % d for d in s1))
PY
done
