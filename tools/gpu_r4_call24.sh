#!/bin/bash
# Round 4, call 24: sweep 1 without its fifth k-step again (timing experiment, wrong results).  The event sums of call 23 mix the sweep
# over all pairs with the re-runs a flooded sub-batch triggers: here every launch of sweep_i8_kernel<1> is listed from a kernel trace
# (the largest grid-filling launches are the sweeps over all pairs).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_EXPERIMENT_NO_DIGITS -shared -o /tmp/libmsfm_nodigits.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error"
cd $ROOT
for job in "" "--u8"; do
  rm -rf /tmp/nodig; 
  timeout 300 rocprofv3 --kernel-trace -d /tmp/nodig -o run -- python tools/ab_multi.py $job --p1 --nocheck --images 40 --rounds 5 tree nodigits=/tmp/libmsfm_nodigits.so > $OUT/r4_nodigits3${job}.log 2>&1; echo "rc=$?"
  DB=$(find /tmp/nodig -name '*.db' | head -1)
  python - "$DB" <<'PY' | tee $OUT/r4_nodigits3${job}.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
s1 = [(s, (e - s) / 1e6) for n, s, e in rows if "sweep_i8_kernel<1>" in n or "sweep_i8_kernelILi1" in n]
print("# %d launches of sweep_i8_kernel<1>; durations in ms in launch order (contexts alternate: tree, nodigits, tree, ...):" % len(s1))
print(" ".join("%.3f" % d for _, d in s1))
PY
done
