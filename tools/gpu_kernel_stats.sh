#!/bin/bash
# rocprofv3 kernel-trace of the bench command (3 steps) -> gpurun_out/kernel_stats.txt ; host laps of one call
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --u8-images 0"
cd /tmp; rm -rf $OUT/prof_stats
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
cd $ROOT
DB=$(find $OUT/prof_stats -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/kernel_stats.txt 2>&1; head -30 $OUT/kernel_stats.txt | cut -c1-200
find $OUT/prof_stats -type f -size +8M -delete
MSFM_DEBUG_TIMING=1 timeout 200 python tools/step_breakdown.py 2>&1 | tail -24
