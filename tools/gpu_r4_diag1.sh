#!/bin/bash
# Round 4, call 1 (no code changes yet): where does the wall clock of a many-sub-batch byte job go beyond sweep 1?
#   (a) baseline bench line of the round on this box; (b) kernel trace of a 400-image config-4-shaped job -> sub-batch timeline;
#   (c) the same job under different scratch budgets (sub-batch sizes), alternated on this box.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python bench.py > $OUT/r4_bench0.json 2> $OUT/r4_bench0.err; echo "bench rc=$?"; head -c 400 $OUT/r4_bench0.json; echo
cd /tmp; rm -rf $OUT/prof_c4
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c4 -- python $ROOT/tools/job_ab.py --images 400 --scratch-gib 48 --warm 1 > $OUT/r4_c4_trace.log 2>&1; echo "trace rc=$?"
cd $ROOT
DB=$(find $OUT/prof_c4 -name '*.db' | head -1)
python tools/subbatch_timeline.py "$DB" 8 4 > $OUT/r4_subbatch_timeline_400.txt 2>&1; head -8 $OUT/r4_subbatch_timeline_400.txt | cut -c1-400
python tools/rocprof_summary.py "$DB" "job_ab --images 400" > $OUT/r4_c4_400_kernel_stats.txt 2>&1
timeout 400 python tools/job_ab.py --images 400 --scratch-gib 48,144,24,48,144,24 > $OUT/r4_scratch_ab.txt 2> $OUT/r4_scratch_ab.err; echo "ab rc=$?"; cat $OUT/r4_scratch_ab.txt | cut -c1-330
MSFM_DEBUG_TIMING=1 timeout 200 python tools/job_ab.py --images 200 --scratch-gib 48 --warm 1 > $OUT/r4_host_timing.txt 2>&1; tail -60 $OUT/r4_host_timing.txt | cut -c1-200
find $OUT/prof_c4 -type f -size +30M -delete
