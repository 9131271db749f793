#!/bin/bash
# kernel stats + timeline of the bench job (main job only), pipelined and unpipelined
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0"
cd /tmp
rm -rf $OUT/prof_stats $OUT/prof_stats_p1
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
MSFM_PIPELINE=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_p1 -- $BENCH > $OUT/prof_stats_p1.log 2>&1; echo "stats p1 rc=$?"
cd $ROOT
DB=$(ls -t $(find $OUT/prof_stats -name '*.db') | head -1)
python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/kernel_stats.txt 2>&1; head -24 $OUT/kernel_stats.txt | cut -c1-170
python tools/step_timeline.py "$DB" 4 > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt | cut -c1-300
DB1=$(ls -t $(find $OUT/prof_stats_p1 -name '*.db') | head -1)
python tools/rocprof_summary.py "$DB1" "MSFM_PIPELINE=1 $BENCH" > $OUT/kernel_stats_p1.txt 2>&1; head -24 $OUT/kernel_stats_p1.txt | cut -c1-170
python tools/step_timeline.py "$DB1" 1 > $OUT/step_timeline_p1.txt 2>&1; tail -1 $OUT/step_timeline_p1.txt | cut -c1-300
find $OUT/prof_stats $OUT/prof_stats_p1 -type f -size +8M -delete
