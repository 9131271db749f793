#!/bin/bash
# One gpurun call: GPU tests, bench line, rocprofv3 kernel stats and the two PMC traffic passes of the same
# bench command.  Everything lands under gpurun_out/ (copy what should be judged into profiles/).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.json
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline"
cd /tmp
rm -rf $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
cd $ROOT
DB=$(find $OUT/prof_stats -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/kernel_stats.txt 2>&1; head -12 $OUT/kernel_stats.txt | cut -c1-170
python tools/pmc_summary.py $OUT/pmc_traffic.json "sweep_kernel<1>,sweep_kernel<3>,sweep_i8_kernel<1>,sweep_i8_kernel<3>,pf_thresholds_kernel,pf_exact_candidates_kernel" $OUT/pmc_fetch $OUT/pmc_write | tail -40
# keep the merge-back small
find $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write -type f -size +8M -delete
