#!/bin/bash
# round 3, final evidence in one GPU call.  Copy what should be judged into profiles/.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
MSFM_Q8=2 timeout 600 python tools/fuzz_routes.py 601 1000 > $OUT/fuzz_q8.txt 2>&1; echo "fuzz q8 rc=$?"; tail -1 $OUT/fuzz_q8.txt
MSFM_Q8=2 MSFM_Q8_DIRECT=2 timeout 600 python tools/fuzz_routes.py 602 600 > $OUT/fuzz_q8_direct.txt 2>&1; echo "fuzz q8 direct rc=$?"; tail -1 $OUT/fuzz_q8_direct.txt
MSFM_Q8=2 MSFM_Q8_DIRECT=0 timeout 600 python tools/fuzz_routes.py 603 400 > $OUT/fuzz_q8_refine.txt 2>&1; echo "fuzz q8 refine rc=$?"; tail -1 $OUT/fuzz_q8_refine.txt
timeout 600 python tools/fuzz_routes.py 604 1000 > $OUT/fuzz_default.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_default.txt
timeout 600 python tools/fuzz_oracle.py 605 300 > $OUT/fuzz_oracle.txt 2>&1; echo "fuzz oracle rc=$?"; tail -1 $OUT/fuzz_oracle.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.json
MSFM_Q8_DIRECT=0 timeout 300 python bench.py --no-cpu-baseline --sustained-steps 0 --u8-images 0 > $OUT/bench_refine.json 2> $OUT/bench_refine.err; echo "bench refine rc=$?"
MSFM_Q8=0 timeout 300 python bench.py --no-cpu-baseline --sustained-steps 0 --u8-images 0 > $OUT/bench_fp16_route.json 2> $OUT/bench_fp16.err; echo "bench fp16 rc=$?"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0 --no-solo"
FULL="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --no-solo"
cd /tmp
rm -rf $OUT/prof_stats $OUT/prof_stats_p1 $OUT/prof_stats_full $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
MSFM_PIPELINE=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_p1 -- $BENCH > $OUT/prof_stats_p1.log 2>&1; echo "stats p1 rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_full -- $FULL > $OUT/prof_stats_full.log 2>&1; echo "stats full rc=$?"
MSFM_PIPELINE=1 timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
MSFM_PIPELINE=1 timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
MSFM_PIPELINE=1 timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $ROOT
DB=$(ls -t $(find $OUT/prof_stats -name '*.db') | head -1)
python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/kernel_stats.txt 2>&1; head -8 $OUT/kernel_stats.txt | cut -c1-150
python tools/step_timeline.py "$DB" 6 > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt | cut -c1-300
DB1=$(ls -t $(find $OUT/prof_stats_p1 -name '*.db') | head -1)
python tools/rocprof_summary.py "$DB1" "MSFM_PIPELINE=1 $BENCH" > $OUT/kernel_stats_p1.txt 2>&1; head -8 $OUT/kernel_stats_p1.txt | cut -c1-150
python tools/step_timeline.py "$DB1" 1 > $OUT/step_timeline_p1.txt 2>&1
DBF=$(ls -t $(find $OUT/prof_stats_full -name '*.db') | head -1)
python tools/rocprof_summary.py "$DBF" "$FULL" > $OUT/kernel_stats_full.txt 2>&1
KERN="sweep_i8_kernel<1>,sweep_kernel<3>,pf_prune_q8_kernel,pf_assign_kernel,pf_exact_candidates_kernel,pf_reduce_second_kernel,pf_finalize_kernel,epilogue_kernel"
python tools/pmc_summary.py $OUT/pmc_traffic.json "$KERN" $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.txt 2>&1; tail -3 $OUT/pmc_traffic.txt
python tools/pmc_summary.py $OUT/pmc_sq.json "sweep_i8_kernel<1>,sweep_kernel<3>" $OUT/pmc_sq > $OUT/pmc_sq.txt 2>&1; tail -3 $OUT/pmc_sq.txt
find $OUT/prof_stats $OUT/prof_stats_p1 $OUT/prof_stats_full $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq -type f -size +8M -delete
timeout 600 python tools/configs_table.py > $OUT/configs.txt 2>&1; echo "configs rc=$?"; cut -c1-220 $OUT/configs.txt
timeout 600 python tools/cli_e2e_bench.py > $OUT/cli_e2e.txt 2>&1; echo "e2e rc=$?"; tail -6 $OUT/cli_e2e.txt
timeout 900 python tools/config4_full.py --int-oracle-pairs 1 > $OUT/config4_full.json 2> $OUT/config4_full.err; echo "config4 rc=$?"; head -c 700 $OUT/config4_full.json
timeout 120 ./tools/ubench_coissue > $OUT/ubench_coissue.txt 2>&1; echo "ubench rc=$?"
