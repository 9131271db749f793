#!/bin/bash
# round 3 evidence in one GPU call: tests, fuzz (route Q forced on small images too), bench line, kernel stats + timeline,
# HBM traffic and SQ counter passes, the route table, CLI end to end, config 4 in full.  Copy what should be judged into profiles/.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
MSFM_Q8=2 timeout 600 python tools/fuzz_routes.py 11 500 > $OUT/fuzz_q8.txt 2>&1; echo "fuzz q8 rc=$?"; tail -3 $OUT/fuzz_q8.txt
timeout 400 python tools/fuzz_routes.py 12 300 > $OUT/fuzz_default.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_default.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.json
MSFM_Q8=0 timeout 300 python bench.py --no-cpu-baseline --sustained-steps 0 > $OUT/bench_fp16_route.json 2> $OUT/bench_fp16.err; echo "bench fp16 rc=$?"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0"
cd /tmp
rm -rf $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $ROOT
DB=$(find $OUT/prof_stats -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/kernel_stats.txt 2>&1; head -16 $OUT/kernel_stats.txt | cut -c1-170
python tools/step_timeline.py "$DB" 4 > $OUT/step_timeline.txt 2>&1; tail -3 $OUT/step_timeline.txt | cut -c1-300
KERN="sweep_i8_kernel<1>,sweep_kernel<4>,sweep_kernel<3>,sweep_i8_kernel<3>,sweep_kernel<1>,pf_thresholds_kernel,pf_prune_q8_kernel,pf_exact_candidates_kernel"
python tools/pmc_summary.py $OUT/pmc_traffic.json "$KERN" $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.txt 2>&1; tail -5 $OUT/pmc_traffic.txt
python tools/pmc_summary.py $OUT/pmc_sq.json "sweep_i8_kernel<1>,sweep_kernel<4>,sweep_kernel<3>,sweep_i8_kernel<3>,sweep_kernel<1>" $OUT/pmc_sq > $OUT/pmc_sq.txt 2>&1; tail -5 $OUT/pmc_sq.txt
find $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq -type f -size +8M -delete
timeout 600 python tools/configs_table.py > $OUT/configs.txt 2>&1; echo "configs rc=$?"; cut -c1-260 $OUT/configs.txt
timeout 600 python tools/cli_e2e_bench.py > $OUT/cli_e2e.txt 2>&1; echo "e2e rc=$?"; tail -8 $OUT/cli_e2e.txt
timeout 900 python tools/config4_full.py --int-oracle-pairs 1 > $OUT/config4_full.json 2> $OUT/config4_full.err; echo "config4 rc=$?"; head -c 900 $OUT/config4_full.json
