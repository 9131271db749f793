#!/bin/bash
# third order + clean PMC passes of the main job (no byte job mixed into the sweep_i8_kernel<1> averages)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
MSFM_Q8=2 timeout 400 python tools/fuzz_routes.py 21 300 > $OUT/fuzz_q8_o3.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_q8_o3.txt
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0"
cd /tmp
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/prof_stats
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $ROOT
DB=$(ls -t $(find $OUT/prof_stats -name '*.db') | head -1)
python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/kernel_stats.txt 2>&1; head -12 $OUT/kernel_stats.txt | cut -c1-170
python tools/step_timeline.py "$DB" 4 > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt | cut -c1-300
KERN="sweep_i8_kernel<1>,sweep_kernel<4>,sweep_kernel<3>,pf_thresholds_kernel,pf_prune_q8_kernel,pf_exact_candidates_kernel,q8_scatter_kernel"
python tools/pmc_summary.py $OUT/pmc_traffic.json "$KERN" $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_sq.json "sweep_i8_kernel<1>,sweep_kernel<4>,sweep_kernel<3>" $OUT/pmc_sq > $OUT/pmc_sq.txt 2>&1
find $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq -type f -size +8M -delete
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_traffic.json"))
for k, v in d.items():
    f = v.get("FETCH_SIZE", {}).get("per_launch_KB_mean", 0); w = v.get("WRITE_SIZE", {}).get("per_launch_KB_mean", 0)
    print("%-32s fetch 2 x %.3f GB, write %.3f GB per launch (%s launches)" % (k, f * 1024 / 1e9, w * 1024 / 1e9, v.get("FETCH_SIZE", {}).get("launches")))
s = json.load(open("gpurun_out/pmc_sq.json"))
for k, v in s.items():
    wc = v["SQ_WAVE_CYCLES"]["per_launch_mean"]
    print("%-24s wave cycles %.3e, MFMA busy %.3e, VALU insts %.3e, wait-any %.2f of wave cycles" % (k, wc, v["SQ_VALU_MFMA_BUSY_CYCLES"]["per_launch_mean"], v["SQ_INSTS_VALU"]["per_launch_mean"], v["SQ_WAIT_ANY"]["per_launch_mean"] / wc))
PY
