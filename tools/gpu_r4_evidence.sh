#!/bin/bash
# Round 4: the round's evidence in one GPU call (copy what should be judged into profiles/ as r04_*).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4ev; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4ev/bench.json")); r=d["roofline"]
print("ms_per_step", round(d["ms_per_step"],3), "value %.4g" % d["value"], "sustained", d["sustained_ms_per_step"], "frac", round(r["frac"],4), "solo", round(r["solo"]["frac"],4), "traffic/algo", r["traffic_over_algorithmic"])
print("strong_u8", {k: d["strong_u8"].get(k) for k in ("value","seconds_per_step","sub_batches_per_step_rank0","matches_per_step","error")}, d["strong_u8"].get("sweep1"))
PY
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0 --no-solo"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
MSFM_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_p1 -- $BENCH > $OUT/prof_stats_p1.log 2>&1; echo "stats p1 rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_u8_p1 -- python $ROOT/tools/job_ab.py --images 64 --pipeline 1 --warm 2 > $OUT/prof_u8_p1.log 2>&1; echo "stats u8 p1 rc=$?"
cd $ROOT
DB=$(ls -t $(find $OUT/prof_stats -name '*.db') | head -1)
python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/bench_kernel_stats.txt 2>&1; head -8 $OUT/bench_kernel_stats.txt | cut -c1-60,150-215
python tools/step_timeline.py "$DB" 2 > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt | cut -c1-300
DB1=$(ls -t $(find $OUT/prof_stats_p1 -name '*.db') | head -1)
python tools/rocprof_summary.py "$DB1" "MSFM_PIPELINE=1 $BENCH" > $OUT/bench_kernel_stats_pipeline1.txt 2>&1; head -12 $OUT/bench_kernel_stats_pipeline1.txt | cut -c1-60,150-215
DBU=$(ls -t $(find $OUT/prof_u8_p1 -name '*.db') | head -1)
python tools/rocprof_summary.py "$DBU" "tools/job_ab.py --images 64 --pipeline 1 (byte job, 2016 pairs of 8192-row images, one sub-batch)" > $OUT/u8_kernel_stats_pipeline1.txt 2>&1; head -14 $OUT/u8_kernel_stats_pipeline1.txt | cut -c1-60,150-215
KERN="sweep_i8_kernel<1>,sweep_kernel<3>,pf_prune_q8_kernel,pf_assign_kernel,pf_exact_candidates_kernel,pf_finalize_kernel,epilogue_kernel,fill_segs_kernel"
PMC_STEPS=5 python tools/pmc_summary.py $OUT/pmc_traffic_approx.json "$KERN" $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.txt 2>&1; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4ev/pmc_traffic_approx.json"))
for k,v in d.items():
    if k.startswith("_"): continue
    print(k, {c:(round(x.get("per_launch_KB_mean",0)*x["launches"]/5/1e6,3),"GB/step",x["launches"]) for c,x in v.items()})
PY
find $OUT -type f -size +8M -delete
timeout 900 python tools/config4_full.py --images 512 --desc 16384 --seed 4096 --warm --oracle-pairs 16 --int-oracle-pairs 1 > $OUT/config5_512.json 2> $OUT/config5_512.err; echo "config5 rc=$?"; head -c 900 $OUT/config5_512.json | tr '\n' ' '; echo
timeout 900 python tools/config4_full.py --warm --int-oracle-pairs 1 > $OUT/config4_full.json 2> $OUT/config4_full.err; echo "config4 rc=$?"; head -c 900 $OUT/config4_full.json | tr '\n' ' '; echo
MSFM_Q8=2 timeout 400 python tools/fuzz_routes.py 701 500 > $OUT/fuzz_q8.txt 2>&1; echo "fuzz q8 rc=$?"; tail -1 $OUT/fuzz_q8.txt
timeout 400 python tools/fuzz_routes.py 704 500 > $OUT/fuzz_default.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_default.txt
timeout 300 python tools/cli_e2e_bench.py > $OUT/cli_e2e.txt 2>&1; echo "e2e rc=$?"; tail -4 $OUT/cli_e2e.txt
