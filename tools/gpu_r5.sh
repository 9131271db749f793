#!/bin/bash
# Round 5: the GPU calls of the round, one stage per `gpurun` call (each box is fresh; A/B comparisons happen inside one stage).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r5.sh <stage>'
# Output: gpurun_out/r5/<stage>/ ; what is judged is copied to profiles/r05_* (index: profiles/r05_README.md).
# tools/_ab/libmsfm_match_r04.so = the library of commit 73e894c (round 4's final build):
#   git worktree add /tmp/w 73e894c && make -C /tmp/w/monocularsfm_amd/csrc && cp /tmp/w/monocularsfm_amd/csrc/libmsfm_match.so tools/_ab/libmsfm_match_r04.so
# tools/_ab/libmsfm_match_head.so = likewise, commit c5c0334 (before the cold matching call was looked at: stage cliab)
set -u
STAGE=${1:-probe}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5/$STAGE; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
R04=tools/_ab/libmsfm_match_r04.so
run() { local name=$1; shift; local t0=$(date +%s); timeout ${TMO:-600} "$@" > $OUT/$name.txt 2>&1; echo "$name rc=$? ($(( $(date +%s) - t0 )) s)"; }

case $STAGE in
probe)   # what an upload can cost, the block-scaled fp6 instruction, parity of the exact-S-in-sweep byte route + its A/B against round 4
    run ubench_upload tools/ubench_upload; cat $OUT/ubench_upload.txt
    run ubench_fp6 tools/ubench_fp6; cat $OUT/ubench_fp6.txt
    TMO=900 run pytest_bytes python -m pytest -m gpu -x -q tests/test_gpu_i8.py tests/test_gpu_jobs.py tests/test_gpu_configs.py tests/test_gpu_certificate.py; tail -3 $OUT/pytest_bytes.txt
    run fuzz_bytes python tools/fuzz_routes.py 811 250; tail -2 $OUT/fuzz_bytes.txt
    run ab_u8 python tools/ab.py --u8 --images 64 r04=$R04 tree; cat $OUT/ab_u8.txt
    run ab_u8_p1 python tools/ab.py --u8 --images 64 --p1 r04=$R04 tree; cat $OUT/ab_u8_p1.txt
    ;;
malloc)  # device / page-locked allocation cost by size
    run ubench_malloc tools/ubench_malloc; cat $OUT/ubench_malloc.txt
    ;;
store)   # the rebuilt store: whole GPU suite, fuzz, the bench line's upload figure, the CLI end to end
    TMO=1500 run pytest_gpu python -m pytest tests -m gpu -x -q; tail -5 $OUT/pytest_gpu.txt
    run fuzz_routes python tools/fuzz_routes.py 821 400; tail -2 $OUT/fuzz_routes.txt
    MSFM_Q8=2 run fuzz_routes_q8 python tools/fuzz_routes.py 822 300; tail -2 $OUT/fuzz_routes_q8.txt
    run fuzz_jobs python tools/fuzz_jobs.py 823 150; tail -2 $OUT/fuzz_jobs.txt
    run bench python bench.py --u8-images 0 --steps 10 --warmup 3 --sustained-steps 0; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/store/bench.txt").read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"],3), "upload_ms", round(d["pcie_inclusive"]["upload_ms"],2), "matches", d["config"]["matches_per_step"], "checksum", d["exchange_checksum"], "gpu/cpu", round(d.get("gpu_over_cpu",0)))
PY
    run cli_e2e python tools/cli_e2e_bench.py; tail -6 $OUT/cli_e2e.txt
    ;;
newtests)   # the tests added in round 5
    TMO=900 run pytest_new python -m pytest -m gpu -x -q tests/test_gpu_stream.py tests/test_gpu_store.py; tail -15 $OUT/pytest_new.txt
    ;;
contract)   # bench.py's contract tests (end_to_end block included) + the two-rank flows
    TMO=1500 run pytest_contract python -m pytest -m gpu -x -q tests/test_bench_contract.py tests/test_gpu_exchange.py; tail -15 $OUT/pytest_contract.txt
    ;;
cfg4stream)   # config 4 in full through the streaming form, cold: no call-wide page-locked result buffer
    TMO=1500 run config4_stream python tools/config4_full.py --stream --int-oracle-pairs 1; head -c 700 $OUT/config4_stream.txt; echo; grep -n "GiB\|mismatch" $OUT/config4_stream.txt
    ;;
stream)   # bounded memory: config 5 at 1024 of its 4096 images through msfm_match_pairs_begin / _next
    TMO=1500 run config5_1024_stream python tools/config4_full.py --images 1024 --desc 16384 --seed 4096 --stream --oracle-pairs 24 --int-oracle-pairs 1; head -c 1500 $OUT/config5_1024_stream.txt; echo
    ;;
cfg4)   # config 4 in full, one call, cold then warm (first call <= 8.3 s, warm <= 7.6 s, store <= 0.4 KB per row)
    TMO=1500 run config4_full python tools/config4_full.py --warm --int-oracle-pairs 1; head -c 1200 $OUT/config4_full.txt; echo; grep -n "store_\|cold_first" $OUT/config4_full.txt
    ;;
tail)   # the exact re-check at 64 VGPRs (co-resident with sweep 1) under 2 / 3 / 4 parts; against round 4's build
    run ab_parts python tools/ab.py --rounds 14 tree "@MSFM_PIPELINE=3" "@MSFM_PIPELINE=4" "@MSFM_PIPELINE=3,MSFM_PIPELINE_TAPER=0.5" "@MSFM_PIPELINE=1"; cat $OUT/ab_parts.txt
    run ab_r04 python tools/ab.py --rounds 14 r04=$R04 tree; cat $OUT/ab_r04.txt
    BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --sustained-steps 0 --u8-images 0 --no-solo"
    cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"; cd $ROOT
    DB=$(ls -t $(find $OUT/prof_stats -name '*.db') | head -1)
    python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/bench_kernel_stats.txt 2>&1; head -16 $OUT/bench_kernel_stats.txt | cut -c1-60,150-215
    python tools/step_timeline.py "$DB" 2 > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt | cut -c1-300
    find $OUT -type f -size +8M -delete
    ;;
cli)   # where the CLI's wall clock goes
    run cli_e2e python tools/cli_e2e_bench.py; head -3 $OUT/cli_e2e.txt
    ;;
evidence)   # the round's evidence in one call on one box (copy what is judged into profiles/ as r05_*)
    TMO=1500 run pytest_gpu python -m pytest tests -m gpu -q; tail -2 $OUT/pytest_gpu.txt
    TMO=900 run bench python bench.py; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5/evidence/bench.txt") if l.startswith("{")][-1]); r=d["roofline"]
json.dump(d, open("gpurun_out/r5/evidence/bench.json","w"), indent=1)
print("ms_per_step", round(d["ms_per_step"],3), "value %.4g" % d["value"], "sustained", d["sustained_ms_per_step"], "frac", round(r["frac"],4), "solo", round(r["solo"]["frac"],4), "upload_ms", round(d["pcie_inclusive"]["upload_ms"],2))
print("strong_u8", {k: d["strong_u8"].get(k) for k in ("value","seconds_per_step","sub_batches_per_step_rank0","matches_per_step","error")})
print("end_to_end", {k: d["end_to_end"].get(k) for k in ("wall_s","phases_s","rows_written","ratio","ratio_vs_single_thread","error")})
PY
    BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --sustained-steps 0 --u8-images 0 --no-solo"
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
    KERN="sweep_i8_kernel<1>,sweep_kernel<3>,pf_prune_q8_kernel,pf_assign_kernel,pf_exact_candidates_kernel,epilogue_kernel,fill_segs_kernel,st_float_kernel,st_i8_kernel,st_classify_kernel"
    PMC_STEPS=5 python tools/pmc_summary.py $OUT/pmc_traffic_approx.json "$KERN" $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.txt 2>&1; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5/evidence/pmc_traffic_approx.json"))
for k,v in d.items():
    if k.startswith("_"): continue
    print(k, {c:(round(x.get("per_launch_KB_mean",0)*x["launches"]/5/1e6,3),"GB/step",x["launches"]) for c,x in v.items()})
PY
    find $OUT -type f -size +8M -delete
    run cli_e2e python tools/cli_e2e_bench.py; head -3 $OUT/cli_e2e.txt
    run upload_timing python tools/upload_timing.py; cat $OUT/upload_timing.txt
    MSFM_DEBUG_TIMING=1 run cold_call_u8 python tools/cold_call.py u8 400; grep "context\|alloc\]" $OUT/cold_call_u8.txt | head -12
    MSFM_DEBUG_TIMING=1 run cold_call_sb python tools/cold_call.py south-building; grep "context\|alloc\]" $OUT/cold_call_sb.txt | head -12
    run two_runtimes python tools/two_runtimes_probe.py monocularsfm_amd/csrc/libmsfm_match.so; tail -2 $OUT/two_runtimes.txt
    MSFM_Q8=2 run fuzz_q8 python tools/fuzz_routes.py 901 1200; tail -1 $OUT/fuzz_q8.txt
    run fuzz_default python tools/fuzz_routes.py 904 1500; tail -1 $OUT/fuzz_default.txt
    run fuzz_jobs python tools/fuzz_jobs.py 905 600; tail -1 $OUT/fuzz_jobs.txt
    ;;
fuzz)   # long seeded fuzz on the final build: routes, jobs (incl. the streaming form and store rebuilds), verification
    TMO=900 run fuzz_jobs python tools/fuzz_jobs.py 951 2500; tail -1 $OUT/fuzz_jobs.txt
    TMO=600 MSFM_Q8=2 run fuzz_q8 python tools/fuzz_routes.py 952 3000; tail -1 $OUT/fuzz_q8.txt
    TMO=600 run fuzz_default python tools/fuzz_routes.py 953 3000; tail -1 $OUT/fuzz_default.txt
    TMO=600 MSFM_Q8=2 MSFM_Q8_DIRECT=0 run fuzz_refine python tools/fuzz_routes.py 954 1000; tail -1 $OUT/fuzz_refine.txt
    TMO=300 run fuzz_verify python tools/fuzz_verify.py 955 300; tail -1 $OUT/fuzz_verify.txt
    ;;
final)   # the final tree once more: the GPU suite, the default bench line, the kernel statistics of the timed region
    TMO=1500 run pytest_gpu python -m pytest tests -m gpu -q; tail -2 $OUT/pytest_gpu.txt
    TMO=900 run bench python bench.py; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5/final/bench.txt") if l.startswith("{")][-1]); r=d["roofline"]
json.dump(d, open("gpurun_out/r5/final/bench.json","w"), indent=1)
print("ms_per_step", round(d["ms_per_step"],3), "value %.4g" % d["value"], "sustained", d["sustained_ms_per_step"], "frac", round(r["frac"],4), "solo", round(r["solo"]["frac"],4), "upload_ms", round(d["pcie_inclusive"]["upload_ms"],2), "traffic_source", r.get("traffic_source"))
print("strong_u8", {k: d["strong_u8"].get(k) for k in ("value","seconds_per_step","matches_per_step","error")})
print("end_to_end", {k: d["end_to_end"].get(k) for k in ("wall_s","walls_s","phases_s","rows_written","ratio","error")})
PY
    BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --sustained-steps 0 --u8-images 0 --no-solo"
    cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"; cd $ROOT
    DB=$(ls -t $(find $OUT/prof_stats -name '*.db') | head -1)
    python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/bench_kernel_stats.txt 2>&1; head -8 $OUT/bench_kernel_stats.txt | cut -c1-60,150-215
    find $OUT -type f -size +8M -delete
    ;;
clidbg)   # where a COLD process's matching call goes: the executable under MSFM_DEBUG_TIMING and under a kernel trace
    python - <<'PY' > $OUT/setup.txt 2>&1
import os, sys
sys.path.insert(0, os.getcwd())
from monocularsfm_amd import synth
import shutil
shutil.rmtree("/tmp/clidbg", ignore_errors=True)
os.makedirs("/tmp/clidbg")
synth.south_building_database("/tmp/clidbg/sb.db", 128, 5000, seed=1234)
open("/tmp/clidbg/cfg.yaml", "w").write('%YAML:1.0\ndatabase_path : "/tmp/clidbg/sb.db"\nSIFTmatch.match_type : 1\n')
PY
    EXE=$ROOT/monocularsfm_amd/host/ComputeMatches
    for k in 1 2; do cp /tmp/clidbg/sb.db /tmp/clidbg/run.db; sed 's/sb.db/run.db/' /tmp/clidbg/cfg.yaml > /tmp/clidbg/run.yaml
        MSFM_CLI_TIMING=1 MSFM_DEBUG_TIMING=1 $EXE /tmp/clidbg/run.yaml > $OUT/cli_$k.out 2> $OUT/cli_$k.err; echo "cli $k rc=$?"; done
    grep -v "^\[msfm alloc\] regrow" $OUT/cli_2.err | tail -60
    cp /tmp/clidbg/sb.db /tmp/clidbg/run.db
    cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -- $EXE /tmp/clidbg/run.yaml > $OUT/prof.log 2>&1; echo "trace rc=$?"; cd $ROOT
    DB=$(ls -t $(find $OUT/prof -name '*.db') | head -1)
    python tools/process_timeline.py "$DB" 150 > $OUT/cli_process_timeline.txt 2>&1; tail -70 $OUT/cli_process_timeline.txt | cut -c1-140
    find $OUT -type f -size +8M -delete
    ;;
cliab)   # the executable, cold, this tree against the library of an earlier commit (tools/_ab/libmsfm_match_head.so: see the header), alternating
    python - <<'PY' > $OUT/setup.txt 2>&1
import os, sys
sys.path.insert(0, os.getcwd())
from monocularsfm_amd import synth
import shutil
shutil.rmtree("/tmp/clidbg", ignore_errors=True)
os.makedirs("/tmp/clidbg")
synth.south_building_database("/tmp/clidbg/sb.db", 128, 5000, seed=1234)
open("/tmp/clidbg/run.yaml", "w").write('%YAML:1.0\ndatabase_path : "/tmp/clidbg/run.db"\nSIFTmatch.match_type : 1\n')
PY
    EXE=$ROOT/monocularsfm_amd/host/ComputeMatches
    for round in 1 2 3 4 5 6; do for v in tree before; do
        cp /tmp/clidbg/sb.db /tmp/clidbg/run.db; sleep 1
        t0=$(date +%s.%N)
        if [ $v = before ]; then LD_PRELOAD=$ROOT/tools/_ab/libmsfm_match_head.so MSFM_CLI_TIMING=1 $EXE /tmp/clidbg/run.yaml > /dev/null 2> /tmp/clidbg/err.txt
        else MSFM_CLI_TIMING=1 $EXE /tmp/clidbg/run.yaml > /dev/null 2> /tmp/clidbg/err.txt; fi
        t1=$(date +%s.%N)
        echo "$v wall $(python -c "print('%.3f' % ($t1 - $t0))") $(grep 'msfm timing' /tmp/clidbg/err.txt | sed 's/\[msfm timing\] //')"
    done; done | tee $OUT/cli_ab.txt
    ;;
settle)   # HIP start-up against the time since the previous GPU process exited
    run hip_init_settle bash tools/hip_init_settle.sh; cat $OUT/hip_init_settle.txt
    ;;
*) echo "unknown stage $STAGE"; exit 2;;
esac
