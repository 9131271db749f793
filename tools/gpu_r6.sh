#!/bin/bash
# Round 6: the GPU calls of the round, one stage per `gpurun` call (each box is fresh; comparisons happen inside one stage).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r6.sh <stage>'
# Output: gpurun_out/r6/<stage>/ ; what is judged is copied to profiles/r06_* (index: profiles/r06_README.md).
set -u
STAGE=${1:-toggle}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6/$STAGE; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift; local t0=$(date +%s); timeout ${TMO:-600} "$@" > $OUT/$name.txt 2>$OUT/$name.err; echo "$name rc=$? ($(( $(date +%s) - t0 )) s)"; }

case $STAGE in
toggle)   # what the operand encoding costs the power-limited sweep 1
    run operand_toggle python tools/operand_toggle.py; cat $OUT/operand_toggle.txt; tail -3 $OUT/operand_toggle.err
    ;;
cfg5full)   # BASELINE configs[4] IN FULL, streamed, one GPU
    free -g | head -2
    TMO=1700 run config5_full_stream python tools/config4_full.py --stream --images 4096 --desc 16384 --seed 4096 --oracle-pairs 24 --int-oracle-pairs 1 --cut-every 100
    head -c 1800 $OUT/config5_full_stream.txt; echo; grep -n "GiB\|mismatch" $OUT/config5_full_stream.txt; tail -5 $OUT/config5_full_stream.err
    ;;
cli)   # the pipelined executable: its tests, the small end-to-end job, then the config-4-shaped database
    TMO=900 run pytest_cli python -m pytest -m gpu -x -q tests/test_cli_gpu.py tests/test_gpu_stream.py tests/test_abi.py; tail -4 $OUT/pytest_cli.txt
    run cli_e2e python tools/cli_e2e_bench.py; tail -8 $OUT/cli_e2e.txt
    TMO=1500 run cli_config4 python tools/cli_e2e_bench.py --config4 --tables ${TABLES:-u8,f32} --json $OUT/cli_config4.json; cat $OUT/cli_config4.txt; tail -5 $OUT/cli_config4.err
    ;;
suite)   # the whole GPU suite
    TMO=1800 run pytest_gpu python -m pytest tests -m gpu -x -q; tail -5 $OUT/pytest_gpu.txt
    ;;
nccl1)   # the exchange path on ONE rank over RCCL, config 4 in full -- the library's copy into the torch-allocated
         # send tensor (the cross-runtime pointer) timed, and the streamed form with the exchange pipelined under the next super-batch
    TMO=900 run bench_nccl1 python bench.py --backend nccl --force-collectives --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --sustained-steps 0 --no-solo
    python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6/nccl1/bench_nccl1.txt") if l.startswith("{")][-1])
print("config 2:", round(d["ms_per_step"],2), "ms/step", d["per_rank_ms"], d.get("exchange_detail_rank0"))
u=d.get("strong_u8",{}); print("config 4:", u.get("seconds_per_step"), "s/step", u.get("per_rank_ms"), u.get("exchange_detail_rank0"))
PY
    TMO=900 run bench_nccl1_stream python bench.py --backend nccl --force-collectives --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --sustained-steps 0 --no-solo --super-batch-pairs 65536
    python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6/nccl1/bench_nccl1_stream.txt") if l.startswith("{")][-1])
print("config 2 streamed:", round(d["ms_per_step"],2), "ms/step", d["per_rank_ms"], d.get("exchange_detail_rank0"))
u=d.get("strong_u8",{}); print("config 4 streamed:", u.get("seconds_per_step"), "s/step", u.get("per_rank_ms"), u.get("exchange_detail_rank0"))
PY
    ;;
multi)   # the executable's device fan-out at config-4 scale with SEVERAL CONTEXTS ON THE ONE GPU (MSFM_DEVICES=0,0 / 0,0,0,0): same rows,
         # and what the phases that do not shrink with the device count cost (bulk load into G stores, contexts, pre-emptive filter)
    TMO=900 run cli_config4_2ctx python tools/cli_e2e_bench.py --config4 --tables u8 --modes off --devices 0,0 --json $OUT/cli_config4_2ctx.json; cat $OUT/cli_config4_2ctx.txt; tail -3 $OUT/cli_config4_2ctx.err
    TMO=900 run cli_config4_4ctx python tools/cli_e2e_bench.py --config4 --tables u8 --modes off --orders pair_id --devices 0,0,0,0 --json $OUT/cli_config4_4ctx.json; cat $OUT/cli_config4_4ctx.txt; tail -3 $OUT/cli_config4_4ctx.err
    ;;
evidence)   # the round's evidence in one call on one box (copy what is judged into profiles/ as r06_*)
    TMO=1500 run pytest_gpu python -m pytest tests -m gpu -q; tail -2 $OUT/pytest_gpu.txt
    TMO=900 run bench python bench.py; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6/evidence/bench.txt") if l.startswith("{")][-1]); r=d["roofline"]
json.dump(d, open("gpurun_out/r6/evidence/bench.json","w"), indent=1)
print("ms_per_step", round(d["ms_per_step"],3), "value %.4g" % d["value"], "sustained", d["sustained_ms_per_step"], "frac", round(r["frac"],4), "solo", round(r["solo"]["frac"],4), "upload_ms", round(d["pcie_inclusive"]["upload_ms"],2), "traffic_source", r.get("traffic_source"))
print("strong_u8", {k: d["strong_u8"].get(k) for k in ("value","seconds_per_step","sub_batches_per_step_rank0","matches_per_step","error")})
print("end_to_end", {k: d["end_to_end"].get(k) for k in ("wall_s","walls_s","phases_s","rows_written","matches_written","matches_kept_by_verification","verification_off","ratio","ratio_vs_single_thread","error")})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value","fast_order_value","fast_order","cores","single_thread_value")}, "gpu_over_cpu", d.get("gpu_over_cpu"), d.get("gpu_over_cpu_fast_order"))
PY
    BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --sustained-steps 0 --u8-images 0 --no-solo"
    cd /tmp
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
    MSFM_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_p1 -- $BENCH > $OUT/prof_stats_p1.log 2>&1; echo "stats p1 rc=$?"
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
    cd $ROOT
    DB=$(ls -t $(find $OUT/prof_stats -name '*.db') | head -1)
    python tools/rocprof_summary.py "$DB" "$BENCH" > $OUT/bench_kernel_stats.txt 2>&1; head -8 $OUT/bench_kernel_stats.txt | cut -c1-60,150-215
    python tools/step_timeline.py "$DB" 2 > $OUT/step_timeline.txt 2>&1; tail -1 $OUT/step_timeline.txt | cut -c1-300
    DB1=$(ls -t $(find $OUT/prof_stats_p1 -name '*.db') | head -1)
    python tools/rocprof_summary.py "$DB1" "MSFM_PIPELINE=1 $BENCH" > $OUT/bench_kernel_stats_pipeline1.txt 2>&1; head -12 $OUT/bench_kernel_stats_pipeline1.txt | cut -c1-60,150-215
    KERN="sweep_i8_kernel<1>,sweep_kernel<3>,pf_prune_q8_kernel,pf_assign_kernel,pf_exact_candidates_kernel,epilogue_kernel,fill_segs_kernel,st_float_kernel,st_i8_kernel,st_classify_kernel"
    PMC_STEPS=5 python tools/pmc_summary.py $OUT/pmc_traffic_approx.json "$KERN" $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.txt 2>&1; tail -3 $OUT/pmc_traffic.txt
    find $OUT -type f -size +8M -delete
    MSFM_Q8=2 run fuzz_q8 python tools/fuzz_routes.py 961 1200; tail -1 $OUT/fuzz_q8.txt
    run fuzz_default python tools/fuzz_routes.py 964 1500; tail -1 $OUT/fuzz_default.txt
    run fuzz_jobs python tools/fuzz_jobs.py 965 600; tail -1 $OUT/fuzz_jobs.txt
    ;;
oom)   # the out-of-memory shrink (tests/test_gpu_jobs.py) + the job-level suite around it, then four contexts on the one GPU at config-4 scale
    TMO=900 MSFM_DEBUG_TIMING= run pytest_oom python -m pytest -m gpu -x -q tests/test_gpu_jobs.py tests/test_gpu_stream.py tests/test_gpu_configs.py; tail -6 $OUT/pytest_oom.txt
    TMO=900 run cli_config4_4ctx python tools/cli_e2e_bench.py --config4 --tables u8 --modes off --orders pair_id --devices 0,0,0,0 --json $OUT/cli_config4_4ctx.json; cat $OUT/cli_config4_4ctx.txt; tail -3 $OUT/cli_config4_4ctx.err
    ;;
check)   # the tree once more: the GPU suite, smoke(), the executable at config-4 scale in both row orders
    TMO=1800 run pytest_gpu python -m pytest tests -m gpu -q; tail -4 $OUT/pytest_gpu.txt
    run smoke python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"; tail -2 $OUT/smoke.txt
    TMO=900 run cli_config4 python tools/cli_e2e_bench.py --config4 --tables u8 --modes off --json $OUT/cli_config4.json; cat $OUT/cli_config4.txt; tail -3 $OUT/cli_config4.err
    ;;
fuzz)   # long seeded fuzz on the final build: routes, whole jobs (incl. the streaming form and store rebuilds), verification
    TMO=900 run fuzz_jobs python tools/fuzz_jobs.py 971 2500; tail -1 $OUT/fuzz_jobs.txt
    TMO=600 MSFM_Q8=2 run fuzz_q8 python tools/fuzz_routes.py 972 3000; tail -1 $OUT/fuzz_q8.txt
    TMO=600 run fuzz_default python tools/fuzz_routes.py 973 3000; tail -1 $OUT/fuzz_default.txt
    TMO=600 MSFM_Q8=2 MSFM_Q8_DIRECT=0 run fuzz_refine python tools/fuzz_routes.py 974 1000; tail -1 $OUT/fuzz_refine.txt
    TMO=300 run fuzz_verify python tools/fuzz_verify.py 975 300; tail -1 $OUT/fuzz_verify.txt
    ;;
small)   # the small-job regime: what one rank of an N = 8 run of config 2 sees (1/8 of the pairs: 46 images -> 1035 pairs)
    for n in 46 64 91; do
      run bench_$n python bench.py --images $n --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --u8-images 0 --sustained-steps 0 --no-solo
      python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6/small/bench_$n.txt") if l.startswith("{")][-1]); r=d["roofline"]
print("images $n: pairs", d["config"]["image_pairs"], "ms_per_step %.3f" % d["ms_per_step"], "sweep1 ms/step %.3f" % r["sweep1_ms_per_step"], "step/sweep1 %.2f" % r["step_over_sweep1"], "device_ms %.3f" % d["device_ms_per_step_rank0"], "sub-batches", d["sub_batches_per_step"])
PY
    done
    BENCH="python $ROOT/bench.py --images 46 --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --sustained-steps 0 --u8-images 0 --no-solo"
    cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"; cd $ROOT
    DB=$(ls -t $(find $OUT/prof_stats -name '*.db') | head -1)
    python tools/step_timeline.py "$DB" 2 > $OUT/step_timeline_46.txt 2>&1; tail -60 $OUT/step_timeline_46.txt | cut -c1-150
    find $OUT -type f -size +8M -delete
    ;;
prefetch3)   # EXPERIMENT: sweep 1 with three k-steps of B fragments in registers (119 VGPRs instead of 111); tools/_ab/libmsfm_prefetch3.so =
             # hipcc ... -DMSFM_I8_PREFETCH3 (built in the container, travels with the snapshot)
    V=tools/_ab/libmsfm_prefetch3.so
    run ab_f32_p1 python tools/ab.py --p1 --rounds 14 tree pre3=$V; cat $OUT/ab_f32_p1.txt
    run ab_f32 python tools/ab.py --rounds 14 tree pre3=$V; cat $OUT/ab_f32.txt
    run ab_u8_p1 python tools/ab.py --u8 --images 64 --p1 --rounds 12 tree pre3=$V; cat $OUT/ab_u8_p1.txt
    run ab_u8 python tools/ab.py --u8 --images 64 --rounds 12 tree pre3=$V; cat $OUT/ab_u8.txt
    run ab_u8_big python tools/ab.py --u8 --images 160 --rounds 6 tree pre3=$V; cat $OUT/ab_u8_big.txt
    ;;
sq)   # SQ counters of the sweeps for the bench command (two --pmc passes of their own: never together with a trace)
    BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --sustained-steps 0 --u8-images 0 --no-solo"
    cd /tmp
    rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | grep -E "MFMA|LDS|BARRIER|WAIT|BUSY|VALU|ACTIVE_INST|WAVE_CYCLES|IFETCH|INST_LEVEL" | tr '\n' ' ' > $OUT/sq_counters_available.txt; wc -w $OUT/sq_counters_available.txt
    timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
    timeout 400 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1; echo "sq2 rc=$?"
    cd $ROOT
    python tools/pmc_summary.py $OUT/pmc_sq.json "sweep_kernel<3>,sweep_i8_kernel<1>" $OUT/pmc_sq | tail -45
    python tools/pmc_summary.py $OUT/pmc_sq2.json "sweep_kernel<3>,sweep_i8_kernel<1>" $OUT/pmc_sq2 | tail -45
    tail -3 $OUT/pmc_sq2.log
    find $OUT -type f -size +8M -delete
    ;;
fourth)
    bash tools/gpu_r6.sh suite
    bash tools/gpu_r6.sh multi
    ;;
third)
    bash tools/gpu_r6.sh suite
    bash tools/gpu_r6.sh cli
    bash tools/gpu_r6.sh nccl1
    ;;
second)
    bash tools/gpu_r6.sh suite
    bash tools/gpu_r6.sh cli
    bash tools/gpu_r6.sh toggle
    ;;
first)   # both of the above in one call
    bash tools/gpu_r6.sh toggle
    bash tools/gpu_r6.sh cfg5full
    ;;
*)
    echo "unknown stage $STAGE"; exit 2
    ;;
esac
