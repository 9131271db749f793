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
