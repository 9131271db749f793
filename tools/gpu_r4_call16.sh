#!/bin/bash
# Round 4, call 16: the N > 1 flow of bench.py AT SCALE on the one GPU there is -- two ranks sharing it over gloo, config 4 in full as strong_u8
# (each rank matches half of the 882 456 pairs, the writer receives 1.4 GB of lists): what the driver's SCALE run will execute, except RCCL.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 2 --backend gloo --share-gpu --steps 2 --warmup 1 --images 32 --no-cpu-baseline ) > $OUT/r4_two_ranks_full_u8.json 2> $OUT/r4_two_ranks_full_u8.err; echo "rc=$?"; tail -5 $OUT/r4_two_ranks_full_u8.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_two_ranks_full_u8.json"))
print("n_gpus", d["n_gpus"], "main matches", d["config"]["matches_per_step"], d["per_rank_ms"])
print(json.dumps(d["strong_u8"])[:1600])
PY
