#!/bin/bash
# Can two ranks share this box's one GPU under RCCL?  (VERDICT r02 task 1c.)  Expected: RCCL refuses duplicate devices
# in one communicator; the outcome is recorded in DESIGN.md section 6.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export MASTER_ADDR=127.0.0.1
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 1 --warmup 0 --images 12 --u8-images 0 --backend nccl --share-gpu --sustained-steps 0 > $OUT/rccl_two_ranks.log 2>&1
echo "two-rank nccl --share-gpu rc=$?"
grep -i -E "duplicate|invalid usage|error|ncclInvalid|Traceback|\{\"metric" $OUT/rccl_two_ranks.log | head -12 | cut -c1-300
