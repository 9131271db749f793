#!/bin/bash
# Round 4, call 34: SQ counters of the two sweeps on the final build, every kernel alone (MSFM_PIPELINE=1), same short bench command
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_sq
MSFM_PIPELINE=1 PMC_STEPS=3 timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d /tmp/pmc_sq -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustained-steps 0 --u8-images 0 --no-solo > $OUT/r4_pmc_sq.log 2>&1; echo "sq rc=$?"
cd $ROOT
PMC_STEPS=3 python tools/pmc_summary.py $OUT/r4_pmc_sq.json "sweep_kernel<3>,sweep_i8_kernel<1>" /tmp/pmc_sq | tail -30
