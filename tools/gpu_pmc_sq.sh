#!/bin/bash
# SQ counters of the sweep kernels for the same bench command (one --pmc pass, csv), summarised into gpurun_out/pmc_sq.json
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/pmc_sq
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $ROOT
python tools/pmc_summary.py $OUT/pmc_sq.json "sweep_kernel<1>,sweep_kernel<3>,sweep_i8_kernel<1>,sweep_i8_kernel<3>" $OUT/pmc_sq | tail -60
find $OUT/pmc_sq -type f -size +8M -delete
