#!/bin/bash
# Round 4, call 18: with ~5 instead of ~7.3 ms of tail kernels per step, do the pipeline's defaults (6 parts, last one 0.3 of the average, 3 in flight) still hold?
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 600 python tools/ab_envs.py --rounds 14 "" "MSFM_PIPELINE=1" "MSFM_PIPELINE=3" "MSFM_PIPELINE=4" "MSFM_PIPELINE=8" "MSFM_PIPELINE=4,MSFM_PIPELINE_TAPER=0.5" "MSFM_PIPELINE=6,MSFM_PIPELINE_TAPER=0.15" "MSFM_PIPELINE=6,MSFM_PIPELINE_TAPER=0.6" "MSFM_IN_FLIGHT=2" "MSFM_IN_FLIGHT=2,MSFM_PIPELINE=4" > $OUT/r4_pipeline_ab.txt 2>&1; echo "rc=$?"; cat $OUT/r4_pipeline_ab.txt
timeout 600 python tools/ab_envs.py --rounds 14 "" "MSFM_PIPELINE=1" "MSFM_PIPELINE=3" "MSFM_PIPELINE=4" "MSFM_PIPELINE=8" "MSFM_PIPELINE=4,MSFM_PIPELINE_TAPER=0.5" "MSFM_PIPELINE=6,MSFM_PIPELINE_TAPER=0.15" "MSFM_PIPELINE=6,MSFM_PIPELINE_TAPER=0.6" "MSFM_IN_FLIGHT=2" "MSFM_IN_FLIGHT=2,MSFM_PIPELINE=4" > $OUT/r4_pipeline_ab2.txt 2>&1; echo "rc=$?"; cat $OUT/r4_pipeline_ab2.txt
