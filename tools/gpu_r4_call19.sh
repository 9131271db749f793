#!/bin/bash
# Round 4, call 19: parts x taper around the best point of call 18 (a dummy first context takes the first-in-the-alternation slot)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
E='MSFM_PIPELINE=6,MSFM_PIPELINE_TAPER=0.31 MSFM_PIPELINE=6,MSFM_PIPELINE_TAPER=0.3 MSFM_PIPELINE=4,MSFM_PIPELINE_TAPER=0.5 MSFM_PIPELINE=4,MSFM_PIPELINE_TAPER=0.7 MSFM_PIPELINE=4,MSFM_PIPELINE_TAPER=1.0 MSFM_PIPELINE=3,MSFM_PIPELINE_TAPER=0.5 MSFM_PIPELINE=3,MSFM_PIPELINE_TAPER=1.0 MSFM_PIPELINE=5,MSFM_PIPELINE_TAPER=0.5 MSFM_PIPELINE=2,MSFM_PIPELINE_TAPER=1.0 MSFM_PIPELINE=2,MSFM_PIPELINE_TAPER=0.5'
timeout 600 python tools/ab_envs.py --rounds 14 $E > $OUT/r4_pipeline_ab3.txt 2>&1; echo "rc=$?"; cat $OUT/r4_pipeline_ab3.txt
timeout 600 python tools/ab_envs.py --u8 --images 160 --rounds 7 $E > $OUT/r4_pipeline_ab3_u8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_pipeline_ab3_u8.txt
