#!/bin/bash
# Round 4, call 20: confirm call 19's best point (two equal parts) against its neighbours and the shipped default, two job sizes, twice
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
E='MSFM_PIPELINE=6,MSFM_PIPELINE_TAPER=0.31 MSFM_PIPELINE=6,MSFM_PIPELINE_TAPER=0.3 MSFM_PIPELINE=2,MSFM_PIPELINE_TAPER=1.0 MSFM_PIPELINE=2,MSFM_PIPELINE_TAPER=0.8 MSFM_PIPELINE=2,MSFM_PIPELINE_TAPER=0.65 MSFM_PIPELINE=3,MSFM_PIPELINE_TAPER=0.7 MSFM_PIPELINE=1'
for rep in 1 2; do
timeout 600 python tools/ab_envs.py --rounds 18 $E > $OUT/r4_pipeline_ab4_$rep.txt 2>&1; echo "rc=$?"; cat $OUT/r4_pipeline_ab4_$rep.txt
done
timeout 600 python tools/ab_envs.py --images 64 --rounds 18 $E > $OUT/r4_pipeline_ab4_64.txt 2>&1; echo "rc=$?"; cat $OUT/r4_pipeline_ab4_64.txt
timeout 600 python tools/ab_envs.py --images 200 --rounds 8 $E > $OUT/r4_pipeline_ab4_200.txt 2>&1; echo "rc=$?"; cat $OUT/r4_pipeline_ab4_200.txt
