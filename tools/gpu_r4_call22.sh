#!/bin/bash
# Round 4, call 22: the many-sub-batch regime (a 400-image config-4-shaped job, cut by memory) -- sets in flight and scratch budget with round 4's tails
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python tools/ab_envs.py --u8 --images 400 --rounds 7 "MSFM_SCRATCH_MIB=49153" "MSFM_SCRATCH_MIB=49152" "MSFM_SCRATCH_MIB=49152,MSFM_IN_FLIGHT=2" "MSFM_SCRATCH_MIB=24576" "MSFM_SCRATCH_MIB=98304" "MSFM_SCRATCH_MIB=24576,MSFM_IN_FLIGHT=2" > $OUT/r4_inflight_u8_400.txt 2>&1; echo "rc=$?"; cat $OUT/r4_inflight_u8_400.txt
