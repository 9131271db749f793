#!/bin/bash
# Round 4, call 23: (1) transposed reverse items of the fp16 sweep 2 (MSFM_S2_TRANSPOSE=1): same results as the plain plan on the bench job
# (tools/ab_envs.py asserts it), sweep-2 time with the pipeline off and on; route / job / config parity tests and the route fuzz with the
# switch on.  (2) sweep 1 without its fifth k-step (timing experiment, wrong results): the ceiling of a norm-free formulation.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 420 python tools/ab_envs.py --rounds 10 "MSFM_PIPELINE=3" "MSFM_PIPELINE=1" "MSFM_PIPELINE=1,MSFM_S2_TRANSPOSE=1" "" "MSFM_S2_TRANSPOSE=1" > $OUT/r4_s2t_ab.txt 2>&1; echo "rc=$?"; tail -8 $OUT/r4_s2t_ab.txt
MSFM_S2_TRANSPOSE=1 timeout 900 python -m pytest tests -x -q -m gpu > $OUT/r4_s2t_pytest.txt 2>&1; echo "rc=$?"; tail -5 $OUT/r4_s2t_pytest.txt
MSFM_S2_TRANSPOSE=1 timeout 600 python tools/fuzz_routes.py 961 300 > $OUT/r4_s2t_fuzz.txt 2>&1; echo "rc=$?"; tail -2 $OUT/r4_s2t_fuzz.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_EXPERIMENT_NO_DIGITS -shared -o /tmp/libmsfm_nodigits.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error"
timeout 300 python tools/ab_multi.py --p1 --nocheck --images 40 --rounds 8 tree nodigits=/tmp/libmsfm_nodigits.so > $OUT/r4_nodigits_ab.txt 2>&1; echo "rc=$?"; tail -4 $OUT/r4_nodigits_ab.txt
timeout 300 python tools/ab_multi.py --u8 --p1 --nocheck --images 40 --rounds 8 tree nodigits=/tmp/libmsfm_nodigits.so > $OUT/r4_nodigits_ab_u8.txt 2>&1; echo "rc=$?"; tail -4 $OUT/r4_nodigits_ab_u8.txt
