#!/bin/bash
# How long a fresh process waits in the HIP runtime's start-up depends on WHEN the previous GPU process of the box exited: within ~0.1 s
# of it the driver is still tearing that process down and hipGetDeviceCount takes 170-240 ms instead of ~52 ms -- with the ComputeMatches
# executable and with a one-line HIP program alike.  (bench.py's end_to_end therefore leaves the device alone for a second before each of
# its two processes.)   Usage, on a GPU box, from the repository root:  bash tools/hip_init_settle.sh  ->  stdout
set -u
mkdir -p /tmp/msfm_settle
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from monocularsfm_amd import synth
synth.south_building_database("/tmp/msfm_settle/sb.db", 128, 5000, seed=1234)
open("/tmp/msfm_settle/run.yaml", "w").write('%YAML:1.0\ndatabase_path : "/tmp/msfm_settle/run.db"\nSIFTmatch.match_type : 1\n')
PY
cat > /tmp/msfm_settle/hipinit.cpp <<'C'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    (void)hipGetDeviceCount(&n);
    printf("hipGetDeviceCount %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
C
/opt/rocm/bin/hipcc -O2 -o /tmp/msfm_settle/hipinit /tmp/msfm_settle/hipinit.cpp 2>/dev/null || exit 1
echo "# a one-line HIP program, started <sleep> seconds after the ComputeMatches executable (South-Building-shaped job) has exited"
for s in 0 0.1 0.3 0.6 1.0 2.0 0 1.0 0 2.0; do
    cp /tmp/msfm_settle/sb.db /tmp/msfm_settle/run.db
    monocularsfm_amd/host/ComputeMatches /tmp/msfm_settle/run.yaml > /dev/null 2>&1
    sleep $s
    echo "after the executable + sleep $s: $(/tmp/msfm_settle/hipinit)"
done
echo "# ... and after itself"
for s in 0 0.5 1.0; do /tmp/msfm_settle/hipinit > /dev/null; sleep $s; echo "after a one-line HIP program + sleep $s: $(/tmp/msfm_settle/hipinit)"; done
echo "# the executable itself, MSFM_CLI_TIMING / MSFM_DEBUG_TIMING: right behind another run of itself, and a second later"
for s in 0 0 1.0 0 1.0; do
    cp /tmp/msfm_settle/sb.db /tmp/msfm_settle/run.db
    sleep $s
    t0=$(date +%s.%N)
    MSFM_CLI_TIMING=1 MSFM_DEBUG_TIMING=1 monocularsfm_amd/host/ComputeMatches /tmp/msfm_settle/run.yaml > /dev/null 2> /tmp/msfm_settle/err.txt
    t1=$(date +%s.%N)
    echo "sleep $s: wall $(python -c "print('%.3f' % ($t1 - $t0))") s | $(grep 'create: hipGetDeviceCount' /tmp/msfm_settle/err.txt | sed 's/.*create: //') | $(grep 'msfm timing' /tmp/msfm_settle/err.txt | sed 's/.*open database/open database/')"
done
rm -rf /tmp/msfm_settle
