mkdir -p gpurun_out/r5/cli_ab /tmp/clidbg
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from monocularsfm_amd import synth
synth.south_building_database("/tmp/clidbg/sb.db", 128, 5000, seed=1234)
open("/tmp/clidbg/run.yaml", "w").write('%YAML:1.0\ndatabase_path : "/tmp/clidbg/run.db"\nSIFTmatch.match_type : 1\n')
PY
for k in 1 2 3 4 5; do cp /tmp/clidbg/sb.db /tmp/clidbg/run.db
MSFM_DEBUG_TIMING=1 MSFM_CLI_TIMING=1 monocularsfm_amd/host/ComputeMatches /tmp/clidbg/run.yaml 2>&1 > /dev/null | grep "create:\|msfm timing" | sed 's/exist-check.*read keypoints/.../'; echo; done
cat > /tmp/hipinit.cpp <<'C'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() { auto t0 = std::chrono::steady_clock::now(); int n = 0; hipGetDeviceCount(&n);
  auto t1 = std::chrono::steady_clock::now(); hipSetDevice(0); void* p; hipMalloc(&p, 1 << 20);
  auto t2 = std::chrono::steady_clock::now();
  printf("bare process: hipGetDeviceCount %.1f ms, set device + first hipMalloc %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count()); return 0; }
C
/opt/rocm/bin/hipcc -O2 -o /tmp/hipinit /tmp/hipinit.cpp 2>/dev/null && for k in 1 2 3 4 5; do /tmp/hipinit; done
