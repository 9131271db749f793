mkdir -p /tmp/clidbg gpurun_out/r5/cli_ab
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from monocularsfm_amd import synth
synth.south_building_database("/tmp/clidbg/sb.db", 128, 5000, seed=1234)
open("/tmp/clidbg/run.yaml", "w").write('%YAML:1.0\ndatabase_path : "/tmp/clidbg/run.db"\nSIFTmatch.match_type : 1\n')
PY
cat > /tmp/hipinit.cpp <<'C'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() { auto t0 = std::chrono::steady_clock::now(); int n = 0; hipGetDeviceCount(&n);
  auto t1 = std::chrono::steady_clock::now();
  printf("hipGetDeviceCount %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count()); return 0; }
C
/opt/rocm/bin/hipcc -O2 -o /tmp/hipinit /tmp/hipinit.cpp 2>/dev/null
for s in 0 0.1 0.3 0.6 1.0 2.0 0 1.0 0 2.0; do
  cp /tmp/clidbg/sb.db /tmp/clidbg/run.db
  monocularsfm_amd/host/ComputeMatches /tmp/clidbg/run.yaml > /dev/null 2>&1
  sleep $s
  echo "after the executable + sleep $s: $(/tmp/hipinit)"
done | tee gpurun_out/r5/cli_ab/init_settle.txt
for s in 0 0.5 1.0; do /tmp/hipinit > /dev/null; sleep $s; echo "after a bare process + sleep $s: $(/tmp/hipinit)"; done | tee -a gpurun_out/r5/cli_ab/init_settle.txt
