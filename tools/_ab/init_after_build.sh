mkdir -p gpurun_out/r5/cli_ab
cat > /tmp/hipinit.cpp <<'C'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() { auto t0 = std::chrono::steady_clock::now(); int n = 0; hipGetDeviceCount(&n);
  auto t1 = std::chrono::steady_clock::now();
  printf("bare: hipGetDeviceCount %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count()); return 0; }
C
/opt/rocm/bin/hipcc -O2 -o /tmp/hipinit /tmp/hipinit.cpp 2>/dev/null
/tmp/hipinit
timeout 300 python -m pytest -m gpu -q tests/test_gpu_store.py 2>&1 | tail -1
python tools/_ab/init_after_build.py 2>&1 | tee gpurun_out/r5/cli_ab/init_after_build.txt
