#!/bin/bash
# byte-valued float uploads: tests + the route table
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
timeout 600 python tools/configs_table.py > $OUT/configs.txt 2>&1; echo "configs rc=$?"; cut -c1-200 $OUT/configs.txt
timeout 300 python tools/fuzz_routes.py 201 400 > $OUT/fuzz_routes.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_routes.txt
