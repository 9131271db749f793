#!/bin/bash
# The round's evidence in one GPU call: gpu_round_profile.sh (tests, bench, kernel stats, HBM traffic passes), the SQ
# counter pass, the three-route config table.  Copy what should be judged from gpurun_out/ into profiles/.
set -u
bash tools/gpu_round_profile.sh
bash tools/gpu_pmc_sq.sh > gpurun_out/pmc_sq_summary.txt 2>&1; tail -5 gpurun_out/pmc_sq_summary.txt
timeout 600 python tools/configs_table.py > gpurun_out/configs.txt 2>&1; echo "configs rc=$?"; cut -c1-250 gpurun_out/configs.txt
timeout 600 python tools/cli_e2e_bench.py > gpurun_out/cli_e2e.txt 2>&1; echo "e2e rc=$?"; tail -12 gpurun_out/cli_e2e.txt
rocm-smi --showclocks --showpower > gpurun_out/smi.txt 2>&1
