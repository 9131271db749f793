set -u
mkdir -p gpurun_out/r5/hints
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_jobs.py -k "plan_regrow or small_call" > gpurun_out/r5/hints/pytest.txt 2>&1; tail -5 gpurun_out/r5/hints/pytest.txt
bash tools/gpu_r5.sh clidbg 2>&1 | grep -v "msfm host\|regrow" | head -40
