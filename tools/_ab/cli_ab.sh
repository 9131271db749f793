set -u
mkdir -p gpurun_out/r5/cli_ab /tmp/clidbg
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from monocularsfm_amd import synth
synth.south_building_database("/tmp/clidbg/sb.db", 128, 5000, seed=1234)
open("/tmp/clidbg/run.yaml", "w").write('%YAML:1.0\ndatabase_path : "/tmp/clidbg/run.db"\nSIFTmatch.match_type : 1\n')
PY
for round in 1 2 3 4 5 6; do
  for v in tree head; do
    cp /tmp/clidbg/sb.db /tmp/clidbg/run.db
    EXE=monocularsfm_amd/host/ComputeMatches
    t0=$(date +%s.%N)
    if [ $v = head ]; then LD_PRELOAD=$PWD/tools/_ab/libmsfm_match_head.so MSFM_CLI_TIMING=1 $EXE /tmp/clidbg/run.yaml > /dev/null 2> /tmp/clidbg/err.txt
    else MSFM_CLI_TIMING=1 $EXE /tmp/clidbg/run.yaml > /dev/null 2> /tmp/clidbg/err.txt; fi
    t1=$(date +%s.%N)
    echo "$v wall $(python -c "print('%.3f' % ($t1 - $t0))") $(grep 'msfm timing' /tmp/clidbg/err.txt | sed 's/exist-check.*upload/upload/; s/read keypoints.*open/open/')"
  done
done | tee gpurun_out/r5/cli_ab/cli_ab.txt
MSFM_DEBUG_TIMING=1 monocularsfm_amd/host/ComputeMatches /tmp/clidbg/run.yaml 2>&1 >/dev/null | grep "alloc\] since"
cp /tmp/clidbg/sb.db /tmp/clidbg/run.db
MSFM_DEBUG_TIMING=1 monocularsfm_amd/host/ComputeMatches /tmp/clidbg/run.yaml 2>&1 >/dev/null | grep "create\|destroy\|alloc\] since"
