python - <<'PY'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
from monocularsfm_amd import synth
tmp = tempfile.mkdtemp(prefix="msfm_cli_ab_")
db = os.path.join(tmp, "sb.db")
synth.south_building_database(db, 128, 5000, seed=1234)
for rep in range(3):
    for mm in ("1", "0"):
        db2 = db + ".%d.%s" % (rep, mm)
        subprocess.check_call(["cp", db, db2])
        cfg = db2 + ".yaml"
        open(cfg, "w").write('%%YAML:1.0\ndatabase_path : "%s"\nSIFTmatch.match_type : 1\n' % db2)
        t0 = time.time()
        r = subprocess.run(["monocularsfm_amd/host/ComputeMatches", cfg], capture_output=True, text=True, env=dict(os.environ, MSFM_CLI_TIMING="1", MSFM_SQLITE_MMAP=mm))
        print("mmap", mm, "wall %.3f" % (time.time() - t0), r.stderr.strip().splitlines()[-1][:200], flush=True)
PY
