"""End-to-end wall-clock of the drop-in `ComputeMatches <yaml>` executable on a South-Building-shaped
synthetic database (128 images x ~5000 float32 descriptors, brute-force mode, pre-emptive filter on as in
the reference), next to the CPU oracle's rate on a sample of the same pairs.
Usage: python tools/cli_e2e_bench.py [n_images] [n_desc]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocularsfm_amd import database, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rng = np.random.default_rng(1234)
t0 = time.time()
tmp = tempfile.mkdtemp(prefix="msfm_e2e_")
db_path = os.path.join(tmp, "south-building-synth.db")
descs, kps = synth.south_building_database(db_path, N, n, seed=1234)
counts = np.array([len(d) for d in descs])
cfg = os.path.join(tmp, "cfg.yaml")
open(cfg, "w").write('%%YAML:1.0\ndatabase_path : "%s"\nSIFTmatch.match_type : 1\n' % db_path)
print("dataset: %d images, %d descriptors, db %.0f MB, built in %.1f s" % (N, int(counts.sum()), os.path.getsize(db_path) / 1e6, time.time() - t0))
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "monocularsfm_amd", "host", "ComputeMatches")
for label, env in (("geometric verification on the device (reference default flow)", {}),
                   ("geometric verification by the host twin", {"MSFM_GEOMETRIC_VERIFICATION": "host"}),
                   ("geometric verification off", {"MSFM_GEOMETRIC_VERIFICATION": "0"})):
    db2 = db_path + "." + ("gv" if not env else ("gvhost" if env["MSFM_GEOMETRIC_VERIFICATION"] == "host" else "nogv"))
    subprocess.check_call(["cp", db_path, db2])
    c2 = cfg + os.path.basename(db2).rsplit(".", 1)[1]
    open(c2, "w").write('%%YAML:1.0\ndatabase_path : "%s"\nSIFTmatch.match_type : 1\n' % db2)
    e = dict(os.environ); e.update(env)
    t0 = time.time()
    e["MSFM_CLI_TIMING"] = "1"
    r = subprocess.run([exe, c2], capture_output=True, text=True, env=e)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr[-500:]
    print("   ", ([l for l in r.stderr.splitlines() if "exist-check" in l] or [""])[-1])
    db = database.Database(db2)
    rows = db.db.execute("SELECT COUNT(*), SUM(rows) FROM matches").fetchone()
    db.Close()
    print("ComputeMatches CLI, %s: wall %.2f s; matches rows %d (of %d pairs), %d matches; last line: %s" % (
        label, dt, rows[0], N * (N - 1) // 2, rows[1] or 0, r.stdout.strip().splitlines()[-1]))
# CPU reference-equivalent rate on a sample (matching only, no DB, no RANSAC)
from oracle import c_oracle as co
pairs = [(i, j) for i in range(N) for j in range(i)]
sample = [pairs[k] for k in rng.choice(len(pairs), 24, replace=False)]
thr = os.cpu_count()
t0 = time.time(); work = 0
for i, j in sample:
    co.match_pair(descs[i], descs[j], nthreads=thr); work += len(descs[i]) * len(descs[j])
dt = time.time() - t0
total = sum(len(descs[i]) * len(descs[j]) for i, j in pairs)
print("CPU oracle (%d threads): %.3e descriptor-pairs/s on 24 pairs -> matching alone would take %.0f s for all %d pairs" % (thr, work / dt, total / (work / dt), len(pairs)))
t0 = time.time(); co.match_pair(descs[sample[0][0]], descs[sample[0][1]], nthreads=1); dt1 = time.time() - t0
print("CPU oracle (1 thread, like the reference's own code): %.2f s per pair -> %.0f s for all pairs" % (dt1, dt1 * len(pairs)))
