import os, re, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd())
from monocularsfm_amd import synth
os.makedirs("/tmp/clidbg", exist_ok=True)
t0 = time.time(); synth.south_building_database("/tmp/clidbg/sb.db", 128, 5000, seed=1234); print("db build %.1f s" % (time.time() - t0))
open("/tmp/clidbg/run.yaml", "w").write('%YAML:1.0\ndatabase_path : "/tmp/clidbg/run.db"\nSIFTmatch.match_type : 1\n')
def run(label, pre_sleep):
    shutil.copyfile("/tmp/clidbg/sb.db", "/tmp/clidbg/run.db")
    time.sleep(pre_sleep)
    t0 = time.perf_counter()
    r = subprocess.run(["monocularsfm_amd/host/ComputeMatches", "/tmp/clidbg/run.yaml"], capture_output=True, text=True, env=dict(os.environ, MSFM_CLI_TIMING="1", MSFM_DEBUG_TIMING="1"))
    w = time.perf_counter() - t0
    cr = [l for l in r.stderr.splitlines() if "create:" in l]
    ph = [l for l in r.stderr.splitlines() if "msfm timing" in l][-1]
    print("%-40s wall %.3f | %s | %s" % (label, w, re.sub(r".*open database", "open database", ph), "; ".join(x.split("create: ")[1] for x in cr)))
run("first after the build", 0)
run("right behind it", 0)
run("0.5 s later", 0.5)
run("right behind it", 0)
run("1 s later", 1.0)
subprocess.run(["/tmp/hipinit"])
