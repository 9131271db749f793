"""End-to-end wall-clock of the drop-in `ComputeMatches <yaml>` executable.

Default: a South-Building-shaped synthetic database (128 images x ~5000 float32 descriptors, brute-force mode, pre-emptive filter on as
in the reference), next to the CPU oracle's rate on a sample of the same pairs.
    python tools/cli_e2e_bench.py [n_images] [n_desc]

`--config4`: the executable at the scale the strong-scaling target is stated on -- a BASELINE-configs[3]-shaped
database (1329 images x 8192 byte descriptors: 882 456 pairs, ~3.6e8 matches, ~2.9 GB of match BLOBs, 2.6 M stdout lines), from the
`descriptors_u8` side table and from the reference's float32 table, geometric verification off and on, with the phases
(MSFM_CLI_TIMING) and the pipeline's own line (device threads vs the calling thread: which one bounds the run):
    python tools/cli_e2e_bench.py --config4 [--images 1329] [--desc 8192] [--tables u8,f32] [--json out.json]
Replaces at this size: /root/reference/src/Feature/FeatureMatching.cpp:102-145 driven by sfm/ComputeMatches.cpp:59-65."""
import json, os, re, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocularsfm_amd import database, synth

EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "monocularsfm_amd", "host", "ComputeMatches")


def run_exe(cfg, env_extra, stdout_path):
    e = dict(os.environ)
    e.update(env_extra)
    e["MSFM_CLI_TIMING"] = "1"
    t0 = time.time()
    with open(stdout_path, "wb") as so:
        r = subprocess.run([EXE, cfg], stdout=so, stderr=subprocess.PIPE, env=e)
    wall = time.time() - t0
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, err[-800:]
    timing = ([l for l in err.splitlines() if "[msfm timing] exist-check" in l] or [""])[-1]
    pipe = ([l for l in err.splitlines() if "[msfm pipeline]" in l] or [""])[-1]
    phases = {k.strip(): float(v) for k, v in re.findall(r"(?:\] |\| )([a-zA-Z+ \-]+?) ([0-9.]+) s", timing)}
    stamps = [float(x) for x in re.findall(r"main (?:entered|left) at ([0-9.]+)", err)]
    return wall, phases, pipe, stamps, t0


def config4_main(argv):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--config4", action="store_true")
    ap.add_argument("--images", type=int, default=1329)
    ap.add_argument("--desc", type=int, default=8192)
    ap.add_argument("--tables", default="u8,f32", help="which descriptor tables to run from: u8 (side table, MSFM_USE_DESCRIPTORS_U8=1), f32")
    ap.add_argument("--modes", default="off,device", help="geometric verification modes: off, device, host")
    ap.add_argument("--devices", default="", help="MSFM_DEVICES for the runs (e.g. 0,0 : two contexts on one GPU)")
    ap.add_argument("--orders", default="reference,pair_id", help="emission orders: reference (default behaviour), pair_id (MSFM_EMIT_ORDER=pair_id)")
    ap.add_argument("--json", default="")
    ap.add_argument("--tmp", default=None)
    args = ap.parse_args(argv)
    tmp = tempfile.mkdtemp(prefix="msfm_e2e4_", dir=args.tmp)
    records = []
    n_pairs = args.images * (args.images - 1) // 2
    try:
        for table in args.tables.split(","):
            base = os.path.join(tmp, "config4_%s.db" % table)
            t0 = time.time()
            counts = synth.u8_database(base, args.images, args.desc, seed=1329, f32_table=(table == "f32"), u8_table=(table == "u8"))
            gen_s = time.time() - t0
            base_mb = os.path.getsize(base) / 1e6
            print("database (%s table): %d images x %d descriptors, %.0f MB, built in %.1f s" % (table, args.images, args.desc, base_mb, gen_s), flush=True)
            for mode, order in [(m, o) for m in args.modes.split(",") for o in args.orders.split(",")]:
                db2 = base + "." + mode + "." + order
                shutil.copyfile(base, db2)
                for ext in ("-wal", "-shm"):
                    if os.path.exists(base + ext):
                        shutil.copyfile(base + ext, db2 + ext)
                cfg = os.path.join(tmp, "cfg_%s_%s_%s.yaml" % (table, mode, order))
                # raw byte descriptors: distances are in byte units (hundreds), the reference's default max_distance 0.7 is for RootSIFT
                open(cfg, "w").write('%%YAML:1.0\ndatabase_path : "%s"\nSIFTmatch.match_type : 1\nSIFTmatch.max_distance : 1000000000.0\n' % db2)
                env = {"MSFM_HONOUR_YAML_MATCH_PARAMS": "1", "MSFM_GEOMETRIC_VERIFICATION": {"off": "0", "device": "1", "host": "host"}[mode]}
                if table == "u8":
                    env["MSFM_USE_DESCRIPTORS_U8"] = "1"
                if args.devices:
                    env["MSFM_DEVICES"] = args.devices
                if order == "pair_id":
                    env["MSFM_EMIT_ORDER"] = "pair_id"
                out_path = os.path.join(tmp, "stdout_%s_%s_%s.txt" % (table, mode, order))
                wall, phases, pipe, stamps, t_start = run_exe(cfg, env, out_path)
                lines = sum(1 for _ in open(out_path, "rb"))
                last = subprocess.run(["tail", "-1", out_path], capture_output=True, text=True).stdout.strip()
                db = database.Database(db2)
                rows, matches, blob = db.db.execute("SELECT COUNT(*), SUM(rows), SUM(LENGTH(data)) FROM matches").fetchone()
                db.Close()
                size_after = sum(os.path.getsize(db2 + e) for e in ("", "-wal") if os.path.exists(db2 + e)) / 1e6
                gpu = phases.get("device match + fetch", 0.0)
                emit = phases.get("stdout + WriteMatches", 0.0)
                rec = {"table": table, "verification": mode, "emit_order": order, "devices": args.devices or "0", "wall_s": wall, "phases_s": phases, "pipeline_line": pipe,
                       "main_entered_after_s": (stamps[0] - t_start) if len(stamps) == 2 else None,
                       "behind_main_s": (t_start + wall - stamps[1]) if len(stamps) == 2 else None,
                       "image_pairs": n_pairs, "rows_written": rows, "matches_written": int(matches or 0), "match_blob_bytes": int(blob or 0),
                       "stdout_lines": lines, "last_line": last, "database_MB_before": base_mb, "database_MB_after": size_after,
                       "wall_over_max_of_device_and_emission": wall / max(1e-9, max(gpu, emit)),
                       "database_build_s": gen_s}
                records.append(rec)
                print("ComputeMatches, %s table, verification %s, rows in %s order: wall %.2f s | rows %d of %d pairs, %d matches (%.2f GB of BLOBs), %d stdout lines | %s" % (
                    table, mode, order, wall, rows, n_pairs, int(matches or 0), (blob or 0) / 1e9, lines, last), flush=True)
                print("    " + ([l for l in [pipe] if l] or ["(no pipeline line)"])[0], flush=True)
                print("    phases: " + " | ".join("%s %.2f" % kv for kv in phases.items()), flush=True)
                for f in (db2, db2 + "-wal", db2 + "-shm", out_path):
                    if os.path.exists(f):
                        os.remove(f)
            for f in (base, base + "-wal", base + "-shm"):
                if os.path.exists(f):
                    os.remove(f)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {"what": "end_to_end_large: the ComputeMatches executable on a BASELINE-configs[3]-shaped database, one MI355X, fresh process per run",
           "images": args.images, "descriptors_per_image": args.desc, "descriptor_pairs": float(n_pairs) * args.desc * args.desc, "runs": records}
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)
    return 0


if "--config4" in sys.argv[1:]:
    sys.exit(config4_main(sys.argv[1:]))

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
exe = EXE
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
