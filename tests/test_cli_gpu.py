"""End-to-end drop-in test on the GPU box: the C++ `ComputeMatches <yaml>` executable against a
synthetic SQLite database, compared row by row with what the reference's control flow
(src/Feature/FeatureMatching.cpp) would write when its per-pair arithmetic is the CPU oracle."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from monocularsfm_amd import database, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "monocularsfm_amd", "host")
EXE = os.path.join(HOST, "ComputeMatches")

YAML = """%YAML:1.0
database_path : "{db}"
SIFTmatch.match_type : {mt}
# the next three are parsed and ignored, exactly as the reference does
SIFTmatch.max_distance : 0.1
SIFTmatch.distance_ratio : 0.5
SIFTmatch.cross_check : 0
"""

SIZES = [700, 650, 300, 90, 512, 640, 128]


@pytest.fixture(scope="module")
def exe(built_lib):
    subprocess.check_call(["make", "-C", HOST, "-s"])
    return EXE


@pytest.fixture(scope="module")
def dataset():
    rng = np.random.default_rng(7)
    descs = synth.rootsift_images(len(SIZES), SIZES, seed=2024, n_proto=1400)
    kps = [synth.keypoints(len(d), seed=50 + i) for i, d in enumerate(descs)]
    # "landmarks": 60 of 80 shared descriptors per image, carried by the LARGEST keypoints, so the
    # top-scale subsets of two images overlap and the pre-emptive test passes ...
    pool = descs[0][:80].copy()
    for i in range(len(descs)):
        if i == 6:
            continue
        pick = rng.choice(80, 60, replace=False)
        rows = rng.choice(len(descs[i]), 60, replace=False)
        v = np.abs(pool[pick] * (1 + 0.03 * rng.standard_normal((60, 128)).astype(np.float32)))
        descs[i][rows] = v / np.linalg.norm(v, axis=1, keepdims=True)
        kps[i][rows, 2] = 100 + rng.uniform(0, 50, 60).astype(np.float32)
    # ... except for image 6, which shares nothing: its pairs must be dropped pre-emptively
    descs[6] = synth.rootsift_images(1, [SIZES[6]], seed=999, n_proto=300, overlap=0.0)[0]
    descs = [np.ascontiguousarray(d, dtype=np.float32) for d in descs]
    return descs, kps


def run_cli(exe, cfg, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([exe, str(cfg)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def expected_brute(oracle, descs, kps):
    """pair -> (m x 2 matches as (query=i, train=j)) for pairs that get a row; pairs dropped by the
    pre-emptive filter are absent."""
    from monocularsfm_amd import _lib
    tops = [d[_lib.topscale_select(k, 100)] for d, k in zip(descs, kps)]
    rows = {}
    for i in range(len(descs)):
        for j in range(i):
            pq, _, _ = oracle.match_pair(tops[i], tops[j], 0.8, True, np.inf)
            if len(pq) < 4:
                continue
            q, t, _ = oracle.match_pair(descs[i], descs[j], 0.8, True, 0.7, nthreads=4)
            rows[(i, j)] = np.stack([q, t], 1).reshape(-1, 2)
    return rows


def test_brute_mode_matches_table(exe, dataset, oracle, tmp_path):
    descs, kps = dataset
    db_path = str(tmp_path / "brute.db")
    database.write_synthetic_database(db_path, descs, kps)
    cfg = tmp_path / "brute.yaml"
    cfg.write_text(YAML.format(db=db_path, mt=1))
    out = run_cli(exe, cfg, {"MSFM_GEOMETRIC_VERIFICATION": "0"})
    exp = expected_brute(oracle, descs, kps)
    assert 5 <= len(exp) < len(SIZES) * (len(SIZES) - 1) // 2, "test data: the pre-emptive filter should drop some pairs"
    db = database.Database(db_path)
    n_rows = db.db.execute("SELECT COUNT(*) FROM matches").fetchone()[0]
    assert n_rows == len(exp)
    for (i, j), m in exp.items():
        assert db.ExistMatches(i, j)
        assert np.array_equal(db.ReadMatches(i, j), m), (i, j)
        # stored canonical form: column 0 indexes the smaller image id
        raw = db.db.execute("SELECT rows, cols, data FROM matches WHERE pair_id = ?", (database.ImagePairToPairId(i, j),)).fetchone()
        assert raw[0] == len(m) and raw[1] == 2
        assert np.array_equal(np.frombuffer(raw[2], np.int32).reshape(-1, 2), m[:, ::-1])
    db.Close()
    # stdout contract (FeatureMatching.cpp:29,63-66; sfm/ComputeMatches.cpp:65)
    for (i, j), m in exp.items():
        assert "Compute Matches %d - %d ... " % (i, j) in out
    assert re.search(r"\t matches num : \d+\n\t Elapsed time: \d+\.\d{5} \[seconds\]\n", out)
    assert re.search(r"Elapsed time: \d+\.\d{5} \[minutes\]\n$", out)

    # resume: a second run recomputes nothing that has a row, retries only pre-emptively dropped pairs
    out2 = run_cli(exe, cfg, {"MSFM_GEOMETRIC_VERIFICATION": "0"})
    assert out2.count("Existing, Continue!") == len(exp)
    assert " ... " not in out2
    db = database.Database(db_path)
    assert db.db.execute("SELECT COUNT(*) FROM matches").fetchone()[0] == len(exp)
    db.Close()


def test_sequential_mode_and_yaml_params_are_ignored(exe, dataset, oracle, tmp_path):
    descs, kps = dataset
    db_path = str(tmp_path / "seq.db")
    database.write_synthetic_database(db_path, descs, kps)
    cfg = tmp_path / "seq.yaml"
    cfg.write_text(YAML.format(db=db_path, mt=0))
    run_cli(exe, cfg, {"MSFM_GEOMETRIC_VERIFICATION": "0"})
    pairs, _ = oracle.enumerate_sequential(len(descs), 3)
    db = database.Database(db_path)
    assert db.db.execute("SELECT COUNT(*) FROM matches").fetchone()[0] == len(pairs)
    for i, j in pairs:
        # defaults 0.8 / cross-check / 0.7, NOT the 0.5 / 0 / 0.1 written in the YAML
        q, t, _ = oracle.match_pair(descs[i], descs[j], 0.8, True, 0.7, nthreads=4)
        assert np.array_equal(db.ReadMatches(int(i), int(j)), np.stack([q, t], 1).reshape(-1, 2)), (i, j)
    db.Close()
    # opt-in: honour the YAML values
    db2 = str(tmp_path / "seq2.db")
    database.write_synthetic_database(db2, descs, kps)
    cfg2 = tmp_path / "seq2.yaml"
    cfg2.write_text(YAML.format(db=db2, mt=0))
    run_cli(exe, cfg2, {"MSFM_GEOMETRIC_VERIFICATION": "0", "MSFM_HONOUR_YAML_MATCH_PARAMS": "1"})
    db = database.Database(db2)
    for i, j in pairs[:6]:
        q, t, _ = oracle.match_pair(descs[i], descs[j], 0.5, False, 0.1, nthreads=4)
        assert np.array_equal(db.ReadMatches(int(i), int(j)), np.stack([q, t], 1).reshape(-1, 2)), (i, j)
    db.Close()


def test_geometric_verification_default_on(exe, dataset, tmp_path):
    """Default run applies the F-matrix RANSAC hand-off (a "next" row, outside bit parity): it may only
    remove matches.  Synthetic keypoints carry no epipolar geometry, so most matches are rejected."""
    descs, kps = dataset
    a, b, c = str(tmp_path / "gv.db"), str(tmp_path / "nogv.db"), str(tmp_path / "gvhost.db")
    database.write_synthetic_database(a, descs, kps)
    shutil.copy(a, b)
    shutil.copy(a, c)
    for path, env in ((a, {}), (b, {"MSFM_GEOMETRIC_VERIFICATION": "0"}), (c, {"MSFM_GEOMETRIC_VERIFICATION": "host"})):
        cfg = tmp_path / (os.path.basename(path) + ".yaml")
        cfg.write_text(YAML.format(db=path, mt=0))
        run_cli(exe, cfg, env)
    da, dbb, dc = database.Database(a), database.Database(b), database.Database(c)
    for i in range(1, len(descs)):
        ma, mb, mc = da.ReadMatches(i, i - 1), dbb.ReadMatches(i, i - 1), dc.ReadMatches(i, i - 1)
        sa = set(map(tuple, ma.tolist()))
        assert sa <= set(map(tuple, mb.tolist())) and len(ma) <= len(mb)
        assert np.array_equal(ma, mc)          # device RANSAC == host twin, row for row
    da.Close()
    dbb.Close()
    dc.Close()


def test_python_matcher_mirror_equals_cli(exe, dataset, gpu_ctx, tmp_path):
    from monocularsfm_amd.matcher import BruteFeatureMatcher
    descs, kps = dataset
    a, b = str(tmp_path / "cli.db"), str(tmp_path / "py.db")
    database.write_synthetic_database(a, descs, kps)
    shutil.copy(a, b)
    cfg = tmp_path / "cli.yaml"
    cfg.write_text(YAML.format(db=a, mt=1))
    run_cli(exe, cfg, {"MSFM_GEOMETRIC_VERIFICATION": "0"})
    gpu_ctx.clear_images()
    BruteFeatureMatcher(b, ctx=gpu_ctx, verbose=False).RunMatching()
    ra = database.Database(a).db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    rb = database.Database(b).db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    assert ra == rb and len(ra) > 4
    # and with the geometric verification on the device on both sides (the CLI's default)
    c, d = str(tmp_path / "cli_gv.db"), str(tmp_path / "py_gv.db")
    database.write_synthetic_database(c, descs, kps)
    shutil.copy(c, d)
    cfg2 = tmp_path / "cli_gv.yaml"
    cfg2.write_text(YAML.format(db=c, mt=1))
    run_cli(exe, cfg2, {})
    gpu_ctx.clear_images()
    BruteFeatureMatcher(d, ctx=gpu_ctx, verbose=False, geometric_verification="device").RunMatching()
    rc = database.Database(c).db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    rd = database.Database(d).db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    assert rc == rd and len(rc) == len(ra)


@pytest.mark.parametrize("gv", ["", "0"])
def test_multi_device_split_writes_the_same_rows(exe, dataset, tmp_path, gv):
    """MSFM_DEVICES splits every super-batch over several contexts (one host thread each).  With one GPU in the box
    the same ordinal is listed three times: three contexts, three threads, same rows and stdout as a single device."""
    descs, kps = dataset
    a, b = str(tmp_path / "one.db"), str(tmp_path / "three.db")
    database.write_synthetic_database(a, descs, kps)
    shutil.copy(a, b)
    outs = []
    for path, env in ((a, {}), (b, {"MSFM_DEVICES": "0,0,0"})):
        cfg = tmp_path / (os.path.basename(path) + ".yaml")
        cfg.write_text(YAML.format(db=path, mt=1))
        e = dict(env)
        if gv:
            e["MSFM_GEOMETRIC_VERIFICATION"] = gv
        outs.append(run_cli(exe, cfg, e))
    da, dbb = database.Database(a), database.Database(b)
    ra = da.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    rb = dbb.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    da.Close()
    dbb.Close()
    assert len(ra) > 5 and ra == rb
    strip = lambda s: re.sub(r"Elapsed time: [0-9.]+", "Elapsed time: X", s)
    assert strip(outs[0]) == strip(outs[1])


def test_multi_device_all_uses_every_gpu_of_the_box(exe, dataset, tmp_path):
    """MSFM_DEVICES=all: one context and one host thread per gfx950 device the box has (msfm_device_count) -- one on the
    test box, eight on a full node (never run there: DESIGN.md section 6 says so) -- and "every device twice" on top of it, so
    that the split / merge code runs over 2 x count contexts wherever this test runs.  Same rows, same stdout."""
    from monocularsfm_amd import _lib
    n = _lib.device_count()
    assert n >= 1
    descs, kps = dataset
    base = str(tmp_path / "one.db")
    database.write_synthetic_database(base, descs, kps)
    rows, outs = [], []
    for tag in ("all", "twice"):
        shutil.copy(base, str(tmp_path / (tag + ".db")))
    for tag, env in (("one", {}), ("all", {"MSFM_DEVICES": "all"}), ("twice", {"MSFM_DEVICES": ",".join([str(d) for d in range(n)] * 2)})):
        path = str(tmp_path / (tag + ".db"))
        cfg = tmp_path / (tag + ".yaml")
        cfg.write_text(YAML.format(db=path, mt=1))
        outs.append(run_cli(exe, cfg, dict(env)))
        d = database.Database(path)
        rows.append(d.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall())
        d.Close()
    assert len(rows[0]) > 5 and rows[0] == rows[1] == rows[2]
    strip = lambda s: re.sub(r"Elapsed time: [0-9.]+", "Elapsed time: X", s)
    assert strip(outs[0]) == strip(outs[1]) == strip(outs[2])


@pytest.mark.parametrize("mt", [0, 1])
def test_images_without_descriptors_and_tiny_images(exe, tmp_path, mt):
    """Images with 0, 1 and 2 descriptors next to normal ones: the reference would hit knnMatch's undefined cases
    (FeatureUtils.cpp:152); the drop-in must finish, give every non-skipped pair a row and never invent matches
    for the degenerate sides."""
    descs = synth.rootsift_images(5, [400, 0, 1, 2, 350], seed=404, n_proto=800)
    descs = [np.ascontiguousarray(d, np.float32).reshape(-1, 128) for d in descs]
    kps = [synth.keypoints(len(d), seed=70 + i) for i, d in enumerate(descs)]
    db_path = str(tmp_path / "tiny.db")
    database.write_synthetic_database(db_path, descs, kps)
    cfg = tmp_path / "tiny.yaml"
    cfg.write_text(YAML.format(db=db_path, mt=mt))
    for env in ({"MSFM_GEOMETRIC_VERIFICATION": "0"}, {}):
        p2 = db_path + (".gv" if not env else ".nogv")
        shutil.copy(db_path, p2)
        c2 = tmp_path / (os.path.basename(p2) + ".yaml")
        c2.write_text(YAML.format(db=p2, mt=mt))
        out = run_cli(exe, c2, env)
        assert re.search(r"Elapsed time: \d+\.\d{5} \[minutes\]\n$", out)
        db = database.Database(p2)
        rows = db.db.execute("SELECT pair_id, rows FROM matches").fetchall()
        db.Close()
        for pid, r in rows:
            i, j = pid % 10000, pid // 10000     # pair_id = 10000 * min + max
            if min(len(descs[i]), len(descs[j])) < 2:
                assert r == 0, (pid, r)


@pytest.mark.parametrize("exist_check", ["per_pair", "index_sweep"])
def test_partial_resume_recomputes_only_missing_rows(exe, dataset, tmp_path, exist_check):
    """Rows are the checkpoint (FeatureMatching.cpp:23-27): after deleting a few rows a rerun recomputes exactly
    those, in place, with the same bytes, and reports every other pair as existing.  Both forms of the exist-check: one
    ExistMatches SELECT per pair (the reference's, used for small jobs) and the one sweep over the table's keys a large job takes."""
    descs, kps = dataset
    sweep_env = {"MSFM_EXIST_SWEEP_MIN": "0" if exist_check == "index_sweep" else "1000000000"}
    db_path = str(tmp_path / "resume.db")
    database.write_synthetic_database(db_path, descs, kps)
    cfg = tmp_path / "resume.yaml"
    cfg.write_text(YAML.format(db=db_path, mt=1))
    run_cli(exe, cfg, sweep_env)
    db = database.Database(db_path)
    before = db.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    victims = [before[1][0], before[len(before) // 2][0], before[-1][0]]
    db.db.execute("DELETE FROM matches WHERE pair_id IN (%s)" % ",".join(str(v) for v in victims))
    db.db.commit()
    db.Close()
    out = run_cli(exe, cfg, sweep_env)
    db = database.Database(db_path)
    after = db.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    db.Close()
    assert after == before
    assert out.count("Existing, Continue!") == len(before) - len(victims)
    assert out.count(" ... \n") == len(victims)


def test_brute_mode_past_the_100_pair_flush_and_the_super_batch(exe, oracle, tmp_path):
    """104 tiny images: rows i >= 100 of the pair matrix cross the reference's 100-pair flush
    (/root/reference/src/Feature/FeatureMatching.cpp:118-139, max_pairs_size_) and, with the super-batch lowered to
    250 pairs, the job takes ~20 device calls.  Rows, stdout order and transaction grouping must be what the
    reference's control flow gives, and must not depend on the super-batch size."""
    rng = np.random.default_rng(104)
    n_img = 104
    sizes = rng.integers(24, 64, n_img).tolist()
    sizes[17] = 0        # an image without features
    sizes[33] = 1
    descs = synth.rootsift_images(n_img, sizes, seed=3104, n_proto=150, sigma=0.03, overlap=0.8)
    descs = [np.ascontiguousarray(d, dtype=np.float32) for d in descs]
    kps = [synth.keypoints(len(d), seed=700 + i) for i, d in enumerate(descs)]
    # expectation: the reference's enumeration + flush boundaries, the pre-emptive test on the (whole: < 100 rows)
    # images, the oracle's per-pair lists
    pairs, batch_end = oracle.enumerate_brute(n_img, 100)
    assert len(pairs) == n_img * (n_img - 1) // 2 and (np.diff(np.concatenate([[0], batch_end])) <= 100).all()
    assert (np.diff(np.concatenate([[0], batch_end])) == 100).sum() >= 4     # rows 100..103 flush at 100 pairs
    offs, q, t, _ = oracle.match_pairs(descs, pairs, 0.8, True, np.inf, nthreads=8)   # pre-emptive: no distance cut
    keep = np.diff(offs) >= 4
    o2, q2, t2, _ = oracle.match_pairs(descs, pairs, 0.8, True, 0.7, nthreads=8)
    assert 0.2 < keep.mean() < 0.98, "test data: the pre-emptive filter should drop some pairs, not all"
    groups, start = [], 0
    for e in batch_end:
        groups.append([p for p in range(start, int(e)) if keep[p]])
        start = int(e)
    outs = {}
    for name, env in (("small", {"MSFM_SUPER_BATCH_PAIRS": "250"}), ("default", {})):
        db_path = str(tmp_path / (name + ".db"))
        database.write_synthetic_database(db_path, descs, kps)
        cfg = tmp_path / (name + ".yaml")
        cfg.write_text(YAML.format(db=db_path, mt=1))
        env = dict(env, MSFM_GEOMETRIC_VERIFICATION="0", MSFM_TRACE_TRANSACTIONS="1")
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([exe, str(cfg)], capture_output=True, text=True, env=e, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        txn = [int(x) for x in re.findall(r"\[msfm txn\] (\d+)", r.stderr)]
        assert txn == [len(g) for g in groups], name                       # one transaction per reference group
        seq = [(int(a), int(b_)) for a, b_ in re.findall(r"Compute Matches (\d+) - (\d+) \.\.\. ", r.stdout)]
        assert seq == [tuple(int(x) for x in pairs[p]) for g in groups for p in g], name
        db = database.Database(db_path)
        assert db.db.execute("SELECT COUNT(*) FROM matches").fetchone()[0] == int(keep.sum())
        rows = db.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
        db.Close()
        outs[name] = rows
        by_id = {r_[0]: r_ for r_ in rows}
        for p in np.nonzero(keep)[0]:
            i, j = int(pairs[p][0]), int(pairs[p][1])
            m = np.stack([q2[o2[p]:o2[p + 1]], t2[o2[p]:o2[p + 1]]], 1).reshape(-1, 2)
            row = by_id[database.ImagePairToPairId(i, j)]
            assert row[1] == len(m) and row[2] == 2
            assert np.array_equal(np.frombuffer(row[3], np.int32).reshape(-1, 2), m[:, ::-1]), (i, j)   # i > j: columns swapped
    assert outs["small"] == outs["default"]


def test_bulk_load_u8_side_table_and_scene_graph_order(exe, oracle, tmp_path):
    """SURVEY 8f-2 / 8f-4 through the CLI: (1) the bulk loader gives the same table as the per-image reads
    (MSFM_BULK_LOAD=0); (2) with a `descriptors_u8` side table the matcher uploads the bytes (integer descriptors: the
    rows equal the exact-integer reference's); (3) MSFM_SCENEGRAPH_ORDER=1 stores every row sorted by column 0 with the
    same match set, default rows are byte-identical to the reference order."""
    import ctypes as C
    from oracle import int_oracle as io
    sizes = [300, 257, 64, 512, 130, 40]
    u = synth.u8_images(len(sizes), sizes, seed=77, dup_frac=0.3, as_float=False)
    kps = [synth.keypoints(len(d), seed=900 + i) for i, d in enumerate(u)]
    host = C.CDLL(os.path.join(HOST, "libmsfm_host.so"))
    host.host_db_write_descriptors_u8.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int]

    def make_db(name, side_table):
        path = str(tmp_path / (name + ".db"))
        # the float table holds HALF the value where a side table exists: a run that ignored the side table would differ
        floats = [(x.astype(np.float32) * (0.5 if side_table else 1.0)) for x in u]
        database.write_synthetic_database(path, floats, kps)
        if side_table:
            for i, x in enumerate(u):
                host.host_db_write_descriptors_u8(path.encode(), i, x.ctypes.data_as(C.c_void_p) if len(x) else None, len(x), 128)
        cfg = tmp_path / (name + ".yaml")
        # the reference's max_distance 0.7 suits unit-norm descriptors; these are 0..255 integers: the YAML carries a wide one
        # and MSFM_HONOUR_YAML_MATCH_PARAMS=1 makes the CLI use it (the reference itself ignores the three values)
        cfg.write_text('%YAML:1.0\ndatabase_path : "{}"\nSIFTmatch.match_type : 0\nSIFTmatch.max_distance : 1000000000.0\n'
                       'SIFTmatch.distance_ratio : 0.8\nSIFTmatch.cross_check : 1\n'.format(path))
        return path, cfg

    def rows_of(path):
        db = database.Database(path)
        r = db.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
        db.Close()
        return r

    env = {"MSFM_GEOMETRIC_VERIFICATION": "0", "MSFM_HONOUR_YAML_MATCH_PARAMS": "1"}
    p_bulk, c_bulk = make_db("bulk", False)
    p_single, c_single = make_db("single", False)
    p_u8, c_u8 = make_db("u8", True)
    p_ord, c_ord = make_db("ord", False)
    run_cli(exe, c_bulk, env)
    run_cli(exe, c_single, dict(env, MSFM_BULK_LOAD="0"))
    out_u8 = run_cli(exe, c_u8, dict(env, MSFM_USE_DESCRIPTORS_U8="1"))
    assert "descriptors_u8 side table" in out_u8            # the switch to byte descriptors is announced
    # the side table is OPT-IN: without the switch a (possibly stale / foreign) table is ignored -- the halved floats rule
    p_off, c_off = make_db("u8_off", True)
    out_off = run_cli(exe, c_off, env)
    assert "descriptors_u8" not in out_off
    run_cli(exe, c_ord, dict(env, MSFM_SCENEGRAPH_ORDER="1"))
    r_bulk, r_single, r_u8, r_ord = rows_of(p_bulk), rows_of(p_single), rows_of(p_u8), rows_of(p_ord)
    assert len(r_bulk) > 3 and r_bulk == r_single          # (1)
    assert r_u8 == r_bulk                                   # (2): the bytes of the side table, not the halved floats
    # (halving every value scales all distances by 0.5: the same matches survive ratio + cross-check, so compare via the oracle)
    db_off = database.Database(p_off)
    i0, j0 = 1, 0
    qh, th, _ = io.match_pair(u[i0], u[j0], 0.8, True, 1e9)
    assert np.array_equal(db_off.ReadMatches(i0, j0), np.stack([qh, th], 1).reshape(-1, 2))   # same indices (scale-invariant tests)
    db_off.Close()
    # against the exact-integer reference, over the sequential mode's pair list (no pre-emptive filter there)
    pairs, _ = oracle.enumerate_sequential(len(sizes), 3)
    assert len(r_bulk) == len(pairs)
    db = database.Database(p_u8)
    for i, j in pairs:
        q, t, _ = io.match_pair(u[int(i)], u[int(j)], 0.8, True, 1e9)
        assert np.array_equal(db.ReadMatches(int(i), int(j)), np.stack([q, t], 1).reshape(-1, 2)), (i, j)
    db.Close()
    # (3) same match sets, column 0 ascending
    assert [x[0] for x in r_ord] == [x[0] for x in r_bulk]
    for a, b_ in zip(r_bulk, r_ord):
        ma = np.frombuffer(a[3], np.int32).reshape(-1, 2)
        mb = np.frombuffer(b_[3], np.int32).reshape(-1, 2)
        assert {tuple(x) for x in ma} == {tuple(x) for x in mb}
        assert (np.diff(mb[:, 0]) > 0).all()


@pytest.mark.parametrize("devices", ["", "0,0"])
def test_key_order_emission_writes_the_same_rows_and_prints_the_same_lines(exe, dataset, tmp_path, devices):
    """MSFM_EMIT_ORDER=pair_id (opt-in): the pairs are computed and their rows written in ascending pair_id -- SQLite appends instead of
    rebalancing between two full leaves for every row of the reference's interleaved order (host/FeatureMatching.cpp) -- in
    transactions of 100 rows; the stdout lines are the reference's, in the reference's order.  Same table, same text (also on a
    partially filled database: the "Existing, Continue!" lines keep their places)."""
    descs, kps = dataset
    a, b = str(tmp_path / "ref_order.db"), str(tmp_path / "key_order.db")
    database.write_synthetic_database(a, descs, kps)
    shutil.copy(a, b)
    strip = lambda s: re.sub(r"Elapsed time: [0-9.]+", "Elapsed time: X", s)
    base = {"MSFM_GEOMETRIC_VERIFICATION": "0", "MSFM_TRACE_TRANSACTIONS": "1"}
    if devices:
        base["MSFM_DEVICES"] = devices
    outs, txns = [], []
    for path, env in ((a, {}), (b, {"MSFM_EMIT_ORDER": "pair_id"})):
        cfg = tmp_path / (os.path.basename(path) + ".yaml")
        cfg.write_text(YAML.format(db=path, mt=1))
        e = dict(os.environ)
        e.update(base)
        e.update(env)
        r = subprocess.run([exe, str(cfg)], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
        txns.append([int(x) for x in re.findall(r"\[msfm txn\] (\d+)", r.stderr)])
    da, dbb = database.Database(a), database.Database(b)
    ra = da.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    rb = dbb.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    assert len(ra) > 5 and ra == rb
    assert strip(outs[0]) == strip(outs[1])
    assert sum(txns[1]) == len(rb) and all(t == 100 for t in txns[1][:-1])      # key order: 100 rows per transaction
    # resume: drop a third of the rows of both, run again -- only those are recomputed, the lines say which
    for d in (da, dbb):
        d.db.execute("DELETE FROM matches WHERE pair_id % 3 = 1")
        d.db.commit()
    da.Close()
    dbb.Close()
    outs2 = []
    for path, env in ((a, {}), (b, {"MSFM_EMIT_ORDER": "pair_id"})):
        cfg = tmp_path / (os.path.basename(path) + ".yaml")
        outs2.append(run_cli(exe, cfg, dict(base, **env)))
    assert "Existing, Continue!" in outs2[0] and strip(outs2[0]) == strip(outs2[1])
    da, dbb = database.Database(a), database.Database(b)
    ra2 = da.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    rb2 = dbb.db.execute("SELECT pair_id, rows, cols, data FROM matches ORDER BY pair_id").fetchall()
    da.Close()
    dbb.Close()
    assert ra2 == rb2 == ra


@pytest.mark.parametrize("order", ["reference", "pair_id"])
def test_killed_run_leaves_whole_rows_and_the_rerun_completes_the_table(exe, tmp_path, order):
    """Rows are the checkpoint of an interrupted run (FeatureMatching.cpp:23-27, 63-72).  The emitter writes a group per transaction
    while the device threads are ahead of it: a process killed (SIGKILL) in the middle leaves a database that opens, holds only whole
    rows -- each byte-identical to the uninterrupted run's -- and the rerun skips those and writes the rest: the same table in the end."""
    import signal
    n_img = 48
    clean, cut = str(tmp_path / "clean.db"), str(tmp_path / "cut.db")
    synth.south_building_database(clean, n_img, 500, seed=77)
    shutil.copy(clean, cut)
    for ext in ("-wal", "-shm"):
        if os.path.exists(clean + ext):
            shutil.copy(clean + ext, cut + ext)
    env = dict(os.environ, MSFM_GEOMETRIC_VERIFICATION="0")
    if order == "pair_id":
        env["MSFM_EMIT_ORDER"] = "pair_id"
    cfg_clean, cfg_cut = tmp_path / "clean.yaml", tmp_path / "cut.yaml"
    cfg_clean.write_text(YAML.format(db=clean, mt=1))
    cfg_cut.write_text(YAML.format(db=cut, mt=1))
    r = subprocess.run([exe, str(cfg_clean)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = database.Database(clean)
    want = dict((row[0], row[1:]) for row in d.db.execute("SELECT pair_id, rows, cols, data FROM matches"))
    d.Close()
    assert len(want) > 600
    # the interrupted run: killed once a third of the pairs have been announced on stdout
    p = subprocess.Popen([exe, str(cfg_cut)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
    seen = 0
    for line in p.stdout:
        if line.startswith("Compute Matches"):
            seen += 1
            if seen >= len(want) // 3:
                break
    p.send_signal(signal.SIGKILL)
    p.stdout.close()
    p.wait()
    d = database.Database(cut)
    part = dict((row[0], row[1:]) for row in d.db.execute("SELECT pair_id, rows, cols, data FROM matches"))
    d.Close()
    assert 0 < len(part) < len(want), (len(part), len(want))
    for k, v in part.items():
        assert want[k] == v
    out = run_cli(exe, cfg_cut, {"MSFM_GEOMETRIC_VERIFICATION": "0", **({"MSFM_EMIT_ORDER": "pair_id"} if order == "pair_id" else {})})
    assert out.count("Existing, Continue!") == len(part)
    d = database.Database(cut)
    full = dict((row[0], row[1:]) for row in d.db.execute("SELECT pair_id, rows, cols, data FROM matches"))
    d.Close()
    assert full == want
