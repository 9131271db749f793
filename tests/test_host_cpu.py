"""Host-side pieces of the drop-in that need no GPU: YAML subset reader, C++ Database against the
Python twin (same file format both ways), F-matrix RANSAC, and the CLI's argv/exit contract
(sfm/ComputeMatches.cpp:15-30)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "monocularsfm_amd", "host")


@pytest.fixture(scope="module")
def host(built_lib):
    subprocess.check_call(["make", "-C", HOST, "-s"])
    L = C.CDLL(os.path.join(HOST, "libmsfm_host.so"))
    L.host_yaml_get_double.restype = C.c_double
    L.host_yaml_get_double.argtypes = [C.c_char_p, C.c_char_p, C.c_double]
    return L


YAML = """%YAML:1.0

images_path : "/data/south-building/images"
database_path : "{db}"

# 0 for sequential match, 1 for brute match
SIFTmatch.match_type :  {mt}
SIFTmatch.max_distance : 0.6   # trailing comment
SIFTmatch.distance_ratio : 0.75
SIFTmatch.cross_check : 0
Reconstruction.Camera.fx: 2559.68
"""


def test_yaml_subset(host, tmp_path):
    p = tmp_path / "cfg.yaml"
    p.write_text(YAML.format(db="/tmp/x y#z.db", mt=0))
    path = str(p).encode()
    assert host.host_yaml_is_opened(path) == 1
    buf = C.create_string_buffer(512)
    assert host.host_yaml_get_string(path, b"database_path", buf, 512) == 1
    assert buf.value == b"/tmp/x y#z.db"            # '#' inside quotes is not a comment
    assert host.host_yaml_get_int(path, b"SIFTmatch.match_type", 1) == 0
    assert host.host_yaml_get_double(path, b"SIFTmatch.max_distance", 0.7) == 0.6
    assert host.host_yaml_get_double(path, b"SIFTmatch.distance_ratio", 0.8) == 0.75
    assert host.host_yaml_get_bool(path, b"SIFTmatch.cross_check", 1) == 0
    assert host.host_yaml_get_double(path, b"Reconstruction.Camera.fx", 0) == 2559.68   # "key: value" without space
    assert host.host_yaml_get_int(path, b"missing.key", 7) == 7                          # missing keeps the default
    assert host.host_yaml_get_string(path, b"missing.key", buf, 512) == 0 and buf.value == b""
    bad = tmp_path / "bad.yaml"
    bad.write_text("database_path : x\n")            # no %YAML directive: FileStorage refuses it
    assert host.host_yaml_is_opened(str(bad).encode()) == 0
    assert host.host_yaml_is_opened(b"/nonexistent/file.yaml") == 0


def test_reference_configs_parse(host):
    ref = "/root/reference/config/south-building.yaml"
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present (GPU box)")
    assert host.host_yaml_is_opened(ref.encode()) == 1
    assert host.host_yaml_get_int(ref.encode(), b"SIFTmatch.match_type", 1) == 0
    buf = C.create_string_buffer(512)
    host.host_yaml_get_string(ref.encode(), b"database_path", buf, 512)
    assert buf.value.endswith(b"south-building.db")


def test_database_cpp_and_python_agree(host, tmp_path):
    from monocularsfm_amd import database, synth
    db_path = str(tmp_path / "t.db")
    descs = synth.rootsift_images(3, [40, 0, 17], seed=5, n_proto=100)
    kps = [synth.keypoints(len(d), seed=i) for i, d in enumerate(descs)]
    database.write_synthetic_database(db_path, descs, kps)
    p = db_path.encode()
    assert host.host_db_num_images(p) == 3
    for i, d in enumerate(descs):
        out = np.zeros(max(d.size, 1), np.float32)
        cols = C.c_int()
        rows = host.host_db_read_descriptors(p, i, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(cols))
        assert rows == len(d) and cols.value == 128
        assert np.array_equal(out[:d.size].reshape(-1, 128), d)
        ko = np.zeros(max(len(d) * 4, 1), np.float32)
        assert host.host_db_read_keypoints(p, i, ko.ctypes.data_as(C.POINTER(C.c_float)), ko.size) == len(d)
        assert np.array_equal(ko[:len(d) * 4].reshape(-1, 4), kps[i])
    # C++ writes matches for (2, 0) [id1 > id2 -> columns swapped on disk]; Python reads them both ways
    qt = np.array([[3, 9], [5, 1], [16, 39]], np.int32)
    assert host.host_db_exist_matches(p, 2, 0) == 0
    host.host_db_write_matches(p, 2, 0, qt.ctypes.data_as(C.POINTER(C.c_int)), len(qt))
    assert host.host_db_exist_matches(p, 2, 0) == 1 and host.host_db_exist_matches(p, 0, 2) == 1
    db = database.Database(db_path)
    assert np.array_equal(db.ReadMatches(2, 0), qt)
    assert np.array_equal(db.ReadMatches(0, 2), qt[:, ::-1])
    (pid, stored), = db.ReadAllMatches()
    assert pid == 2 == host.host_pair_id(2, 0) and np.array_equal(stored, qt[:, ::-1])  # col 0 = smaller image id
    # a rows=0 result still gets a row (resume marker) but is invisible to ReadAllMatches
    db.WriteMatches(1, 0, np.zeros((0, 2), np.int32))
    assert db.ExistMatches(0, 1) and len(db.ReadAllMatches()) == 1
    db.Close()
    # ... and its data column is a zero-length BLOB, not NULL, from both writers (the reference binds malloc(0))
    host.host_db_write_matches(p, 2, 1, qt.ctypes.data_as(C.POINTER(C.c_int)), 0)
    db = database.Database(db_path)
    kinds = db.db.execute("SELECT pair_id, rows, typeof(data), length(data) FROM matches WHERE rows = 0 ORDER BY pair_id").fetchall()
    assert kinds == [(database.ImagePairToPairId(1, 0), 0, "blob", 0), (database.ImagePairToPairId(2, 1), 0, "blob", 0)]
    db.Close()
    back = np.zeros((8, 2), np.int32)
    assert host.host_db_read_matches(p, 0, 2, back.ctypes.data_as(C.POINTER(C.c_int)), 8) == 3
    assert np.array_equal(back[:3], qt[:, ::-1])
    assert host.host_db_read_matches(p, 1, 0, back.ctypes.data_as(C.POINTER(C.c_int)), 8) == 0


def _two_views(n_in, n_out, seed):
    rng = np.random.default_rng(seed)
    X = np.c_[rng.uniform(-2, 2, n_in), rng.uniform(-1.5, 1.5, n_in), rng.uniform(4, 9, n_in)]
    K = np.array([[2559.68, 0, 1536], [0, 2559.68, 1152], [0, 0, 1]])
    a = 0.12
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([0.8, 0.05, 0.1])
    x1 = (K @ X.T).T
    x1 = x1[:, :2] / x1[:, 2:]
    x2 = (K @ (R @ X.T + t[:, None])).T
    x2 = x2[:, :2] / x2[:, 2:]
    x1 += rng.normal(0, 0.4, x1.shape)
    x2 += rng.normal(0, 0.4, x2.shape)
    o1 = np.c_[rng.uniform(0, 3072, n_out), rng.uniform(0, 2304, n_out)]
    o2 = np.c_[rng.uniform(0, 3072, n_out), rng.uniform(0, 2304, n_out)]
    p1 = np.r_[x1, o1].astype(np.float32)
    p2 = np.r_[x2, o2].astype(np.float32)
    perm = rng.permutation(len(p1))
    truth = np.r_[np.ones(n_in, bool), np.zeros(n_out, bool)][perm]
    return np.ascontiguousarray(p1[perm]), np.ascontiguousarray(p2[perm]), truth


def test_fundamental_ransac_recovers_the_epipolar_inliers(host):
    p1, p2, truth = _two_views(300, 120, seed=3)
    mask = np.zeros(len(p1), np.uint8)
    fp = C.POINTER(C.c_float)
    n = host.host_fundamental_ransac(p1.ctypes.data_as(fp), p2.ctypes.data_as(fp), len(p1), mask.ctypes.data_as(C.POINTER(C.c_ubyte)))
    assert n == len(p1)
    got = mask.astype(bool)
    assert (got & truth).sum() >= 0.95 * truth.sum()          # keeps the true correspondences
    assert (got & ~truth).sum() <= 0.1 * (~truth).sum()       # random outliers rarely sit on an epipolar line
    # deterministic
    mask2 = np.zeros_like(mask)
    host.host_fundamental_ransac(p1.ctypes.data_as(fp), p2.ctypes.data_as(fp), len(p1), mask2.ctypes.data_as(C.POINTER(C.c_ubyte)))
    assert np.array_equal(mask, mask2)
    # findFundamentalMat's small-count cases: < 7 -> no model (nothing kept), == 7 -> all ones
    assert host.host_fundamental_ransac(p1.ctypes.data_as(fp), p2.ctypes.data_as(fp), 6, mask.ctypes.data_as(C.POINTER(C.c_ubyte))) == 0
    mask[:] = 0
    assert host.host_fundamental_ransac(p1.ctypes.data_as(fp), p2.ctypes.data_as(fp), 7, mask.ctypes.data_as(C.POINTER(C.c_ubyte))) == 7
    assert mask[:7].all()


def test_synthetic_scene_keypoints_have_the_geometry_the_verification_keeps(host):
    """synth.scene_cameras / scene_keypoints (round 6: the bench's end_to_end database): rows that observe the same scene point in two
    images are a true epipolar correspondence (+ 0.7 px noise) -- FeatureUtils::FilterMatches' twin keeps them and drops the rows that
    observe nothing (random positions), which is what makes the write phase of the end-to-end figure a representative one."""
    from monocularsfm_amd import synth
    rng = np.random.default_rng(12)
    n_points, n = 400, 900
    ids = []
    for _ in range(2):
        a = np.full(n, -1, np.int64)
        a[rng.choice(n, 300, replace=False)] = rng.choice(n_points, 300, replace=False)
        ids.append(a)
    cams = synth.scene_cameras(2, seed=5)
    kps = synth.scene_keypoints(ids, cams, n_points, seed=6)
    assert all(k.shape == (n, 4) and k.dtype == np.float32 for k in kps)
    # the "matches": every scene point seen by both images (true), plus as many pairs of rows that observe nothing (false)
    both = np.intersect1d(ids[0][ids[0] >= 0], ids[1][ids[1] >= 0])
    assert len(both) >= 150
    r0 = {int(v): i for i, v in enumerate(ids[0]) if v >= 0}
    r1 = {int(v): i for i, v in enumerate(ids[1]) if v >= 0}
    true_q, true_t = [r0[int(v)] for v in both], [r1[int(v)] for v in both]
    free0, free1 = np.nonzero(ids[0] < 0)[0][:len(both)], np.nonzero(ids[1] < 0)[0][:len(both)]
    p1 = np.r_[kps[0][true_q, :2], kps[0][free0, :2]].astype(np.float32)
    p2 = np.r_[kps[1][true_t, :2], kps[1][free1, :2]].astype(np.float32)
    truth = np.r_[np.ones(len(both), bool), np.zeros(len(both), bool)]
    mask = np.zeros(len(p1), np.uint8)
    fp = C.POINTER(C.c_float)
    p1, p2 = np.ascontiguousarray(p1), np.ascontiguousarray(p2)
    assert host.host_fundamental_ransac(p1.ctypes.data_as(fp), p2.ctypes.data_as(fp), len(p1), mask.ctypes.data_as(C.POINTER(C.c_ubyte))) == len(p1)
    got = mask.astype(bool)
    assert (got & truth).sum() >= 0.9 * truth.sum(), (got & truth).sum()
    assert (got & ~truth).sum() <= 0.15 * (~truth).sum()


def test_fundamental_ransac_runs_its_iterations_on_unrelated_points(host):
    """A consensus of ~8 of 358 makes 1 - w^8 round to 1.0: the adaptive bound must then stay at the cap instead
    of collapsing (log(1) = 0 in the denominator).  With all 1000 hypotheses tried, some 8-point model always
    gathers >= 8 of 358 random correspondences within 3 px of its epipolar lines."""
    rng = np.random.default_rng(0)
    n = 358
    p1 = np.c_[rng.uniform(0, 3072, n), rng.uniform(0, 2304, n)].astype(np.float32)
    p2 = np.c_[rng.uniform(0, 3072, n), rng.uniform(0, 2304, n)].astype(np.float32)
    mask = np.zeros(n, np.uint8)
    fp = C.POINTER(C.c_float)
    assert host.host_fundamental_ransac(p1.ctypes.data_as(fp), p2.ctypes.data_as(fp), n, mask.ctypes.data_as(C.POINTER(C.c_ubyte))) == n
    assert 8 <= mask.sum() <= 40


def test_cli_argv_contract(host, tmp_path):
    exe = os.path.join(HOST, "ComputeMatches")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 255 and "You need specify the YAML file path!" in r.stdout      # exit(-1)
    r = subprocess.run([exe, "/nonexistent.yaml"], capture_output=True, text=True)
    assert r.returncode == 255 and "YAML file : /nonexistent.yaml can't not open!" in r.stdout


def test_cli_fails_loudly_without_gpu(host, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from monocularsfm_amd import database, synth
    db_path = str(tmp_path / "t.db")
    database.write_synthetic_database(db_path, synth.rootsift_images(2, [30, 30], seed=1, n_proto=50))
    cfg = tmp_path / "c.yaml"
    cfg.write_text(YAML.format(db=db_path, mt=1))
    r = subprocess.run([os.path.join(HOST, "ComputeMatches"), str(cfg)], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
    db = database.Database(db_path)
    assert len(db.ReadAllMatches()) == 0 and not db.ExistMatches(1, 0)   # nothing was written
    db.Close()


# ---- bulk loader, u8 side table ---------------------------------------------------------------------------

def _fnv1a(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_bulk_visitors_see_every_row_once_in_id_order(host, tmp_path):
    """One SELECT per table instead of one prepared-statement read per image (the reference: two per PAIR,
    /root/reference/src/Feature/FeatureMatching.cpp:32-33, src/Database/Database.cpp:482-523): same bytes."""
    from monocularsfm_amd import database, synth
    sizes = [40, 0, 17, 300, 1]
    descs = [np.ascontiguousarray(d, np.float32) for d in synth.rootsift_images(len(sizes), sizes, seed=5, n_proto=400)]
    kps = [synth.keypoints(len(d), seed=10 + i) for i, d in enumerate(descs)]
    path = str(tmp_path / "bulk.db")
    database.write_synthetic_database(path, descs, kps)
    host.host_db_visit_all.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_ulonglong), C.c_int]
    for which, arrays, cols_expected in ((0, descs, 128), (1, kps, 4)):
        ids, rows, cols = (np.zeros(16, np.int32) for _ in range(3))
        sums = np.zeros(16, np.uint64)
        n = host.host_db_visit_all(path.encode(), which, ids.ctypes.data_as(C.POINTER(C.c_int)), rows.ctypes.data_as(C.POINTER(C.c_int)),
                                   cols.ctypes.data_as(C.POINTER(C.c_int)), sums.ctypes.data_as(C.POINTER(C.c_ulonglong)), 16)
        assert n == len(sizes) and list(ids[:n]) == list(range(len(sizes))) and list(rows[:n]) == sizes
        for i in range(n):
            assert int(sums[i]) == _fnv1a(np.ascontiguousarray(arrays[i], np.float32).tobytes())
            assert rows[i] == 0 or cols[i] == cols_expected
    # no side table yet
    assert host.host_db_visit_all(path.encode(), 2, None, None, None, None, 0) == -1


def test_u8_side_table_round_trip(host, tmp_path):
    from monocularsfm_amd import database, synth
    u = synth.u8_images(3, [50, 33, 0], seed=6, as_float=False)
    path = str(tmp_path / "u8.db")
    database.write_synthetic_database(path, [x.astype(np.float32) for x in u], None)
    host.host_db_write_descriptors_u8.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    for i, x in enumerate(u):
        host.host_db_write_descriptors_u8(path.encode(), i, x.ctypes.data_as(C.c_void_p) if len(x) else None, len(x), 128)
    host.host_db_visit_all.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_ulonglong), C.c_int]
    ids, rows, cols = (np.zeros(8, np.int32) for _ in range(3))
    sums = np.zeros(8, np.uint64)
    n = host.host_db_visit_all(path.encode(), 2, ids.ctypes.data_as(C.POINTER(C.c_int)), rows.ctypes.data_as(C.POINTER(C.c_int)),
                               cols.ctypes.data_as(C.POINTER(C.c_int)), sums.ctypes.data_as(C.POINTER(C.c_ulonglong)), 8)
    assert n == 3 and list(rows[:3]) == [50, 33, 0]
    for i in range(3):
        assert int(sums[i]) == _fnv1a(u[i].tobytes())
    # the reference's own tables are untouched: the Python twin still reads the float descriptors
    db = database.Database(path)
    assert np.array_equal(db.ReadDescriptors(0), u[0].astype(np.float32))
    db.Close()


# ---- SceneGraph-friendly emission ---------------------------------------------------------------------------

def scene_graph_add_correspondences(rows, num_keypoints):
    """SceneGraph::Load + AddCorrespondences (/root/reference/src/Reconstruction/SceneGraph.cpp:59-76, 170-251) on stored
    rows {pair_id: m x 2 (column 0 = index in the smaller image id)}: -> (corrs per (image, point), warnings, find_if steps)."""
    corrs = {}
    warnings = steps = 0
    for pair_id in sorted(rows):                                   # ReadAllMatches: primary-key order
        id2 = pair_id % 10000
        id1 = (pair_id - id2) // 10000
        for a, b in rows[pair_id]:
            if not (0 <= a < num_keypoints[id1] and 0 <= b < num_keypoints[id2]):
                warnings += 1
                continue
            lst = corrs.setdefault((id1, int(a)), [])
            steps += len(lst)                                       # the linear std::find_if
            if (id2, int(b)) in lst:
                warnings += 1
                continue
            lst.append((id2, int(b)))
            corrs.setdefault((id2, int(b)), []).append((id1, int(a)))
    return corrs, warnings, steps


def test_emission_default_is_identity_and_scene_graph_order_sorts_column_zero(host):
    rng = np.random.default_rng(3)
    host.host_apply_emission.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    host.host_check_row_contract.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
    q = np.sort(rng.choice(500, 120, replace=False)).astype(np.int32)       # a matcher list: ascending distinct queryIdx
    t = rng.choice(400, 120, replace=False).astype(np.int32)                # cross-checked: distinct trainIdx
    qt = np.ascontiguousarray(np.stack([q, t], 1))
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    assert host.host_check_row_contract(p(qt), 120, 500, 400) == 0
    same = qt.copy()
    assert host.host_apply_emission(7, 3, 0, 0, p(same), 120) == 120 and np.array_equal(same, qt)      # default: untouched
    srt = qt.copy()
    assert host.host_apply_emission(7, 3, 1, 0, p(srt), 120) == 120
    # id1 = 7 > id2 = 3: column 0 of the stored row is the trainIdx
    assert (np.diff(srt[:, 1]) > 0).all() and {tuple(x) for x in srt} == {tuple(x) for x in qt}
    srt2 = qt.copy()
    assert host.host_apply_emission(3, 7, 1, 0, p(srt2), 120) == 120 and np.array_equal(srt2, qt)       # already by queryIdx
    few = qt[:5].copy()
    assert host.host_apply_emission(7, 3, 0, 16, p(few), 5) == 0                                          # below min_num_matches
    # the contract check sees what AddCorrespondences would warn about
    bad = qt.copy()
    bad[3] = bad[2]
    assert host.host_check_row_contract(p(bad), 120, 500, 400) == 2
    assert host.host_check_row_contract(p(qt), 120, 100, 400) == 1


def test_scene_graph_consumer_sees_the_same_graph_with_fewer_steps(host):
    """The consumer emulation on reference-order rows and on MSFM_SCENEGRAPH_ORDER rows: identical correspondences, no
    warnings, and the ordered rows never pay for a find_if longer than the unordered ones."""
    rng = np.random.default_rng(9)
    host.host_apply_emission.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    nk = {i: 300 for i in range(5)}
    rows_ref, rows_ord = {}, {}
    for i in range(5):
        for j in range(i):
            m = int(rng.integers(20, 120))
            q = np.sort(rng.choice(300, m, replace=False)).astype(np.int32)
            t = rng.choice(300, m, replace=False).astype(np.int32)
            for order, store in ((0, rows_ref), (1, rows_ord)):
                qt = np.ascontiguousarray(np.stack([q, t], 1))
                n = host.host_apply_emission(i, j, order, 0, qt.ctypes.data_as(C.POINTER(C.c_int)), m)
                store[10000 * j + i] = qt[:n, ::-1].copy()          # i > j: WriteMatches swaps the columns
    g_ref, w_ref, s_ref = scene_graph_add_correspondences(rows_ref, nk)
    g_ord, w_ord, s_ord = scene_graph_add_correspondences(rows_ord, nk)
    assert w_ref == w_ord == 0
    assert {k: sorted(v) for k, v in g_ref.items()} == {k: sorted(v) for k, v in g_ord.items()}
    assert s_ord == s_ref                                           # same work, sequential instead of scattered access
    for pid, r in rows_ord.items():
        assert (np.diff(r[:, 0]) > 0).all()
