"""Seeded synthetic descriptor / keypoint generators for the ComputeMatches hot path.

Shapes follow SURVEY.md section 8(d); nothing here comes from reference code or data.

* ``rootsift_images``  -- "clustered RootSIFT": a shared pool of unit-norm, non-negative
  prototypes (what FeatureExtraction's L1-root normalisation produces,
  /root/reference/src/Feature/FeatureExtraction.cpp:260-270) plus per-image jitter, so that
  the Lowe ratio test, the cross-check and max_distance=0.7 all have something to keep.
* ``u8_images``        -- integer-valued SIFT-like descriptors, i.i.d. round(clip(|N(0,48^2)|,0,255)),
  with a planted near-duplicate subset so matches are non-empty (BASELINE configs 4-5).
* ``keypoints``        -- n x 4 float32 (x, y, size, angle) rows like Database keypoint blobs
  (/root/reference/src/Database/Database.cpp:114-126).
"""
import numpy as np

F32 = np.float32


def _l1_root(x):
    """L1RootNormalized: divide by the L1 norm, then elementwise sqrt -> unit L2 norm."""
    x = np.abs(x)
    s = x.sum(axis=1, keepdims=True)
    s[s == 0] = 1.0
    return np.sqrt(x / s).astype(F32)


def rootsift_images(n_images, n_desc, seed=1234, n_proto=20000, sigma=0.05, overlap=0.5, return_proto=False):
    """List of n_images float32 arrays [n_i, 128] with unit L2 norm, values in [0, 1].

    ``n_desc`` may be an int or a per-image sequence.  Every image draws ``overlap`` of its
    rows from the shared prototype pool (with jitter sigma on the pre-normalised histogram,
    relative to its mean) and the rest from fresh random histograms.
    ``return_proto``: also the prototype index of every row (-1: a fresh row) -- the scene point a row observes
    (``scene_keypoints``); the descriptors are the same with and without it.
    """
    rng = np.random.default_rng(seed)
    if np.isscalar(n_desc):
        n_desc = [int(n_desc)] * n_images
    # SIFT-like gradient histograms: sparse-ish non-negative values
    proto = rng.gamma(shape=0.6, scale=1.0, size=(n_proto, 128)).astype(F32)
    out, protos = [], []
    for i in range(n_images):
        n = int(n_desc[i])
        n_shared = min(int(round(n * overlap)), n_proto)
        pick = rng.choice(n_proto, size=n_shared, replace=False)
        base = proto[pick]
        jit = base * (1.0 + sigma * rng.standard_normal(base.shape).astype(F32))
        fresh = rng.gamma(shape=0.6, scale=1.0, size=(n - n_shared, 128)).astype(F32)
        d = np.concatenate([jit, fresh], axis=0)
        perm = rng.permutation(n)
        d = d[perm]
        out.append(np.ascontiguousarray(_l1_root(d)))
        protos.append(np.concatenate([pick, np.full(n - n_shared, -1, np.int64)])[perm])
    return (out, protos) if return_proto else out


def u8_images(n_images, n_desc, seed=1329, dup_frac=0.05, as_float=True, return_planted=False):
    """Integer-valued descriptors in 0..255 (uint8, or float32 holding the same integers).
    ``return_planted``: also, per image, the rows that carry the planted near-duplicates (entry r = the row of pool descriptor r)."""
    rng = np.random.default_rng(seed)
    if np.isscalar(n_desc):
        n_desc = [int(n_desc)] * n_images
    n_pool = max(int(max(n_desc) * dup_frac), 1)
    pool = np.clip(np.rint(np.abs(rng.normal(0.0, 48.0, size=(n_pool, 128)))), 0, 255)
    out, planted = [], []
    for i in range(n_images):
        n = int(n_desc[i])
        d = np.clip(np.rint(np.abs(rng.normal(0.0, 48.0, size=(n, 128)))), 0, 255)
        k = min(n_pool, n)
        rows = rng.choice(n, size=k, replace=False)
        noise = np.rint(rng.normal(0.0, 2.0, size=(k, 128)))
        d[rows] = np.clip(pool[:k] + noise, 0, 255)
        d = d.astype(np.uint8)
        out.append(np.ascontiguousarray(d.astype(F32) if as_float else d))
        planted.append(rows)
    return (out, planted) if return_planted else out


def keypoints(n, seed=0, width=3072, height=2304):
    """n x 4 float32 keypoint rows (x, y, size, angle)."""
    rng = np.random.default_rng(seed)
    k = np.empty((n, 4), F32)
    k[:, 0] = rng.uniform(0, width, n)
    k[:, 1] = rng.uniform(0, height, n)
    k[:, 2] = rng.gamma(2.0, 2.0, n) + 1.0
    k[:, 3] = rng.uniform(0, 360, n)
    return k


def scene_cameras(n_images, seed=0, width=3072, height=2304, focal=2500.0):
    """n_images pinhole cameras on an arc around the origin, all looking at it: (R [3,3], t [3], f, cx, cy) per image."""
    rng = np.random.default_rng(seed)
    cams = []
    for _ in range(n_images):
        yaw, pitch, roll = rng.uniform(-0.6, 0.6), rng.uniform(-0.15, 0.15), rng.uniform(-0.05, 0.05)
        cy_, sy_ = np.cos(yaw), np.sin(yaw)
        cp, sp = np.cos(pitch), np.sin(pitch)
        cr, sr = np.cos(roll), np.sin(roll)
        R = (np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1.0]]) @ np.array([[1.0, 0, 0], [0, cp, -sp], [0, sp, cp]]) @
             np.array([[cy_, 0, sy_], [0, 1.0, 0], [-sy_, 0, cy_]]))
        t = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), rng.uniform(5.5, 7.0)])
        cams.append((R, t, focal, width / 2.0, height / 2.0))
    return cams


def scene_keypoints(point_ids, cams, n_points, seed=0, noise_px=0.7, base=None):
    """Keypoints with a true epipolar geometry: rows whose ``point_ids`` entry is >= 0 observe that shared 3-D point (a box around the
    origin) through their image's camera, + ``noise_px`` of Gaussian pixel noise -- what F-RANSAC (FeatureUtils::FilterMatches,
    /root/reference/src/Feature/FeatureUtils.cpp:176-206) keeps on real overlapping photographs; rows with -1 keep ``base``'s random
    positions (or get fresh ones).  -> list of n x 4 float32 (x, y, size, angle)."""
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-1.6, 1.6, n_points), rng.uniform(-1.1, 1.1, n_points), rng.uniform(-1.0, 1.0, n_points)], 1)
    out = []
    for i, ids in enumerate(point_ids):
        ids = np.asarray(ids)
        k = base[i].copy() if base is not None else keypoints(len(ids), seed=seed + 1000 + i)
        sel = np.nonzero(ids >= 0)[0]
        if len(sel):
            R, t, f, cx, cy = cams[i]
            Xc = X[ids[sel]] @ R.T + t
            k[sel, 0] = (f * Xc[:, 0] / Xc[:, 2] + cx + rng.normal(0, noise_px, len(sel))).astype(F32)
            k[sel, 1] = (f * Xc[:, 1] / Xc[:, 2] + cy + rng.normal(0, noise_px, len(sel))).astype(F32)
        out.append(k)
    return out


def all_pairs(n_images):
    """BruteFeatureMatcher::RunMatching's enumeration (FeatureMatching.cpp:110-139): (i, j), j < i, i-major."""
    i, j = np.tril_indices(int(n_images), -1)      # row-major over the lower triangle = i-major, j ascending
    return np.ascontiguousarray(np.stack([i, j], 1).astype(np.int32)).reshape(-1, 2)


def job(workload="south-building", n_images=None, n_desc=None, seed=1234):
    """The seeded jobs of BASELINE.json -> (images, pairs, description).

    south-building : configs[1] / [2]: n_images (default 128; 330 = Person-Hall-shaped) x 4600..5400 float32
                     RootSIFT-like descriptors, brute-force all pairs.  bench.py's N = 1 workload.
    synthetic-u8   : configs[3] / [4]: n_images x n_desc (default 8192) u8-valued descriptors; the full configs
                     have 1329 x 8192 and 4096 x 16384 -- benches and tests run seeded subsets of the image set."""
    rng = np.random.default_rng(seed)
    if workload == "south-building":
        n_images = n_images or 128
        counts = rng.integers(4600, 5401, n_images) if n_desc is None else np.full(n_images, n_desc)
        imgs = rootsift_images(n_images, counts.tolist(), seed=seed, n_proto=20000, sigma=0.05)
        name = "south-building-shaped synthetic: %d images x ~%d f32 RootSIFT-like desc, brute-force all pairs" % (
            n_images, int(np.mean(counts)))
    elif workload == "synthetic-u8":
        n_images = n_images or 64
        nd = n_desc or 8192
        imgs = u8_images(n_images, nd, seed=seed, as_float=False)
        name = "synthetic u8 descriptors: %d images x %d desc, brute-force all pairs" % (n_images, nd)
    else:
        raise ValueError("unknown workload " + workload)
    return imgs, all_pairs(n_images), name


def south_building_database(path, n_images=128, n_desc=5000, seed=1234):
    """A South-Building-shaped SQLite database as FeatureExtraction would leave it (BASELINE configs[1]; SURVEY.md 8(d): synthetic
    when the dataset is unavailable): n_images x ~n_desc float32 RootSIFT-like descriptors + keypoints, written through the build's
    Database twin.  A shared pool of "landmarks" carried by the largest keypoints makes the reference's pre-emptive test (top-100
    scales, >= 4 cross-matches; src/Feature/FeatureMatching.cpp:148-179) keep the pairs, as it does on real overlapping photographs.
    Round 6: every row drawn from the shared prototype pool (and every landmark) OBSERVES A SCENE POINT -- its keypoint is that
    point's projection through the image's camera + 0.7 px of noise (``scene_keypoints``) -- so the geometric verification keeps the
    true matches (hundreds per pair) instead of the minimal-sample consensus it found on uniformly random positions.
    -> (descriptors, keypoints)"""
    from . import database
    rng = np.random.default_rng(seed)
    n_proto = 20000
    counts = rng.integers(int(n_desc * 0.92), int(n_desc * 1.08) + 1, n_images)
    descs, protos = rootsift_images(n_images, counts.tolist(), seed=seed, n_proto=n_proto, return_proto=True)
    kps = [keypoints(len(d), seed=50 + i) for i, d in enumerate(descs)]
    pool = descs[0][:120].copy()
    for i in range(n_images):
        k = min(80, len(descs[i]))
        pick = rng.choice(120, k, replace=False)
        rows = rng.choice(len(descs[i]), k, replace=False)
        v = np.abs(pool[pick] * (1 + 0.03 * rng.standard_normal((k, 128)).astype(F32)))
        descs[i][rows] = v / np.linalg.norm(v, axis=1, keepdims=True)
        kps[i][rows, 2] = 100 + rng.uniform(0, 50, k).astype(F32)
        protos[i][rows] = n_proto + pick                 # landmarks are scene points too
    kps = scene_keypoints(protos, scene_cameras(n_images, seed=seed + 7), n_proto + 120, seed=seed + 9, base=kps)
    database.write_synthetic_database(path, descs, kps)
    return descs, kps


def u8_database(path, n_images=1329, n_desc=8192, seed=1329, f32_table=True, u8_table=True, progress=None):
    """A BASELINE-configs[3]-shaped database (1329 x 8192 raw byte SIFT-like descriptors) for the ComputeMatches EXECUTABLE at the
    scale the strong-scaling target is stated on: the images of ``job("synthetic-u8", ...)`` (same seed -> same descriptors), written as
    the reference's float32 `descriptors` table (integers 0..255: the library recognises byte images) and / or the `descriptors_u8`
    side table.  The planted near-duplicates observe scene points (``scene_keypoints``) and the first 100 of them carry the largest
    keypoint scales, so the pre-emptive filter keeps every pair and the geometric verification keeps the planted matches.
    Images are generated and written one at a time (the float table is 5.6 GB for the full config).  -> per-image row counts"""
    from . import database
    imgs, planted = u8_images(n_images, n_desc, seed=seed, as_float=False, return_planted=True)
    n_pool = max(len(r) for r in planted)
    cams = scene_cameras(n_images, seed=seed + 7)
    db = database.Database(path)
    if u8_table:
        db.db.execute("CREATE TABLE IF NOT EXISTS descriptors_u8 (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, "
                      "cols INTEGER NOT NULL, data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE)")
    db.BeginTransaction()
    for i, d in enumerate(imgs):
        ids = np.full(len(d), -1, np.int64)
        ids[planted[i]] = np.arange(len(planted[i]))
        k = scene_keypoints([ids], [cams[i]], n_pool, seed=seed + 9, base=[keypoints(len(d), seed=seed + 50 + i)])[0]
        k[planted[i][:100], 2] = 100 + 50.0 * np.arange(len(planted[i][:100]), 0, -1, dtype=F32) / 100.0
        db.WriteImage(i, "image_%05d.jpg" % i)
        db.WriteKeyPoints(i, k)
        if f32_table:
            db.WriteDescriptors(i, d.astype(F32))
        if u8_table:
            db.db.execute("INSERT INTO descriptors_u8 VALUES(?, ?, ?, ?);", (i, d.shape[0], d.shape[1], d.tobytes()))
        if progress and i % 100 == 0:
            progress(i)
    db.EndTransaction()
    db.Close()
    return [len(d) for d in imgs]
