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


def rootsift_images(n_images, n_desc, seed=1234, n_proto=20000, sigma=0.05, overlap=0.5):
    """List of n_images float32 arrays [n_i, 128] with unit L2 norm, values in [0, 1].

    ``n_desc`` may be an int or a per-image sequence.  Every image draws ``overlap`` of its
    rows from the shared prototype pool (with jitter sigma on the pre-normalised histogram,
    relative to its mean) and the rest from fresh random histograms.
    """
    rng = np.random.default_rng(seed)
    if np.isscalar(n_desc):
        n_desc = [int(n_desc)] * n_images
    # SIFT-like gradient histograms: sparse-ish non-negative values
    proto = rng.gamma(shape=0.6, scale=1.0, size=(n_proto, 128)).astype(F32)
    out = []
    for i in range(n_images):
        n = int(n_desc[i])
        n_shared = min(int(round(n * overlap)), n_proto)
        pick = rng.choice(n_proto, size=n_shared, replace=False)
        base = proto[pick]
        jit = base * (1.0 + sigma * rng.standard_normal(base.shape).astype(F32))
        fresh = rng.gamma(shape=0.6, scale=1.0, size=(n - n_shared, 128)).astype(F32)
        d = np.concatenate([jit, fresh], axis=0)
        d = d[rng.permutation(n)]
        out.append(np.ascontiguousarray(_l1_root(d)))
    return out


def u8_images(n_images, n_desc, seed=1329, dup_frac=0.05, as_float=True):
    """Integer-valued descriptors in 0..255 (uint8, or float32 holding the same integers)."""
    rng = np.random.default_rng(seed)
    if np.isscalar(n_desc):
        n_desc = [int(n_desc)] * n_images
    n_pool = max(int(max(n_desc) * dup_frac), 1)
    pool = np.clip(np.rint(np.abs(rng.normal(0.0, 48.0, size=(n_pool, 128)))), 0, 255)
    out = []
    for i in range(n_images):
        n = int(n_desc[i])
        d = np.clip(np.rint(np.abs(rng.normal(0.0, 48.0, size=(n, 128)))), 0, 255)
        k = min(n_pool, n)
        rows = rng.choice(n, size=k, replace=False)
        noise = np.rint(rng.normal(0.0, 2.0, size=(k, 128)))
        d[rows] = np.clip(pool[:k] + noise, 0, 255)
        d = d.astype(np.uint8)
        out.append(np.ascontiguousarray(d.astype(F32) if as_float else d))
    return out


def keypoints(n, seed=0, width=3072, height=2304):
    """n x 4 float32 keypoint rows (x, y, size, angle)."""
    rng = np.random.default_rng(seed)
    k = np.empty((n, 4), F32)
    k[:, 0] = rng.uniform(0, width, n)
    k[:, 1] = rng.uniform(0, height, n)
    k[:, 2] = rng.gamma(2.0, 2.0, n) + 1.0
    k[:, 3] = rng.uniform(0, 360, n)
    return k


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
    -> (descriptors, keypoints)"""
    from . import database
    rng = np.random.default_rng(seed)
    counts = rng.integers(int(n_desc * 0.92), int(n_desc * 1.08) + 1, n_images)
    descs = rootsift_images(n_images, counts.tolist(), seed=seed, n_proto=20000)
    kps = [keypoints(len(d), seed=50 + i) for i, d in enumerate(descs)]
    pool = descs[0][:120].copy()
    for i in range(n_images):
        k = min(80, len(descs[i]))
        pick = rng.choice(120, k, replace=False)
        rows = rng.choice(len(descs[i]), k, replace=False)
        v = np.abs(pool[pick] * (1 + 0.03 * rng.standard_normal((k, 128)).astype(F32)))
        descs[i][rows] = v / np.linalg.norm(v, axis=1, keepdims=True)
        kps[i][rows, 2] = 100 + rng.uniform(0, 50, k).astype(F32)
    database.write_synthetic_database(path, descs, kps)
    return descs, kps

