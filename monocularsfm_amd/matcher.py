"""Host-side mirror of the reference's matcher interface, running on the gfx950 C ABI.

Same names, argument meaning and defaults as the reference:
  FeatureUtils.ComputeMatches / ComputeCrossMatches / FilterMatchesByDistance /
  ExtractTopScaleDescriptors      (/root/reference/include/Feature/FeatureUtils.h:94-108, src :68-96, :141-218)
  FeatureMatcher / SequentialFeatureMatcher / BruteFeatureMatcher
                                  (/root/reference/include/Feature/FeatureMatching.h:18-130, src :10-203)
Matches are returned as (queryIdx int32[m], trainIdx int32[m], distance float32[m]) -- the three
DMatch fields the reference uses (imgIdx is always 0).

The C++ host in monocularsfm_amd/host/ is the drop-in ComputeMatches executable; this module is
what the parity tests, bench.py and the multi-GPU driver call.  Nothing here touches the oracle.
"""
import sys
import time

import numpy as np

from . import _lib
from .database import Database

TOPSCALE_BASE = _lib.MAX_IMAGES  # + image id: cached top-scale subset of that image (slots MAX .. 2 MAX - 1)
AUX0 = 2 * _lib.MAX_IMAGES       # two scratch slots behind them: operator-level calls on raw arrays


class FeatureUtils:
    """Stateless operators of the reference, bound to one GPU context."""

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx if ctx is not None else _lib.Context(device)

    def ComputeMatches(self, desc1, desc2, distance_ratio=0.8):
        """knnMatch(k=2) + Lowe ratio (FeatureUtils.cpp:141-157)."""
        return self._match(desc1, desc2, distance_ratio, cross_check=False)

    def ComputeCrossMatches(self, desc1, desc2, distance_ratio=0.8):
        """Both directions + CrossCheck (FeatureUtils.cpp:160-174, :281-310)."""
        return self._match(desc1, desc2, distance_ratio, cross_check=True)

    def _match(self, desc1, desc2, distance_ratio, cross_check):
        self.ctx.upload_image(AUX0, desc1)
        self.ctx.upload_image(AUX0 + 1, desc2)
        return self.ctx.match_pair(AUX0, AUX0 + 1, ratio=distance_ratio, cross_check=cross_check,
                                   max_distance=float("inf"))

    @staticmethod
    def FilterMatchesByDistance(matches, max_distance=0.7):
        """Drop iff (double)distance > max_distance (FeatureUtils.cpp:208-218)."""
        q, t, d = matches
        keep = ~(d.astype(np.float64) > float(max_distance))
        return q[keep], t[keep], d[keep]

    @staticmethod
    def ExtractTopScaleDescriptors(kpts, descriptors, num_features):
        """Rows of the num_features largest keypoint sizes (FeatureUtils.cpp:68-96); the whole
        matrix if num_features > n.  Tie rule: size descending, index ascending."""
        idx = _lib.topscale_select(kpts, num_features)
        return np.ascontiguousarray(np.asarray(descriptors)[idx])


class FeatureMatcher:
    """FeatureMatcher (FeatureMatching.h:18-57).  geometric_verification: "device" runs FeatureUtils::FilterMatches
    (F-matrix RANSAC, a "next" row of the scope table) on the GPU inside the matching call
    (msfm_match_pairs_verified, what the C++ CLI does by default); a callable is called as
    f(kpts1, kpts2, matches) -> matches instead; None keeps the distance-filtered matches."""

    def __init__(self, database_path, max_num_matches=10240, max_distance=0.7, distance_ratio=0.8,
                 cross_check=True, ctx=None, device=0, geometric_verification=None, verbose=True):
        self.database_path_ = database_path
        self.max_num_matches_ = max_num_matches  # stored, never read (as in the reference)
        self.max_distance_ = float(max_distance)
        self.distance_ratio_ = float(distance_ratio)
        self.cross_check_ = bool(cross_check)
        self.database_ = None
        self.ctx = ctx if ctx is not None else _lib.Context(device)
        self.geometric_verification = geometric_verification
        self.verbose = verbose
        self._resident = set()

    def _out(self, s):
        if self.verbose:
            sys.stdout.write(s)

    def _ensure_resident(self, image_id):
        # replaces the per-pair Database::ReadDescriptors re-read (FeatureMatching.cpp:31-33)
        if image_id not in self._resident:
            self.ctx.upload_image(image_id, self.database_.ReadDescriptors(image_id))
            if self.geometric_verification == "device":
                self.ctx.upload_keypoints(image_id, self.database_.ReadKeyPoints(image_id))
            self._resident.add(image_id)

    def MatchImagePairs(self, image_pairs):
        """FeatureMatching.cpp:10-73: skip existing rows, match, filter, write one row per pair."""
        db = self.database_
        db.BeginTransaction()
        todo = []
        for (id1, id2) in image_pairs:
            if db.ExistMatches(id1, id2):
                self._out("Compute Matches %d - %d Existing, Continue!\n" % (id1, id2))
                continue
            todo.append((int(id1), int(id2)))
        if todo:
            for id1, id2 in todo:
                self._ensure_resident(id1)
                self._ensure_resident(id2)
            t0 = time.perf_counter()
            match = self.ctx.match_pairs_verified if self.geometric_verification == "device" else self.ctx.match_pairs
            offs, qt, dist = match(np.asarray(todo, np.int32), ratio=self.distance_ratio_,
                                   cross_check=self.cross_check_, max_distance=self.max_distance_)
            per_pair = (time.perf_counter() - t0) / len(todo)
            for p, (id1, id2) in enumerate(todo):
                self._out("Compute Matches %d - %d ... \n" % (id1, id2))
                m = qt[offs[p]:offs[p + 1]]
                if callable(self.geometric_verification):
                    k1, k2 = db.ReadKeyPoints(id1), db.ReadKeyPoints(id2)
                    q, t, _ = self.geometric_verification(k1, k2, (m[:, 0], m[:, 1], dist[offs[p]:offs[p + 1]]))
                    m = np.stack([q, t], axis=1) if len(q) else np.zeros((0, 2), np.int32)
                self._out("\t matches num : %d\n" % len(m))
                self._out("\t Elapsed time: %.5f [seconds]\n\n" % per_pair)
                db.WriteMatches(id1, id2, m)
        db.EndTransaction()

    def RunMatching(self):
        raise NotImplementedError


class SequentialFeatureMatcher(FeatureMatcher):
    """Pairs (i, i-k), k = 1..overlap (FeatureMatching.cpp:75-100)."""

    def __init__(self, database_path, overlap=3, **kw):
        super().__init__(database_path, **kw)
        self.overlap_ = overlap

    def RunMatching(self):
        self.database_ = Database(self.database_path_)
        n = len(self.database_.ReadAllImages())
        for i in range(1, n):
            image_pairs = []
            for k in range(1, self.overlap_ + 1):
                j = i - k
                if j < 0:
                    break
                image_pairs.append((i, j))
            self.MatchImagePairs(image_pairs)
        self.database_.Close()


class BruteFeatureMatcher(FeatureMatcher):
    """All pairs (i, j<i), flushed every max_pairs_size pairs and at the end of each row, with the
    pre-emptive top-scale filter (FeatureMatching.cpp:102-203)."""

    def __init__(self, database_path, max_pairs_size=100, is_preemtive=True, preemtive_num_features=100,
                 preemtive_min_num_matches=4, **kw):
        super().__init__(database_path, **kw)
        self.max_pairs_size_ = max_pairs_size
        self.is_preemtive_ = is_preemtive
        self.preemtive_num_features_ = preemtive_num_features
        self.preemtive_min_num_matches_ = preemtive_min_num_matches
        self.top_scale_descriptors_cache_ = set()

    def RunMatching(self):
        self.database_ = Database(self.database_path_)
        n = len(self.database_.ReadAllImages())
        for i in range(n):
            image_pairs = []
            for j in range(i):
                image_pairs.append((i, j))
                if len(image_pairs) == self.max_pairs_size_:
                    if self.is_preemtive_:
                        image_pairs = self.PreemptivelyFilterImagePairs(image_pairs)
                    self.MatchImagePairs(image_pairs)
                    image_pairs = []
            if image_pairs:
                if self.is_preemtive_:
                    image_pairs = self.PreemptivelyFilterImagePairs(image_pairs)
                self.MatchImagePairs(image_pairs)
        self.database_.Close()

    def GetTopScaleDescriptors(self, image_id):
        """Uploads (once) the top-scale subset of image_id to its auxiliary slot; returns the slot."""
        slot = TOPSCALE_BASE + image_id
        if image_id not in self.top_scale_descriptors_cache_:
            kpts = self.database_.ReadKeyPoints(image_id)
            desc = self.database_.ReadDescriptors(image_id)
            top = FeatureUtils.ExtractTopScaleDescriptors(kpts, desc, self.preemtive_num_features_)
            self.ctx.upload_image(slot, top)
            self.top_scale_descriptors_cache_.add(image_id)
        return slot

    def PreemptivelyFilterImagePairs(self, image_pairs):
        if not image_pairs:
            return []
        slots = np.asarray([(self.GetTopScaleDescriptors(a), self.GetTopScaleDescriptors(b))
                            for a, b in image_pairs], np.int32)
        # the pre-emptive test does not apply FilterMatchesByDistance (FeatureMatching.cpp:163-172)
        offs, _, _ = self.ctx.match_pairs(slots, ratio=self.distance_ratio_, cross_check=self.cross_check_,
                                          max_distance=float("inf"), fetch=False)
        counts = np.diff(offs)
        return [p for p, c in zip(image_pairs, counts) if c >= self.preemtive_min_num_matches_]
