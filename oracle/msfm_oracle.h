/*
 * msfm_oracle.h -- CPU oracle for the ComputeMatches hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (monocularsfm_amd/,
 * include/, the ComputeMatches executable) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the CPU baseline, never as the thing shipped.
 *
 * PARITY UNPINNED: the hot arithmetic of the reference is one call into
 * OpenCV (cv::BFMatcher::knnMatch, /root/reference/src/Feature/FeatureUtils.cpp:146-149).
 * OpenCV is not vendored, not version-pinned (/root/reference/CMakeLists.txt:32)
 * and not installed in this image, and the reference holds no golden vector or
 * test for this path.  This file therefore restates
 *   (i)  the reference's own code on the path, cited per function, and
 *   (ii) the published behaviour of OpenCV 4.x cv::batchDistance /
 *        cv::hal::normL2Sqr_ (core/src/batch_distance.cpp, core/src/norm.cpp),
 *        restated from its documentation and public source, not copied.
 * The fp32 accumulation order of normL2Sqr_ depends on the OpenCV build; the
 * order is a named parameter here (MSFM_ORC_ORDER_*).  For integer-valued
 * descriptors (raw SIFT, 0..255) every order gives identical bits because all
 * partial sums are integers < 2^24.
 */
#ifndef MSFM_ORACLE_H
#define MSFM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSFM_ORC_DIM 128

/* accumulation orders of S(q,t) = sum_c (a_c - b_c)^2 in fp32 */
enum {
    /* OpenCV 4.x universal intrinsics, SSE2/SSE3 baseline (the default x86-64
     * build): 4 accumulators x 4 lanes, 16 floats / iteration, multiply and add
     * rounded separately (v_muladd without FMA3), then ((d0+d1)+d2)+d3 lane-wise
     * and the horizontal sum (l0+l2)+(l1+l3). */
    MSFM_ORC_ORDER_SSE4X4 = 0,
    /* OpenCV 4.x built with an AVX2+FMA3 baseline: 4 accumulators x 8 lanes,
     * 32 floats / iteration, fused multiply-add, ((d0+d1)+d2)+d3 lane-wise and
     * the AVX2 v_reduce_sum tree ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)). */
    MSFM_ORC_ORDER_AVX2_FMA = 1,
    /* plain scalar loop d += t*t, c ascending, no FMA (OpenCV built without SIMD) */
    MSFM_ORC_ORDER_SCALAR = 2,
    /* AVX-512 + FMA3 build of the universal-intrinsics loop: 4 accumulators x 16 lanes, 64 floats / iteration, fused,
     * ((d0+d1)+d2)+d3 lane-wise, then halves / halves / the four-lane sum of the SSE order (msfm_oracle.c) */
    MSFM_ORC_ORDER_AVX512_FMA = 3
};

/* S(a,b) in the given order (never sqrt'ed). */
float orc_l2sqr(const float* a, const float* b, int order);
/* bit 0: the AVX2+FMA3 order runs on real intrinsics on this host, bit 1: the AVX-512F one (else their plain-C statements);
 * orders 101 / 103 of orc_l2sqr are those plain-C statements (test hooks, like 100 for the SSE order) */
int orc_simd_level(void);

/*
 * knnMatch(query, train, k=2) of cv::BFMatcher(NORM_L2):
 * per query row the two smallest sqrtf(S) under (distance asc, train index asc).
 * idx0/idx1 = -1 and d = FLT_MAX where fewer than 1/2 neighbours exist
 * (K = min(2, nt); candidates whose distance bit pattern is >= FLT_MAX's are
 * never inserted, as in batchDistance).
 */
void orc_knn2(const float* q, int nq, const float* t, int nt, int order,
              int32_t* idx0, float* d0, int32_t* idx1, float* d1);

/* same results from a cache-blocked loop order (32 query rows x 128 train rows at a time, train tiles ascending):
 * what the pair-parallel baseline runs (orc_match_pairs_mt). */
void orc_knn2_blocked(const float* q, int nq, const float* t, int nt, int order,
                      int32_t* idx0, float* d0, int32_t* idx1, float* d1);

/* same, query rows split over nthreads pthreads (mirrors OpenCV's parallel_for_). */
void orc_knn2_mt(const float* q, int nq, const float* t, int nt, int order, int nthreads,
                 int32_t* idx0, float* d0, int32_t* idx1, float* d1);

/* FeatureUtils::ComputeMatches (FeatureUtils.cpp:141-157): knn2 + Lowe ratio.
 * out_q/out_t/out_d need capacity n1.  nt < 2 => 0 matches (reference is UB). */
int orc_compute_matches(const float* d1, int n1, const float* d2, int n2, float ratio,
                        int order, int nthreads, int32_t* out_q, int32_t* out_t, float* out_d);

/* FeatureUtils::CrossCheck (FeatureUtils.cpp:281-310) incl. the operator[] quirk:
 * a train index without a surviving reverse match reads as 0. */
int orc_cross_check(const int32_t* q12, const int32_t* t12, const float* d12, int m12,
                    const int32_t* q21, const int32_t* t21, int m21,
                    int32_t* out_q, int32_t* out_t, float* out_d);

/* FeatureUtils::FilterMatchesByDistance (FeatureUtils.cpp:208-218): drop iff (double)d > max. */
int orc_filter_by_distance(const int32_t* q, const int32_t* t, const float* d, int m,
                           double max_distance, int32_t* out_q, int32_t* out_t, float* out_d);

/* ComputeCrossMatches/ComputeMatches + FilterMatchesByDistance as MatchImagePairs
 * chains them (FeatureMatching.cpp:36-49).  Returns the match count; out_* need capacity n1. */
int orc_match_pair(const float* d1, int n1, const float* d2, int n2,
                   float ratio, int cross_check, double max_distance, int order, int nthreads,
                   int32_t* out_q, int32_t* out_t, float* out_d);

/* The loop of FeatureMatcher::MatchImagePairs (FeatureMatching.cpp:14-49) over INDEPENDENT image pairs, one
 * pair per worker at a time: `nthreads` pthreads pull pair indices from a shared counter and run orc_match_pair
 * single-threaded (the reference runs its pairs one after the other; OpenCV fans each knnMatch out over query
 * rows -- parallelising over pairs instead keeps all cores busy without a thread create/join per pair).
 * images[id] -> n x 128 descriptors, rows[id] -> n; pairs = P x 2 ids (query, train).
 * out_offsets: P+1 CSR offsets; out_q/out_t/out_d: capacity sum over pairs of rows[query].  budget_s > 0: no new
 * pair is started after that many seconds (the timed CPU baseline); *n_done (nullable) receives the number of pairs
 * computed, always a prefix of the list -- the rest get empty lists.  Returns the number of matches. */
int64_t orc_match_pairs_mt(const float* const* images, const int32_t* rows, const int32_t* pairs, int n_pairs,
                           float ratio, int cross_check, double max_distance, int order, int nthreads, double budget_s,
                           int* n_done, int64_t* out_offsets, int32_t* out_q, int32_t* out_t, float* out_d);

/* FeatureUtils::ExtractTopScaleDescriptors' selection (FeatureUtils.cpp:68-96):
 * indices of the k largest KeyPoint.size (kpts = n x 4 floats x,y,size,angle).
 * The reference uses std::partial_sort (unspecified order among equal sizes);
 * the build's documented rule is size descending, index ascending.
 * Returns number of indices written: n if k > n (whole matrix, identity order) else k. */
int orc_topscale_select(const float* kpts, int n, int k, int32_t* out_idx);

/* Database::ImagePairToPairId / PairIdToImagePair / SwapImagePair (Database.cpp:656-694). */
int32_t orc_pair_id(int32_t id1, int32_t id2);
void    orc_pair_from_id(int32_t pair_id, int32_t* id1, int32_t* id2);
int     orc_swap_image_pair(int32_t id1, int32_t id2);

/* BruteFeatureMatcher::RunMatching pair enumeration (FeatureMatching.cpp:110-139):
 * pairs (i,j), j<i, i-major, flushed every max_pairs (100) pairs and at the end of each
 * row i.  Writes pairs (2 ints each) and batch_end[] (exclusive end offsets, in pairs).
 * Returns number of pairs; *n_batches receives the number of flushes. */
int64_t orc_enumerate_brute(int n_images, int max_pairs, int32_t* pairs, int64_t* batch_end,
                            int64_t* n_batches);
/* SequentialFeatureMatcher::RunMatching (FeatureMatching.cpp:82-97): one batch per i>=1. */
int64_t orc_enumerate_sequential(int n_images, int overlap, int32_t* pairs, int64_t* batch_end,
                                 int64_t* n_batches);

#ifdef __cplusplus
}
#endif
#endif
