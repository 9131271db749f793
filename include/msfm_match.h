/*
 * msfm_match.h -- C ABI of the MI355X (gfx950) ComputeMatches hot path.
 *
 * Drop-in boundary for nebula-beta/MonocularSfM's descriptor matcher: every entry point
 * replaces one C++ interface of the reference (cited per function, paths relative to the
 * reference checkout).  The reference has no FFI layer of its own; its seams are the static
 * FeatureUtils operators (include/Feature/FeatureUtils.h:94-108) called from
 * src/Feature/FeatureMatching.cpp:36-49 and :163-170, and the Database blobs on either side.
 *
 * Conventions
 *   - plain C types only; every call returns an int status (MSFM_OK == 0), never throws
 *     (every entry point runs behind one exception barrier, csrc/msfm_guard.h: an allocation
 *     failure inside the library comes back as MSFM_E_DEVICE, the context stays usable),
 *     never exits.  msfm_last_error() gives the text of the last failure on a context.
 *   - the caller owns all host buffers; the library owns all device memory.
 *   - descriptors are row-major, contiguous, 128 columns (cv::Mat CV_32F n x 128 as read by
 *     Database::ReadDescriptors, src/Database/Database.cpp:510-523), or uint8 with the same shape.
 *   - a context is bound to one GPU and is not thread-safe; distinct contexts may be driven
 *     from distinct threads (one per GPU).
 *   - there is NO CPU fallback: without a usable gfx950 device msfm_create() fails.
 */
#ifndef MSFM_MATCH_H
#define MSFM_MATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSFM_DIM 128
#define MSFM_MAX_IMAGES 10000 /* kMaxNumImages, src/Database/Database.cpp:6 */

enum {
    MSFM_OK = 0,
    MSFM_E_INVALID = 1,   /* bad argument */
    MSFM_E_DEVICE = 2,    /* HIP error (text in msfm_last_error) */
    MSFM_E_NOIMAGE = 3,   /* image id not uploaded */
    MSFM_E_CAPACITY = 4,  /* caller buffer too small */
    MSFM_E_STATE = 5      /* call sequence error (e.g. fetch without a prior match) */
};

enum { MSFM_DTYPE_F32 = 0, MSFM_DTYPE_U8 = 1 };

/* fp32 accumulation order of S(q,t) = sum_c (a_c-b_c)^2, i.e. which build of
 * cv::hal::normL2Sqr_ (called through BFMatcher::knnMatch at src/Feature/FeatureUtils.cpp:149)
 * the bits are identical to.  Integer-valued descriptors give identical bits under all. */
enum {
    MSFM_ORDER_SSE4X4 = 0,   /* OpenCV 4.x SSE baseline: 16 lane partials, mul and add rounded apart */
    MSFM_ORDER_AVX2_FMA = 1, /* OpenCV 4.x AVX2+FMA3 baseline: 32 lane partials, fused */
    MSFM_ORDER_AVX512_FMA = 3 /* AVX-512+FMA3 build of the same loop: 64 lane partials, fused (a third named order: the
                                 cross-check of the order-invariance certificate, msfm_fetch_order_certificate) */
};

typedef struct msfm_ctx msfm_ctx;

/* Matching parameters = the FeatureMatcher constructor arguments that reach the operators
 * (include/Feature/FeatureMatching.h:28-32): distance_ratio (passed on as float),
 * cross_check, max_distance (double). */
typedef struct msfm_match_params {
    float ratio;          /* default 0.8f */
    int cross_check;      /* default 1    */
    double max_distance;  /* default 0.7  */
} msfm_match_params;

/* Kernel-side timing of the last msfm_match_pairs / msfm_match_pair / msfm_knn2_pair call,
 * measured with HIP events on the library's own stream. */
typedef struct msfm_profile {
    double dist_kernel_ms;     /* sum over launches of the distance/top-2 kernel */
    int dist_kernel_launches;
    double total_device_ms;    /* first launch -> last result copy of the call */
    int64_t descriptor_pairs;  /* sum n1*n2 over the pairs of the call */
    int64_t dist_algo_bytes;   /* compulsory HBM bytes of the distance kernel */
    /* MFMA prefilter path (two sweep_kernel launches per batch + exact re-check of the candidates) */
    double approx_kernel_ms;   /* sum over batches of sweep 1 (sweep_kernel<1> / sweep_i8_kernel<1>: every descriptor pair once; the field keeps its round-1 name) */
    int approx_kernel_launches;
    int prefilter_pairs;       /* pairs answered through the prefilter path */
    int fallback_pairs;        /* pairs whose candidate list overflowed -> brute-force exact kernel */
    int64_t candidates;        /* exact distances evaluated for prefiltered pairs */
    int64_t prefilter_descriptor_pairs;
    int64_t exact_descriptor_pairs; /* descriptor pairs that went through the brute-force kernel */
    int64_t tie_rows;          /* rows re-scanned by the sqrt-space tie fix-up */
    double sweep2_ms;          /* sum over batches of sweep 2 (sweep_kernel<2> dense / <3> compacted live rows, sweep_i8_kernel<3>) */
    int sweep2_launches;
    int compacted_pairs;       /* pairs whose sweep 2 ran on the compacted live rows only */
    int64_t sweep2_descriptor_pairs; /* descriptor pairs sweep 2 actually multiplied (padded rows included) */
    double verify_ms;          /* geometric verification kernels (msfm_match_pairs_verified) */
    int sub_batches;           /* device sub-batches the call was cut into (msfm_set_limits) */
    int tie_queue_regrows;     /* sub-batches re-run because the sqrt-space tie queue had to grow */
    int plan_regrows;          /* sub-batches re-run because the device-side sweep-2 plan outgrew its predicted buffers */
    int sweep1_i8_launches;     /* sweep-1 launches on the integer matrix cores (byte stores, msfm_sweep_i8.hip.h) */
    /* route Q: float images with byte twins (all values in [0, 1]) -- sweep 1 on the twins, on the integer matrix cores */
    int sweep1_q8_launches;     /* sweep-1 launches on byte TWINS of float images (counted in sweep1_i8_launches too) */
    int sweep1b_launches;       /* fp16 sweep 1' launches (coarse twins only -- values beyond 0.625, or MSFM_Q8_DIRECT=0: the S~ top-2 of the
                                   rows the twins' sweep left alive, sweep_kernel<4>); 0 when the twins give sweep 2 its thresholds directly */
    double sweep1b_ms;
    int64_t sweep1b_descriptor_pairs; /* descriptor pairs sweep 1' multiplied (padded compacted rows included) */
    int64_t order_sensitive_rows; /* rows / columns of the call WITHOUT an order-invariance certificate (see
                                     msfm_fetch_order_certificate); 0 => the stored (queryIdx, trainIdx) rows are the same
                                     under any conforming fp32 evaluation order of hal::normL2Sqr_ */
    int demoted_pairs;            /* pairs of two byte images (or of two images with byte twins) whose first sweep ran on the fp16 cores
                                     all the same, because their SUB-BATCH also held a pair that could not take the integer route (a
                                     mixed store: byte stores and coarse twins choose the route per sub-batch; fine twins do not, see
                                     mixed_route_sub_batches).  Same results, ~1.6 x the sweep time; 0 on a homogeneous store */
    int mixed_route_sub_batches;  /* sub-batches that ran TWO first sweeps: the integer one on the pairs whose images both have fine byte
                                     twins, the fp16 one on the rest (an image with a value beyond [0, 1] among twinned ones) -- the twins'
                                     pairs are not demoted then.  Byte stores and coarse twins still choose one route per sub-batch */
    int memory_shrinks;           /* times the call found the device OUT OF MEMORY while sizing a sub-batch's scratch (another tenant of the GPU,
                                     several contexts on one device: the budget of msfm_set_limits is derived from the free memory once per
                                     state of the store), gave every idle buffer back, halved the scratch share of a sub-batch and cut again from
                                     the same pair.  Same results, more sub-batches; 0 normally.  After four of them the call fails with MSFM_E_DEVICE */
} msfm_profile;

/* ---- context ------------------------------------------------------------------------- */
/* msfm_create: ~85 ms in a fresh process, 52 of them the HIP runtime's own start-up (its first call), the rest one stream (a hardware
 * queue), the kernels' attributes and a first allocation; the streams of the second and third scratch set are created by the first call
 * that keeps more than one sub-batch in flight.  MSFM_DEBUG_TIMING=1 in the environment: the library's host-side clocks on stderr
 * (context creation and destruction, store build, every sub-batch's phases, allocations per call). */
int msfm_create(int device_ordinal, msfm_ctx** out_ctx);
void msfm_destroy(msfm_ctx* ctx);
const char* msfm_last_error(const msfm_ctx* ctx);
/* device name + CU count of the context's GPU (for bench reports); name_cap >= 64 */
int msfm_device_info(const msfm_ctx* ctx, char* name, int name_cap, int* cu_count, int* clock_mhz);
int msfm_set_accum_order(msfm_ctx* ctx, int order);
/* 1 (default): MFMA prefilter + exact re-check where safe -- byte images (MSFM_DTYPE_U8 uploads, and MSFM_DTYPE_F32
 * uploads whose every value is an integer in [0, 255]: raw OpenCV SIFT stored as CV_32F, recognised on the device at
 * upload; MSFM_BYTE_DETECT=0 at msfm_create: off) on the integer matrix cores (v_mfma_i32_32x32x32_i8); float images whose values all lie in [0, 1] (RootSIFT) get a byte twin at upload and
 * their FIRST sweep on the integer cores too; fine twins (the store's values stay below 0.625) hand the second sweep its thresholds
 * directly, coarse ones are refined by an fp16 sweep of the ~6 % of rows left alive first (route Q; MSFM_Q8=0 in the environment
 * at msfm_create: off, MSFM_Q8_DIRECT=0: always refine); everything else on the fp16 cores; 2: fp16 matrix cores for every
 * image; 0: always the brute-force exact kernel.  Results are bit-identical in all.
 * Env: MSFM_PREFILTER=0|1|2. */
int msfm_set_prefilter(msfm_ctx* ctx, int enable);
int msfm_get_profile(const msfm_ctx* ctx, msfm_profile* out);
/* msfm_match_pairs cuts a call into device sub-batches of at most `max_pairs_per_batch` image pairs (default 16384) and of
 * at most `scratch_bytes` of device scratch for ALL sub-batches in flight TOGETHER (three scratch sets share it, a third each;
 * what a pair needs: msfm_pair_scratch_bytes in csrc/msfm_hostutil.h -- on the matrix-core route ~2 MB per pair of 8192-row
 * images).  scratch_bytes <= 0 (default): 64 GiB, but never more than a quarter of what the device has free when the call starts
 * (hipMemGetInfo + what the sets already hold); an explicit value is honoured up to half of that (the per-pair figure is what a call is CUT by, not a cap: buffers that turn out too small are re-grown and the sub-batch re-run).  The device is asked once per state of the store, not per call.  The call's result
 * lists (12 bytes per match, device + page-locked host) and the descriptor store are NOT part of this budget.  Also
 * MSFM_MAX_PAIRS_PER_BATCH and MSFM_SCRATCH_MIB in the environment at msfm_create.  Results do not depend on the cut (tests force
 * small limits to cross it).  The reference's counterpart is the 100-pair flush of BruteFeatureMatcher::RunMatching
 * (src/Feature/FeatureMatching.cpp:118-139, max_pairs_size_). */
int msfm_set_limits(msfm_ctx* ctx, int max_pairs_per_batch, int64_t scratch_bytes);
/* A call of enough work (>= 1.5e10 descriptor pairs per part) is cut into at least `min_sub_batches` sub-batches, launched round-robin
 * on three streams / scratch sets: the bandwidth-bound tail of one (thresholds, sweep-2 plan, exact re-check, epilogue, copy-out)
 * runs while the sweeps of the next ones own the matrix cores.  Default 2 equal parts (round 3: 6 parts shrinking to 0.3 of the
 * average; with round 4's smaller tails a further cut costs more than it hides); 1 = one sub-batch where memory allows (no overlap);
 * <= 0 restores the default.  Env: MSFM_PIPELINE, MSFM_PIPELINE_TAPER (size of the last part relative to the average, default 1),
 * MSFM_IN_FLIGHT.  Results do not depend on any of it. */
int msfm_set_pipeline(msfm_ctx* ctx, int min_sub_batches);

/* ---- descriptor store -------------------------------------------------------------------
 * Replaces the per-pair Database::ReadDescriptors calls of MatchImagePairs
 * (src/Feature/FeatureMatching.cpp:32-33, "TODO: cache"): every image is uploaded once and
 * stays resident in HBM.  ids are Database image ids, 0 <= id < MSFM_MAX_IMAGES.
 * msfm_upload_image COPIES the rows (the caller's buffer is free when it returns) and returns without building anything: the
 * images of all uploads since the last use are built together -- classification (byte store? values in [0, 1]?), ONE device
 * allocation, table-driven layout kernels -- by the next call that needs them (any matching call, msfm_subset_image of a
 * pending source) or by msfm_finalize_store.  A device error of that build (out of memory) is therefore reported by THAT call.
 * Resident per row: 184 B for a byte image (the 176-byte operand row of the integer matrix cores + norms; the fp32 / fp16 forms
 * are derived on the device the first time a route needs them: a pair with a float image, msfm_knn2_pair, ratio > 0.95,
 * msfm_set_prefilter 0 / 2), 788 B for a float image (+ 184 B with a byte twin); + 512 B per row of 128-row panels for
 * images that take the brute-force route.  csrc/msfm_store.hip.h has the table. */
int msfm_upload_image(msfm_ctx* ctx, int image_id, const void* desc, int n, int dim, int dtype);
/* A new store entry from rows of a resident one, entirely on the device: the sub-matrix
 * FeatureUtils::ExtractTopScaleDescriptors builds (src/Feature/FeatureUtils.cpp:84-95, rows picked by
 * msfm_topscale_select) without reading the descriptors from the database a second time
 * (BruteFeatureMatcher::GetTopScaleDescriptors, FeatureMatching.cpp:181-196, does re-read them).
 * rows: `count` indices into src, any order, repeats allowed; dst != src (usually MSFM_MAX_IMAGES + id). */
int msfm_subset_image(msfm_ctx* ctx, int src_image_id, int dst_image_id, const int32_t* rows, int count);
int msfm_image_rows(const msfm_ctx* ctx, int image_id, int* out_n);
int msfm_clear_images(msfm_ctx* ctx);
/* Build everything uploaded so far now (otherwise the first matching call does it): the end of a bulk load, e.g. of
 * FeatureMatcher::PreloadAllImages' one SELECT sweep over the descriptors table (host/FeatureMatching.cpp). */
int msfm_finalize_store(msfm_ctx* ctx);
/* Device bytes the store holds (its chunks, incl. forms derived on demand and keypoints), descriptor rows resident, images still
 * waiting for msfm_finalize_store.  Any pointer may be NULL. */
int msfm_store_info(const msfm_ctx* ctx, int64_t* out_device_bytes, int64_t* out_rows, int64_t* out_pending_images);

/* ---- one pair ---------------------------------------------------------------------------
 * Twin of FeatureUtils::ComputeCrossMatches / ComputeMatches (src/Feature/FeatureUtils.cpp:
 * 141-174) followed by FilterMatchesByDistance (:208-218), i.e. lines 36-49 of
 * FeatureMatching.cpp.  query = id1, train = id2.  out_qt receives (queryIdx, trainIdx) int32
 * pairs in ascending queryIdx (capacity: rows(id1) pairs); out_dist (nullable) the DMatch
 * distances.  The reference indexes the 2nd neighbour unconditionally (FeatureUtils.cpp:152), so a
 * direction whose train set has < 2 rows is undefined there; here it yields no matches, and a
 * cross-checked pair with such a direction yields none at all.  Empty sides => no matches. */
int msfm_match_pair(msfm_ctx* ctx, int id1, int id2, float ratio, int cross_check,
                    double max_distance, int32_t* out_qt, float* out_dist, int* out_count);

/* ---- batch of pairs ---------------------------------------------------------------------
 * Twin of the loop body of FeatureMatcher::MatchImagePairs (FeatureMatching.cpp:14-49) over
 * independent pairs; results keep the input order.  pairs = P x 2 int32 (id1, id2).
 * out_offsets (P+1 int64) receives the CSR offsets of each pair's match list; the lists
 * themselves stay on the context until msfm_fetch_matches copies them out
 * (out_qt: 2*out_offsets[P] int32, out_dist nullable: out_offsets[P] float). */
int msfm_match_pairs(msfm_ctx* ctx, const int32_t* pairs, int n_pairs,
                     const msfm_match_params* params, int64_t* out_offsets);
int msfm_fetch_matches(msfm_ctx* ctx, int32_t* out_qt, float* out_dist);
/* Order-invariance certificate of the last msfm_match_pairs / _verified call: per pair the number of query rows (and,
 * with cross_check, train rows) for which a decision the reference makes -- which element is the first neighbour,
 * d0 < ratio * d1, d0 <= max_distance -- has a margin below the worst-case fp32 reassociation bound of a 128-term sum
 * of squares (relative 1e-5 on a distance, derivation in csrc/msfm_kernels.hip.h).  The reference delegates S(q,t) to an
 * unpinned OpenCV (cv::BFMatcher::knnMatch, src/Feature/FeatureUtils.cpp:146-149) whose accumulation order depends on the
 * build; a pair with 0 sensitive rows has the same (queryIdx, trainIdx) list under ANY such build.  Pairs of byte images
 * are exact integers under every order (always 0).  out_sensitive_rows: n_pairs int32.
 * The TIE RULE, the other thing SURVEY App. C restates from memory (batchDistance keeps the lower train index among equal
 * distances): with ratio <= 1 the match LISTS do not depend on it either.  A row whose two smallest distances are equal
 * (d0 == d1, which is when the rule decides the first neighbour) fails `d0 < ratio * d1` under either index order, in both
 * directions, so its index never reaches a list; a tie for the SECOND place leaves the value d1 unchanged.  So a job with 0
 * order-sensitive rows and ratio <= 1 stores the same rows under any conforming accumulation order AND either tie order
 * (tests/test_gpu_certificate.py::test_match_lists_do_not_depend_on_the_tie_rule flips the rule in the integer reference on
 * the planted-tie fixture; only the knnMatch-level API msfm_knn2_pair and ratio > 1 see the rule, and implement the lower
 * index: tie_fixup_kernel). */
int msfm_fetch_order_certificate(msfm_ctx* ctx, int32_t* out_sensitive_rows);
/* The same lists without the copy: pointers into the context's page-locked result buffers
 * (2 * count int32, count float), valid until the next matching call on this context or msfm_destroy. */
int msfm_view_matches(msfm_ctx* ctx, const int32_t** out_qt, const float** out_dist, int64_t* out_count);
/* The same lists copied device-to-device into CALLER-OWNED DEVICE memory on the context's GPU (2 * count int32 /
 * count float; either pointer may be NULL): the multi-GPU exchange step sends them over RCCL straight from HBM
 * instead of staging them through the host.  Complete when the call returns. */
int msfm_fetch_matches_device(msfm_ctx* ctx, int32_t* d_out_qt, float* d_out_dist);

/* ---- batch of pairs, STREAMING form (bounded memory) ----------------------------------------
 * msfm_match_pairs keeps the whole call's lists resident (12 bytes per match in HBM and the same page-locked): BASELINE's
 * largest config produces 6.9e9 matches.  The reference streams by construction -- one transaction per <= 100 pairs,
 * BruteFeatureMatcher::RunMatching (src/Feature/FeatureMatching.cpp:13, 70-72, 118-139).  Same here:
 *   msfm_match_pairs_begin   takes the pair list (copied) and the parameters (geometric_verification != 0: the lists are verified
 *                            on the device as in msfm_match_pairs_verified, `verify` NULL = the reference's constants);
 *   msfm_match_pairs_next    completes ONE device sub-batch -- the next `n_pairs` pairs of the list, in order -- and hands out its
 *                            lists: CSR offsets relative to the chunk, (q, t) rows and distances in page-locked host memory AND in
 *                            device memory (for an RCCL send straight from HBM), the order-certificate counts.  The pointers are valid
 *                            until the next msfm_match_pairs_next / any other matching call on the context.  n_pairs == 0: done.
 * Between two calls the library keeps up to three sub-batches in flight exactly as msfm_match_pairs does; what is resident at any time
 * is the scratch of those (msfm_set_limits) plus their lists.  msfm_get_profile accumulates over the series.  While a series is open
 * (until the call that returns n_pairs == 0) the store must not change: uploads, msfm_subset_image and msfm_clear_images return
 * MSFM_E_STATE; another matching call abandons the series (what it has in flight is drained first). */
typedef struct msfm_chunk {
    int first_pair;                 /* index into the pair list given to msfm_match_pairs_begin */
    int n_pairs;                    /* 0: the series is complete */
    int64_t count;                  /* matches of the chunk = offsets[n_pairs] */
    const int64_t* offsets;         /* n_pairs + 1 */
    const int32_t* qt;              /* host (page-locked): 2 * count */
    const float* dist;              /* host: count */
    const int32_t* d_qt;            /* device: 2 * count */
    const float* d_dist;            /* device: count */
    const int32_t* sensitive_rows;  /* n_pairs (msfm_fetch_order_certificate) */
} msfm_chunk;
struct msfm_verify_params;
int msfm_match_pairs_begin(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params,
                           int geometric_verification, const struct msfm_verify_params* verify);
int msfm_match_pairs_next(msfm_ctx* ctx, msfm_chunk* out);
/* Ends a series before its last chunk (a consumer that stops early, an error in the consumer's own loop): drains what the series has
 * in flight, unlocks the store; a following msfm_match_pairs_next returns MSFM_E_STATE.  No series open: MSFM_OK, nothing happens.
 * (The reference's counterpart is leaving the pair loop of FeatureMatcher::MatchImagePairs, src/Feature/FeatureMatching.cpp:14-72.) */
int msfm_match_pairs_end(msfm_ctx* ctx);
/* What the context holds right now, in bytes: the descriptor store, uploads waiting in the inbox, the scratch of the sub-batches in
 * flight (grow-only: its high-water mark), the call-wide result lists on the device (msfm_match_pairs; the streaming form has none),
 * page-locked host memory (result lists, staging); and the device's free / total memory (hipMemGetInfo). */
typedef struct msfm_memory {
    int64_t device_free, device_total;
    int64_t store, inbox, scratch, results_device;
    int64_t page_locked_host;
} msfm_memory;
int msfm_memory_info(msfm_ctx* ctx, msfm_memory* out);
/* Plain device -> host copy through the library's own runtime (a caller without a HIP runtime of its own -- a ctypes binding --
 * reading msfm_chunk::d_qt / d_dist or a buffer msfm_fetch_matches_device filled). */
int msfm_read_device(msfm_ctx* ctx, void* host_dst, const void* device_src, int64_t bytes);

/* ---- batch of pairs with the geometric verification hand-off -------------------------------
 * Lines 36-60 of FeatureMatching.cpp in one call: matching as above, then FeatureUtils::FilterMatches
 * (src/Feature/FeatureUtils.cpp:176-206: GetAlignedPointsFromMatches + cv::findFundamentalMat(FM_RANSAC,
 * 3.0, 0.99) + keep the inliers) for every pair, on the device, before the lists are copied out.
 * Needs the keypoint coordinates of every image of the batch: msfm_upload_keypoints after msfm_upload_image
 * (kpts: n rows of `stride_floats` floats, x and y first -- the Database's keypoint blob has stride 4:
 * x, y, size, angle; n >= descriptor rows).  NULL `verify` = the reference's constants (3.0 px, 0.99, OpenCV's
 * 1000-iteration cap).  OpenCV's RANSAC (RNG, 7-point solver) cannot be reproduced without OpenCV: this entry
 * point is OUTSIDE the bit-parity claim; it is bit-identical to the host twin
 * monocularsfm_amd/host/GeometricVerification.cpp (shared arithmetic, csrc/msfm_fmat.h).
 * Cases as in findFundamentalMat: no matches -> none; < 7 -> none; exactly 7 -> all; otherwise RANSAC with
 * the adaptive iteration bound, and a consensus set below 8 keeps none. */
typedef struct msfm_verify_params {
    double threshold;            /* pixels (reference: 3.0) */
    double confidence;           /* (reference: 0.99) */
    int max_iters;               /* hypotheses per pair at most (OpenCV default: 1000) */
    unsigned long long seed;     /* sampling stream, the same for every pair */
} msfm_verify_params;
int msfm_upload_keypoints(msfm_ctx* ctx, int image_id, const float* kpts, int n, int stride_floats);
int msfm_match_pairs_verified(msfm_ctx* ctx, const int32_t* pairs, int n_pairs,
                              const msfm_match_params* params, const msfm_verify_params* verify,
                              int64_t* out_offsets);

/* ---- knnMatch(k=2) twin (parity/debug) ----------------------------------------------------
 * Both directions of cv::BFMatcher(NORM_L2).knnMatch(.., 2) for one pair
 * (call site src/Feature/FeatureUtils.cpp:146-149), from ONE pass over the distance matrix.
 * fwd_* have rows(id1) entries (train index into id2), rev_* rows(id2) entries.
 * idx = -1 and d = FLT_MAX where fewer than 1 / 2 neighbours exist.  Any pointer may be NULL. */
int msfm_knn2_pair(msfm_ctx* ctx, int id1, int id2,
                   int32_t* fwd_idx0, float* fwd_d0, float* fwd_d1,
                   int32_t* rev_idx0, float* rev_d0, float* rev_d1);

/* ---- host-side helpers (no device work) ------------------------------------------------- */
/* FeatureUtils::ExtractTopScaleDescriptors' row selection (FeatureUtils.cpp:68-96):
 * kpts = n x 4 float (x, y, size, angle); writes min(k, n) indices, k > n => identity.
 * Tie rule (reference: unspecified, std::partial_sort): size descending, index ascending. */
int msfm_topscale_select(const float* kpts, int n, int k, int32_t* out_idx, int* out_count);
/* Database::ImagePairToPairId / PairIdToImagePair / SwapImagePair (Database.cpp:656-694) */
int msfm_pair_id(int id1, int id2, int32_t* out_pair_id);
int msfm_pair_from_id(int32_t pair_id, int* out_id1, int* out_id2);
int msfm_swap_image_pair(int id1, int id2);

const char* msfm_version(void);
/* gfx950 devices this process can open with msfm_create (ordinals 0 .. n-1); 0 without a GPU.  The reference is single-device;
 * the drop-in's node-level fan-out (MSFM_DEVICES=all in host/FeatureMatching.cpp) asks it how many contexts to create. */
int msfm_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
