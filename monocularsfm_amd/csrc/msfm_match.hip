// msfm_match.hip -- host side of the C ABI declared in include/msfm_match.h.
//
// Owns the device-resident descriptor store, schedules image pairs onto the gfx950 kernels in
// msfm_kernels.hip.h and returns match lists.  Mirrors what FeatureMatcher::MatchImagePairs
// (src/Feature/FeatureMatching.cpp:10-73 of the reference) does between its two
// Database::ReadDescriptors calls and FeatureUtils::FilterMatches, without the per-pair
// descriptor re-read.  No CPU fallback: every entry point that needs the GPU fails loudly
// when there is none.
#include "msfm_match.h"
#include "msfm_hostutil.h"
#include "msfm_kernels.hip.h"
#include "msfm_prefilter.hip.h"
#include "msfm_verify.hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

using namespace msfm;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    // grow and keep the first `keep` bytes (the match lists of a call accumulate over its sub-batches)
    hipError_t ensure_keep(size_t bytes, size_t keep, hipStream_t stream, size_t hint = 0) {
        if (bytes <= cap) return hipSuccess;
        const size_t want = std::max(bytes + bytes / 2 + 4096, hint);
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess) return e;
        if (p && keep) {
            e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {
                (void)hipFree(q);
                return e;
            }
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// page-locked host memory, grow-only and content-preserving (result lists of a call accumulate over its sub-batches)
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, size_t keep, size_t hint = 0) {
        if (bytes <= cap) return hipSuccess;
        size_t want = std::max(bytes + bytes / 2 + 4096, hint);
        void* q = nullptr;
        hipError_t e = hipHostMalloc(&q, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        if (p && keep) std::memcpy(q, p, keep);
        if (p) (void)hipHostFree(p);
        p = q;
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// a table inside an upload arena (Scratch::d_up): not owned, set by UploadPlan::place
struct Ref {
    void* p = nullptr;
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// The host tables a sub-batch uploads, packed: every table is written into ONE page-locked staging buffer and travels in ONE
// hipMemcpyAsync into a device arena of the same layout (round 3: nine pageable copies per sub-batch, each a 5 us link of
// the launch chain at the head of the sub-batch, from vectors that died with the issuing function -- ADVICE r03).  The
// staging buffer belongs to the scratch set and is rewritten only after the set's previous sub-batch has been completed.
struct UploadPlan {
    struct Item { Ref* dst; const void* src; size_t bytes, off; };
    std::vector<Item> items;
    size_t total = 0;
    void add(Ref& dst, const void* src, size_t bytes) {
        items.push_back(Item{&dst, src, bytes, total});
        total += (bytes + 255) & ~(size_t)255;
    }
    hipError_t place_and_copy(DevBuf& arena, PinnedBuf& staging, hipStream_t stream) {
        const size_t need = std::max<size_t>(total, 256);
        hipError_t e = arena.ensure(need);
        if (e != hipSuccess) return e;
        e = staging.ensure(need, 0);
        if (e != hipSuccess) return e;
        for (const Item& it : items) {
            it.dst->p = static_cast<char*>(arena.p) + it.off;
            if (it.bytes) std::memcpy(static_cast<char*>(staging.p) + it.off, it.src, it.bytes);
        }
        return total ? hipMemcpyAsync(arena.p, staging.p, total, hipMemcpyHostToDevice, stream) : hipSuccess;
    }
};

// Every table a sub-batch clears before its kernels run, in ONE launch (round 3: ~27 hipMemsetAsync per sub-batch = 27 runtime
// fill kernels of ~5 us each, without wave priority: they starved under the other stream's sweep).
struct FillSeg {
    void* p;
    unsigned long long bytes;
    unsigned value;   // the byte, replicated
};
constexpr int kFillSegs = 12;
struct FillSegs {
    FillSeg s[kFillSegs];
    int n;
};

struct Image {
    int n = -1;  // -1: not uploaded
    int nblk = 0;    // 128-row blocks holding data
    int nalloc = 0;  // allocated blocks (even: the prefilter walks 256-row A blocks); padding is zero-filled
    float* panel = nullptr;
    float* raw = nullptr;
    float* rawp = nullptr;   // raw, every row permuted for the exact re-check (PairDesc::a_rawp)
    // prefilter operands: fp16 rows of 272 B (128 halfs + the norm quadruple of the ninth MFMA k-step), row norms
    // (+inf padded), maxima
    _Float16* h16 = nullptr;
    float* nrm = nullptr;
    float c = 1.f;            // scale of the quadruples (power of two)
    float nrm_max = 0.f, abs_max = 0.f;
    bool pf_safe = false;
    // byte stores (MSFM_DTYPE_U8 uploads and their subsets): signed operand rows of 176 B (kI8RowBytes: 128 operand bytes, 16 digits, 16 constants, padding) for the integer matrix cores
    // and the float "norms" 2 floor(|x - 128|^2 / 2) (msfm_sweep_i8.hip.h)
    bool is_u8 = false;
    bool from_u8 = false;     // uploaded as MSFM_DTYPE_U8 (or a subset of such an image): integer values 0..255
    signed char* i8 = nullptr;
    float* nrm_i8 = nullptr;
    int* n2_i8 = nullptr;     // n' = |x - 128|^2 per row, exactly (nrm_i8 holds 2 floor(n' / 2)): the exact S of a byte pair's candidates
    float nrm_i8_max = 0.f;
    int h0_i8 = 0;            // centre of the rows' h = floor(|x - 128|^2 / 2): the digit k-step carries H0 - h
    // route Q (msfm_q8.hip.h): the byte twin q = rint(x 255 / m) of a FLOAT image whose values all lie in [0, 1] -- operand rows,
    // norms 2h and centre as for a byte image, plus the rows' quantisation error norms and their maximum
    signed char* q8 = nullptr;
    float* nrm_q8 = nullptr;
    float* err_q8 = nullptr;
    float err_q8_max = 0.f;
    int h0_q8 = 0;
    float q8_level = 0.f;     // the context's twin level m (scale 255 / m) this twin was built with
    // keypoint coordinates (x, y) for the geometric verification; nk = -1: not uploaded
    float2* kxy = nullptr;
    int nk = -1;
};

void free_image(Image& im) {
    if (im.panel) (void)hipFree(im.panel);
    if (im.raw) (void)hipFree(im.raw);
    if (im.rawp) (void)hipFree(im.rawp);
    if (im.h16) (void)hipFree(im.h16);
    if (im.nrm) (void)hipFree(im.nrm);
    if (im.i8) (void)hipFree(im.i8);
    if (im.nrm_i8) (void)hipFree(im.nrm_i8);
    if (im.n2_i8) (void)hipFree(im.n2_i8);
    if (im.q8) (void)hipFree(im.q8);
    if (im.nrm_q8) (void)hipFree(im.nrm_q8);
    if (im.err_q8) (void)hipFree(im.err_q8);
    if (im.kxy) (void)hipFree(im.kxy);
    im = Image{};
}

constexpr int kSlots = 2 * MSFM_MAX_IMAGES + 2;  // ids >= MSFM_MAX_IMAGES: auxiliary (top-scale subsets, two operator-level scratch slots)
// Sub-batches of msfm_match_pairs are bounded by a pair count and by the device scratch of ALL scratch sets in flight together
// (msfm_pair_scratch_bytes per pair, msfm_hostutil.h): at most kDefaultScratchBytes, and never more than a quarter of what the
// device has free when the call starts (hipMemGetInfo + what the sets already hold) -- a 9-second job gains ~2.5 % from 7
// instead of 19 sub-batches per 80 000 pairs, but every GiB of scratch costs ~10 ms the first time it is allocated
// (profiles/r04_scratch_ab.txt).
constexpr long long kDefaultScratchBytes = (long long)64 << 30;
constexpr int kDefaultMaxPairsPerBatch = 16384;
constexpr int kMaxPairsPerBatchLimit = 65535;   // gridDim.y
// A call large enough is cut into at least this many sub-batches so that the bandwidth-bound tail of one (thresholds, plan,
// exact re-check, epilogue, copy-out) runs under the next one's sweep 1; every sub-batch keeps >= kMinPipelineCost descriptor pairs
// (a few ms of sweep 1) so that the fixed costs of a sub-batch stay small.  TWO EQUAL parts since round 4: with ~5 ms of tail
// kernels per 8128-pair job (7.3 in round 3, when six parts shrinking to 0.3 of the average were best) the first part's tail hides
// under the second part's sweep and every further cut costs more -- another sub-batch's fill and drain, sweeps stretched by the
// tails beside them -- than it hides: 39.5 -> 37.3 ms against six parts, 39.4 against one (profiles/r04_pipeline_ab.txt, one box,
// alternated, twice).  Jobs cut by memory or by the pair limit anyway (config 3, config 4) are not affected.
constexpr int kDefaultPipeline = 2;
constexpr double kDefaultTaper = 1.0;   // size of a call's last part relative to the average part
// Sub-batches in flight (streams / scratch sets).  With three, sweep 1 of sub-batch k + 2 is ordered behind sweep 2 of sub-batch k
// (Scratch::sweep2_done): the matrix pipes see S1(k+1) S2(k) S1(k+2) S2(k+1) ... and every bandwidth-bound tail has a sweep to run
// beside -- what the many-sub-batch jobs (config 4: 105 of them) live on.
constexpr int kInFlight = 3;
constexpr long long kMinPipelineCost = 15000000000LL;

}  // namespace

// Everything ONE device sub-batch in flight owns: its stream, the partial / plan / candidate / result scratch, the upload arenas
// with their page-locked staging, the page-locked words the host reads at the end of the sub-batch, its share of the profile.  A
// context has kInFlight (three) of them: while the tail of sub-batch k (thresholds, plan, sweep 2, exact re-check, epilogue,
// copy-out) runs on one stream, the sweeps of sub-batches k + 1 and k + 2 are already queued on the others (match_pairs_impl).
// Buffers grow on demand (DevBuf::ensure = hipFree + hipMalloc, both of which synchronise the DEVICE: a growth inside issue()
// serialises the pipeline once -- in the first call of a job shape, and whenever a later sub-batch is larger than any before --
// and is also what makes re-using a buffer safe that kernels queued earlier still read; steady state allocates nothing).
struct PfPending {                // what the end-of-batch synchronisation has to look at
    bool active = false, compact = false, i8 = false, q8 = false;
    size_t n_lists = 0, P = 0;
    long long rows_cap = 0, cand_cap = 0, items_cap = 0;
    int compact_pairs = 0;
    long long dense_swept = 0;
    size_t ev_base = 0;
};

struct Scratch {
    hipStream_t stream = nullptr;
    // upload arenas + their page-locked staging: [0] pair tables of the matrix-core route, [1] plan tables of sweep 2, [2] pair tables
    // of the brute-force route (a sub-batch may run both routes: the first route's copy may still be in flight)
    DevBuf d_up[3];
    PinnedBuf h_up[3];
    Ref d_pairs, d_pf, d_pfq, d_pf16, d_item_base, d_item_base16;                                   // in d_up[0] / d_up[2]
    Ref d_groups, d_gmembers, d_member_pair, d_member_group, d_ppair;       // in d_up[1]
    DevBuf d_items;
    DevBuf d_rp_s0, d_rp_i0, d_rp_s1, d_cp_s0, d_cp_i0, d_cp_s1;
    DevBuf d_k_i0, d_k_d0, d_k_d1;
    DevBuf d_st_qt, d_st_d, d_counts, d_offsets, d_sens;
    DevBuf d_sub_qt, d_sub_d;         // the sub-batch's match lists, compact (CSR order), before they join the call's lists
    DevBuf d_fix_count, d_fix_list;
    int fix_cap_eff = 0;              // capacity handed to the kernels of the current batch (0: ties need no fix-up)
    bool keys_epilogue = false;       // the epilogue reads the reduce slots of the exact re-check itself: no pf_finalize_kernel, no kNN arrays
    // prefilter path
    DevBuf d_tu, d_tv, d_cand, d_cand_count, d_best, d_second;
    DevBuf d_cmp_tu, d_live_idx, d_row_pair, d_row_src, d_vpairs, d_vpf, d_vitems, d_lists, d_items16;
    DevBuf d_cmp_s0, d_cmp_s1, d_summary_a;   // route Q: sweep 1' row results, summary of plan A
    DevBuf d_cand_val, d_cmp_n2;              // integer route: the candidates' accumulators (parallel to d_cand), n' per compacted row
    // device-side plan of the compacted sweep 2 (msfm_plan.hip.h)
    DevBuf d_colmask, d_gtot, d_grow0, d_cnt, d_mrow, d_summary, d_overflow, d_totals;
    PfPending pf_pending;
    PinnedBuf h_summary;              // PlanSummary | totals[2] | overflow bytes
    PinnedBuf h_tail;                 // tie-queue count | CSR offsets [P + 1] | certificate counts [P]: read at the end of the sub-batch
    // geometric verification
    DevBuf d_vf_pairs, d_vf_x1, d_vf_y1, d_vf_x2, d_vf_y2, d_vf_hyp, d_vf_best_it, d_vf_best_count, d_vf_flags,
        d_st2_qt, d_st2_d, d_counts2;
    msfm_profile prof = {};           // this sub-batch's share; joins the call's profile when the sub-batch is accepted
    hipEvent_t sweep1_done = nullptr; // recorded behind sweep 1: the other stream's next sweep 1 waits for it
    bool sweep1_recorded = false;
    hipEvent_t sweep2_done = nullptr; // recorded behind sweep 2: the sweep 1 two sub-batches later waits for it (three sets in flight)
    bool sweep2_recorded = false;
    long long seq = 0;                // number of the sub-batch this set works on (msfm_ctx::issue_seq)
    void for_each_buf(void (*fn)(DevBuf&, void*), void* arg) {
        DevBuf* bufs[] = {&d_up[0], &d_up[1], &d_up[2], &d_items, &d_rp_s0, &d_rp_i0, &d_rp_s1, &d_cp_s0, &d_cp_i0, &d_cp_s1, &d_k_i0, &d_k_d0,
                          &d_k_d1, &d_st_qt, &d_st_d, &d_counts, &d_offsets, &d_sens, &d_sub_qt, &d_sub_d, &d_fix_count, &d_fix_list, &d_tu,
                          &d_tv, &d_cand, &d_cand_count, &d_best, &d_second, &d_cmp_tu, &d_live_idx, &d_row_pair, &d_row_src,
                          &d_vpairs, &d_vpf, &d_vitems, &d_lists, &d_colmask, &d_gtot, &d_grow0, &d_cnt, &d_mrow, &d_summary, &d_overflow,
                          &d_totals, &d_vf_pairs,
                          &d_vf_x1, &d_vf_y1, &d_vf_x2, &d_vf_y2, &d_vf_hyp, &d_vf_best_it, &d_vf_best_count, &d_vf_flags, &d_st2_qt,
                          &d_st2_d, &d_counts2, &d_cmp_s0, &d_cmp_s1, &d_summary_a, &d_items16, &d_cand_val, &d_cmp_n2};
        for (DevBuf* b : bufs) fn(*b, arg);
    }
    long long device_bytes() {
        long long sum = 0;
        for_each_buf([](DevBuf& b, void* a) { *static_cast<long long*>(a) += (long long)b.cap; }, &sum);
        return sum;
    }
    void release_all() {
        for_each_buf([](DevBuf& b, void*) { b.release(); }, nullptr);
        for (PinnedBuf& h : h_up) h.release();
        h_summary.release();
        h_tail.release();
    }
};

struct msfm_ctx {
    int device = 0;
    int order = MSFM_ORDER_SSE4X4;
    int cu_count = 0, clock_mhz = 0;
    char dev_name[256] = {0};
    std::vector<Image> images;
    std::string err;

    Scratch sc[kInFlight];
    Scratch* cur = &sc[0];            // the scratch set (and stream) the batch functions work on
    Scratch* last_sweep1 = nullptr;   // the set whose sweep 1 was launched last in this call: the next sweep 1 waits for it
    DevBuf d_stage, d_maxima, d_zero_row;   // upload staging, upload-time maxima, the all-zero operand row
    DevBuf d_out_qt, d_out_d;         // the match lists of the whole call (msfm_fetch_matches_device)
    int fix_cap = 1 << 16;            // entries of the sqrt-space tie queue; grows on overflow (the sub-batch is re-run)
    // sub-batch limits of msfm_match_pairs (msfm_set_limits / MSFM_MAX_PAIRS_PER_BATCH / MSFM_SCRATCH_MIB)
    int max_pairs_per_batch = kDefaultMaxPairsPerBatch;
    long long scratch_bytes = 0;      // msfm_set_limits / MSFM_SCRATCH_MIB: total for the scratch sets in flight; 0 = automatic (above)
    long long issue_seq = 0;          // sub-batches issued so far
    double pipeline_taper = kDefaultTaper;   // size of a call's last part relative to the average part (MSFM_PIPELINE_TAPER; 1: equal parts)
    int in_flight = kInFlight;        // scratch sets used (MSFM_IN_FLIGHT=1 at msfm_create: no sub-batch overlap, for A/B measurements)
    int pipeline = kDefaultPipeline;  // sub-batches a large call is cut into at least, so that tails overlap sweeps (1: off)
    int prefilter = 1;                // 0: brute force only; 1: MFMA prefilter, integer matrix cores for byte stores; 2: fp16 MFMA only
    int byte_detect = 1;              // a float upload holding only integers in [0, 255] is a byte store (MSFM_BYTE_DETECT=0: off)
    float q8_level = 0.f;             // m: largest value of the twinned images so far, rounded up to a multiple of 1/16 (msfm_q8.hip.h)
    int q8_direct = 1;                // thresholds for sweep 2 straight from the twins' sweep when they are fine enough (MSFM_Q8_DIRECT=0: never, 2: always)
    int q8_route = 1;                 // float images in [0, 1] get byte twins and their first sweep on the integer cores (MSFM_Q8=0: off)
    long long cmp_rows_hint = 0;      // compacted rows the previous batch needed (sizes the next batch's buffers)
    long long items_hint = 0, cand_hint = 0;   // likewise: work items, candidate-list capacity

    // results of the last msfm_match_pairs call
    bool have_results = false;
    std::vector<int64_t> res_offsets;
    std::vector<int32_t> res_sens;   // per pair: rows / columns without an order-invariance certificate
    PinnedBuf res_qt, res_dist;  // (q, t) int32 pairs and distances of res_count matches
    size_t res_count = 0;

    msfm_profile prof = {};
    std::vector<hipEvent_t> ev_pool;
};

#define SC (*ctx->cur)

namespace {

int fail(msfm_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(ctx, MSFM_E_DEVICE,                                                   \
                        std::string(#call) + ": " + hipGetErrorString(e__));                  \
    } while (0)


// MSFM_DEBUG_TIMING=1: host-side wall clock of the orchestration phases of each batch on stderr
struct HostClock {
    bool on = std::getenv("MSFM_DEBUG_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[msfm host] %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// MSFM_DEBUG_SYNC=1: synchronise after every launch of the prefilter path and name it on stderr
// (a faulting kernel is then the one named last)
#define DBGSYNC(ctx, name)                                                        \
    do {                                                                          \
        static const bool on__ = std::getenv("MSFM_DEBUG_SYNC") != nullptr;       \
        if (on__) {                                                               \
            std::fprintf(stderr, "[msfm] %s ...", name);                          \
            hipError_t e__ = hipStreamSynchronize((ctx)->cur->stream);                \
            std::fprintf(stderr, " %s\n", hipGetErrorString(e__));                \
        }                                                                         \
    } while (0)

struct Batch {
    std::vector<PairDesc> pairs;
    std::vector<PfPair> pf;
    std::vector<int> id1, id2;       // store slots of the pairs' images
    std::vector<int> item_base;      // per pair: index of its first work item in the linear list (-1: none on this path)
    size_t n_items = 0, items_per_xcd = 0;   // length of the XCD-interleaved list (a multiple of 8), items per XCD chunk
    long long rp_elems = 0, cp_elems = 0, kf_elems = 0, kr_elems = 0, out_elems = 0, cand_elems = 0;
    int max_npad = 0;
    int64_t desc_pairs = 0;
    int64_t algo_bytes = 0;
    // host tables of copies that are still in flight when the issuing function returns (they live as long as the sub-batch)
    std::vector<CandList> dense_lists;
    std::vector<VerifyPair> verify_pairs;
};

// Work items of the pairs whose `path` matches: only their NUMBERING is made on the host -- per pair the index of its
// first item in the linear list -- the list itself (85 000 items of 32 B for the bench job) is written by
// build_items_kernel from the pair descriptors.  Items of one pair are contiguous; the linear list is cut into 8 chunks,
// one per XCD (workgroup b runs on XCD b % 8): linear item k sits at position (k % per) * 8 + k / per.
// (`only`: a subset of the pairs -- the list of one route of a mixed sub-batch -- into the given outputs instead of the batch's own)
void build_items(Batch& b, int path, const std::vector<char>* only = nullptr, std::vector<int>* base_out = nullptr, size_t* per_out = nullptr) {
    // Pairs in the order of their STREAMED image (id2): the 32 workgroups of an XCD walk 32 consecutive items of their
    // chunk at a time, and those then stream the same B image -- one image (1.4 MB at 5000 rows) stays in the XCD's 4 MB
    // L2 while ~30 workgroups read it, instead of three or four images evicting one another (pair order = id1-major:
    // measured 69 GB of L2 misses per sweep-1 launch on the bench job against 11 GB of distinct B bytes).
    std::vector<int> order(b.pairs.size());
    for (size_t p = 0; p < order.size(); ++p) order[p] = (int)p;
    if (b.id2.size() == b.pairs.size())
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b.id2[(size_t)x] < b.id2[(size_t)y]; });
    std::vector<int>& base = base_out ? *base_out : b.item_base;
    base.assign(b.pairs.size(), -1);
    long long n = 0;
    for (size_t q = 0; q < order.size(); ++q) {
        const size_t p = (size_t)order[q];
        const PairDesc& pd = b.pairs[p];
        if (!pd.valid || pd.path != path) continue;
        if (only && !(*only)[p]) continue;
        base[p] = (int)n;
        n += (long long)pd.ranges * (path == 1 ? pd.a_blocks256 : pd.a_blocks);
    }
    if (per_out) {
        *per_out = (size_t)((n + 7) / 8);
        return;
    }
    b.items_per_xcd = (size_t)((n + 7) / 8);
    b.n_items = b.items_per_xcd * 8;
}

// one workgroup per pair: its items, in the order (range, A block), at their XCD-interleaved positions
__global__ void build_items_kernel(const PairDesc* __restrict__ pairs, const int* __restrict__ item_base, int path, int per,
                                   WorkItem* __restrict__ items) {
    MSFM_TAIL_PRIO();
    const int p = blockIdx.x;
    const int base = item_base[p];
    if (base < 0) return;
    const PairDesc pd = pairs[p];
    const int nab = path == 1 ? pd.a_blocks256 : pd.a_blocks;
    for (int i = threadIdx.x; i < pd.ranges * nab; i += blockDim.x) {
        const int r = i / nab, ab = i - r * nab;
        WorkItem w = {};
        w.pair = p;
        w.a_blk = ab;
        w.bt_begin = (int)((long long)pd.b_tiles * r / pd.ranges);
        w.bt_end = (int)((long long)pd.b_tiles * (r + 1) / pd.ranges);
        w.range = r;
        const int k = base + i;
        items[(size_t)(k % per) * 8 + (size_t)(k / per)] = w;
    }
}

int fill_pair(msfm_ctx* ctx, int id1, int id2, PairDesc& pd, PfPair& pp) {
    if (id1 < 0 || id1 >= kSlots || id2 < 0 || id2 >= kSlots)
        return fail(ctx, MSFM_E_INVALID, "image id out of range");
    const Image& a = ctx->images[id1];
    const Image& b = ctx->images[id2];
    if (a.n < 0 || b.n < 0) return fail(ctx, MSFM_E_NOIMAGE, "image not uploaded: " + std::to_string(a.n < 0 ? id1 : id2));
    pd = PairDesc{};
    pd.a_panel = a.panel;
    pd.b_panel = b.panel;
    pd.a_raw = a.raw;
    pd.b_raw = b.raw;
    pd.a_rawp = a.rawp;
    pd.b_rawp = b.rawp;
    pd.n1 = a.n;
    pd.n2 = b.n;
    pd.a_blocks = a.nblk;
    pd.b_tiles = b.nblk;
    pd.n1pad = a.nalloc * kBM;
    pd.n2pad = b.nalloc * kBN;
    pd.a_blocks256 = a.nalloc * kBM / kPfWgRows;  // sweep work items: kPfWgRows A rows each (the name dates from 256)
    pd.ranges = 1;
    // empty query or train set: knnMatch returns nothing, no device work
    pd.valid = (a.n >= 1 && b.n >= 1) ? 1 : 0;
    pd.exact_int = (a.from_u8 && b.from_u8) ? 1 : 0;
    pp = PfPair{};
    pp.a_h = a.h16;
    pp.b_h = b.h16;
    pp.a_nrm = a.nrm;
    pp.b_nrm = b.nrm;
    pp.a_nrm_max = a.nrm_max;
    pp.b_nrm_max = b.nrm_max;
    pp.a_c = a.c;
    pp.b_c = b.c;
    // the MFMA prefilter needs fp16-representable magnitudes on both sides, and norms of comparable scale
    // (one image's norms are expressed in units of the other's c)
    const bool scales_ok = a.nrm_max <= 8.f * b.nrm_max && b.nrm_max <= 8.f * a.nrm_max;
    pp.use = (ctx->prefilter && pd.valid && a.pf_safe && b.pf_safe && scales_ok) ? 1 : 0;
    pd.path = pp.use;
    return MSFM_OK;
}

// offsets every path shares: final kNN arrays and staged match lists
void assign_common(Batch& b) {
    for (auto& pd : b.pairs) {
        pd.kf_off = b.kf_elems;
        pd.kr_off = b.kr_elems;
        pd.out_off = b.out_elems;
        if (!pd.valid) continue;
        b.kf_elems += pd.n1pad;
        b.kr_elems += pd.n2pad;
        b.out_elems += pd.n1;
        b.desc_pairs += (int64_t)pd.n1 * pd.n2;
        // compulsory traffic, no cross-pair reuse: both descriptor sets once + both knn lists
        b.algo_bytes += ((int64_t)pd.n1 + pd.n2) * kDim * 4 + ((int64_t)pd.n1 + pd.n2) * 12;
        b.max_npad = std::max(b.max_npad, std::max(pd.n1pad, pd.n2pad));
    }
    // reverse arrays live behind the forward ones in the same buffers
    for (auto& pd : b.pairs) pd.kr_off += b.kf_elems;
}

// partial-result offsets + B-range split of the pairs on `path` (1: prefilter doubles the partial
// slots: one column partial per wave of an A block)
void assign_partials(Batch& b, int path, int target_items) {
    b.rp_elems = b.cp_elems = 0;
    long long total_ablocks = 0;
    for (auto& pd : b.pairs)
        if (pd.valid && pd.path == path) total_ablocks += (path == 1 ? pd.a_blocks256 : pd.a_blocks);
    const int rmult = 1;
    for (auto& pd : b.pairs) {
        if (!pd.valid || pd.path != path) continue;
        pd.ranges = 1;
        if (total_ablocks > 0 && total_ablocks < target_items) {
            long long r = (target_items + total_ablocks - 1) / total_ablocks;
            pd.ranges = (int)std::max<long long>(1, std::min<long long>(r, pd.b_tiles));
        }
        pd.rp_off = b.rp_elems;
        pd.cp_off = b.cp_elems;
        b.rp_elems += (long long)pd.ranges * rmult * pd.n1pad;
        // prefilter: one column partial per 256-row A block (the four waves are merged in LDS)
        b.cp_elems += (long long)(path == 1 ? pd.a_blocks256 : pd.a_blocks) * pd.n2pad;
    }
}

hipEvent_t get_event(msfm_ctx* ctx, size_t i) {
    while (ctx->ev_pool.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_pool[i];
}

__global__ void fill_segs_kernel(FillSegs segs) {
    MSFM_TAIL_PRIO();
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (unsigned long long)gridDim.x * blockDim.x;
    for (int k = 0; k < segs.n; ++k) {
        const FillSeg e = segs.s[k];
        const unsigned w = e.value * 0x01010101u;
        // (device allocations and arena offsets are 256-byte aligned: 16-byte stores, then the tail)
        const unsigned long long n16 = e.bytes >> 4;
        uint4* p16 = reinterpret_cast<uint4*>(e.p);
        for (unsigned long long i = tid; i < n16; i += nt) p16[i] = make_uint4(w, w, w, w);
        for (unsigned long long i = (n16 << 4) + tid; i < e.bytes; i += nt) reinterpret_cast<unsigned char*>(e.p)[i] = (unsigned char)e.value;
    }
}

struct FillBatch {
    FillSegs segs = {};
    unsigned long long bytes = 0;
    std::vector<FillSegs> full;
    void add(void* p, size_t n, unsigned value) {
        if (!p || n == 0) return;
        if (segs.n == kFillSegs) {
            full.push_back(segs);
            segs = FillSegs{};
        }
        segs.s[segs.n++] = FillSeg{p, (unsigned long long)n, value & 255u};
        bytes += n;
    }
    hipError_t launch(hipStream_t stream) {
        if (segs.n) full.push_back(segs);
        // one wave per SIMD and CU at most: small enough to run beside a resident sweep workgroup, enough stores in flight for the
        // tens of MB the row-source table of a sub-batch takes
        const unsigned grid = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(256, (bytes + 65535) / 65536));
        for (const FillSegs& f : full) hipLaunchKernelGGL(fill_segs_kernel, dim3(grid), dim3(256), 0, stream, f);
        full.clear();
        segs = FillSegs{};
        bytes = 0;
        return hipGetLastError();
    }
};

// The pair tables of one route of the sub-batch (slot 0: matrix-core route, 2: brute-force route) in ONE copy, the work-item list
// cleared (pair = -1: padding item) together with `more` in ONE fill launch, the items written by build_items_kernel.
int upload_pair_tables(msfm_ctx* ctx, Batch& b, int path, const std::vector<PfPair>* pfq, FillBatch& fills, const std::vector<PfPair>* pf16 = nullptr,
                       const std::vector<int>* item_base16 = nullptr, size_t per16 = 0) {
    const size_t P = b.pairs.size();
    const int slot = path == 1 ? 0 : 2;
    UploadPlan up;
    up.add(SC.d_pairs, b.pairs.data(), P * sizeof(PairDesc));
    up.add(SC.d_pf, b.pf.data(), P * sizeof(PfPair));
    if (pfq) up.add(SC.d_pfq, pfq->data(), P * sizeof(PfPair));
    if (pf16) up.add(SC.d_pf16, pf16->data(), P * sizeof(PfPair));
    if (item_base16) up.add(SC.d_item_base16, item_base16->data(), item_base16->size() * 4);
    up.add(SC.d_item_base, b.item_base.data(), b.item_base.size() * 4);
    HIPCHK(ctx, up.place_and_copy(SC.d_up[slot], SC.h_up[slot], SC.stream));
    HIPCHK(ctx, SC.d_items.ensure(std::max<size_t>(1, b.n_items) * sizeof(WorkItem)));
    fills.add(SC.d_items.p, b.n_items * sizeof(WorkItem), 0xff);
    if (item_base16 && per16) {
        HIPCHK(ctx, SC.d_items16.ensure(per16 * 8 * sizeof(WorkItem)));
        fills.add(SC.d_items16.p, per16 * 8 * sizeof(WorkItem), 0xff);
    }
    HIPCHK(ctx, fills.launch(SC.stream));
    if (b.n_items == 0) return MSFM_OK;
    hipLaunchKernelGGL(build_items_kernel, dim3((unsigned)P), dim3(64), 0, SC.stream, (const PairDesc*)SC.d_pairs.as<PairDesc>(),
                       (const int*)SC.d_item_base.as<int>(), path, (int)b.items_per_xcd, SC.d_items.as<WorkItem>());
    HIPCHK(ctx, hipGetLastError());
    if (item_base16 && per16) {   // the second route's own list: a kernel that walks the common list and skips pays ~17 us per skipped item
        hipLaunchKernelGGL(build_items_kernel, dim3((unsigned)P), dim3(64), 0, SC.stream, (const PairDesc*)SC.d_pairs.as<PairDesc>(),
                           (const int*)SC.d_item_base16.as<int>(), path, (int)per16, SC.d_items16.as<WorkItem>());
        HIPCHK(ctx, hipGetLastError());
    }
    return MSFM_OK;
}

// ---- static tables of the device-side plan of sweep 2 (msfm_plan.hip.h) -----------------------------------------
struct CompactPlan {
    std::vector<PlanGroup> groups;
    std::vector<int> member_group;  // member -> group
    std::vector<int> gmembers;      // member ids ordered by group
    std::vector<int> member_pair;   // member -> pair of the batch
    std::vector<PlanPair> ppair;    // per pair
    long long rows_ub = 0;          // compacted rows if every row were alive (each live column counted once)
    long long rows_ub_all_bits = 0; // ... each column once per block group (plan A of route Q)
    int pairs = 0;
};

// Which groups exist and which (pair, direction[, block bit]) members they consist of: a function of the pair list
// alone.  O(pairs + members): counting sort over dense image indices, no maps (this runs on the host while the GPU
// is busy with sweep 1).
void build_compact_plan(msfm_ctx* ctx, const Batch& b, CompactPlan& cp, bool no_ranges) {
    const size_t P = b.pairs.size();
    cp.ppair.assign(P, PlanPair{-1, 0, 0, 0});
    static thread_local std::vector<int> dense;   // store slot -> dense image index of this batch (-1: absent)
    dense.assign((size_t)kSlots, -1);
    std::vector<int> img_slot;                     // dense index -> store slot
    auto dense_of = [&](int slot) {
        if (dense[(size_t)slot] < 0) {
            dense[(size_t)slot] = (int)img_slot.size();
            img_slot.push_back(slot);
        }
        return dense[(size_t)slot];
    };
    // group keys: forward = streamed image id2; reverse = (streamed image id1, bit)
    std::vector<int> fwd_group_of_img, rev_group0_of_img;   // dense image -> group id (-1: none yet)
    std::vector<int>& member_group = cp.member_group;
    for (size_t p = 0; p < P; ++p) {
        const PairDesc& pd = b.pairs[p];
        const PfPair& pp = b.pf[p];
        if (!pd.valid || !pp.use) continue;
        ++cp.pairs;
        const int di = dense_of(b.id1[p]), dj = dense_of(b.id2[p]);
        const size_t need = img_slot.size();
        if (fwd_group_of_img.size() < need) {
            fwd_group_of_img.resize(need, -1);
            rev_group0_of_img.resize(need, -1);
        }
        // forward: live rows of image 1 against all of image 2
        if (fwd_group_of_img[(size_t)dj] < 0) {
            fwd_group_of_img[(size_t)dj] = (int)cp.groups.size();
            PlanGroup g = {};
            g.b_h = pp.b_h;
            g.b_nrm = pp.b_nrm;
            g.b_c = pp.b_c;
            g.a_c = pp.a_c;     // (per image: every image that meets image j in this batch has a compatible scale, see fill_pair)
            g.b_h0 = pp.b_h0;
            g.b_n2 = pp.b_n2;
            g.dir = 0;
            g.bt_begin = 0;
            g.bt_end = pd.b_tiles;
            g.n2 = pd.n2;
            g.n2pad = pd.n2pad;
            g.b_tiles = pd.b_tiles;
            g.ranges = 1;
            cp.groups.push_back(g);
        }
        cp.ppair[p].fwd_member = (int)cp.member_pair.size();
        cp.member_pair.push_back((int)p);
        member_group.push_back(fwd_group_of_img[(size_t)dj]);
        cp.rows_ub += pd.n1;
        // reverse: live columns against the 512-row blocks of image 1 their mask names
        const int nb = pd.a_blocks256, gshift = (nb + 31) / 32, bits = (nb + gshift - 1) / gshift;
        if (rev_group0_of_img[(size_t)di] < 0) {
            rev_group0_of_img[(size_t)di] = (int)cp.groups.size();
            for (int bit = 0; bit < bits; ++bit) {
                PlanGroup g = {};
                g.b_h = pp.a_h;
                g.b_nrm = pp.a_nrm;
                g.b_c = pp.a_c;
                g.a_c = pp.b_c;
                g.b_h0 = pp.a_h0;
                g.b_n2 = pp.a_n2;
                g.dir = 1;
                g.bt_begin = std::min(pd.a_blocks, bit * gshift * (kPfWgRows / kBM));
                g.bt_end = std::min(pd.a_blocks, (bit + 1) * gshift * (kPfWgRows / kBM));
                g.n2 = pd.n1;
                g.n2pad = pd.n1pad;
                g.b_tiles = pd.a_blocks;
                g.ranges = 1;
                cp.groups.push_back(g);
            }
        }
        cp.ppair[p].rev_member0 = (int)cp.member_pair.size();
        cp.ppair[p].rev_bits = bits;
        for (int bit = 0; bit < bits; ++bit) {
            cp.member_pair.push_back((int)p);
            member_group.push_back(rev_group0_of_img[(size_t)di] + bit);
        }
        cp.rows_ub += pd.n2;
        cp.rows_ub_all_bits += (long long)pd.n1 + (long long)pd.n2 * bits;
    }
    // counting sort of the members by group
    const size_t G = cp.groups.size(), M = cp.member_pair.size();
    for (size_t m = 0; m < M; ++m) cp.groups[(size_t)member_group[m]].count += 1;
    int at = 0;
    for (size_t g = 0; g < G; ++g) {
        cp.groups[g].first = at;
        at += cp.groups[g].count;
        cp.groups[g].count = 0;
    }
    cp.gmembers.assign(M, 0);
    for (size_t m = 0; m < M; ++m) {
        PlanGroup& g = cp.groups[(size_t)member_group[m]];
        cp.gmembers[(size_t)(g.first + g.count++)] = (int)m;
    }
    // a small batch would leave most CUs idle with one work item per 512 compacted rows: split the streamed ranges
    // (route Q keeps whole streamed ranges: sweep 1' writes one row result per compacted row)
    const long long target = 4LL * ctx->cu_count;
    if ((long long)G < target && !no_ranges)
        for (PlanGroup& g : cp.groups) {
            const long long r = (target + (long long)G - 1) / (long long)G;
            g.ranges = (int)std::max<long long>(1, std::min<long long>(r, g.bt_end - g.bt_begin));
        }
}

// layout of Scratch::h_summary: PlanSummary (plan B, the one sweep 2 ran on) | PlanSummary (plan A of route Q) | totals[2] | overflow bytes
constexpr size_t kHsTotals = 2 * sizeof(PlanSummary), kHsOverflow = kHsTotals + 16;

// MFMA prefilter + exact re-check for the pairs on path 1, WITHOUT a host synchronisation: the caller looks at
// SC.pf_pending at the end of the batch (finish_prefilter) and re-runs the batch if a capacity was exceeded or a
// candidate list overflowed.
//   sweep 1 (sweep_kernel<1>): S~ minima per row / column -> thresholds (with pruning for match lists)
//   sweep 2: match lists: only the rows / columns pruning left alive, compacted and grouped per streamed image
//            (sweep_kernel<3>, plan built on the device); kNN-level API: everything again (sweep_kernel<2>)
//   exact pinned-order S of the candidates, 64-bit atomicMin reduce, finalize
int run_prefilter(msfm_ctx* ctx, Batch& b, size_t ev_base, PruneParams prune) {
    const size_t P = b.pairs.size();
    HostClock hc;
    SC.pf_pending = PfPending{};
    assign_partials(b, 1, 4 * ctx->cu_count);
    // Sweep 2 on the compacted live rows, or on everything again?  Decided per batch, before any result exists: the Lowe
    // test is what kills rows (~94 % at ratio 0.8 on SIFT-like data); with a ratio near or above 1 almost every row stays
    // alive and the compacted sweep (both directions separately) would multiply up to twice what the dense one does.
    const bool compact = prune.prune != 0 && prune.ratio > 0.f && prune.ratio <= 0.95f;
    // Byte stores: both sweeps on the integer matrix cores (msfm_sweep_i8.hip.h) when every prefiltered pair of the batch
    // joins two byte images (a store is bytes throughout or not at all; a mixed batch takes the fp16 kernels)
    bool i8 = compact && ctx->prefilter == 1;
    for (size_t p = 0; p < P && i8; ++p)
        if (b.pairs[p].valid && b.pf[p].use) i8 = ctx->images[b.id1[p]].is_u8 && ctx->images[b.id2[p]].is_u8;
    if (i8)
        for (size_t p = 0; p < P; ++p) {
            if (!b.pairs[p].valid || !b.pf[p].use) continue;
            const Image& ia = ctx->images[b.id1[p]];
            const Image& ib = ctx->images[b.id2[p]];
            PfPair& pp = b.pf[p];
            pp.i8 = 1;
            pp.a_h = reinterpret_cast<const _Float16*>(ia.i8);   // 176-byte rows behind the same pointers
            pp.b_h = reinterpret_cast<const _Float16*>(ib.i8);
            pp.a_nrm = ia.nrm_i8;
            pp.b_nrm = ib.nrm_i8;
            pp.a_nrm_max = ia.nrm_i8_max;
            pp.b_nrm_max = ib.nrm_i8_max;
            pp.a_c = pp.b_c = 0.f;
            pp.a_h0 = ia.h0_i8;
            pp.b_h0 = ib.h0_i8;
            pp.a_n2 = ia.n2_i8;
            pp.b_n2 = ib.n2_i8;
        }
    SC.pf_pending.i8 = i8;
    // Route Q (msfm_q8.hip.h): float images with byte twins -- sweep 1 on the twins (integer matrix cores); coarse twins: an fp16
    // sweep 1' on the rows that survive.  A pair takes it when both its images have twins; a sub-batch in which only SOME pairs do runs
    // two first sweeps (fine twins: q8_mixed below) or keeps the fp16 route for all of them (coarse twins).
    bool q8 = compact && !i8 && ctx->prefilter == 1 && ctx->q8_route;
    long long q8_rows = 0, q8_pairs = 0, twin_pairs = 0, twin_rows = 0;
    std::vector<char> twin;
    if (q8) {
        twin.assign(P, 0);
        for (size_t p = 0; p < P; ++p)
            if (b.pairs[p].valid && b.pf[p].use) {
                q8_rows += b.pairs[p].n1 + b.pairs[p].n2;
                q8_pairs += 1;
                if (ctx->images[b.id1[p]].q8 != nullptr && ctx->images[b.id2[p]].q8 != nullptr) {
                    twin[p] = 1;
                    twin_pairs += 1;
                    twin_rows += b.pairs[p].n1 + b.pairs[p].n2;
                }
            }
    }
    // (two plans and three sweeps only pay on real images: batches of small ones -- the pre-emptive filter's 100-row
    // subsets -- keep the fp16 route; MSFM_Q8=2 lifts the limit, for the tests)
    const bool fine_twins = ctx->q8_direct == 2 || (ctx->q8_direct == 1 && ctx->q8_level <= kQ8DirectMaxLevel);
    bool q8_mixed = false;   // some pairs join images without twins: THEIR sweep 1 runs on the fp16 cores, the twins' on the integer cores
    if (q8 && twin_pairs < q8_pairs) {
        // (fine twins only: the coarse route's plan A / sweep 1' cover whole sub-batches; and only when a quarter of the work or more has twins)
        q8_mixed = fine_twins && twin_pairs > 0 && 4 * twin_rows >= q8_rows;
        if (!q8_mixed) q8 = false;
    }
    if (q8 && ctx->q8_route < 2 && (twin_pairs == 0 || twin_rows < 2 * 1024 * twin_pairs)) q8 = q8_mixed = false;
    std::vector<PfPair> pfq, pf16;
    if (q8) {
        pfq = b.pf;
        for (size_t p = 0; p < P; ++p) {
            if (!b.pairs[p].valid || !b.pf[p].use) continue;
            if (!twin[p]) {   // (mixed sub-batch: not a pair of the twins' sweep)
                pfq[p].use = 0;
                continue;
            }
            const Image& ia = ctx->images[b.id1[p]];
            const Image& ib = ctx->images[b.id2[p]];
            PfPair& pp = pfq[p];
            pp.i8 = 1;
            pp.a_h = reinterpret_cast<const _Float16*>(ia.q8);
            pp.b_h = reinterpret_cast<const _Float16*>(ib.q8);
            pp.a_nrm = ia.nrm_q8;
            pp.b_nrm = ib.nrm_q8;
            pp.a_c = ia.err_q8_max;     // (the twin pair carries the images' largest quantisation errors here)
            pp.b_c = ib.err_q8_max;
            pp.a_h0 = ia.h0_q8;
            pp.b_h0 = ib.h0_q8;
            pp.a_err = ia.err_q8;
            pp.b_err = ib.err_q8;
        }
    }
    // The route is chosen per sub-batch: one pair that cannot take an integer route sends all of them to the fp16 kernels (same
    // results, ~1.6 x the sweep time).  Counted, so that a mixed store shows up in the profile instead of only in the clock.
    if (compact && ctx->prefilter == 1 && !i8 && !q8)   // (a mixed sub-batch of fine twins keeps the twins' pairs on the integer cores: q8_mixed)
        for (size_t p = 0; p < P; ++p) {
            if (!b.pairs[p].valid || !b.pf[p].use) continue;
            const Image& ia = ctx->images[b.id1[p]];
            const Image& ib = ctx->images[b.id2[p]];
            const bool twins = ctx->q8_route && ia.q8 && ib.q8 && (ctx->q8_route == 2 || b.pairs[p].n1 + b.pairs[p].n2 >= 2048);
            if ((ia.is_u8 && ib.is_u8) || twins) SC.prof.demoted_pairs += 1;
        }
    // fine twins: their sweep's bounds are the thresholds of sweep 2; coarse ones (a store with values near 1): an fp16 sweep 1'
    // of the live rows refines them first
    const bool q8_direct = q8 && fine_twins;
    const bool q8_refine = q8 && !q8_direct;
    SC.pf_pending.q8 = q8_refine;   // (a plan A and a sweep 1' to account for at the end of the batch)
    long long dense_cand = 0;
    for (size_t p = 0; p < P; ++p) {
        b.pf[p].tu_off = b.pairs[p].kf_off;
        b.pf[p].tv_off = b.pairs[p].kr_off;  // same combined index space as the kNN arrays
        b.pf[p].cand_off = 0;
        b.pf[p].cand_cap = 0;
        if (!compact && b.pairs[p].valid && b.pf[p].use) {
            b.pf[p].cand_off = dense_cand;
            b.pf[p].cand_cap = 16 * (b.pairs[p].n1 + b.pairs[p].n2) + 2048;
            dense_cand += b.pf[p].cand_cap;
            SC.pf_pending.dense_swept += (long long)b.pairs[p].n1pad * b.pairs[p].n2;
        }
    }
    for (size_t p = 0; p < P && q8; ++p) {
        pfq[p].tu_off = b.pf[p].tu_off;
        pfq[p].tv_off = b.pf[p].tv_off;
    }
    if (q8_mixed) {   // the table of the fp16 first sweep and its thresholds: the pairs WITHOUT twins (after the offsets above are final)
        pf16 = b.pf;
        for (size_t p = 0; p < P; ++p)
            if (twin[p]) pf16[p].use = 0;
        // The column partials share one buffer: the fp16 sweep stores 8-byte entries at ELEMENT cp_off, the integer sweep 4-byte entries
        // at the same element numbers -- in a homogeneous sub-batch either is consistent, mixed they would overlap.  The twins' pairs
        // count their offset in 4-byte entries of their own 8-byte region.  (Only the integer sweep and the prune kernel read it; the
        // batch is rebuilt before a re-run.)
        for (size_t p = 0; p < P; ++p)
            if (twin[p]) b.pairs[p].cp_off *= 2;
    }
    std::vector<int> item_base16;
    size_t per16 = 0;
    build_items(b, 1);
    if (b.n_items == 0) return MSFM_OK;
    if (q8_mixed) {   // the fp16 first sweep's own item list: the pairs without twins
        std::vector<char> only(P, 0);
        for (size_t p = 0; p < P; ++p) only[p] = (b.pairs[p].valid && b.pf[p].use && !twin[p]) ? 1 : 0;
        build_items(b, 1, &only, &item_base16, &per16);
    }
    const long long kn = std::max<long long>(1, b.kf_elems + b.kr_elems);
    HIPCHK(ctx, SC.d_rp_s0.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, SC.d_rp_s1.ensure(std::max<long long>(1, b.rp_elems) * 4));
    // column partials of sweep 1: one float2 (the two largest of four row-class maxima) per 512-row A block and column
    HIPCHK(ctx, SC.d_cp_s0.ensure(std::max<long long>(1, b.cp_elems) * 8));
    HIPCHK(ctx, SC.d_tu.ensure(kn * 4));
    HIPCHK(ctx, SC.d_colmask.ensure(kn * 4));
    HIPCHK(ctx, SC.d_best.ensure(kn * 8));
    HIPCHK(ctx, SC.d_second.ensure(kn * 8));
    HIPCHK(ctx, SC.d_overflow.ensure(P));
    // [0..1] candidate / overflow totals (64-bit), ints [8..15]: per-XCD item cursors of sweep 2, ints [16..23]: per-XCD span cursors
    // of the exact re-check
    HIPCHK(ctx, SC.d_totals.ensure(128));
    FillBatch fills;
    if (!compact) {   // (compacted sweep 2: pf_assign_kernel initialises the live slots only)
        fills.add(SC.d_best.p, (size_t)kn * 8, 0xff);
        fills.add(SC.d_second.p, (size_t)kn * 8, 0xff);
    }
    fills.add(SC.d_totals.p, 128, 0);
    fills.add(SC.d_overflow.p, P, 0);   // (which pairs own an overflowed list: pf_overflow_kernel at the end of the chain)
    int rc = upload_pair_tables(ctx, b, 1, q8 ? &pfq : nullptr, fills, q8_mixed ? &pf16 : nullptr, q8_mixed ? &item_base16 : nullptr, per16);
    if (rc != MSFM_OK) return rc;

    hipEvent_t e0 = get_event(ctx, ev_base), e1 = get_event(ctx, ev_base + 1);
    hipEvent_t e2 = get_event(ctx, ev_base + 2), e3 = get_event(ctx, ev_base + 3);
    if (!e0 || !e1 || !e2 || !e3) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
    const dim3 block(kPfThreads);
    const unsigned sweep_grid = (unsigned)std::max(8, (ctx->cu_count / 8) * 8);   // one persistent workgroup per CU, a multiple of the 8 XCDs
    const PairDesc* dp = SC.d_pairs.as<PairDesc>();
    const PfPair* dpf = SC.d_pf.as<PfPair>();
    float* tuv = SC.d_tu.as<float>();  // rows at kf offsets, columns at kr offsets (one buffer)
    unsigned* colmask = SC.d_colmask.as<unsigned>();
    hc.lap("sweep-1 setup + uploads");
    // Sweeps 1 of consecutive sub-batches are persistent one-workgroup-per-CU kernels: two of them cannot share the chip, and
    // a launch that merely queues behind the other stream's sweep would be timed (events) with its wait.  So this one
    // starts when the other stream's sweep 1 is done; what DOES overlap with it is that stream's tail.
    if (ctx->last_sweep1 && ctx->last_sweep1 != ctx->cur)
        HIPCHK(ctx, hipStreamWaitEvent(SC.stream, ctx->last_sweep1->sweep1_done, 0));
    // ... and, with three sub-batches in flight, behind sweep 2 of the one before that: the matrix pipes see
    // S1(k+1) S2(k) S1(k+2) S2(k+1) ..., every bandwidth-bound tail runs beside a sweep, and two persistent kernels never split the CUs
    for (Scratch& other : ctx->sc)
        if (&other != ctx->cur && other.sweep2_recorded && other.seq + 2 <= SC.seq)
            HIPCHK(ctx, hipStreamWaitEvent(SC.stream, other.sweep2_done, 0));
    HIPCHK(ctx, hipEventRecord(e0, SC.stream));
    if (i8 || q8)
        hipLaunchKernelGGL(sweep_i8_kernel<1>, dim3(std::min<unsigned>(sweep_grid, (unsigned)b.n_items)), dim3(kI8Threads), kI8LdsBytes, SC.stream, dp,
                           q8 ? (const PfPair*)SC.d_pfq.as<PfPair>() : dpf,
                           SC.d_items.as<WorkItem>(), SC.d_rp_s0.as<float>(), SC.d_rp_s1.as<float>(), SC.d_cp_s0.as<float>(),
                           (const float*)nullptr, (int2*)nullptr, (unsigned long long*)nullptr, (const int*)nullptr, (int)b.n_items, (int*)nullptr,
                           (int*)nullptr);
    else
        hipLaunchKernelGGL(sweep_kernel<1>, dim3(std::min<unsigned>(sweep_grid, (unsigned)b.n_items)), block, kPfLdsBytes, SC.stream, dp, dpf,
                           SC.d_items.as<WorkItem>(), SC.d_rp_s0.as<float>(), SC.d_rp_s1.as<float>(),
                           SC.d_cp_s0.as<float>(), (float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           (int2*)nullptr, (unsigned long long*)nullptr, (const int*)nullptr, (int)b.n_items, (int*)nullptr);
    HIPCHK(ctx, hipGetLastError());
    if (q8_mixed && per16) {   // the pairs without twins: their own item list (the twins' sweep skipped them: not in use in ITS table)
        hipLaunchKernelGGL(sweep_kernel<1>, dim3(std::min<unsigned>(sweep_grid, (unsigned)(per16 * 8))), block, kPfLdsBytes, SC.stream, dp,
                           (const PfPair*)SC.d_pf16.as<PfPair>(), SC.d_items16.as<WorkItem>(), SC.d_rp_s0.as<float>(), SC.d_rp_s1.as<float>(),
                           SC.d_cp_s0.as<float>(), (float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           (int2*)nullptr, (unsigned long long*)nullptr, (const int*)nullptr, (int)(per16 * 8), (int*)nullptr);
        HIPCHK(ctx, hipGetLastError());
        SC.prof.mixed_route_sub_batches += 1;
    }
    DBGSYNC(ctx, "sweep_kernel<1>");
    HIPCHK(ctx, hipEventRecord(e1, SC.stream));
    HIPCHK(ctx, hipEventRecord(SC.sweep1_done, SC.stream));
    SC.sweep1_recorded = true;
    ctx->last_sweep1 = ctx->cur;
    SC.prof.approx_kernel_launches += 1;
    if (i8 || q8) SC.prof.sweep1_i8_launches += 1;
    if (q8) SC.prof.sweep1_q8_launches += 1;
    const dim3 mgrid((unsigned)((b.max_npad + 255) / 256), (unsigned)P);
    if (!compact) {
        hipLaunchKernelGGL(pf_thresholds_kernel, mgrid, dim3(256), 0, SC.stream, dp, dpf, SC.d_rp_s0.as<float>(),
                           SC.d_rp_s1.as<float>(), SC.d_cp_s0.as<float>(), (unsigned*)nullptr, tuv, tuv, prune, PlanCounts{}, 0);
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "pf_thresholds_kernel");
    }
    hc.lap("launch sweep 1 (+ thresholds)");

    size_t n_lists = 0;
    const CandList* dl = nullptr;
    if (compact) {
        // ---- static plan tables (the GPU is busy with sweep 1 meanwhile), buffers from the prediction ----------
        CompactPlan cp;
        build_compact_plan(ctx, b, cp, q8_refine);
        const size_t G = cp.groups.size(), M = cp.member_pair.size();
        n_lists = G;
        long long max_ranges = 1;
        for (const PlanGroup& g : cp.groups) max_ranges = std::max<long long>(max_ranges, g.ranges);
        const long long slack = (long long)kPfWgRows * (long long)G + kPfWgRows;
        // (route Q: plan A holds every live column once per 512-row block group of the other image)
        const long long rows_cap = std::max<long long>(ctx->cmp_rows_hint + ctx->cmp_rows_hint / 2, (q8_refine ? cp.rows_ub_all_bits / 8 : cp.rows_ub / 4)) + slack;
        // (per group: 8 entries per compacted row rounded up to 1024, + 1024 -- plan_group_cap_units)
        const long long cand_cap = std::max<long long>(8 * rows_cap + 2048LL * (long long)G, ctx->cand_hint);
        // (the list holds 8 x the longest per-XCD sub-list: twice the balanced size leaves room for skew)
        const long long items_cap = std::max<long long>(2 * ((rows_cap / kPfWgRows + (long long)G) * max_ranges + 64) / 8 * 8, (ctx->items_hint + 64) / 8 * 8);
        HIPCHK(ctx, SC.d_gtot.ensure(std::max<size_t>(1, G) * 4));
        HIPCHK(ctx, SC.d_grow0.ensure((4 * std::max<size_t>(1, G) + 8) * 8));   // grow0 | gpos[3] | fwd_items_x[8]
        HIPCHK(ctx, SC.d_cnt.ensure(std::max<size_t>(1, M) * 4));
        HIPCHK(ctx, SC.d_mrow.ensure(std::max<size_t>(1, M) * 8));
        HIPCHK(ctx, SC.d_summary.ensure(sizeof(PlanSummary)));
        HIPCHK(ctx, SC.d_vpairs.ensure(std::max<size_t>(1, G) * sizeof(PairDesc)));
        HIPCHK(ctx, SC.d_vpf.ensure(std::max<size_t>(1, G) * sizeof(PfPair)));
        HIPCHK(ctx, SC.d_lists.ensure(std::max<size_t>(1, G) * sizeof(CandList)));
        HIPCHK(ctx, SC.d_vitems.ensure((size_t)items_cap * sizeof(WorkItem)));
        HIPCHK(ctx, SC.d_cmp_tu.ensure((size_t)rows_cap * 4));
        HIPCHK(ctx, SC.d_live_idx.ensure((size_t)rows_cap * 4));
        HIPCHK(ctx, SC.d_row_pair.ensure((size_t)rows_cap * 4));
        HIPCHK(ctx, SC.d_row_src.ensure((size_t)rows_cap * 8));
        HIPCHK(ctx, SC.d_cand.ensure((size_t)cand_cap * sizeof(int2)));
        if (i8) {
            HIPCHK(ctx, SC.d_cand_val.ensure((size_t)cand_cap * 4));
            HIPCHK(ctx, SC.d_cmp_n2.ensure((size_t)rows_cap * 4));
        }
        HIPCHK(ctx, SC.d_cand_count.ensure(std::max<size_t>(1, G) * 8));
        {   // the plan's static tables in one copy (page-locked staging of this scratch set), everything it clears in one launch
            UploadPlan up;
            up.add(SC.d_groups, cp.groups.data(), G * sizeof(PlanGroup));
            up.add(SC.d_gmembers, cp.gmembers.data(), M * 4);
            up.add(SC.d_member_pair, cp.member_pair.data(), M * 4);
            up.add(SC.d_member_group, cp.member_group.data(), M * 4);
            up.add(SC.d_ppair, cp.ppair.data(), P * sizeof(PlanPair));
            HIPCHK(ctx, up.place_and_copy(SC.d_up[1], SC.h_up[1], SC.stream));
            FillBatch fb;
            fb.add(SC.d_gtot.p, std::max<size_t>(1, G) * 4, 0);
            fb.add(SC.d_cnt.p, std::max<size_t>(1, M) * 4, 0);
            // (pf_plan_write_kernel stores only the non-zero fields of the groups' descriptors)
            fb.add(SC.d_vpairs.p, std::max<size_t>(1, G) * sizeof(PairDesc), 0);
            fb.add(SC.d_vpf.p, std::max<size_t>(1, G) * sizeof(PfPair), 0);
            fb.add(SC.d_lists.p, std::max<size_t>(1, G) * sizeof(CandList), 0);
            fb.add(SC.d_summary.p, sizeof(PlanSummary), 0);
            fb.add(SC.d_vitems.p, (size_t)items_cap * sizeof(WorkItem), 0xff);
            fb.add(SC.d_row_src.p, (size_t)rows_cap * 8, 0);
            fb.add(SC.d_cand_count.p, std::max<size_t>(1, G) * 8, 0);
            HIPCHK(ctx, fb.launch(SC.stream));
        }
        hc.lap("plan tables + uploads");
        const PlanPair* dpp = SC.d_ppair.as<PlanPair>();
        PlanCounts pc = {dpp, (const int*)SC.d_member_group.as<int>(), SC.d_cnt.as<int>(), SC.d_gtot.as<int>()};
        HIPCHK(ctx, SC.d_summary_a.ensure(sizeof(PlanSummary)));
        PlanOut po = {};
        po.vpairs = SC.d_vpairs.as<PairDesc>();
        po.vpf = SC.d_vpf.as<PfPair>();
        po.lists = SC.d_lists.as<CandList>();
        po.items = SC.d_vitems.as<WorkItem>();
        po.grow0 = SC.d_grow0.as<long long>();
        po.gpos = SC.d_grow0.as<long long>() + std::max<size_t>(1, G);
        po.fwd_items_x = SC.d_grow0.as<long long>() + 4 * std::max<size_t>(1, G);
        po.summary = SC.d_summary.as<PlanSummary>();
        po.row_src = SC.d_row_src.as<const _Float16*>();
        po.zero_row = ctx->d_zero_row.as<_Float16>();
        po.live_idx = SC.d_live_idx.as<int>();
        po.row_pair = SC.d_row_pair.as<int>();
        po.rows_cap = rows_cap;
        po.cand_cap = cand_cap;
        po.items_cap = items_cap;
        po.cmp_tu = SC.d_cmp_tu.as<float>();
        po.cmp_n2 = i8 ? SC.d_cmp_n2.as<int>() : nullptr;
        // the plan from the live counts in d_cnt / d_gtot: scan, descriptors + work items, member rows, slot assignment
        auto launch_plan = [&](PlanSummary* summary, int norms_only) -> int {
            po.summary = summary;
            hipLaunchKernelGGL(pf_plan_scan_kernel, dim3(1), dim3(64), 0, SC.stream, SC.d_groups.as<PlanGroup>(), (int)G,
                               (const int*)SC.d_gtot.as<int>(), po);
            HIPCHK(ctx, hipGetLastError());
            DBGSYNC(ctx, "pf_plan_scan_kernel");
            if (G > 0)
                hipLaunchKernelGGL(pf_plan_write_kernel, dim3((unsigned)((G + 63) / 64)), dim3(64), 0, SC.stream, SC.d_groups.as<PlanGroup>(), (int)G,
                                   (const int*)SC.d_gtot.as<int>(), po);
            HIPCHK(ctx, hipGetLastError());
            DBGSYNC(ctx, "pf_plan_write_kernel");
            if (G > 0)
                hipLaunchKernelGGL(pf_member_rows_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, SC.stream, SC.d_groups.as<PlanGroup>(), (int)G,
                                   (const int*)SC.d_gmembers.as<int>(), (const int*)SC.d_cnt.as<int>(),
                                   (const long long*)SC.d_grow0.as<long long>(), SC.d_mrow.as<long long>());
            HIPCHK(ctx, hipGetLastError());
            DBGSYNC(ctx, "pf_member_rows_kernel");
            hipLaunchKernelGGL(pf_assign_kernel, dim3((unsigned)(2 * P), (unsigned)((b.max_npad + kAssignChunk - 1) / kAssignChunk)), dim3(256), 0,
                               SC.stream, dp, dpf, dpp, (const float*)tuv,
                               (const unsigned*)colmask, (const long long*)SC.d_mrow.as<long long>(), SC.d_cnt.as<int>(),
                               SC.d_live_idx.as<int>(), SC.d_row_pair.as<int>(), SC.d_cmp_tu.as<float>(),
                               SC.d_row_src.as<const _Float16*>(), i8 ? kI8RowBytes / 2 : kPfRowHalfs,
                               SC.d_best.as<unsigned long long>(), SC.d_second.as<unsigned long long>(), norms_only,
                               i8 ? SC.d_cmp_n2.as<int>() : (int*)nullptr);
            HIPCHK(ctx, hipGetLastError());
            DBGSYNC(ctx, "pf_assign_kernel");
            return MSFM_OK;
        };
        if (q8) {
            // ---- route Q: live / dead (fine twins: and the thresholds, the block masks, the counts of the plan) from the twins' sweep
            hipLaunchKernelGGL(pf_prune_q8_kernel, mgrid, dim3(256), 0, SC.stream, dp, dpf, (const PfPair*)SC.d_pfq.as<PfPair>(),
                               (const float*)SC.d_rp_s0.as<float>(), (const float*)SC.d_rp_s1.as<float>(), (const float*)SC.d_cp_s0.as<float>(),
                               colmask, tuv, prune, pc, ctx->q8_level / 255.f, q8_direct ? 1 : 0);
            HIPCHK(ctx, hipGetLastError());
            DBGSYNC(ctx, "pf_prune_q8_kernel");
        }
        if (q8_refine) {
            // ---- coarse twins: plan A, fp16 sweep 1' on the live rows, scatter ------------------------------------------------
            HIPCHK(ctx, hipMemsetAsync(SC.d_summary_a.p, 0, sizeof(PlanSummary), SC.stream));
            rc = launch_plan(SC.d_summary_a.as<PlanSummary>(), 1);
            if (rc != MSFM_OK) return rc;
            HIPCHK(ctx, SC.d_cmp_s0.ensure((size_t)rows_cap * 4));
            HIPCHK(ctx, SC.d_cmp_s1.ensure((size_t)rows_cap * 4));
            hipEvent_t e4 = get_event(ctx, ev_base + 6), e5 = get_event(ctx, ev_base + 7);
            if (!e4 || !e5) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
            HIPCHK(ctx, hipEventRecord(e4, SC.stream));
            hipLaunchKernelGGL(sweep_kernel<4>, dim3(sweep_grid), block, kPfLdsBytes, SC.stream,
                               (const PairDesc*)SC.d_vpairs.as<PairDesc>(), (const PfPair*)SC.d_vpf.as<PfPair>(),
                               (const WorkItem*)SC.d_vitems.as<WorkItem>(), SC.d_cmp_s0.as<float>(), SC.d_cmp_s1.as<float>(), (float*)nullptr,
                               (float*)nullptr, (const float*)SC.d_cmp_tu.as<float>(), (const float*)nullptr, (int2*)nullptr,
                               (unsigned long long*)nullptr, (const int*)&SC.d_summary_a.as<PlanSummary>()->n_items, 0,
                               SC.d_totals.as<int>() + 8);
            HIPCHK(ctx, hipGetLastError());
            DBGSYNC(ctx, "sweep_kernel<4>");
            HIPCHK(ctx, hipEventRecord(e5, SC.stream));
            SC.prof.sweep1b_launches += 1;
            if (G > 0)
                hipLaunchKernelGGL(q8_scatter_kernel, dim3(16, (unsigned)std::min<size_t>(G, 65535)), dim3(256), 0, SC.stream, dp,
                                   (const PairDesc*)SC.d_vpairs.as<PairDesc>(), (const CandList*)SC.d_lists.as<CandList>(),
                                   (const PlanGroup*)SC.d_groups.as<PlanGroup>(), (int)G, (const long long*)SC.d_grow0.as<long long>(),
                                   (const float*)SC.d_cmp_s0.as<float>(), (const float*)SC.d_cmp_s1.as<float>(),
                                   SC.d_rp_s0.as<float>(), SC.d_rp_s1.as<float>(), SC.d_cp_s0.as<float>());
            HIPCHK(ctx, hipGetLastError());
            DBGSYNC(ctx, "q8_scatter_kernel");
            // plan B starts from clean counters, work items, row sources and item cursors
            FillBatch fb;
            fb.add(SC.d_gtot.p, std::max<size_t>(1, G) * 4, 0);
            fb.add(SC.d_cnt.p, std::max<size_t>(1, M) * 4, 0);
            fb.add(SC.d_vpairs.p, std::max<size_t>(1, G) * sizeof(PairDesc), 0);
            fb.add(SC.d_vpf.p, std::max<size_t>(1, G) * sizeof(PfPair), 0);
            fb.add(SC.d_lists.p, std::max<size_t>(1, G) * sizeof(CandList), 0);
            fb.add(SC.d_vitems.p, (size_t)items_cap * sizeof(WorkItem), 0xff);
            fb.add(SC.d_row_src.p, (size_t)rows_cap * 8, 0);
            fb.add(SC.d_totals.p, 64, 0);
            HIPCHK(ctx, fb.launch(SC.stream));
        }
        // thresholds + live counts per member / group (the plan tables above are uploaded by now; sweep 1 is still running)
        if (!q8_direct)
            hipLaunchKernelGGL(pf_thresholds_kernel, mgrid, dim3(256), 0, SC.stream, dp, dpf, SC.d_rp_s0.as<float>(),
                               SC.d_rp_s1.as<float>(), SC.d_cp_s0.as<float>(), colmask, tuv, tuv, prune, pc, q8 ? 1 : 0);
        else if (q8_mixed)   // (the pairs of the fp16 sweep 1: thresholds and plan counts the usual way; the prune kernel did the twins')
            hipLaunchKernelGGL(pf_thresholds_kernel, mgrid, dim3(256), 0, SC.stream, dp, (const PfPair*)SC.d_pf16.as<PfPair>(), SC.d_rp_s0.as<float>(),
                               SC.d_rp_s1.as<float>(), SC.d_cp_s0.as<float>(), colmask, tuv, tuv, prune, pc, 0);
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "pf_thresholds_kernel");
        rc = launch_plan(SC.d_summary.as<PlanSummary>(), 0);
        if (rc != MSFM_OK) return rc;
        HIPCHK(ctx, hipEventRecord(e2, SC.stream));
        if (i8)
            hipLaunchKernelGGL(sweep_i8_kernel<3>, dim3(sweep_grid), dim3(kI8Threads), kI8LdsBytes3, SC.stream,
                               (const PairDesc*)SC.d_vpairs.as<PairDesc>(), (const PfPair*)SC.d_vpf.as<PfPair>(),
                               (const WorkItem*)SC.d_vitems.as<WorkItem>(), (float*)nullptr, (float*)nullptr, (float*)nullptr,
                               (const float*)SC.d_cmp_tu.as<float>(), SC.d_cand.as<int2>(),
                               SC.d_cand_count.as<unsigned long long>(), (const int*)&SC.d_summary.as<PlanSummary>()->n_items, 0,
                               SC.d_totals.as<int>() + 8, SC.d_cand_val.as<int>());
        else
            hipLaunchKernelGGL(sweep_kernel<3>, dim3(sweep_grid), block, kPfLdsBytes, SC.stream,
                               (const PairDesc*)SC.d_vpairs.as<PairDesc>(), (const PfPair*)SC.d_vpf.as<PfPair>(),
                               (const WorkItem*)SC.d_vitems.as<WorkItem>(), (float*)nullptr, (float*)nullptr, (float*)nullptr,
                               (float*)nullptr, (const float*)SC.d_cmp_tu.as<float>(), (const float*)nullptr, SC.d_cand.as<int2>(),
                               SC.d_cand_count.as<unsigned long long>(), (const int*)&SC.d_summary.as<PlanSummary>()->n_items, 0,
                               SC.d_totals.as<int>() + 8);
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "sweep_kernel<3>");
        SC.prof.sweep2_launches += 1;
        HIPCHK(ctx, hipEventRecord(e3, SC.stream));
        HIPCHK(ctx, hipEventRecord(SC.sweep2_done, SC.stream));
        SC.sweep2_recorded = true;
        dl = SC.d_lists.as<CandList>();
        SC.pf_pending.compact = true;
        SC.pf_pending.rows_cap = rows_cap;
        SC.pf_pending.cand_cap = cand_cap;
        SC.pf_pending.items_cap = items_cap;
        SC.pf_pending.compact_pairs = cp.pairs;
    } else {
        // ---- dense sweep 2: the pairs' own lists, the sweep-1 items again ----------------------------------------
        n_lists = P;
        std::vector<CandList>& lists = b.dense_lists;   // (lives as long as the sub-batch: the copy below may still be in flight)
        lists.resize(P);
        for (size_t p = 0; p < P; ++p)
            lists[p] = CandList{(int)p, 0, b.pf[p].cand_off, b.pf[p].cand_cap, 0, nullptr, nullptr};
        HIPCHK(ctx, SC.d_cand.ensure(std::max<long long>(1, dense_cand) * sizeof(int2)));
        HIPCHK(ctx, SC.d_cand_count.ensure(P * 8));
        HIPCHK(ctx, SC.d_lists.ensure(P * sizeof(CandList)));
        HIPCHK(ctx, hipMemsetAsync(SC.d_cand_count.p, 0, P * 8, SC.stream));
        HIPCHK(ctx, hipMemcpyAsync(SC.d_lists.p, lists.data(), P * sizeof(CandList), hipMemcpyHostToDevice, SC.stream));
        HIPCHK(ctx, hipEventRecord(e2, SC.stream));
        hipLaunchKernelGGL(sweep_kernel<2>, dim3(std::min<unsigned>(sweep_grid, (unsigned)b.n_items)), block, kPfLdsBytes, SC.stream, dp, dpf,
                           SC.d_items.as<WorkItem>(), (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (const float*)tuv, (const float*)tuv, SC.d_cand.as<int2>(), SC.d_cand_count.as<unsigned long long>(),
                           (const int*)nullptr, (int)b.n_items, (int*)nullptr);
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "sweep_kernel<2>");
        SC.prof.sweep2_launches += 1;
        HIPCHK(ctx, hipEventRecord(e3, SC.stream));
        HIPCHK(ctx, hipEventRecord(SC.sweep2_done, SC.stream));
        SC.sweep2_recorded = true;
        dl = SC.d_lists.as<CandList>();
    }

    if (n_lists > 0) {
        // list l on XCD l mod 8, its spans of 256 candidates handed out by a per-XCD cursor to that XCD's persistent workgroups
        // (see the kernel): 8 workgroups of 4 waves per CU when it has the chip to itself, one per CU fits next to a sweep workgroup
        const int wgs_per_xcd = std::max(1, ctx->cu_count / 8) * 8;
        const dim3 cgrid((unsigned)(8 * wgs_per_xcd));
        const unsigned long long* dcount = SC.d_cand_count.as<unsigned long long>();
#define MSFM_LAUNCH_EXACT(O)                                                                                             \
    hipLaunchKernelGGL(pf_exact_candidates_kernel<O>, cgrid, dim3(kExSpan), 0, SC.stream, dp, dl, dcount,                  \
                       (const int2*)SC.d_cand.as<int2>(), SC.d_best.as<unsigned long long>(), SC.d_second.as<unsigned long long>(), \
                       (int)n_lists, SC.d_totals.as<int>() + 16, (const int*)SC.d_cand_val.as<int>(), 0)
        // byte pairs on the integer route: the sweep handed over exact integers -- no rows are read, the named order does not matter
        if (i8 && compact) MSFM_LAUNCH_EXACT(4);
        else if (ctx->order == MSFM_ORDER_SSE4X4) MSFM_LAUNCH_EXACT(0);
        else if (ctx->order == MSFM_ORDER_AVX2_FMA) MSFM_LAUNCH_EXACT(1);
        else MSFM_LAUNCH_EXACT(3);
#undef MSFM_LAUNCH_EXACT
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "pf_exact_candidates_kernel");
    }
    if (!SC.keys_epilogue) {
        hipLaunchKernelGGL(pf_finalize_kernel, mgrid, dim3(256), 0, SC.stream, dp, dpf, (const float*)tuv, SC.d_best.as<unsigned long long>(),
                           SC.d_second.as<unsigned long long>(), SC.d_k_i0.as<int>(), SC.d_k_d0.as<float>(),
                           SC.d_k_d1.as<float>(), SC.d_fix_count.as<int>(), SC.d_fix_list.as<int4>(), SC.fix_cap_eff);
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "pf_finalize_kernel");
    }
    // which pairs own an overflowed list, how many candidates were evaluated: read at the end of the batch
    if (n_lists > 0) {
        hipLaunchKernelGGL(pf_overflow_kernel, dim3((unsigned)((n_lists + 255) / 256)), dim3(256), 0, SC.stream, dl, (int)n_lists,
                           (const unsigned long long*)SC.d_cand_count.as<unsigned long long>(), (const PlanGroup*)SC.d_groups.as<PlanGroup>(),
                           (const int*)SC.d_gmembers.as<int>(), (const int*)SC.d_member_pair.as<int>(),
                           SC.d_overflow.as<unsigned char>(), SC.d_totals.as<unsigned long long>());
        HIPCHK(ctx, hipGetLastError());
    }
    // (summary / totals / overflow bytes travel to the host with the other end-of-batch words: queue_tail_copies)
    SC.pf_pending.active = true;
    SC.pf_pending.n_lists = n_lists;
    SC.pf_pending.P = P;
    SC.pf_pending.ev_base = ev_base;
    hc.lap("launch sweep 2 .. finalize");
    return MSFM_OK;
}

// After the batch's stream synchronisation: did the prefilter path complete?  *retry: run the batch again (buffers
// grown / overflowed pairs moved to the brute-force path in `force_exact`).
int finish_prefilter(msfm_ctx* ctx, Batch& b, std::vector<char>& force_exact, bool* retry) {
    *retry = false;
    PfPending& pe = SC.pf_pending;
    if (!pe.active) return MSFM_OK;
    pe.active = false;
    const char* hs = SC.h_summary.as<char>();
    PlanSummary sm = {};
    unsigned long long totals[2] = {0, 0};
    std::memcpy(totals, hs + kHsTotals, 16);
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[pe.ev_base], ctx->ev_pool[pe.ev_base + 1]));
    SC.prof.approx_kernel_ms += ms;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[pe.ev_base + 2], ctx->ev_pool[pe.ev_base + 3]));
    SC.prof.sweep2_ms += ms;
    if (std::getenv("MSFM_DEBUG_TIMING") && pe.compact) {
        PlanSummary d;
        std::memcpy(&d, hs, sizeof(PlanSummary));
        std::fprintf(stderr, "[msfm plan] ok %d, items %d, compacted rows %lld, candidate capacity %lld, swept descriptor pairs %lld; sweep 1 %.3f ms, sweep 2 %.3f ms\n",
                     d.ok, d.n_items, d.cmp_rows, d.cand_elems, d.swept_desc_pairs, SC.prof.approx_kernel_ms, ms);
    }
    if (pe.compact) {
        std::memcpy(&sm, hs, sizeof(PlanSummary));
        ctx->cmp_rows_hint = sm.cmp_rows;
        ctx->items_hint = sm.items_needed;
        ctx->cand_hint = sm.cand_elems;
        if (pe.q8) {   // plan A (every live column in every block group) is the larger one
            PlanSummary sa;
            std::memcpy(&sa, hs + sizeof(PlanSummary), sizeof(PlanSummary));
            ctx->cmp_rows_hint = std::max(ctx->cmp_rows_hint, sa.cmp_rows);
            ctx->items_hint = std::max(ctx->items_hint, sa.items_needed);
            ctx->cand_hint = std::max(ctx->cand_hint, sa.cand_elems);
            if (!sa.ok) sm.ok = 0;
            float ms1b = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&ms1b, ctx->ev_pool[pe.ev_base + 6], ctx->ev_pool[pe.ev_base + 7]));
            SC.prof.sweep1b_ms += ms1b;
            SC.prof.sweep1b_descriptor_pairs += sa.swept_desc_pairs;
        }
        if (!sm.ok) {   // the prediction was too small: the buffers are sized from the need now
            SC.prof.plan_regrows += 1;
            *retry = true;
            return MSFM_OK;
        }
        SC.prof.sweep2_descriptor_pairs += sm.swept_desc_pairs;
    } else {
        SC.prof.sweep2_descriptor_pairs += pe.dense_swept;
    }
    const unsigned char* ov = reinterpret_cast<const unsigned char*>(hs + kHsOverflow);
    int n_over = 0;
    for (size_t p = 0; p < pe.P; ++p) {
        if (!b.pairs[p].valid || !b.pf[p].use) continue;
        if (ov[p]) {
            force_exact[p] = 1;
            ++n_over;
        }
    }
    if (n_over > 0) {   // candidate-list overflow -> those pairs take the brute-force exact path in a second run
        SC.prof.fallback_pairs += n_over;
        *retry = true;
        return MSFM_OK;
    }
    SC.prof.candidates += (int64_t)totals[0];
    for (size_t p = 0; p < pe.P; ++p)
        if (b.pairs[p].valid && b.pf[p].use) {
            SC.prof.prefilter_pairs += 1;
            SC.prof.prefilter_descriptor_pairs += (int64_t)b.pairs[p].n1 * b.pairs[p].n2;
        }
    if (pe.compact) SC.prof.compacted_pairs += pe.compact_pairs;
    return MSFM_OK;
}

// brute-force exact distance kernel + merge for the pairs on path 0
int run_exact(msfm_ctx* ctx, Batch& b, size_t ev_base) {
    const size_t P = b.pairs.size();
    assign_partials(b, 0, 4 * ctx->cu_count);
    build_items(b, 0);
    if (b.n_items == 0) return MSFM_OK;
    HIPCHK(ctx, SC.d_rp_s0.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, SC.d_rp_i0.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, SC.d_rp_s1.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, SC.d_cp_s0.ensure(std::max<long long>(1, b.cp_elems) * 4));
    HIPCHK(ctx, SC.d_cp_i0.ensure(std::max<long long>(1, b.cp_elems) * 4));
    HIPCHK(ctx, SC.d_cp_s1.ensure(std::max<long long>(1, b.cp_elems) * 4));
    FillBatch fills;
    int rc = upload_pair_tables(ctx, b, 0, nullptr, fills);
    if (rc != MSFM_OK) return rc;
    hipEvent_t e0 = get_event(ctx, ev_base), e1 = get_event(ctx, ev_base + 1);
    if (!e0 || !e1) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
    // (like sweep 1: the brute-force kernels of two sub-batches in flight take turns, so that the event span is the kernel's)
    if (ctx->last_sweep1 && ctx->last_sweep1 != ctx->cur)
        HIPCHK(ctx, hipStreamWaitEvent(SC.stream, ctx->last_sweep1->sweep1_done, 0));
    HIPCHK(ctx, hipEventRecord(e0, SC.stream));
    const dim3 grid((unsigned)b.n_items), block(kThreads);
#define MSFM_LAUNCH_DIST(O)                                                                                               \
    hipLaunchKernelGGL(dist_top2_kernel<O>, grid, block, (O) == 3 ? kLdsBytesIdxStash : kLdsBytes, SC.stream, SC.d_pairs.as<PairDesc>(), SC.d_items.as<WorkItem>(), \
                       SC.d_rp_s0.as<float>(), SC.d_rp_i0.as<int>(), SC.d_rp_s1.as<float>(), SC.d_cp_s0.as<float>(),             \
                       SC.d_cp_i0.as<int>(), SC.d_cp_s1.as<float>())
    if (ctx->order == MSFM_ORDER_SSE4X4) MSFM_LAUNCH_DIST(0);
    else if (ctx->order == MSFM_ORDER_AVX2_FMA) MSFM_LAUNCH_DIST(1);
    else MSFM_LAUNCH_DIST(3);
#undef MSFM_LAUNCH_DIST
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(e1, SC.stream));
    HIPCHK(ctx, hipEventRecord(SC.sweep1_done, SC.stream));
    SC.sweep1_recorded = true;
    ctx->last_sweep1 = ctx->cur;
    SC.prof.dist_kernel_launches += 1;
    for (auto& pd : b.pairs)
        if (pd.valid && pd.path == 0) SC.prof.exact_descriptor_pairs += (int64_t)pd.n1 * pd.n2;
    const dim3 mgrid((unsigned)((b.max_npad + 255) / 256), (unsigned)P);
    hipLaunchKernelGGL(merge_knn_kernel, mgrid, dim3(256), 0, SC.stream, SC.d_pairs.as<PairDesc>(),
                       SC.d_rp_s0.as<float>(), SC.d_rp_i0.as<int>(), SC.d_rp_s1.as<float>(),
                       SC.d_cp_s0.as<float>(), SC.d_cp_i0.as<int>(), SC.d_cp_s1.as<float>(),
                       SC.d_k_i0.as<int>(), SC.d_k_d0.as<float>(), SC.d_k_d1.as<float>(),
                       SC.d_fix_count.as<int>(), SC.d_fix_list.as<int4>(), SC.fix_cap_eff);
    HIPCHK(ctx, hipGetLastError());
    return MSFM_OK;
}

// kNN-2 of both directions for every pair of the batch (device arrays left in the ctx buffers):
// prefilter path where eligible, brute-force exact path for the rest, then the sqrt-space tie fix-up
//   need_fix: the caller can observe WHICH index a sqrt-space tie resolves to (knnMatch-level API, ratio > 1).
//   For match lists with ratio <= 1 a row with d0 == d1 fails `d0 < ratio * d1` in both directions, so its
//   index never reaches a list: the queue is not filled and nothing is re-scanned.
int run_knn(msfm_ctx* ctx, Batch& b, size_t ev_base, bool* exact_launched, PruneParams prune, bool need_fix, bool lists_only) {
    assign_common(b);
    SC.fix_cap_eff = need_fix ? ctx->fix_cap : 0;
    const long long kn = std::max<long long>(1, b.kf_elems + b.kr_elems);
    HIPCHK(ctx, SC.d_k_i0.ensure(kn * 4));
    HIPCHK(ctx, SC.d_k_d0.ensure(kn * 4));
    HIPCHK(ctx, SC.d_k_d1.ensure(kn * 4));
    HIPCHK(ctx, SC.d_fix_count.ensure(4));
    HIPCHK(ctx, SC.d_fix_list.ensure((size_t)ctx->fix_cap * sizeof(int4)));
    HIPCHK(ctx, hipMemsetAsync(SC.d_fix_count.p, 0, 4, SC.stream));
    bool any_pf = false, any_exact = false;
    for (auto& pd : b.pairs) any_pf |= (pd.valid && pd.path == 1);
    for (auto& pd : b.pairs) any_exact |= (pd.valid && pd.path == 0);
    // match lists of a batch that is on the matrix-core route throughout, no sqrt-space tie queue: the epilogue reads best / second
    // keys directly (KnnFromKeys); the knnMatch-level API and mixed batches keep the kNN arrays
    SC.keys_epilogue = lists_only && any_pf && !any_exact && SC.fix_cap_eff == 0;
    int rc;
    if (any_pf) {
        rc = run_prefilter(ctx, b, ev_base + 2, prune);  // events ev_base+2 .. ev_base+5
        if (rc != MSFM_OK) return rc;
    }
    *exact_launched = false;
    if (any_exact) {
        rc = run_exact(ctx, b, ev_base);
        if (rc != MSFM_OK) return rc;
        *exact_launched = b.n_items != 0;
    } else if (!any_pf) {
        b.item_base.assign(b.pairs.size(), -1);
        b.n_items = 0;
        FillBatch fills;
        rc = upload_pair_tables(ctx, b, 0, nullptr, fills);  // later kernels still read the (all-invalid) pair table
        if (rc != MSFM_OK) return rc;
    }
    // (no queue -- match lists with ratio <= 1 -- no fix-up launch: the kernel's 86 registers would not fit next to the other
    // stream's sweep and the tail would wait for that sweep's end)
    if ((any_pf || any_exact) && SC.fix_cap_eff > 0) {
#define MSFM_LAUNCH_FIX(O)                                                                                          \
    hipLaunchKernelGGL(tie_fixup_kernel<O>, dim3(256), dim3(64), 0, SC.stream, SC.d_pairs.as<PairDesc>(),               \
                       SC.d_fix_count.as<int>(), SC.d_fix_list.as<int4>(), SC.fix_cap_eff, SC.d_k_i0.as<int>(), SC.d_k_d0.as<float>())
        if (ctx->order == MSFM_ORDER_SSE4X4) MSFM_LAUNCH_FIX(0);
        else if (ctx->order == MSFM_ORDER_AVX2_FMA) MSFM_LAUNCH_FIX(1);
        else MSFM_LAUNCH_FIX(3);
#undef MSFM_LAUNCH_FIX
        HIPCHK(ctx, hipGetLastError());
    }
    return MSFM_OK;
}

// The words the host reads at the end of a sub-batch -- tie-queue count, CSR offsets, certificate counts, and on the
// prefilter path the plan summary, candidate totals and overflow bytes -- are WRITTEN INTO PAGE-LOCKED HOST MEMORY BY A
// KERNEL.  A copy into pageable memory would block the host until the whole sub-batch has run (and with it the launch
// of the next sub-batch on the other stream); and the runtime's own copy kernels for such small transfers have no wave
// priority: under the other stream's persistent sweep one 16-byte copy was measured at 9.7 ms.
struct ExportSeg {
    const char* src;
    char* dst;
    unsigned bytes;
};
struct ExportSegs {
    ExportSeg s[8];
};

__global__ void export_tail_kernel(ExportSegs segs) {
    MSFM_TAIL_PRIO();
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int k = 0; k < 8; ++k) {
        const ExportSeg e = segs.s[k];
        const unsigned words = e.bytes >> 2;
        for (unsigned i = tid; i < words; i += nt) reinterpret_cast<unsigned*>(e.dst)[i] = reinterpret_cast<const unsigned*>(e.src)[i];
        for (unsigned i = (words << 2) + tid; i < e.bytes; i += nt) e.dst[i] = e.src[i];
    }
    __threadfence_system();
}

int queue_tail_copies(msfm_ctx* ctx, size_t P) {
    HIPCHK(ctx, SC.h_tail.ensure(8 + (P + 1) * 8 + P * 4 + 64, 0));
    HIPCHK(ctx, SC.h_summary.ensure(kHsOverflow + P + 64, 0));
    char *h = nullptr, *hs = nullptr;
    HIPCHK(ctx, hipHostGetDevicePointer((void**)&h, SC.h_tail.p, 0));
    HIPCHK(ctx, hipHostGetDevicePointer((void**)&hs, SC.h_summary.p, 0));
    ExportSegs segs = {};
    segs.s[0] = ExportSeg{SC.d_fix_count.as<char>(), h, 4};
    if (SC.d_offsets.p) segs.s[1] = ExportSeg{SC.d_offsets.as<char>(), h + 8, (unsigned)((P + 1) * 8)};
    if (SC.d_sens.p) segs.s[2] = ExportSeg{SC.d_sens.as<char>(), h + 8 + (P + 1) * 8, (unsigned)(P * 4)};
    if (SC.pf_pending.active) {
        if (SC.pf_pending.compact) segs.s[3] = ExportSeg{SC.d_summary.as<char>(), hs, (unsigned)sizeof(PlanSummary)};
        if (SC.pf_pending.q8) segs.s[6] = ExportSeg{SC.d_summary_a.as<char>(), hs + sizeof(PlanSummary), (unsigned)sizeof(PlanSummary)};
        segs.s[4] = ExportSeg{SC.d_totals.as<char>(), hs + kHsTotals, 16};
        segs.s[5] = ExportSeg{SC.d_overflow.as<char>(), hs + kHsOverflow, (unsigned)P};
    }
    hipLaunchKernelGGL(export_tail_kernel, dim3(32), dim3(256), 0, SC.stream, segs);
    HIPCHK(ctx, hipGetLastError());
    return MSFM_OK;
}

// After the sub-batch's stream synchronisation.  *retry = true: more tied rows than the queue holds -- the queue has
// been grown to fit, the caller re-runs the batch (rare: duplicate descriptors on the brute-force path with ratio > 1 or
// through the knnMatch-level API).
int check_fix_overflow(msfm_ctx* ctx, bool* retry) {
    int nfix = 0;
    *retry = false;
    std::memcpy(&nfix, SC.h_tail.as<char>(), 4);
    if (SC.fix_cap_eff > 0 && nfix > SC.fix_cap_eff) {
        ctx->fix_cap = nfix + nfix / 8 + 1024;
        SC.prof.tie_queue_regrows += 1;
        *retry = true;
        return MSFM_OK;
    }
    SC.prof.tie_rows += nfix;
    return MSFM_OK;
}

int accumulate_kernel_time(msfm_ctx* ctx, size_t ev_base, bool launched) {
    if (!launched) return MSFM_OK;
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[ev_base], ctx->ev_pool[ev_base + 1]));
    SC.prof.dist_kernel_ms += ms;
    return MSFM_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char* msfm_version(void) { return "msfm-match 0.1 (gfx950)"; }

int msfm_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return 0;
    int usable = 0;
    for (int d = 0; d < count; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) break;
        ++usable;   // (ordinals are contiguous: the count of leading gfx950 devices)
    }
    return usable;
}

static void destroy_streams(msfm_ctx* ctx);

int msfm_create(int device_ordinal, msfm_ctx** out_ctx) {
    if (!out_ctx) return MSFM_E_INVALID;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MSFM_E_DEVICE;  // no GPU, no fallback
    if (device_ordinal < 0 || device_ordinal >= count) return MSFM_E_INVALID;
    if (hipSetDevice(device_ordinal) != hipSuccess) return MSFM_E_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return MSFM_E_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "msfm_create: device %d is %s, this library is built for gfx950 only\n",
                     device_ordinal, prop.gcnArchName);
        return MSFM_E_DEVICE;
    }
    msfm_ctx* ctx = new (std::nothrow) msfm_ctx();
    if (!ctx) return MSFM_E_DEVICE;
    ctx->device = device_ordinal;
    ctx->images.resize(kSlots);
    ctx->cu_count = prop.multiProcessorCount;
    ctx->clock_mhz = prop.clockRate / 1000;
    std::snprintf(ctx->dev_name, sizeof(ctx->dev_name), "%s (%s)", prop.name, prop.gcnArchName);
    for (Scratch& sc : ctx->sc)
        if (hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sc.sweep1_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sc.sweep2_done, hipEventDisableTiming) != hipSuccess) {
            destroy_streams(ctx);
            delete ctx;
            return MSFM_E_DEVICE;
        }
    // the distance kernel needs 108 KiB of dynamic LDS
    hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(dist_top2_kernel<0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(dist_top2_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e1 == hipSuccess)
        e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(dist_top2_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesIdxStash);
    if (e0 != hipSuccess || e1 != hipSuccess) {
        std::fprintf(stderr, "msfm_create: cannot reserve %d bytes of LDS: %s\n", kLdsBytes,
                     hipGetErrorString(e0 != hipSuccess ? e0 : e1));
        destroy_streams(ctx);
        delete ctx;
        return MSFM_E_DEVICE;
    }
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kPfLdsBytes);
    hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kPfLdsBytes);
    hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kPfLdsBytes);
    if (e4 == hipSuccess)
        e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kPfLdsBytes);
    hipError_t e5 = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_i8_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kI8LdsBytes);
    hipError_t e6 = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_i8_kernel<3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kI8LdsBytes3);
    if (e5 != hipSuccess || e6 != hipSuccess) e2 = e5 != hipSuccess ? e5 : e6;
    if (e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
        std::fprintf(stderr, "msfm_create: cannot reserve %d bytes of LDS for the prefilter kernels\n", kPfLdsBytes);
        destroy_streams(ctx);
        delete ctx;
        return MSFM_E_DEVICE;
    }
    // the all-zero operand row the compacted sweep reads for rows without a source
    if (ctx->d_zero_row.ensure(kPfRowBytes) != hipSuccess || hipMemset(ctx->d_zero_row.p, 0, kPfRowBytes) != hipSuccess) {
        destroy_streams(ctx);
        delete ctx;
        return MSFM_E_DEVICE;
    }
    if (const char* e = std::getenv("MSFM_PREFILTER")) ctx->prefilter = e[0] == '2' ? 2 : (e[0] != '0');
    if (const char* e = std::getenv("MSFM_MAX_PAIRS_PER_BATCH"))
        if (std::atoi(e) > 0) ctx->max_pairs_per_batch = std::min(std::atoi(e), kMaxPairsPerBatchLimit);
    if (const char* e = std::getenv("MSFM_PIPELINE_TAPER")) {
        const double t = std::atof(e);
        if (t >= 0.05 && t <= 1.0) ctx->pipeline_taper = t;
    }
    if (const char* e = std::getenv("MSFM_BYTE_DETECT")) ctx->byte_detect = e[0] != '0';
    if (const char* e = std::getenv("MSFM_Q8_DIRECT")) ctx->q8_direct = e[0] == '2' ? 2 : (e[0] != '0');
    if (const char* e = std::getenv("MSFM_Q8")) ctx->q8_route = e[0] == '2' ? 2 : (e[0] != '0');
    if (const char* e = std::getenv("MSFM_IN_FLIGHT"))
        if (std::atoi(e) >= 1 && std::atoi(e) <= kInFlight) ctx->in_flight = std::atoi(e);
    if (const char* e = std::getenv("MSFM_PIPELINE"))
        if (std::atoi(e) > 0) ctx->pipeline = std::min(std::atoi(e), 64);
    if (const char* e = std::getenv("MSFM_SCRATCH_MIB"))
        if (std::atoll(e) > 0) ctx->scratch_bytes = std::atoll(e) * (1LL << 20);
    *out_ctx = ctx;
    return MSFM_OK;
}

static void destroy_streams(msfm_ctx* ctx) {
    for (Scratch& sc : ctx->sc) {
        if (sc.sweep1_done) (void)hipEventDestroy(sc.sweep1_done);
        if (sc.sweep2_done) (void)hipEventDestroy(sc.sweep2_done);
        sc.sweep2_done = nullptr;
        if (sc.stream) (void)hipStreamDestroy(sc.stream);
        sc.sweep1_done = nullptr;
        sc.stream = nullptr;
    }
}

void msfm_destroy(msfm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (Scratch& sc : ctx->sc)
        if (sc.stream) (void)hipStreamSynchronize(sc.stream);
    for (auto& im : ctx->images) free_image(im);
    for (Scratch& sc : ctx->sc) sc.release_all();
    DevBuf* bufs[] = {&ctx->d_stage, &ctx->d_maxima, &ctx->d_zero_row, &ctx->d_out_qt, &ctx->d_out_d};
    for (DevBuf* b : bufs) b->release();
    ctx->res_qt.release();
    ctx->res_dist.release();
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    destroy_streams(ctx);
    delete ctx;
}

const char* msfm_last_error(const msfm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int msfm_device_info(const msfm_ctx* ctx, char* name, int name_cap, int* cu_count, int* clock_mhz) {
    if (!ctx) return MSFM_E_INVALID;
    if (name && name_cap > 0) std::snprintf(name, (size_t)name_cap, "%s", ctx->dev_name);
    if (cu_count) *cu_count = ctx->cu_count;
    if (clock_mhz) *clock_mhz = ctx->clock_mhz;
    return MSFM_OK;
}

int msfm_set_accum_order(msfm_ctx* ctx, int order) {
    if (!ctx) return MSFM_E_INVALID;
    if (order != MSFM_ORDER_SSE4X4 && order != MSFM_ORDER_AVX2_FMA && order != MSFM_ORDER_AVX512_FMA)
        return fail(ctx, MSFM_E_INVALID, "unknown accumulation order");
    if (order == ctx->order) return MSFM_OK;
    // the panel layout stores dimensions in accumulation order: re-lay every resident image
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->order = order;
    for (auto& im : ctx->images) {
        if (im.n <= 0) continue;
        const int blocks = std::min(4096, im.nalloc * 16);
        if (order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL((layout_kernel<0, float>), dim3(blocks), dim3(256), 0, SC.stream, im.raw, (float*)nullptr, im.panel, im.n, im.nalloc);
        else if (order == MSFM_ORDER_AVX2_FMA)
            hipLaunchKernelGGL((layout_kernel<1, float>), dim3(blocks), dim3(256), 0, SC.stream, im.raw, (float*)nullptr, im.panel, im.n, im.nalloc);
        else
            hipLaunchKernelGGL((layout_kernel<3, float>), dim3(blocks), dim3(256), 0, SC.stream, im.raw, (float*)nullptr, im.panel, im.n, im.nalloc);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    return MSFM_OK;
}

int msfm_set_prefilter(msfm_ctx* ctx, int enable) {
    if (!ctx) return MSFM_E_INVALID;
    ctx->prefilter = enable == 2 ? 2 : (enable ? 1 : 0);
    return MSFM_OK;
}

int msfm_set_limits(msfm_ctx* ctx, int max_pairs_per_batch, int64_t scratch_bytes) {
    if (!ctx) return MSFM_E_INVALID;
    // (the pair index of a sub-batch is gridDim.y of several kernels: at most 65535)
    ctx->max_pairs_per_batch = max_pairs_per_batch > 0 ? std::min(max_pairs_per_batch, kMaxPairsPerBatchLimit) : kDefaultMaxPairsPerBatch;
    ctx->scratch_bytes = scratch_bytes > 0 ? scratch_bytes : 0;
    return MSFM_OK;
}

int msfm_set_pipeline(msfm_ctx* ctx, int min_sub_batches) {
    if (!ctx) return MSFM_E_INVALID;
    ctx->pipeline = min_sub_batches > 0 ? std::min(min_sub_batches, 64) : kDefaultPipeline;
    return MSFM_OK;
}

int msfm_get_profile(const msfm_ctx* ctx, msfm_profile* out) {
    if (!ctx || !out) return MSFM_E_INVALID;
    *out = ctx->prof;
    return MSFM_OK;
}

// allocate the store of an n-row image (panels + row-major copy); rows are filled by the caller
static int alloc_image(msfm_ctx* ctx, Image& im, int n) {
    free_image(im);
    im.n = n;
    im.nblk = (n + kBM - 1) / kBM;
    im.nalloc = (im.nblk + kPfWgRows / kBM - 1) / (kPfWgRows / kBM) * (kPfWgRows / kBM);
    if (n == 0) return MSFM_OK;
    HIPCHK(ctx, hipMalloc((void**)&im.panel, (size_t)im.nalloc * kPanelFloats * 4));
    HIPCHK(ctx, hipMalloc((void**)&im.raw, (size_t)n * kDim * 4));
    HIPCHK(ctx, hipMalloc((void**)&im.rawp, (size_t)n * kDim * 4));
    return MSFM_OK;
}

// route Q: byte twin of a float image with values in [0, 1] (msfm_q8.hip.h); no twin (nothing allocated) when a value
// is negative / not finite or the rows' norms spread beyond the digit range
static void free_q8_twin(Image& im) {
    if (im.q8) (void)hipFree(im.q8);
    if (im.nrm_q8) (void)hipFree(im.nrm_q8);
    if (im.err_q8) (void)hipFree(im.err_q8);
    im.q8 = nullptr;
    im.nrm_q8 = nullptr;
    im.err_q8 = nullptr;
}

// (the image's values lie in [0, ctx->q8_level]: the caller has raised the level)
static int build_q8_twin(msfm_ctx* ctx, Image& im) {
    const int n = im.n, npad = im.nalloc * kBM;
    free_q8_twin(im);
    im.q8_level = ctx->q8_level;
    const float scale = 255.f / ctx->q8_level, inv = ctx->q8_level / 255.f;
    HIPCHK(ctx, ctx->d_stage.ensure((size_t)n * kDim * 4));
    HIPCHK(ctx, ctx->d_maxima.ensure(32));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_maxima.p, 0, 32, SC.stream));
    HIPCHK(ctx, hipMalloc((void**)&im.err_q8, (size_t)std::max(n, 1) * 4));
    unsigned* mx_d = ctx->d_maxima.as<unsigned>();
    hipLaunchKernelGGL(pf_quantise_q8_kernel, dim3(std::min(2048, (n + 3) / 4)), dim3(256), 0, SC.stream, (const float*)im.raw,
                       ctx->d_stage.as<float>(), im.err_q8, mx_d + 4, mx_d + 5, n, scale, inv);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMalloc((void**)&im.q8, (size_t)npad * kI8RowBytes));
    HIPCHK(ctx, hipMalloc((void**)&im.nrm_q8, (size_t)npad * 4));
    hipLaunchKernelGGL(pf_prepare_i8_kernel, dim3(std::min(2048, (npad * 11 + 255) / 256)), dim3(256), 0, SC.stream,
                       (const float*)ctx->d_stage.as<float>(), im.q8, im.nrm_q8, mx_d, n, npad, (int*)nullptr);
    HIPCHK(ctx, hipGetLastError());
    unsigned mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(mx, ctx->d_maxima.p, 32, hipMemcpyDeviceToHost, SC.stream));
    HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    float nmax = 0.f, nmin = 0.f;
    const unsigned min_bits = ~mx[3];
    std::memcpy(&nmax, &mx[2], 4);
    std::memcpy(&nmin, &min_bits, 4);
    std::memcpy(&im.err_q8_max, &mx[5], 4);
    const long long hmax = (long long)(0.5f * nmax), hmin = (long long)(0.5f * nmin);
    const long long h0 = (hmin + hmax) / 2;
    const bool ok = mx[4] == 0 && hmin <= hmax && h0 - hmax >= kI8DigitLo && h0 - hmin <= kI8DigitHi;
    if (!ok) {
        free_q8_twin(im);
        return MSFM_OK;
    }
    im.h0_q8 = (int)h0;
    hipLaunchKernelGGL(pf_digits_i8_kernel, dim3(std::min(1024, (n + 255) / 256)), dim3(256), 0, SC.stream,
                       (const float*)im.nrm_q8, im.q8, n, im.h0_q8);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    return MSFM_OK;
}

namespace {
// rawp: position 64 h + 4 L + c of a row <- its element 16 (4 h + c) + L (the exact re-check's 16-lane groups, msfm_prefilter.hip.h:
// lane L's float4 number h sits at float offset 64 h + 4 L -- the sixteen lanes of a group read 256 contiguous bytes per load)
__global__ void permute_rows_kernel(const float* __restrict__ raw, float* __restrict__ rawp, int n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < (long long)n * kDim; e += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(e & (kDim - 1)), h = k >> 6, L = (k >> 2) & 15, c = k & 3;
        rawp[e] = raw[(e & ~(long long)(kDim - 1)) + 16 * (4 * h + c) + L];
    }
}
}  // namespace

// everything derived from the row-major fp32 copy `im.raw` (src8 != nullptr: u8 rows still to be widened
// into im.raw by the layout kernel): panels in accumulation order, prefilter operands
static int build_image(msfm_ctx* ctx, Image& im, const unsigned char* src8, bool is_u8) {
    const int n = im.n;
    im.is_u8 = is_u8;
    im.from_u8 = is_u8;
    const int blocks = std::min(4096, im.nalloc * 16);
    if (!src8) {
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL((layout_kernel<0, float>), dim3(blocks), dim3(256), 0, SC.stream, im.raw, (float*)nullptr, im.panel, n, im.nalloc);
        else if (ctx->order == MSFM_ORDER_AVX2_FMA)
            hipLaunchKernelGGL((layout_kernel<1, float>), dim3(blocks), dim3(256), 0, SC.stream, im.raw, (float*)nullptr, im.panel, n, im.nalloc);
        else
            hipLaunchKernelGGL((layout_kernel<3, float>), dim3(blocks), dim3(256), 0, SC.stream, im.raw, (float*)nullptr, im.panel, n, im.nalloc);
    } else {
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL((layout_kernel<0, unsigned char>), dim3(blocks), dim3(256), 0, SC.stream, src8, im.raw, im.panel, n, im.nalloc);
        else if (ctx->order == MSFM_ORDER_AVX2_FMA)
            hipLaunchKernelGGL((layout_kernel<1, unsigned char>), dim3(blocks), dim3(256), 0, SC.stream, src8, im.raw, im.panel, n, im.nalloc);
        else
            hipLaunchKernelGGL((layout_kernel<3, unsigned char>), dim3(blocks), dim3(256), 0, SC.stream, src8, im.raw, im.panel, n, im.nalloc);
    }
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(permute_rows_kernel, dim3(std::min(2048, (n * kDim + 255) / 256)), dim3(256), 0, SC.stream, (const float*)im.raw, im.rawp, n);
    HIPCHK(ctx, hipGetLastError());
    // prefilter operands (order-independent): fp16 swizzled blocks, norms, maxima
    const int npad = im.nalloc * kBM;
    HIPCHK(ctx, hipMalloc((void**)&im.h16, (size_t)npad * kPfRowBytes));
    HIPCHK(ctx, hipMalloc((void**)&im.nrm, (size_t)npad * 4));
    HIPCHK(ctx, ctx->d_maxima.ensure(32));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_maxima.p, 0, 32, SC.stream));
    hipLaunchKernelGGL(pf_prepare_kernel, dim3(std::min(2048, (npad * 16 + 255) / 256)), dim3(256), 0, SC.stream,
                       im.raw, im.h16, im.nrm, ctx->d_maxima.as<unsigned>(), n, npad);
    HIPCHK(ctx, hipGetLastError());
    unsigned mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool detected = false;
    auto prepare_bytes = [&]() -> int {
        HIPCHK(ctx, hipMalloc((void**)&im.i8, (size_t)npad * kI8RowBytes));
        HIPCHK(ctx, hipMalloc((void**)&im.nrm_i8, (size_t)npad * 4));
        HIPCHK(ctx, hipMalloc((void**)&im.n2_i8, (size_t)npad * 4));
        hipLaunchKernelGGL(pf_prepare_i8_kernel, dim3(std::min(2048, (npad * 11 + 255) / 256)), dim3(256), 0, SC.stream,
                           (const float*)im.raw, im.i8, im.nrm_i8, ctx->d_maxima.as<unsigned>(), n, npad, im.n2_i8);
        HIPCHK(ctx, hipGetLastError());
        return MSFM_OK;
    };
    if (is_u8) {
        const int rc = prepare_bytes();
        if (rc != MSFM_OK) return rc;
    }
    HIPCHK(ctx, hipMemcpyAsync(mx, ctx->d_maxima.p, 32, hipMemcpyDeviceToHost, SC.stream));
    // the caller may free/reuse its buffer (and we reuse d_stage) as soon as we return
    HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    if (!is_u8 && ctx->byte_detect && mx[6] == 0 && n > 0) {
        // A FLOAT upload whose every value is an integer in [0, 255] (raw OpenCV SIFT stored as CV_32F, the reference's
        // Database::WriteDescriptors format before RootSIFT): the same store as a byte upload.  Every partial sum of
        // (a - b)^2 stays below 2^24, so S is an exact integer under any accumulation order and the image rides the
        // integer matrix cores.
        is_u8 = detected = true;
        im.is_u8 = im.from_u8 = true;
        const int rc = prepare_bytes();
        if (rc != MSFM_OK) return rc;
        HIPCHK(ctx, hipMemcpyAsync(mx, ctx->d_maxima.p, 32, hipMemcpyDeviceToHost, SC.stream));
        HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    }
    std::memcpy(&im.nrm_max, &mx[0], 4);
    std::memcpy(&im.abs_max, &mx[1], 4);
    std::memcpy(&im.nrm_i8_max, &mx[2], 4);
    if (is_u8) {
        // the digit k-step represents H0 - h in [kI8DigitLo, kI8DigitHi]: centre H0 between the smallest and the largest h
        const unsigned min_bits = ~mx[3];
        float nrm_i8_min = 0.f;
        std::memcpy(&nrm_i8_min, &min_bits, 4);
        const long long hmax = (long long)(0.5f * im.nrm_i8_max), hmin = (long long)(0.5f * nrm_i8_min);
        const long long h0 = (hmin + hmax) / 2;
        if (hmin <= hmax && h0 - hmax >= kI8DigitLo && h0 - hmin <= kI8DigitHi) {
            im.h0_i8 = (int)h0;
            hipLaunchKernelGGL(pf_digits_i8_kernel, dim3(std::min(1024, (n + 255) / 256)), dim3(256), 0, SC.stream,
                               (const float*)im.nrm_i8, im.i8, n, im.h0_i8);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipStreamSynchronize(SC.stream));
        } else {   // (an all-zero next to an all-128 descriptor: not SIFT) -- the image is served by the fp16 kernels
            (void)hipFree(im.i8);
            (void)hipFree(im.nrm_i8);
            (void)hipFree(im.n2_i8);
            im.i8 = nullptr;
            im.nrm_i8 = nullptr;
            im.n2_i8 = nullptr;
            im.is_u8 = false;
        }
    }
    if ((!is_u8 || detected) && ctx->q8_route && im.abs_max <= 1.f) {   // (a float image of 0 / 1 entries is both)
        ctx->q8_level = std::max(ctx->q8_level, std::max(kQ8LevelStep, std::ceil(im.abs_max / kQ8LevelStep) * kQ8LevelStep));
        int rc = build_q8_twin(ctx, im);
        if (rc != MSFM_OK) return rc;
    }
    im.pf_safe = (im.abs_max <= kF16Safe) && (im.nrm_max < 3.0e38f);  // NaN/inf compare false
    if (im.pf_safe) {
        // c = 2^k with max|row|^2 / 2 / c in (2^11, 2^12]; k must keep c an exact fp16 value
        int e = 0;
        (void)std::frexp(0.5f * im.nrm_max, &e);  // 0.5 nrm_max = m * 2^e, m in [0.5, 1)
        int k = (im.nrm_max > 0.f ? e : -24) - 12;
        if (k < -24) k = -24;
        if (k > 15) im.pf_safe = false;
        else {
            im.c = std::ldexp(1.f, k);
            hipLaunchKernelGGL(pf_ext_kernel, dim3((npad + 255) / 256), dim3(256), 0, SC.stream, im.nrm, im.h16, npad, im.c);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipStreamSynchronize(SC.stream));
        }
    }
    return MSFM_OK;
}

int msfm_upload_image(msfm_ctx* ctx, int image_id, const void* desc, int n, int dim, int dtype) {
    if (!ctx) return MSFM_E_INVALID;
    if (image_id < 0 || image_id >= kSlots) return fail(ctx, MSFM_E_INVALID, "image id out of range");
    if (n < 0 || dim != MSFM_DIM) return fail(ctx, MSFM_E_INVALID, "descriptors must be n x 128");
    if (n >= (1 << 18)) return fail(ctx, MSFM_E_INVALID, "more than 2^18 - 1 rows (BFMatcher packs the train index in 18 bits)");
    if (dtype != MSFM_DTYPE_F32 && dtype != MSFM_DTYPE_U8) return fail(ctx, MSFM_E_INVALID, "dtype must be F32 or U8");
    if (n > 0 && !desc) return fail(ctx, MSFM_E_INVALID, "null descriptor pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Image& im = ctx->images[image_id];
    int rc = alloc_image(ctx, im, n);
    if (rc != MSFM_OK || n == 0) return rc;
    if (dtype == MSFM_DTYPE_F32) {
        HIPCHK(ctx, hipMemcpyAsync(im.raw, desc, (size_t)n * kDim * 4, hipMemcpyHostToDevice, SC.stream));
        return build_image(ctx, im, nullptr, false);
    }
    HIPCHK(ctx, ctx->d_stage.ensure((size_t)n * kDim));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage.p, desc, (size_t)n * kDim, hipMemcpyHostToDevice, SC.stream));
    return build_image(ctx, im, ctx->d_stage.as<unsigned char>(), true);
}

namespace {
__global__ void subset_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int count) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < (long long)count * kDim; e += (long long)gridDim.x * blockDim.x)
        dst[e] = src[(size_t)idx[e >> 7] * kDim + (e & (kDim - 1))];
}
}  // namespace

int msfm_subset_image(msfm_ctx* ctx, int src_image_id, int dst_image_id, const int32_t* rows, int count) {
    if (!ctx) return MSFM_E_INVALID;
    if (src_image_id < 0 || src_image_id >= kSlots || dst_image_id < 0 || dst_image_id >= kSlots || src_image_id == dst_image_id)
        return fail(ctx, MSFM_E_INVALID, "bad image ids for msfm_subset_image");
    if (count < 0 || (count > 0 && !rows)) return fail(ctx, MSFM_E_INVALID, "bad row list");
    const Image& src = ctx->images[src_image_id];
    if (src.n < 0) return fail(ctx, MSFM_E_NOIMAGE, "image not uploaded: " + std::to_string(src_image_id));
    for (int i = 0; i < count; ++i)
        if (rows[i] < 0 || rows[i] >= src.n) return fail(ctx, MSFM_E_INVALID, "row index out of range in msfm_subset_image");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Image& im = ctx->images[dst_image_id];
    int rc = alloc_image(ctx, im, count);
    if (rc != MSFM_OK || count == 0) return rc;
    HIPCHK(ctx, ctx->d_stage.ensure((size_t)count * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage.p, rows, (size_t)count * 4, hipMemcpyHostToDevice, SC.stream));
    hipLaunchKernelGGL(subset_rows_kernel, dim3(std::min(1024, (count * kDim + 255) / 256)), dim3(256), 0, SC.stream,
                       (const float*)ctx->images[src_image_id].raw, (const int*)ctx->d_stage.as<int>(), im.raw, count);
    HIPCHK(ctx, hipGetLastError());
    return build_image(ctx, im, nullptr, ctx->images[src_image_id].is_u8);   // rows of a byte image are bytes
}

int msfm_image_rows(const msfm_ctx* ctx, int image_id, int* out_n) {
    if (!ctx || !out_n) return MSFM_E_INVALID;
    if (image_id < 0 || image_id >= kSlots) return MSFM_E_INVALID;
    if (ctx->images[image_id].n < 0) return MSFM_E_NOIMAGE;
    *out_n = ctx->images[image_id].n;
    return MSFM_OK;
}

int msfm_clear_images(msfm_ctx* ctx) {
    if (!ctx) return MSFM_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    for (auto& im : ctx->images) free_image(im);
    ctx->q8_level = 0.f;
    return MSFM_OK;
}

// An error return may leave launches of the failed batch in flight: drain the stream before handing control back, so
// that the caller can free or reuse its buffers and a following call starts from an idle stream.
static int drained(msfm_ctx* ctx, int rc) {
    if (rc != MSFM_OK && ctx) {
        for (Scratch& sc : ctx->sc)
            if (sc.stream) (void)hipStreamSynchronize(sc.stream);
        ctx->cur = &ctx->sc[0];
    }
    return rc;
}

namespace {

// host state of one sub-batch in flight (parallel to ctx->sc[slot])
struct SubBatch {
    bool active = false;
    Batch b;
    int begin = 0, end = 0;
    size_t ev_base = 0;
    bool exact_launched = false;
    int begin_of_cost = -1;       // the pair index cost_begin belongs to (a re-built sub-batch keeps its start cost)
    long long cost_begin = 0;
};

void add_profile(msfm_profile& to, const msfm_profile& d) {
    to.dist_kernel_ms += d.dist_kernel_ms;
    to.dist_kernel_launches += d.dist_kernel_launches;
    to.descriptor_pairs += d.descriptor_pairs;
    to.dist_algo_bytes += d.dist_algo_bytes;
    to.approx_kernel_ms += d.approx_kernel_ms;
    to.approx_kernel_launches += d.approx_kernel_launches;
    to.prefilter_pairs += d.prefilter_pairs;
    to.fallback_pairs += d.fallback_pairs;
    to.candidates += d.candidates;
    to.prefilter_descriptor_pairs += d.prefilter_descriptor_pairs;
    to.exact_descriptor_pairs += d.exact_descriptor_pairs;
    to.tie_rows += d.tie_rows;
    to.sweep2_ms += d.sweep2_ms;
    to.sweep2_launches += d.sweep2_launches;
    to.compacted_pairs += d.compacted_pairs;
    to.sweep2_descriptor_pairs += d.sweep2_descriptor_pairs;
    to.verify_ms += d.verify_ms;
    to.sub_batches += d.sub_batches;
    to.tie_queue_regrows += d.tie_queue_regrows;
    to.plan_regrows += d.plan_regrows;
    to.sweep1_i8_launches += d.sweep1_i8_launches;
    to.order_sensitive_rows += d.order_sensitive_rows;
    to.sweep1_q8_launches += d.sweep1_q8_launches;
    to.sweep1b_launches += d.sweep1b_launches;
    to.sweep1b_ms += d.sweep1b_ms;
    to.sweep1b_descriptor_pairs += d.sweep1b_descriptor_pairs;
    to.demoted_pairs += d.demoted_pairs;
    to.mixed_route_sub_batches += d.mixed_route_sub_batches;
}

int drain_streams(msfm_ctx* ctx) {
    for (Scratch& s : ctx->sc) HIPCHK(ctx, hipStreamSynchronize(s.stream));
    return MSFM_OK;
}

}  // namespace

// msfm_match_pairs / msfm_match_pairs_verified.  The call is cut into device sub-batches (memory, pair count, and -- for
// a large call -- at least ctx->pipeline of them); sub-batch k + 1 is LAUNCHED on the other stream / scratch set before
// the host waits for sub-batch k, so that the bandwidth-bound tail of k runs under sweep 1 of k + 1:
//
//      issue(0)  issue(1) complete(0)  issue(2) complete(1)  issue(3) complete(2)  ...  complete(last)
//
// issue(k)    = every launch of the sub-batch (sweeps, plan, exact re-check, epilogue, [verification], CSR gather into the
//               scratch set's own list buffer) + the copies of the words the host needs into page-locked memory;
// complete(k) = wait for k's stream; a queue / plan buffer was too small -> drain everything, re-run k alone (grown by
//               then), carry on behind it; else append k's lists to the call's lists (device-to-device for
//               msfm_fetch_matches_device, device-to-host into the page-locked result buffers: asynchronous, on k's stream).
// Results do not depend on the cut (tests force it every which way).
static int match_pairs_impl(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params,
                            const msfm_verify_params* verify, int64_t* out_offsets) {
    if (!ctx) return MSFM_E_INVALID;
    if (n_pairs < 0 || (n_pairs > 0 && !pairs) || !out_offsets) return fail(ctx, MSFM_E_INVALID, "bad pair list");
    msfm_match_params prm = {0.8f, 1, 0.7};
    if (params) prm = *params;
    // rows / columns that provably fail the ratio test or the distance cut need no exact neighbours
    PruneParams prune = {1, prm.ratio, (float)prm.max_distance};
    if ((double)prune.max_distance < prm.max_distance) prune.max_distance = nextafterf(prune.max_distance, __builtin_huge_valf());
    if (!(prm.ratio > 0.f) || !(prm.ratio <= 1.f)) prune.ratio = 0.f;  // outside (0, 1]: no ratio-based pruning
    if (!(prm.max_distance >= 0.0)) prune.max_distance = __builtin_huge_valf();  // NaN / negative: no distance-based pruning
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->cur = &ctx->sc[0];
    ctx->have_results = false;
    for (Scratch& s : ctx->sc) {
        s.pf_pending = PfPending{};
        s.sweep1_recorded = false;
        s.sweep2_recorded = false;
    }
    ctx->last_sweep1 = nullptr;
    // twins built before a later upload raised the context's level (msfm_q8.hip.h): rebuilt here, nothing is in flight
    if (ctx->q8_route && ctx->prefilter == 1)
        for (int k = 0; k < 2 * n_pairs; ++k) {
            const int id = pairs[k];
            if (id < 0 || id >= kSlots) continue;
            Image& im = ctx->images[id];
            if (im.q8 && im.q8_level != ctx->q8_level) {
                const int rc = build_q8_twin(ctx, im);
                if (rc != MSFM_OK) return rc;
            }
        }
    ctx->res_offsets.assign((size_t)n_pairs + 1, 0);
    ctx->res_sens.assign((size_t)n_pairs, 0);
    ctx->res_count = 0;
    ctx->prof = msfm_profile{};

    hipEvent_t ev_begin = get_event(ctx, 0), ev_end = get_event(ctx, 1);
    if (!ev_begin || !ev_end) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
    HIPCHK(ctx, hipEventRecord(ev_begin, ctx->sc[0].stream));

    // ---- the cut: scratch memory per set, pair count, and a cost limit that gives a large call >= `pipeline` sub-batches
    const int kSets = ctx->in_flight;
    // the scratch sets in flight SHARE the budget (three sets at half of it each were 1.5 x the documented limit)
    long long budget = ctx->scratch_bytes > 0 ? ctx->scratch_bytes : kDefaultScratchBytes;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            long long held = 0;   // what the scratch sets hold already is theirs to use
            for (Scratch& sc : ctx->sc) held += sc.device_bytes();
            const long long avail = (long long)free_b + held;
            budget = std::min(budget, ctx->scratch_bytes > 0 ? avail * 3 / 4 : avail / 4);
        }
    }
    const long long kScratchBytes = std::max<long long>(1, budget / kSets);
    // which buffers a pair needs (msfm_pair_scratch_bytes): 0 brute force, 1 matrix cores + compacted sweep 2, 2 + dense sweep 2
    const int scratch_route = !ctx->prefilter ? 0 : ((prune.ratio > 0.f && prune.ratio <= 0.95f) ? 1 : 2);
    const int kMaxPairsPerBatch = ctx->max_pairs_per_batch;
    // cumulative-cost marks of the parts (empty: no cost cut): msfm_pipeline_marks (msfm_hostutil.h) -- shrinking parts
    std::vector<long long> marks;
    if (ctx->pipeline > 1 && n_pairs > 1) {
        long long total = 0;
        for (int k = 0; k < n_pairs; ++k) {
            const int i = pairs[2 * k], j = pairs[2 * k + 1];
            if (i < 0 || i >= kSlots || j < 0 || j >= kSlots) return fail(ctx, MSFM_E_INVALID, "image id out of range");
            const long long n1 = ctx->images[i].n, n2 = ctx->images[j].n;
            if (n1 > 0 && n2 > 0) total += n1 * n2;
        }
        marks = msfm_pipeline_marks(total, std::min<long long>(ctx->pipeline, total / kMinPipelineCost), ctx->pipeline_taper);
    }
    long long cost_done = 0;   // cost of the sub-batches built so far (a re-built sub-batch starts from its own begin: see build)
    // a tie in sqrt space can only surface in a match list when a row with d0 == d1 can pass the ratio test
    const bool need_fix = !(prm.ratio <= 1.f);

    SubBatch sb[kInFlight];
    // pairs [begin, ...) -> sb.b (host tables); force_exact: pairs of this sub-batch whose candidate list overflowed in an
    // earlier attempt take the brute-force path
    auto build = [&](SubBatch& w, int begin, const std::vector<char>& force_exact) -> int {
        w.b = Batch{};
        w.begin = begin;
        if (begin == 0) cost_done = 0;
        const long long cost_begin = (begin == w.begin_of_cost) ? w.cost_begin : cost_done;
        w.begin_of_cost = begin;
        w.cost_begin = cost_begin;
        // the first cumulative-cost mark behind this sub-batch's start
        // (a part ends at the pair nearest to its mark, so its successor may start a little before or behind one)
        long long mark = 0;
        if (!marks.empty()) {
            size_t i = 0;
            while (i + 2 < marks.size() && cost_begin >= (marks[i] + marks[i + 1]) / 2) ++i;
            mark = marks[i + 1];
        }
        long long est = 0, cost = 0;
        int end = begin;
        while (end < n_pairs && (end - begin) < kMaxPairsPerBatch) {
            PairDesc pd;
            PfPair pp;
            int rc = fill_pair(ctx, pairs[2 * end], pairs[2 * end + 1], pd, pp);
            if (rc != MSFM_OK) return rc;
            if (verify) {
                const Image& ia = ctx->images[pairs[2 * end]];
                const Image& ib = ctx->images[pairs[2 * end + 1]];
                if (ia.nk < ia.n || ib.nk < ib.n)
                    return fail(ctx, MSFM_E_STATE, "geometric verification needs msfm_upload_keypoints for image " +
                                                       std::to_string(ia.nk < ia.n ? pairs[2 * end] : pairs[2 * end + 1]));
            }
            const long long need = pd.valid ? msfm_pair_scratch_bytes(pd.n1, pd.n2, pd.n1pad, pd.n2pad, pd.a_blocks, pd.a_blocks256,
                                                                     pp.use ? scratch_route : 0) : 0;
            const long long c = pd.valid ? (long long)pd.n1 * pd.n2 : 0;
            if (end > begin && est + need > kScratchBytes) break;
            if (end > begin && !marks.empty() && cost_begin + cost + c / 2 > mark) break;
            est += need;
            cost += c;
            const size_t k = w.b.pairs.size();
            if (k < force_exact.size() && force_exact[k]) {
                pp.use = 0;
                pd.path = 0;
            }
            w.b.pairs.push_back(pd);
            w.b.pf.push_back(pp);
            w.b.id1.push_back(pairs[2 * end]);
            w.b.id2.push_back(pairs[2 * end + 1]);
            ++end;
        }
        w.end = end;
        cost_done = cost_begin + cost;
        return MSFM_OK;
    };

    // every launch of the sub-batch on the CURRENT scratch set (ctx->cur), nothing waits
    auto issue = [&](SubBatch& w, size_t ev_base) -> int {
        Batch& b = w.b;
        const size_t P = b.pairs.size();
        const int begin = w.begin;
        w.ev_base = ev_base;
        w.exact_launched = false;
        SC.prof = msfm_profile{};
        SC.seq = ctx->issue_seq++;
        SC.sweep2_recorded = false;
        int rc = run_knn(ctx, b, ev_base, &w.exact_launched, prune, need_fix, true);
        if (rc != MSFM_OK) return rc;
        const long long oe = std::max<long long>(1, b.out_elems);
        HIPCHK(ctx, SC.d_st_qt.ensure(oe * sizeof(int2)));
        HIPCHK(ctx, SC.d_st_d.ensure(oe * 4));
        HIPCHK(ctx, SC.d_sub_qt.ensure(oe * sizeof(int2)));
        HIPCHK(ctx, SC.d_sub_d.ensure(oe * 4));
        HIPCHK(ctx, SC.d_counts.ensure(P * 4));
        HIPCHK(ctx, SC.d_sens.ensure(P * 4));
        HIPCHK(ctx, SC.d_offsets.ensure((P + 1) * 8));
        EpiParams ep = {prm.ratio, prm.cross_check, prm.max_distance};
        if (SC.keys_epilogue)
            hipLaunchKernelGGL(epilogue_kernel<KnnFromKeys>, dim3((unsigned)P), dim3(256), 0, SC.stream, SC.d_pairs.as<PairDesc>(), ep,
                               KnnFromKeys{SC.d_tu.as<float>(), SC.d_best.as<unsigned long long>(), SC.d_second.as<unsigned long long>()},
                               SC.d_st_qt.as<int2>(), SC.d_st_d.as<float>(), SC.d_counts.as<int>(), SC.d_sens.as<int>());
        else
            hipLaunchKernelGGL(epilogue_kernel<KnnFromArrays>, dim3((unsigned)P), dim3(256), 0, SC.stream, SC.d_pairs.as<PairDesc>(), ep,
                               KnnFromArrays{SC.d_k_i0.as<int>(), SC.d_k_d0.as<float>(), SC.d_k_d1.as<float>()},
                               SC.d_st_qt.as<int2>(), SC.d_st_d.as<float>(), SC.d_counts.as<int>(), SC.d_sens.as<int>());
        HIPCHK(ctx, hipGetLastError());
        const int* d_counts = SC.d_counts.as<int>();
        const int2* d_st_qt = SC.d_st_qt.as<int2>();
        const float* d_st_d = SC.d_st_d.as<float>();
        if (verify) {
            // FeatureUtils::FilterMatches on the staged lists: all hypotheses of all pairs at once
            VerifyParams vprm = {verify->threshold * verify->threshold, verify->confidence, verify->max_iters, 0, verify->seed};
            std::vector<VerifyPair>& vpairs = b.verify_pairs;   // (lives as long as the sub-batch: the copy below may still be in flight)
            vpairs.resize(P);
            for (size_t p = 0; p < P; ++p)
                vpairs[p] = VerifyPair{ctx->images[pairs[2 * (begin + (int)p)]].kxy, ctx->images[pairs[2 * (begin + (int)p) + 1]].kxy};
            HIPCHK(ctx, SC.d_vf_pairs.ensure(P * sizeof(VerifyPair)));
            HIPCHK(ctx, SC.d_vf_x1.ensure(oe * 4));
            HIPCHK(ctx, SC.d_vf_y1.ensure(oe * 4));
            HIPCHK(ctx, SC.d_vf_x2.ensure(oe * 4));
            HIPCHK(ctx, SC.d_vf_y2.ensure(oe * 4));
            HIPCHK(ctx, SC.d_vf_flags.ensure(oe));
            HIPCHK(ctx, SC.d_vf_hyp.ensure(P * (size_t)vprm.max_iters * 4));
            HIPCHK(ctx, SC.d_vf_best_it.ensure(P * 4));
            HIPCHK(ctx, SC.d_vf_best_count.ensure(P * 4));
            HIPCHK(ctx, SC.d_st2_qt.ensure(oe * sizeof(int2)));
            HIPCHK(ctx, SC.d_st2_d.ensure(oe * 4));
            HIPCHK(ctx, SC.d_counts2.ensure(P * 4));
            HIPCHK(ctx, hipMemcpyAsync(SC.d_vf_pairs.p, vpairs.data(), P * sizeof(VerifyPair), hipMemcpyHostToDevice, SC.stream));
            hipEvent_t v0 = get_event(ctx, ev_base + 6), v1 = get_event(ctx, ev_base + 7);
            if (!v0 || !v1) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
            HIPCHK(ctx, hipEventRecord(v0, SC.stream));
            const PairDesc* dp = SC.d_pairs.as<PairDesc>();
            float *x1 = SC.d_vf_x1.as<float>(), *y1 = SC.d_vf_y1.as<float>(), *x2 = SC.d_vf_x2.as<float>(), *y2 = SC.d_vf_y2.as<float>();
            hipLaunchKernelGGL(vf_points_kernel, dim3((unsigned)P), dim3(256), 0, SC.stream, dp, SC.d_vf_pairs.as<VerifyPair>(),
                               d_counts, d_st_qt, x1, y1, x2, y2);
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(vf_hypotheses_kernel, dim3((unsigned)((vprm.max_iters + 255) / 256), (unsigned)P), dim3(256), 0, SC.stream,
                               dp, d_counts, (const float*)x1, (const float*)y1, (const float*)x2, (const float*)y2,
                               SC.d_vf_hyp.as<int>(), vprm);
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(vf_select_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, SC.stream, d_counts,
                               (const int*)SC.d_vf_hyp.as<int>(), (int)P, vprm, SC.d_vf_best_it.as<int>(), SC.d_vf_best_count.as<int>());
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(vf_mask_compact_kernel, dim3((unsigned)P), dim3(256), 0, SC.stream, dp, d_counts, d_st_qt, d_st_d,
                               (const float*)x1, (const float*)y1, (const float*)x2, (const float*)y2,
                               (const int*)SC.d_vf_best_it.as<int>(), (const int*)SC.d_vf_best_count.as<int>(),
                               SC.d_vf_flags.as<unsigned char>(), vprm, SC.d_st2_qt.as<int2>(), SC.d_st2_d.as<float>(),
                               SC.d_counts2.as<int>());
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipEventRecord(v1, SC.stream));
            d_counts = SC.d_counts2.as<int>();
            d_st_qt = SC.d_st2_qt.as<int2>();
            d_st_d = SC.d_st2_d.as<float>();
        }
        hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(256), 0, SC.stream, d_counts,
                           SC.d_offsets.as<long long>(), (int)P);
        HIPCHK(ctx, hipGetLastError());
        // CSR order, into this scratch set's own list buffer: where the lists go in the call's buffers is only known when
        // the sub-batches before this one have been completed
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)P), dim3(256), 0, SC.stream, SC.d_pairs.as<PairDesc>(),
                           d_counts, SC.d_offsets.as<long long>(), d_st_qt,
                           d_st_d, SC.d_sub_qt.as<int2>(), SC.d_sub_d.as<float>());
        HIPCHK(ctx, hipGetLastError());
        rc = queue_tail_copies(ctx, P);
        if (rc != MSFM_OK) return rc;
        w.active = true;
        return MSFM_OK;
    };

    // wait for the sub-batch of the CURRENT scratch set; *retry: a queue / plan buffer was too small (grown by now)
    auto complete = [&](SubBatch& w, std::vector<char>& force_exact, bool* retry_out) -> int {
        Batch& b = w.b;
        const size_t P = b.pairs.size();
        w.active = false;
        *retry_out = false;
        HIPCHK(ctx, hipStreamSynchronize(SC.stream));
        bool retry = false, retry_pf = false;
        force_exact.resize(P, 0);
        int rc = check_fix_overflow(ctx, &retry);
        if (rc != MSFM_OK) return rc;
        rc = finish_prefilter(ctx, b, force_exact, &retry_pf);
        if (rc != MSFM_OK) return rc;
        if (retry || retry_pf) {   // only the re-run counters of a dropped attempt count
            ctx->prof.tie_queue_regrows += SC.prof.tie_queue_regrows;
            ctx->prof.plan_regrows += SC.prof.plan_regrows;
            ctx->prof.fallback_pairs += SC.prof.fallback_pairs;
            *retry_out = true;
            return MSFM_OK;
        }
        SC.prof.fallback_pairs = 0;   // (counted when the attempt that found them was dropped)
        const char* h = SC.h_tail.as<char>();
        const long long* offs = reinterpret_cast<const long long*>(h + 8);
        const int32_t* sens = reinterpret_cast<const int32_t*>(h + 8 + (P + 1) * 8);
        const long long total = offs[P];
        const size_t base = ctx->res_count;
        // the call's lists: grown with the end of the call in mind (matches per pair so far x pairs to come), so that a call
        // of a hundred sub-batches re-allocates -- and re-pins gigabytes of host memory -- once or twice, not a dozen times
        const size_t need = base + (size_t)total + 1;
        if (need * 8 > ctx->res_qt.cap || need * 4 > ctx->res_dist.cap || need * 8 > ctx->d_out_qt.cap || need * 4 > ctx->d_out_d.cap) {
            rc = drain_streams(ctx);   // copies of earlier sub-batches may still be writing into the buffers that move
            if (rc != MSFM_OK) return rc;
            const double per_pair = (double)(base + (size_t)total) / (double)std::max(1, w.end);
            const size_t hint = (size_t)(per_pair * (double)n_pairs * 1.08) + 4096;
            HIPCHK(ctx, ctx->res_qt.ensure(need * 8, base * 8, hint * 8));
            HIPCHK(ctx, ctx->res_dist.ensure(need * 4, base * 4, hint * 4));
            HIPCHK(ctx, ctx->d_out_qt.ensure_keep(need * sizeof(int2), base * sizeof(int2), SC.stream, hint * sizeof(int2)));
            HIPCHK(ctx, ctx->d_out_d.ensure_keep(need * 4, base * 4, SC.stream, hint * 4));
        }
        if (total > 0) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->res_qt.as<int32_t>() + 2 * base, SC.d_sub_qt.p, (size_t)total * 8, hipMemcpyDeviceToHost, SC.stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->res_dist.as<float>() + base, SC.d_sub_d.p, (size_t)total * 4, hipMemcpyDeviceToHost, SC.stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_out_qt.as<int2>() + base, SC.d_sub_qt.p, (size_t)total * 8, hipMemcpyDeviceToDevice, SC.stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_out_d.as<float>() + base, SC.d_sub_d.p, (size_t)total * 4, hipMemcpyDeviceToDevice, SC.stream));
        }
        ctx->res_count = base + (size_t)total;
        for (size_t p = 0; p < P; ++p) {
            ctx->res_offsets[(size_t)w.begin + p + 1] = (int64_t)base + offs[p + 1];
            ctx->res_sens[(size_t)w.begin + p] = sens[p];
            SC.prof.order_sensitive_rows += sens[p];
        }
        SC.prof.descriptor_pairs += b.desc_pairs;
        SC.prof.dist_algo_bytes += b.algo_bytes;
        rc = accumulate_kernel_time(ctx, w.ev_base, w.exact_launched);
        if (rc != MSFM_OK) return rc;
        if (verify) {
            float vms = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&vms, ctx->ev_pool[w.ev_base + 6], ctx->ev_pool[w.ev_base + 7]));
            SC.prof.verify_ms += vms;
        }
        SC.prof.sub_batches += 1;
        add_profile(ctx->prof, SC.prof);
        return MSFM_OK;
    };

    const std::vector<char> no_force;
    int next_begin = 0;
    long long issued = 0, completed = 0;   // sub-batch k lives in slot k % kSets
    while (completed < issued || next_begin < n_pairs) {
        // launch ahead: as many sub-batches as there are free scratch sets
        while (next_begin < n_pairs && issued - completed < kSets) {
            const int slot = (int)(issued % kSets);
            ctx->cur = &ctx->sc[slot];
            int rc = build(sb[slot], next_begin, no_force);
            if (rc != MSFM_OK) return rc;
            rc = issue(sb[slot], 2 + 12 * (size_t)slot);
            if (rc != MSFM_OK) return rc;
            next_begin = sb[slot].end;
            ++issued;
        }
        // wait for the oldest one
        const int slot = (int)(completed % kSets);
        ctx->cur = &ctx->sc[slot];
        std::vector<char> force_exact;
        bool retry = false;
        int rc = complete(sb[slot], force_exact, &retry);
        if (rc != MSFM_OK) return rc;
        if (retry) {
            // drop what is in flight behind it, re-run this sub-batch alone until it fits, carry on from its end
            rc = drain_streams(ctx);
            if (rc != MSFM_OK) return rc;
            for (int k = 0; k < kSets; ++k)
                if (k != slot) {
                    sb[k].active = false;
                    ctx->sc[k].pf_pending = PfPending{};
                    ctx->sc[k].sweep1_recorded = false;
                    ctx->sc[k].sweep2_recorded = false;
                }
            ctx->last_sweep1 = nullptr;
            for (int attempt = 1;; ++attempt) {
                rc = build(sb[slot], sb[slot].begin, force_exact);
                if (rc != MSFM_OK) return rc;
                rc = issue(sb[slot], 2 + 12 * (size_t)slot);
                if (rc != MSFM_OK) return rc;
                rc = complete(sb[slot], force_exact, &retry);
                if (rc != MSFM_OK) return rc;
                if (!retry) break;
                if (attempt >= 6) return fail(ctx, MSFM_E_DEVICE, "batch kept overflowing its queues");
            }
            next_begin = sb[slot].end;
            issued = completed + 1;
        }
        ++completed;
    }
    int rc = drain_streams(ctx);
    if (rc != MSFM_OK) return rc;
    ctx->cur = &ctx->sc[0];
    HIPCHK(ctx, hipEventRecord(ev_end, SC.stream));
    HIPCHK(ctx, hipEventSynchronize(ev_end));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ev_begin, ev_end));
    ctx->prof.total_device_ms = ms;
    std::memcpy(out_offsets, ctx->res_offsets.data(), ((size_t)n_pairs + 1) * sizeof(int64_t));
    ctx->have_results = true;
    return MSFM_OK;
}

int msfm_match_pairs(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params,
                     int64_t* out_offsets) {
    return drained(ctx, match_pairs_impl(ctx, pairs, n_pairs, params, nullptr, out_offsets));
}

int msfm_match_pairs_verified(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params,
                              const msfm_verify_params* verify, int64_t* out_offsets) {
    msfm_verify_params v = {3.0, 0.99, 1000, 0x5eed5eedULL};  // FeatureUtils.cpp:196: FM_RANSAC, 3.0, 0.99; OpenCV's maxIters
    if (verify) v = *verify;
    if (!(v.threshold >= 0.0) || !(v.confidence > 0.0) || !(v.confidence < 1.0) || v.max_iters < 1 || v.max_iters > (1 << 16))
        return fail(ctx, MSFM_E_INVALID, "bad verification parameters");
    return drained(ctx, match_pairs_impl(ctx, pairs, n_pairs, params, &v, out_offsets));
}

int msfm_upload_keypoints(msfm_ctx* ctx, int image_id, const float* kpts, int n, int stride_floats) {
    if (!ctx) return MSFM_E_INVALID;
    if (image_id < 0 || image_id >= kSlots) return fail(ctx, MSFM_E_INVALID, "image id out of range");
    if (n < 0 || (n > 0 && !kpts) || stride_floats < 2) return fail(ctx, MSFM_E_INVALID, "bad keypoint array");
    Image& im = ctx->images[image_id];
    if (im.n < 0) return fail(ctx, MSFM_E_NOIMAGE, "msfm_upload_keypoints before msfm_upload_image for image " + std::to_string(image_id));
    if (n < im.n) return fail(ctx, MSFM_E_INVALID, "fewer keypoints than descriptor rows");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (im.kxy) (void)hipFree(im.kxy);
    im.kxy = nullptr;
    im.nk = -1;
    std::vector<float2> xy((size_t)std::max(n, 1));
    for (int i = 0; i < n; ++i) xy[(size_t)i] = make_float2(kpts[(size_t)i * stride_floats], kpts[(size_t)i * stride_floats + 1]);
    HIPCHK(ctx, hipMalloc((void**)&im.kxy, xy.size() * sizeof(float2)));
    HIPCHK(ctx, hipMemcpyAsync(im.kxy, xy.data(), xy.size() * sizeof(float2), hipMemcpyHostToDevice, SC.stream));
    HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    im.nk = n;
    return MSFM_OK;
}

int msfm_fetch_matches(msfm_ctx* ctx, int32_t* out_qt, float* out_dist) {
    if (!ctx) return MSFM_E_INVALID;
    if (!ctx->have_results) return fail(ctx, MSFM_E_STATE, "msfm_fetch_matches without a completed msfm_match_pairs");
    if (out_qt && ctx->res_count) std::memcpy(out_qt, ctx->res_qt.p, ctx->res_count * 8);
    if (out_dist && ctx->res_count) std::memcpy(out_dist, ctx->res_dist.p, ctx->res_count * 4);
    return MSFM_OK;
}

int msfm_fetch_matches_device(msfm_ctx* ctx, int32_t* d_out_qt, float* d_out_dist) {
    if (!ctx) return MSFM_E_INVALID;
    if (!ctx->have_results) return fail(ctx, MSFM_E_STATE, "msfm_fetch_matches_device without a completed msfm_match_pairs");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (d_out_qt && ctx->res_count)
        HIPCHK(ctx, hipMemcpyAsync(d_out_qt, ctx->d_out_qt.p, ctx->res_count * 8, hipMemcpyDeviceToDevice, SC.stream));
    if (d_out_dist && ctx->res_count)
        HIPCHK(ctx, hipMemcpyAsync(d_out_dist, ctx->d_out_d.p, ctx->res_count * 4, hipMemcpyDeviceToDevice, SC.stream));
    HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    return MSFM_OK;
}

int msfm_fetch_order_certificate(msfm_ctx* ctx, int32_t* out_sensitive_rows) {
    if (!ctx) return MSFM_E_INVALID;
    if (!ctx->have_results) return fail(ctx, MSFM_E_STATE, "msfm_fetch_order_certificate without a completed msfm_match_pairs");
    if (out_sensitive_rows && !ctx->res_sens.empty()) std::memcpy(out_sensitive_rows, ctx->res_sens.data(), ctx->res_sens.size() * 4);
    return MSFM_OK;
}

int msfm_view_matches(msfm_ctx* ctx, const int32_t** out_qt, const float** out_dist, int64_t* out_count) {
    if (!ctx) return MSFM_E_INVALID;
    if (!ctx->have_results) return fail(ctx, MSFM_E_STATE, "msfm_view_matches without a completed msfm_match_pairs");
    if (out_qt) *out_qt = ctx->res_qt.as<int32_t>();
    if (out_dist) *out_dist = ctx->res_dist.as<float>();
    if (out_count) *out_count = (int64_t)ctx->res_count;
    return MSFM_OK;
}

int msfm_match_pair(msfm_ctx* ctx, int id1, int id2, float ratio, int cross_check, double max_distance,
                    int32_t* out_qt, float* out_dist, int* out_count) {
    if (!ctx || !out_count) return MSFM_E_INVALID;
    const int32_t pr[2] = {id1, id2};
    msfm_match_params prm = {ratio, cross_check, max_distance};
    int64_t offs[2] = {0, 0};
    int rc = msfm_match_pairs(ctx, pr, 1, &prm, offs);
    if (rc != MSFM_OK) return rc;
    *out_count = (int)offs[1];
    return msfm_fetch_matches(ctx, out_qt, out_dist);
}

static int knn2_pair_impl(msfm_ctx* ctx, int id1, int id2, int32_t* fwd_idx0, float* fwd_d0, float* fwd_d1,
                          int32_t* rev_idx0, float* rev_d0, float* rev_d1);

int msfm_knn2_pair(msfm_ctx* ctx, int id1, int id2, int32_t* fwd_idx0, float* fwd_d0, float* fwd_d1,
                   int32_t* rev_idx0, float* rev_d0, float* rev_d1) {
    if (!ctx) return MSFM_E_INVALID;
    return drained(ctx, knn2_pair_impl(ctx, id1, id2, fwd_idx0, fwd_d0, fwd_d1, rev_idx0, rev_d0, rev_d1));
}

static int knn2_pair_impl(msfm_ctx* ctx, int id1, int id2, int32_t* fwd_idx0, float* fwd_d0, float* fwd_d1,
                          int32_t* rev_idx0, float* rev_d0, float* rev_d1) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->cur = &ctx->sc[0];
    ctx->prof = msfm_profile{};
    for (Scratch& sc : ctx->sc) {   // (a failed earlier batch may have left its end-of-batch state behind)
        sc.pf_pending = PfPending{};
        sc.sweep1_recorded = false;
        sc.sweep2_recorded = false;
    }
    ctx->last_sweep1 = nullptr;
    Batch b;
    PairDesc pd;
    PfPair pp;
    int rc = fill_pair(ctx, id1, id2, pd, pp);
    if (rc != MSFM_OK) return rc;
    b.pairs.push_back(pd);
    b.pf.push_back(pp);
    b.id1.push_back(id1);
    b.id2.push_back(id2);
    bool exact_launched = false;
    int regrows = 0, fallbacks = 0;
    std::vector<char> force_exact(1, 0);
    for (int attempt = 0;; ++attempt) {
        b.n_items = 0;
        b.rp_elems = b.cp_elems = b.kf_elems = b.kr_elems = b.out_elems = b.cand_elems = 0;
        b.desc_pairs = b.algo_bytes = 0;
        b.pairs[0].path = b.pf[0].use = force_exact[0] ? 0 : pp.use;
        SC.prof = msfm_profile{};   // only the attempt that is kept counts
        rc = run_knn(ctx, b, 2, &exact_launched, PruneParams{0, 0.f, 0.f}, true, false);  // knnMatch twin: every row keeps its neighbours
        if (rc != MSFM_OK) return rc;
        SC.d_offsets.release();   // (no CSR on this path: the export kernel skips absent segments)
        SC.d_sens.release();
        rc = queue_tail_copies(ctx, 1);
        if (rc != MSFM_OK) return rc;
        HIPCHK(ctx, hipStreamSynchronize(SC.stream));
        bool retry = false, retry_pf = false;
        rc = check_fix_overflow(ctx, &retry);
        if (rc != MSFM_OK) return rc;
        rc = finish_prefilter(ctx, b, force_exact, &retry_pf);
        if (rc != MSFM_OK) return rc;
        if (!retry && !retry_pf) break;
        if (retry) ++regrows;
        if (retry_pf) fallbacks += SC.prof.fallback_pairs;
        if (attempt >= 6) return fail(ctx, MSFM_E_DEVICE, "batch kept overflowing its queues");
    }
    rc = accumulate_kernel_time(ctx, 2, exact_launched);
    if (rc != MSFM_OK) return rc;
    ctx->prof = SC.prof;
    ctx->prof.tie_queue_regrows = regrows;
    ctx->prof.fallback_pairs = fallbacks;
    ctx->prof.sub_batches = 1;
    ctx->prof.descriptor_pairs += b.desc_pairs;
    ctx->prof.dist_algo_bytes += b.algo_bytes;
    const PairDesc& q = b.pairs[0];
    const int n1 = ctx->images[id1].n, n2 = ctx->images[id2].n;
    if (!q.valid) {
        // an empty side: no neighbours
        for (int i = 0; i < n1; ++i) {
            if (fwd_idx0) fwd_idx0[i] = -1;
            if (fwd_d0) fwd_d0[i] = 3.402823466e+38f;
            if (fwd_d1) fwd_d1[i] = 3.402823466e+38f;
        }
        for (int i = 0; i < n2; ++i) {
            if (rev_idx0) rev_idx0[i] = -1;
            if (rev_d0) rev_d0[i] = 3.402823466e+38f;
            if (rev_d1) rev_d1[i] = 3.402823466e+38f;
        }
        return MSFM_OK;
    }
    struct Cp { void* dst; const DevBuf* src; long long off; int n; };
    const Cp cps[6] = {{fwd_idx0, &SC.d_k_i0, q.kf_off, n1}, {fwd_d0, &SC.d_k_d0, q.kf_off, n1},
                       {fwd_d1, &SC.d_k_d1, q.kf_off, n1},   {rev_idx0, &SC.d_k_i0, q.kr_off, n2},
                       {rev_d0, &SC.d_k_d0, q.kr_off, n2},   {rev_d1, &SC.d_k_d1, q.kr_off, n2}};
    for (const Cp& c : cps)
        if (c.dst && c.n > 0)
            HIPCHK(ctx, hipMemcpyAsync(c.dst, c.src->as<char>() + c.off * 4, (size_t)c.n * 4, hipMemcpyDeviceToHost, SC.stream));
    HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    return MSFM_OK;
}

// ---- host-only helpers --------------------------------------------------------------------

int msfm_topscale_select(const float* kpts, int n, int k, int32_t* out_idx, int* out_count) {
    if (n < 0 || k < 0 || !out_idx || !out_count || (n > 0 && !kpts)) return MSFM_E_INVALID;
    if (k > n) {  // "if(num_features > kpts.size()) top_scale_descriptors = descriptors"
        for (int i = 0; i < n; ++i) out_idx[i] = i;
        *out_count = n;
        return MSFM_OK;
    }
    std::vector<int32_t> order((size_t)n);
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    // documented tie rule: size descending, then index ascending (the reference's
    // std::partial_sort leaves the order of equal sizes unspecified)
    std::partial_sort(order.begin(), order.begin() + k, order.end(), [kpts](int32_t a, int32_t b) {
        const float sa = kpts[(size_t)a * 4 + 2], sb = kpts[(size_t)b * 4 + 2];
        if (sa != sb) return sa > sb;
        return a < b;
    });
    for (int i = 0; i < k; ++i) out_idx[i] = order[(size_t)i];
    *out_count = k;
    return MSFM_OK;
}

int msfm_swap_image_pair(int id1, int id2) { return id1 > id2 ? 1 : 0; }

int msfm_pair_id(int id1, int id2, int32_t* out_pair_id) {
    if (!out_pair_id || id1 < 0 || id2 < 0 || id1 >= MSFM_MAX_IMAGES || id2 >= MSFM_MAX_IMAGES) return MSFM_E_INVALID;
    *out_pair_id = msfm_swap_image_pair(id1, id2) ? MSFM_MAX_IMAGES * id2 + id1 : MSFM_MAX_IMAGES * id1 + id2;
    return MSFM_OK;
}

int msfm_pair_from_id(int32_t pair_id, int* out_id1, int* out_id2) {
    if (!out_id1 || !out_id2 || pair_id < 0) return MSFM_E_INVALID;
    *out_id2 = pair_id % MSFM_MAX_IMAGES;
    *out_id1 = (pair_id - *out_id2) / MSFM_MAX_IMAGES;
    return MSFM_OK;
}

}  // extern "C"
