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

#include "msfm_ctx.hip.h"
#include "msfm_store_host.hip.h"
#include "msfm_batch.hip.h"

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
    ctx->store.release_all();
    ctx->inbox.release_all();
    for (Scratch& sc : ctx->sc) sc.release_all();
    DevBuf* bufs[] = {&ctx->d_jobs, &ctx->d_store_maxima, &ctx->d_zero_row, &ctx->d_out_qt, &ctx->d_out_d};
    for (DevBuf* b : bufs) b->release();
    ctx->res_qt.release();
    ctx->res_dist.release();
    ctx->up_ring.release();
    ctx->h_jobs.release();
    ctx->h_store_maxima.release();
    for (hipEvent_t e : ctx->up_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->jobs_ev) (void)hipEventDestroy(ctx->jobs_ev);
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
    // uploads since the last use are built now (msfm_store_host.hip.h); twins built before a later upload raised the context's level
    // (msfm_q8.hip.h) are rebuilt: nothing is in flight
    {
        int rc = settle_store(ctx);
        if (rc != MSFM_OK) return rc;
        rc = rebuild_stale_twins(ctx, pairs, 2 * n_pairs);
        if (rc != MSFM_OK) return rc;
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
        // the forms of the images this sub-batch's routes read (derived on demand for byte images), then the pairs' pointers
        return prepare_batch_images(ctx, w.b, prune);
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
    int rc = settle_store(ctx);
    if (rc != MSFM_OK) return rc;
    rc = fill_pair(ctx, id1, id2, pd, pp);
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
        rc = prepare_batch_images(ctx, b, PruneParams{0, 0.f, 0.f});   // (the kNN-level API reads the float forms: nothing is pruned)
        if (rc != MSFM_OK) return rc;
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
