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
#include "msfm_guard.h"
#include "msfm_kernels.hip.h"
#include "msfm_prefilter.hip.h"
#include "msfm_verify.hip.h"

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
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
    MSFM_API_BEGIN(nullptr)
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return 0;
    int usable = 0;
    for (int d = 0; d < count; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) break;
        ++usable;   // (ordinals are contiguous: the count of leading gfx950 devices)
    }
    return usable;
    MSFM_API_END
}

static void destroy_streams(msfm_ctx* ctx);
static MatchJob* new_match_job();
static void delete_match_job(MatchJob* j);

// The stream and the events of scratch set k exist.  A stream is a hardware queue: ~11 ms to create on this part -- msfm_create makes
// the first, a call that puts a second / third sub-batch in flight the others, at a moment when the device is busy with the sub-batch before
// and the host is ahead of it (the ComputeMatches executable on the South-Building job: one set for the pre-emptive filter, two for the
// pairs: 22 ms less in front of the first upload, 11 of them never spent).
int ensure_scratch_set(msfm_ctx* ctx, int k) {
    Scratch& sc = ctx->sc[k];
    if (sc.stream) return MSFM_OK;
    hipError_t e = hipSuccess;
    if ((e = hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&sc.sweep1_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&sc.sweep2_done, hipEventDisableTiming)) != hipSuccess)
        return fail(ctx, MSFM_E_DEVICE, std::string("stream / events of a scratch set: ") + hipGetErrorString(e));
    return MSFM_OK;
}

// msfm_create behind the device checks: streams and events of the scratch sets, the kernels' LDS attributes, the zero row
static int create_rest(msfm_ctx* ctx) {
    auto bad = [ctx](const char* what, hipError_t e) { return fail(ctx, MSFM_E_DEVICE, std::string("msfm_create: ") + what + ": " + hipGetErrorString(e)); };
    hipError_t e = hipSuccess;
    // (the first scratch set's stream -- uploads and the store build use it as well; the other sets get theirs when a call first has
    // more than one sub-batch in flight: ensure_scratch_set)
    if (ensure_scratch_set(ctx, 0) != MSFM_OK) return MSFM_E_DEVICE;
    // dynamic LDS beyond 64 KiB: the brute-force kernel (108 KiB), the sweeps (94 / 124 KiB)
    struct { const void* f; int bytes; } attrs[] = {
        {reinterpret_cast<const void*>(dist_top2_kernel<0>), kLdsBytes},     {reinterpret_cast<const void*>(dist_top2_kernel<1>), kLdsBytes},
        {reinterpret_cast<const void*>(dist_top2_kernel<3>), kLdsBytesIdxStash},
        {reinterpret_cast<const void*>(sweep_kernel<1>), kPfLdsBytes},       {reinterpret_cast<const void*>(sweep_kernel<2>), kPfLdsBytes},
        {reinterpret_cast<const void*>(sweep_kernel<3>), kPfLdsBytes},       {reinterpret_cast<const void*>(sweep_kernel<4>), kPfLdsBytes},
        {reinterpret_cast<const void*>(sweep_i8_kernel<1>), kI8LdsBytes},    {reinterpret_cast<const void*>(sweep_i8_kernel<3>), kI8LdsBytes3},
    };
    for (const auto& a : attrs)
        if ((e = hipFuncSetAttribute(a.f, hipFuncAttributeMaxDynamicSharedMemorySize, a.bytes)) != hipSuccess)
            return bad("cannot reserve the kernels' LDS", e);
    // the all-zero operand row the compacted sweep reads for rows without a source
    if ((e = ctx->d_zero_row.ensure(kPfRowBytes)) != hipSuccess || (e = hipMemset(ctx->d_zero_row.p, 0, kPfRowBytes)) != hipSuccess)
        return bad("the zero row", e);
    return MSFM_OK;
}

int msfm_create(int device_ordinal, msfm_ctx** out_ctx) {
    MSFM_API_BEGIN(nullptr)
    if (!out_ctx) return MSFM_E_INVALID;
    *out_ctx = nullptr;
    int count = 0;
    HostClock hc;   // MSFM_DEBUG_TIMING=1: where a context's creation goes (the runtime's own start-up is most of it)
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MSFM_E_DEVICE;  // no GPU, no fallback
    hc.lap("create: hipGetDeviceCount");
    if (device_ordinal < 0 || device_ordinal >= count) return MSFM_E_INVALID;
    if (hipSetDevice(device_ordinal) != hipSuccess) return MSFM_E_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return MSFM_E_DEVICE;
    hc.lap("create: set device, properties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "msfm_create: device %d is %s, this library is built for gfx950 only\n",
                     device_ordinal, prop.gcnArchName);
        return MSFM_E_DEVICE;
    }
    msfm_ctx* ctx = new (std::nothrow) msfm_ctx();
    if (!ctx) return MSFM_E_DEVICE;
    ctx->device = device_ordinal;
    ctx->images.resize(kSlots);
    ctx->inbox.recycle = true;
    ctx->job = new_match_job();
    for (Scratch& sc : ctx->sc) {
        for (PinnedBuf& h : sc.h_up) h.pool = &ctx->pinned_pool;
        sc.h_summary.pool = sc.h_tail.pool = &ctx->pinned_pool;
    }
    ctx->h_jobs.pool = ctx->h_store_maxima.pool = &ctx->pinned_pool;
    ctx->cu_count = prop.multiProcessorCount;
    ctx->clock_mhz = prop.clockRate / 1000;
    std::snprintf(ctx->dev_name, sizeof(ctx->dev_name), "%s (%s)", prop.name, prop.gcnArchName);
    // (Round 5 tried the streams of the other scratch sets, the kernel attributes and the zero row on a helper thread, joined by the first
    // matching call: 45 of msfm_create's 95 ms.  The runtime serialises them with the uploads the caller makes meanwhile -- the
    // ComputeMatches executable's bulk load went from 48 to 72 ms --, so nothing was gained: profiles/r05_cli_cold_call.txt.)
    if (create_rest(ctx) != MSFM_OK) {
        std::fprintf(stderr, "%s\n", ctx->err.c_str());
        destroy_streams(ctx);
        ctx->d_zero_row.release();
        delete_match_job(ctx->job);
        delete ctx;
        return MSFM_E_DEVICE;
    }
    hc.lap("create: streams, events, kernel attributes, zero row");
    if (const char* e = std::getenv("MSFM_PREFILTER")) ctx->prefilter = e[0] == '2' ? 2 : (e[0] != '0');
    if (const char* e = std::getenv("MSFM_MAX_PAIRS_PER_BATCH"))
        if (std::atoi(e) > 0) ctx->max_pairs_per_batch = std::min(std::atoi(e), kMaxPairsPerBatchLimit);
    if (const char* e = std::getenv("MSFM_PIPELINE_TAPER")) {
        const double t = std::atof(e);
        if (t >= 0.05 && t <= 1.0) ctx->pipeline_taper = t;
    }
    if (const char* e = std::getenv("MSFM_BYTE_DETECT")) ctx->byte_detect = e[0] != '0';
    if (const char* e = std::getenv("MSFM_UPLOAD_THREADS")) ctx->copy_helper = std::atoi(e) != 1;
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
    MSFM_API_END
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
    try {
    HostClock hc;   // MSFM_DEBUG_TIMING=1
    (void)hipSetDevice(ctx->device);
    for (Scratch& sc : ctx->sc)
        if (sc.stream) (void)hipStreamSynchronize(sc.stream);
    ctx->deferred.flush();
    ctx->store.release_all();
    ctx->inbox.release_all();
    hc.lap("destroy: store");
    for (Scratch& sc : ctx->sc) sc.release_all();
    hc.lap("destroy: scratch sets");
    DevBuf* bufs[] = {&ctx->d_jobs, &ctx->d_store_maxima, &ctx->d_zero_row};
    for (DevBuf* b : bufs) b->release();
    for (OutSeg& s : ctx->out_segs) {
        s.qt.release();
        s.d.release();
    }
    ctx->res_qt.release();
    ctx->res_dist.release();
    ctx->up_ring.release();
    ctx->h_jobs.release();
    ctx->h_store_maxima.release();
    ctx->pinned_pool.release();   // (behind every buffer that holds a piece of it: the scratch sets' above)
    hc.lap("destroy: result lists, page-locked memory");
    for (hipEvent_t e : ctx->up_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->jobs_ev) (void)hipEventDestroy(ctx->jobs_ev);
    delete_match_job(ctx->job);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    destroy_streams(ctx);
    hc.lap("destroy: events, streams");
    delete ctx;
    } catch (...) {   // (the C ABI never throws; nothing above is expected to)
    }
}

const char* msfm_last_error(const msfm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int msfm_device_info(const msfm_ctx* ctx, char* name, int name_cap, int* cu_count, int* clock_mhz) {
    MSFM_API_BEGIN(nullptr)
    if (!ctx) return MSFM_E_INVALID;
    if (name && name_cap > 0) std::snprintf(name, (size_t)name_cap, "%s", ctx->dev_name);
    if (cu_count) *cu_count = ctx->cu_count;
    if (clock_mhz) *clock_mhz = ctx->clock_mhz;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_set_prefilter(msfm_ctx* ctx, int enable) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    ctx->prefilter = enable == 2 ? 2 : (enable ? 1 : 0);
    return MSFM_OK;
    MSFM_API_END
}

int msfm_set_limits(msfm_ctx* ctx, int max_pairs_per_batch, int64_t scratch_bytes) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    // (the pair index of a sub-batch is gridDim.y of several kernels: at most 65535)
    ctx->max_pairs_per_batch = max_pairs_per_batch > 0 ? std::min(max_pairs_per_batch, kMaxPairsPerBatchLimit) : kDefaultMaxPairsPerBatch;
    ctx->scratch_bytes = scratch_bytes > 0 ? scratch_bytes : 0;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_set_pipeline(msfm_ctx* ctx, int min_sub_batches) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    ctx->pipeline = min_sub_batches > 0 ? std::min(min_sub_batches, 64) : kDefaultPipeline;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_get_profile(const msfm_ctx* ctx, msfm_profile* out) {
    MSFM_API_BEGIN(nullptr)
    if (!ctx || !out) return MSFM_E_INVALID;
    *out = ctx->prof;
    return MSFM_OK;
    MSFM_API_END
}

// An error return may leave launches of the failed batch in flight: drain the stream before handing control back, so
// that the caller can free or reuse its buffers and a following call starts from an idle stream.
static int drained(msfm_ctx* ctx, int rc) {
    if (rc != MSFM_OK && ctx) {
        ctx->series_open = false;
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
    std::vector<int64_t> offsets; // streaming form: the completed sub-batch's CSR offsets (relative to the sub-batch) and
    std::vector<int32_t> sens;    //   order-certificate counts, until the caller asks for the next chunk
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
    to.memory_shrinks += d.memory_shrinks;
}

int drain_streams(msfm_ctx* ctx) {
    for (Scratch& s : ctx->sc)
        if (s.stream) HIPCHK(ctx, hipStreamSynchronize(s.stream));
    return MSFM_OK;
}

}  // namespace

#include "msfm_job.hip.h"

namespace {
// behind an exception caught at the C ABI (msfm_guard.h): the text for msfm_last_error, nothing left in flight, no series left open
void after_api_exception(msfm_ctx* ctx, const char* text) noexcept {
    if (!ctx) return;
    try {
        ctx->err = text ? text : "C++ exception";
    } catch (...) {
    }
    ctx->series_open = false;
    ctx->have_results = false;
    if (ctx->job) ctx->job->open = false;
    for (Scratch& sc : ctx->sc)
        if (sc.stream) (void)hipStreamSynchronize(sc.stream);
    ctx->cur = &ctx->sc[0];
}
}  // namespace

// Any matching call ends a streaming series that was left open (msfm_match_pairs_begin without the msfm_match_pairs_next that returns
// n_pairs == 0): its sub-batches in flight are drained, the store is unlocked, a later msfm_match_pairs_next returns MSFM_E_STATE.
static int abandon_series(msfm_ctx* ctx) {
    MatchJob& job = *ctx->job;
    if (ctx->series_open || (job.open && job.streaming)) {
        const int rc = drain_streams(ctx);
        ctx->series_open = false;
        job.open = false;
        if (rc != MSFM_OK) return rc;
        for (SubBatch& w : job.sb) w.active = false;
    }
    job.open = false;
    return MSFM_OK;
}

static int match_pairs_impl(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params,
                            const msfm_verify_params* verify, int64_t* out_offsets) {
    if (!ctx) return MSFM_E_INVALID;
    if (n_pairs < 0 || (n_pairs > 0 && !pairs) || !out_offsets) return fail(ctx, MSFM_E_INVALID, "bad pair list");
    MatchJob& job = *ctx->job;
    const int rc0 = abandon_series(ctx);   // a streaming series left open is abandoned: what it has in flight is drained first
    if (rc0 != MSFM_OK) return rc0;
    int rc = job.start(ctx, pairs, n_pairs, params, verify, false);
    if (rc != MSFM_OK) return rc;
    while (job.more()) {
        int slot = 0;
        rc = job.step(&slot);
        if (rc != MSFM_OK) return rc;
    }
    rc = job.finish();
    if (rc != MSFM_OK) return rc;
    std::memcpy(out_offsets, ctx->res_offsets.data(), ((size_t)n_pairs + 1) * sizeof(int64_t));
    ctx->have_results = true;
    return MSFM_OK;
}

int msfm_match_pairs(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params,
                     int64_t* out_offsets) {
    MSFM_API_BEGIN(ctx)
    return drained(ctx, match_pairs_impl(ctx, pairs, n_pairs, params, nullptr, out_offsets));
    MSFM_API_END
}

int msfm_match_pairs_verified(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params,
                              const msfm_verify_params* verify, int64_t* out_offsets) {
    MSFM_API_BEGIN(ctx)
    msfm_verify_params v = {3.0, 0.99, 1000, 0x5eed5eedULL};  // FeatureUtils.cpp:196: FM_RANSAC, 3.0, 0.99; OpenCV's maxIters
    if (verify) v = *verify;
    if (!(v.threshold >= 0.0) || !(v.confidence > 0.0) || !(v.confidence < 1.0) || v.max_iters < 1 || v.max_iters > (1 << 16))
        return fail(ctx, MSFM_E_INVALID, "bad verification parameters");
    return drained(ctx, match_pairs_impl(ctx, pairs, n_pairs, params, &v, out_offsets));
    MSFM_API_END
}

static MatchJob* new_match_job() { return new (std::nothrow) MatchJob(); }
static void delete_match_job(MatchJob* j) { delete j; }

// ---- streaming form: one device sub-batch per call, nothing accumulates (include/msfm_match.h) -------------------------------------
static int begin_impl(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params, int geometric_verification,
                      const msfm_verify_params* verify) {
    if (n_pairs < 0 || (n_pairs > 0 && !pairs)) return fail(ctx, MSFM_E_INVALID, "bad pair list");
    msfm_verify_params v = {3.0, 0.99, 1000, 0x5eed5eedULL};
    if (verify) v = *verify;
    if (geometric_verification &&
        (!(v.threshold >= 0.0) || !(v.confidence > 0.0) || !(v.confidence < 1.0) || v.max_iters < 1 || v.max_iters > (1 << 16)))
        return fail(ctx, MSFM_E_INVALID, "bad verification parameters");
    MatchJob& job = *ctx->job;
    const int rc0 = abandon_series(ctx);
    if (rc0 != MSFM_OK) return rc0;
    job.pairs_own.assign(pairs, pairs + 2 * (size_t)n_pairs);
    return job.start(ctx, job.pairs_own.data(), n_pairs, params, geometric_verification ? &v : nullptr, true);
}

int msfm_match_pairs_begin(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params, int geometric_verification,
                           const msfm_verify_params* verify) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    const int rc = drained(ctx, begin_impl(ctx, pairs, n_pairs, params, geometric_verification, verify));
    if (rc != MSFM_OK) ctx->job->open = false;
    return rc;
    MSFM_API_END
}

static int next_impl(msfm_ctx* ctx, msfm_chunk* out) {
    MatchJob& job = *ctx->job;
    if (!job.open || !job.streaming) return fail(ctx, MSFM_E_STATE, "msfm_match_pairs_next without msfm_match_pairs_begin");
    *out = msfm_chunk{};
    if (!job.more()) return job.finish();   // n_pairs == 0: the series is complete
    int slot = 0;
    const int rc = job.step(&slot);
    if (rc != MSFM_OK) return rc;
    const SubBatch& w = job.sb[slot];
    const Scratch& sc = ctx->sc[slot];
    out->first_pair = w.begin;
    out->n_pairs = w.end - w.begin;
    out->offsets = w.offsets.data();
    out->count = w.offsets.empty() ? 0 : w.offsets.back();
    out->qt = sc.h_sub_qt.as<int32_t>();
    out->dist = sc.h_sub_d.as<float>();
    out->d_qt = sc.d_sub_qt.as<int32_t>();
    out->d_dist = sc.d_sub_d.as<float>();
    out->sensitive_rows = w.sens.data();
    return MSFM_OK;
}

int msfm_match_pairs_next(msfm_ctx* ctx, msfm_chunk* out) {
    MSFM_API_BEGIN(ctx)
    if (!ctx || !out) return MSFM_E_INVALID;
    const int rc = drained(ctx, next_impl(ctx, out));
    if (rc != MSFM_OK) ctx->job->open = false;
    return rc;
    MSFM_API_END
}

int msfm_match_pairs_end(msfm_ctx* ctx) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return abandon_series(ctx);
    MSFM_API_END
}

int msfm_fetch_matches(msfm_ctx* ctx, int32_t* out_qt, float* out_dist) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    if (!ctx->have_results) return fail(ctx, MSFM_E_STATE, "msfm_fetch_matches without a completed msfm_match_pairs");
    if (out_qt && ctx->res_count) std::memcpy(out_qt, ctx->res_qt.base, ctx->res_count * 8);
    if (out_dist && ctx->res_count) std::memcpy(out_dist, ctx->res_dist.base, ctx->res_count * 4);
    return MSFM_OK;
    MSFM_API_END
}

int msfm_fetch_matches_device(msfm_ctx* ctx, int32_t* d_out_qt, float* d_out_dist) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    if (!ctx->have_results) return fail(ctx, MSFM_E_STATE, "msfm_fetch_matches_device without a completed msfm_match_pairs");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (size_t k = 0; k < ctx->out_used; ++k) {
        const OutSeg& s = ctx->out_segs[k];
        if (!s.count) continue;
        if (d_out_qt) HIPCHK(ctx, hipMemcpyAsync(d_out_qt + 2 * s.first, s.qt.p, s.count * 8, hipMemcpyDeviceToDevice, SC.stream));
        if (d_out_dist) HIPCHK(ctx, hipMemcpyAsync(d_out_dist + s.first, s.d.p, s.count * 4, hipMemcpyDeviceToDevice, SC.stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(SC.stream));
    return MSFM_OK;
    MSFM_API_END
}

int msfm_memory_info(msfm_ctx* ctx, msfm_memory* out) {
    MSFM_API_BEGIN(ctx)
    if (!ctx || !out) return MSFM_E_INVALID;
    *out = msfm_memory{};
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t free_b = 0, total_b = 0;
    HIPCHK(ctx, hipMemGetInfo(&free_b, &total_b));
    out->device_free = (int64_t)free_b;
    out->device_total = (int64_t)total_b;
    out->store = (int64_t)ctx->store.bytes();
    out->inbox = (int64_t)ctx->inbox.bytes();
    // (buffers that are pieces of the context's pool count once, with the pool)
    auto own = [](const PinnedBuf& h) { return h.pooled ? (size_t)0 : h.cap; };
    size_t pinned = ctx->pinned_pool.cap + ctx->up_ring.cap + own(ctx->h_jobs) + own(ctx->h_store_maxima) + ctx->res_qt.pinned + ctx->res_dist.pinned;
    long long scratch = 0;
    for (Scratch& sc : ctx->sc) {
        scratch += sc.device_bytes();
        for (const PinnedBuf& h : sc.h_up) pinned += own(h);
        pinned += own(sc.h_summary) + own(sc.h_tail) + sc.h_sub_qt.cap + sc.h_sub_d.cap;
    }
    out->scratch = scratch;
    for (const OutSeg& s : ctx->out_segs) out->results_device += (int64_t)(s.qt.cap + s.d.cap);
    out->page_locked_host = (int64_t)pinned;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_read_device(msfm_ctx* ctx, void* host_dst, const void* device_src, int64_t bytes) {
    MSFM_API_BEGIN(ctx)
    if (!ctx || bytes < 0 || (bytes > 0 && (!host_dst || !device_src))) return MSFM_E_INVALID;
    if (bytes == 0) return MSFM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpy(host_dst, device_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return MSFM_OK;
    MSFM_API_END
}

int msfm_fetch_order_certificate(msfm_ctx* ctx, int32_t* out_sensitive_rows) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    if (!ctx->have_results) return fail(ctx, MSFM_E_STATE, "msfm_fetch_order_certificate without a completed msfm_match_pairs");
    if (out_sensitive_rows && !ctx->res_sens.empty()) std::memcpy(out_sensitive_rows, ctx->res_sens.data(), ctx->res_sens.size() * 4);
    return MSFM_OK;
    MSFM_API_END
}

int msfm_view_matches(msfm_ctx* ctx, const int32_t** out_qt, const float** out_dist, int64_t* out_count) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    if (!ctx->have_results) return fail(ctx, MSFM_E_STATE, "msfm_view_matches without a completed msfm_match_pairs");
    if (out_qt) *out_qt = ctx->res_qt.as<int32_t>();
    if (out_dist) *out_dist = ctx->res_dist.as<float>();
    if (out_count) *out_count = (int64_t)ctx->res_count;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_match_pair(msfm_ctx* ctx, int id1, int id2, float ratio, int cross_check, double max_distance,
                    int32_t* out_qt, float* out_dist, int* out_count) {
    MSFM_API_BEGIN(ctx)
    if (!ctx || !out_count) return MSFM_E_INVALID;
    const int32_t pr[2] = {id1, id2};
    msfm_match_params prm = {ratio, cross_check, max_distance};
    int64_t offs[2] = {0, 0};
    int rc = msfm_match_pairs(ctx, pr, 1, &prm, offs);
    if (rc != MSFM_OK) return rc;
    *out_count = (int)offs[1];
    return msfm_fetch_matches(ctx, out_qt, out_dist);
    MSFM_API_END
}

static int knn2_pair_impl(msfm_ctx* ctx, int id1, int id2, int32_t* fwd_idx0, float* fwd_d0, float* fwd_d1,
                          int32_t* rev_idx0, float* rev_d0, float* rev_d1);

int msfm_knn2_pair(msfm_ctx* ctx, int id1, int id2, int32_t* fwd_idx0, float* fwd_d0, float* fwd_d1,
                   int32_t* rev_idx0, float* rev_d0, float* rev_d1) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    // a streaming series left open is abandoned (include/msfm_match.h): what it has in flight is drained first -- this call runs on
    // scratch set 0 and would otherwise overwrite the tail words of a sub-batch the series still has to complete
    const int rc0 = abandon_series(ctx);
    if (rc0 != MSFM_OK) return rc0;
    const int rc = drained(ctx, knn2_pair_impl(ctx, id1, id2, fwd_idx0, fwd_d0, fwd_d1, rev_idx0, rev_d0, rev_d1));
    if (rc != MSFM_OK) ctx->job->open = false;
    return rc;
    MSFM_API_END
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
    MSFM_API_BEGIN(nullptr)
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
    MSFM_API_END
}

int msfm_swap_image_pair(int id1, int id2) { return id1 > id2 ? 1 : 0; }

int msfm_pair_id(int id1, int id2, int32_t* out_pair_id) {
    MSFM_API_BEGIN(nullptr)
    if (!out_pair_id || id1 < 0 || id2 < 0 || id1 >= MSFM_MAX_IMAGES || id2 >= MSFM_MAX_IMAGES) return MSFM_E_INVALID;
    *out_pair_id = msfm_swap_image_pair(id1, id2) ? MSFM_MAX_IMAGES * id2 + id1 : MSFM_MAX_IMAGES * id1 + id2;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_pair_from_id(int32_t pair_id, int* out_id1, int* out_id2) {
    MSFM_API_BEGIN(nullptr)
    if (!out_id1 || !out_id2 || pair_id < 0) return MSFM_E_INVALID;
    *out_id2 = pair_id % MSFM_MAX_IMAGES;
    *out_id1 = (pair_id - *out_id2) / MSFM_MAX_IMAGES;
    return MSFM_OK;
    MSFM_API_END
}

}  // extern "C"
