// msfm_store_host.hip.h -- host side of the descriptor store: uploads into the inbox, finalize_store (classification + one allocation
// + table-driven build kernels for everything uploaded since the last use), the forms derived on demand, and the store's C ABI entry
// points.  Kernels and the residency table: msfm_store.hip.h.  Included by msfm_match.hip only.
//
// Replaces the per-pair Database::ReadDescriptors of the reference (src/Feature/FeatureMatching.cpp:32-33) and, since round 5, the
// build-inside-upload of rounds 1-4.  What the pieces cost on this part (profiles/r05_ubench_upload.txt, r05_ubench_malloc.txt):
// a pageable hipMemcpyAsync followed by a synchronisation 8 GB/s the first time a buffer is seen, memcpy into page-locked memory
// 32 GB/s on one core, page-locked -> device 38-46 GB/s, kernel + synchronisation 12 us (27 us with a pageable read-back), hipMalloc
// 1-7 us for a few MB but with 20-70 ms outliers when the runtime has to grow its heap, hipFree 90 us, hipHostMalloc 0.2 ms per MiB.
// Hence: a small page-locked ring (two slots), no synchronisation per image, one allocation per finalize_store call, nothing freed
// one array at a time.
#pragma once

namespace {

constexpr size_t kUpSlotBytes = (size_t)4 << 20;     // one ring slot: an image of up to 8192 float rows travels in one piece
constexpr int kUpSlots = 2;                          // memcpy into one slot while the other one's DMA runs
constexpr size_t kInboxWaveBytes = (size_t)256 << 20;   // uploads waiting beyond this are built before the next one is accepted
constexpr size_t kInboxMinChunk = (size_t)64 << 20;

inline hipStream_t store_stream(msfm_ctx* ctx) { return ctx->sc[0].stream; }

int finalize_store(msfm_ctx* ctx);
int settle_store(msfm_ctx* ctx);

void free_image(msfm_ctx* ctx, Image& im) {
    ctx->store.drop(im.chunk_core);
    ctx->store.drop(im.chunk_wide);
    ctx->store.drop(im.chunk_panel);
    ctx->store.drop(im.chunk_kp);
    ctx->inbox.drop(im.inbox_chunk);
    im = Image{};
}

// `bytes` produced piecewise by fill(dst, offset, piece_bytes) into the page-locked ring, each piece sent to dev + offset on the store's
// stream.  When this returns the producer's source is no longer needed; the copies may still be in flight (same stream as every build
// kernel, so nothing reads `dev` before they land).
template <class Fill>
int stage_h2d(msfm_ctx* ctx, void* dev, size_t bytes, Fill fill) {
    if (!ctx->up_ring.p) {
        // (exactly the ring: page-locked memory costs 0.2 ms per MiB to get)
        HIPCHK(ctx, hipHostMalloc(&ctx->up_ring.p, kUpSlots * kUpSlotBytes, hipHostMallocDefault));
        ctx->up_ring.cap = kUpSlots * kUpSlotBytes;
        for (int s = 0; s < kUpSlots; ++s) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->up_ev[s], hipEventDisableTiming));
    }
    hipStream_t st = store_stream(ctx);
    for (size_t off = 0; off < bytes; off += kUpSlotBytes) {
        const size_t piece = std::min(kUpSlotBytes, bytes - off);
        const int s = (int)(ctx->up_seq++ % kUpSlots);
        if (ctx->up_ev_recorded[s]) HIPCHK(ctx, hipEventSynchronize(ctx->up_ev[s]));
        char* slot = ctx->up_ring.as<char>() + (size_t)s * kUpSlotBytes;
        fill(slot, off, piece);
        HIPCHK(ctx, hipMemcpyAsync(static_cast<char*>(dev) + off, slot, piece, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipEventRecord(ctx->up_ev[s], st));
        ctx->up_ev_recorded[s] = true;
    }
    ctx->store_async = true;
    return MSFM_OK;
}

// before a matching call launches on its other streams: whatever the store's stream still carries (keypoint copies, gathers) has landed
int settle_store(msfm_ctx* ctx) {
    int rc = finalize_store(ctx);
    if (rc != MSFM_OK) return rc;
    if (ctx->store_async) {
        HIPCHK(ctx, hipStreamSynchronize(store_stream(ctx)));
        ctx->store_async = false;
    }
    return MSFM_OK;
}

// A new image `image_id` of n rows whose row-major rows (kind: kSrcF32 / kSrcU8) will be written to *inbox_ptr by the caller (plus
// `extra` bytes behind them, for the caller's own use).  The image is pending until finalize_store.
int begin_pending(msfm_ctx* ctx, int image_id, int n, int kind, bool no_twin, size_t extra, void** inbox_ptr) {
    Image& im = ctx->images[(size_t)image_id];
    free_image(ctx, im);
    im.n = n;
    im.nblk = (n + kBM - 1) / kBM;
    im.nalloc = (im.nblk + kPfWgRows / kBM - 1) / (kPfWgRows / kBM) * (kPfWgRows / kBM);
    *inbox_ptr = nullptr;
    if (n == 0) return MSFM_OK;   // (an empty image: nothing to build, no pair of it does device work)
    const size_t bytes = (((size_t)n * kDim * (kind == kSrcU8 ? 1 : 4) + 255) & ~(size_t)255) + extra;
    // a long series of uploads is built in waves: the inbox stays small (allocation cost, memory), the build kernels of a wave run while
    // the host copies the next one
    if (!ctx->pending.empty() && ctx->inbox_waiting + bytes > kInboxWaveBytes) {
        const int rc = finalize_store(ctx);
        if (rc != MSFM_OK) return rc;
    }
    ctx->inbox_waiting += bytes;
    HIPCHK(ctx, ctx->inbox.reserve(bytes, kInboxMinChunk));
    void* p = ctx->inbox.take(bytes, &im.inbox_chunk);
    if (!p) return fail(ctx, MSFM_E_DEVICE, "inbox allocation failed");
    im.pending = true;
    im.inbox = p;
    im.inbox_kind = kind;
    im.no_twin = no_twin;
    ctx->pending.push_back(image_id);
    *inbox_ptr = p;
    return MSFM_OK;
}

// job table -> device (page-locked staging of its own: the table is a few KB)
int upload_jobs(msfm_ctx* ctx, const std::vector<StoreJob>& jobs) {
    const size_t bytes = jobs.size() * sizeof(StoreJob);
    // the staging is rewritten by the HOST: the previous table's copy must have left it (the device-side table is safe by stream
    // order: this copy runs behind the kernels that read the previous one)
    if (!ctx->jobs_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->jobs_ev, hipEventDisableTiming));
    else HIPCHK(ctx, hipEventSynchronize(ctx->jobs_ev));
    if (bytes > ctx->d_jobs.cap) {   // (growing frees the old table: nothing may still read it)
        HIPCHK(ctx, hipStreamSynchronize(store_stream(ctx)));
        HIPCHK(ctx, ctx->d_jobs.ensure(bytes));
    }
    HIPCHK(ctx, ctx->h_jobs.ensure(bytes, 0));
    std::memcpy(ctx->h_jobs.p, jobs.data(), bytes);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_jobs.p, ctx->h_jobs.p, bytes, hipMemcpyHostToDevice, store_stream(ctx)));
    HIPCHK(ctx, hipEventRecord(ctx->jobs_ev, store_stream(ctx)));
    return MSFM_OK;
}

inline unsigned store_grid_x(int max_rows) { return (unsigned)std::max(1, std::min(64, (max_rows + 15) / 16)); }

// per-job maxima (16 words each: [0..7] classification, [8..15] the twin's) -> host
int read_maxima(msfm_ctx* ctx, size_t n_jobs) {
    HIPCHK(ctx, ctx->h_store_maxima.ensure(n_jobs * 64, 0));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_store_maxima.p, ctx->d_store_maxima.p, n_jobs * 64, hipMemcpyDeviceToHost, store_stream(ctx)));
    HIPCHK(ctx, hipStreamSynchronize(store_stream(ctx)));
    return MSFM_OK;
}

inline float bits_to_float(unsigned u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// centre of the digit k-step from the range of 2h (maxima words [2] = max, [3] = ~min): false when the rows' norms spread beyond what
// sixteen digits represent around any centre (msfm_sweep_i8.hip.h)
inline bool digit_centre(unsigned max_bits, unsigned inv_min_bits, int* h0) {
    const float nmax = bits_to_float(max_bits), nmin = bits_to_float(~inv_min_bits);
    const long long hmax = (long long)(0.5f * nmax), hmin = (long long)(0.5f * nmin);
    const long long c = (hmin + hmax) / 2;
    if (!(hmin <= hmax && c - hmax >= kI8DigitLo && c - hmin <= kI8DigitHi)) return false;
    *h0 = (int)c;
    return true;
}

// twins (msfm_q8.hip.h) of `twins` (image ids) from src_kind / their source, into their existing q8 / nrm_q8 / err_q8 arrays, under the
// context's current level: quantise + rows + error norms, read the norm ranges back, digits.  One synchronisation.
int build_twins(msfm_ctx* ctx, const std::vector<int>& twins, bool from_inbox) {
    if (twins.empty()) return MSFM_OK;
    hipStream_t st = store_stream(ctx);
    const size_t T = twins.size();
    HIPCHK(ctx, ctx->d_store_maxima.ensure(T * 64));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_store_maxima.p, 0, T * 64, st));
    std::vector<StoreJob> jobs(T);
    int max_rows = 1;
    const float scale = 255.f / ctx->q8_level, inv = ctx->q8_level / 255.f;
    for (size_t j = 0; j < T; ++j) {
        Image& im = ctx->images[(size_t)twins[j]];
        StoreJob& J = jobs[j];
        J = StoreJob{};
        J.src = from_inbox ? im.inbox : (const void*)im.rawp;
        J.src_kind = from_inbox ? im.inbox_kind : kSrcRawp;
        J.n = im.n;
        J.npad = im.nalloc * kBM;
        J.nalloc = im.nalloc;
        J.i8 = im.q8;
        J.nrm_i8 = im.nrm_q8;
        J.err = im.err_q8;
        J.scale = scale;
        J.inv = inv;
        J.maxima = ctx->d_store_maxima.as<unsigned>() + 16 * j + 8;
        max_rows = std::max(max_rows, J.npad);
        im.q8_level = ctx->q8_level;
    }
    int rc = upload_jobs(ctx, jobs);
    if (rc != MSFM_OK) return rc;
    hipLaunchKernelGGL(st_i8_kernel, dim3(store_grid_x(max_rows), (unsigned)T), dim3(256), 0, st, (const StoreJob*)ctx->d_jobs.as<StoreJob>());
    HIPCHK(ctx, hipGetLastError());
    rc = read_maxima(ctx, T);
    if (rc != MSFM_OK) return rc;
    bool any = false;
    for (size_t j = 0; j < T; ++j) {
        Image& im = ctx->images[(size_t)twins[j]];
        const unsigned* mx = ctx->h_store_maxima.as<unsigned>() + 16 * j + 8;
        StoreJob& J = jobs[j];
        J.fuse_digits = 0;
        if (digit_centre(mx[2], mx[3], &im.h0_q8)) {
            im.err_q8_max = bits_to_float(mx[5]);
            J.h0 = im.h0_q8;
            J.fuse_digits = 1;
            any = true;
        } else {   // no twin (its memory stays with the image's chunk)
            im.q8 = nullptr;
            im.nrm_q8 = nullptr;
            im.err_q8 = nullptr;
        }
    }
    if (any) {
        rc = upload_jobs(ctx, jobs);
        if (rc != MSFM_OK) return rc;
        hipLaunchKernelGGL(st_digits_kernel, dim3(store_grid_x(max_rows / 16 + 1), (unsigned)T), dim3(256), 0, st, (const StoreJob*)ctx->d_jobs.as<StoreJob>());
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));   // (the job table's staging is rewritten by the next caller)
    }
    return MSFM_OK;
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// Everything uploaded since the last call becomes a built image: classification, decisions, ONE allocation, the build kernels.
int finalize_store(msfm_ctx* ctx) {
    if (ctx->pending.empty()) return MSFM_OK;
    HostClock hc;   // MSFM_DEBUG_TIMING=1
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = store_stream(ctx);
    std::vector<int> ids;
    {
        std::vector<int> p = ctx->pending;
        std::sort(p.begin(), p.end());
        p.erase(std::unique(p.begin(), p.end()), p.end());
        for (int id : p)
            if (ctx->images[(size_t)id].pending) ids.push_back(id);
    }
    // The images leave the pending list only when the build has succeeded: a build that fails (out of device memory in store.reserve,
    // reported by THIS call) leaves them pending with their inbox rows, so a retry -- after the caller freed memory -- builds them
    // instead of returning MSFM_OK with nothing built.
    struct RestorePending {
        msfm_ctx* ctx;
        const std::vector<int>* ids;
        size_t waiting;
        bool done = false;
        ~RestorePending() {
            if (done) return;
            for (int id : *ids)
                if (ctx->images[(size_t)id].pending) ctx->pending.push_back(id);
            ctx->inbox_waiting += waiting;
        }
    } restore{ctx, &ids, ctx->inbox_waiting};
    ctx->pending.clear();
    ctx->inbox_waiting = 0;
    if (ids.empty()) {
        restore.done = true;
        return MSFM_OK;
    }
    const size_t P = ids.size();
    HIPCHK(ctx, ctx->d_store_maxima.ensure(P * 64));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_store_maxima.p, 0, P * 64, st));
    std::vector<StoreJob> jobs(P);
    int max_rows = 1;
    for (size_t j = 0; j < P; ++j) {
        const Image& im = ctx->images[(size_t)ids[j]];
        StoreJob& J = jobs[j];
        J = StoreJob{};
        J.src = im.inbox;
        J.src_kind = im.inbox_kind;
        J.n = im.n;
        J.npad = im.nalloc * kBM;
        J.nalloc = im.nalloc;
        J.maxima = ctx->d_store_maxima.as<unsigned>() + 16 * j;
        max_rows = std::max(max_rows, J.npad);
    }
    int rc = upload_jobs(ctx, jobs);
    if (rc != MSFM_OK) return rc;
    hipLaunchKernelGGL(st_classify_kernel, dim3(store_grid_x(max_rows), (unsigned)P), dim3(256), 0, st, (const StoreJob*)ctx->d_jobs.as<StoreJob>());
    HIPCHK(ctx, hipGetLastError());
    rc = read_maxima(ctx, P);
    if (rc != MSFM_OK) return rc;
    hc.lap("store: classify + read-back");

    // ---- decisions per image, sizes
    std::vector<char> want_float(P, 0), want_twin(P, 0);
    size_t total = 0;
    for (size_t j = 0; j < P; ++j) {
        Image& im = ctx->images[(size_t)ids[j]];
        const unsigned* mx = ctx->h_store_maxima.as<unsigned>() + 16 * j;
        im.nrm_max = bits_to_float(mx[0]);
        im.abs_max = bits_to_float(mx[1]);
        // A FLOAT upload whose every value is an integer in [0, 255] (raw OpenCV SIFT stored as CV_32F, the reference's
        // Database::WriteDescriptors format before RootSIFT) is the same store as a byte upload: every partial sum of (a - b)^2 stays
        // below 2^24, S is an exact integer under any accumulation order, the image rides the integer matrix cores.
        const bool bytes = im.inbox_kind == kSrcU8 || (ctx->byte_detect && mx[6] == 0);
        im.from_u8 = bytes;
        im.is_u8 = false;
        if (bytes) {
            im.nrm_i8_max = bits_to_float(mx[2]);
            // (an all-zero next to an all-128 descriptor -- not SIFT -- has no centre: the image is served by the fp16 kernels)
            im.is_u8 = digit_centre(mx[2], mx[3], &im.h0_i8);
        }
        im.pf_safe = (im.abs_max <= kF16Safe) && (im.nrm_max < 3.0e38f);   // NaN / inf compare false
        im.c = 1.f;
        if (im.pf_safe) {
            // c = 2^k with max|row|^2 / 2 / c in (2^11, 2^12]; k must keep c an exact fp16 value
            int e = 0;
            (void)std::frexp(0.5f * im.nrm_max, &e);   // 0.5 nrm_max = m * 2^e, m in [0.5, 1)
            int k = (im.nrm_max > 0.f ? e : -24) - 12;
            if (k < -24) k = -24;
            if (k > 15) im.pf_safe = false;
            else im.c = std::ldexp(1.f, k);
        }
        want_twin[j] = (!im.no_twin && ctx->q8_route && mx[4] == 0) ? 1 : 0;   // every value in [0, 1] (a float image of 0 / 1 entries is both)
        if (want_twin[j])
            ctx->q8_level = std::max(ctx->q8_level, std::max(kQ8LevelStep, std::ceil(im.abs_max / kQ8LevelStep) * kQ8LevelStep));
        want_float[j] = (!im.is_u8 || want_twin[j]) ? 1 : 0;
        const size_t npad = (size_t)im.nalloc * kBM, n = (size_t)im.n;
        if (want_float[j]) total += al256(n * kDim * 4) + al256(npad * kPfRowBytes) + al256(npad * 4);
        if (im.is_u8) total += al256(npad * kI8RowBytes) + 2 * al256(npad * 4);
        if (want_twin[j]) total += al256(npad * kI8RowBytes) + al256(npad * 4) + al256(std::max<size_t>(n, 1) * 4);
    }
    HIPCHK(ctx, ctx->store.reserve(total, 0));
    hc.lap("store: decisions + allocation");
    std::vector<int> twins;
    for (size_t j = 0; j < P; ++j) {
        Image& im = ctx->images[(size_t)ids[j]];
        StoreJob& J = jobs[j];
        const size_t npad = (size_t)im.nalloc * kBM, n = (size_t)im.n;
        int chunk = -1;
        auto take = [&](size_t bytes) -> void* {
            int c = -1;
            void* p = ctx->store.take(bytes, &c);
            if (chunk >= 0) ctx->store.chunks[(size_t)c].live -= 1;   // one reference per image, not per array
            chunk = c;
            return p;
        };
        if (want_float[j]) {
            im.rawp = static_cast<float*>(take(n * kDim * 4));
            im.h16 = static_cast<_Float16*>(take(npad * kPfRowBytes));
            im.nrm = static_cast<float*>(take(npad * 4));
            J.rawp = im.rawp;
            J.h16 = im.h16;
            J.nrm = im.nrm;
            J.c = im.pf_safe ? im.c : 0.f;
        }
        if (im.is_u8) {
            im.i8 = static_cast<signed char*>(take(npad * kI8RowBytes));
            im.nrm_i8 = static_cast<float*>(take(npad * 4));
            im.n2_i8 = static_cast<int*>(take(npad * 4));
            J.i8 = im.i8;
            J.nrm_i8 = im.nrm_i8;
            J.n2 = im.n2_i8;
            J.h0 = im.h0_i8;
            J.fuse_digits = 1;
        }
        if (want_twin[j]) {
            im.q8 = static_cast<signed char*>(take(npad * kI8RowBytes));
            im.nrm_q8 = static_cast<float*>(take(npad * 4));
            im.err_q8 = static_cast<float*>(take(std::max<size_t>(n, 1) * 4));
            twins.push_back(ids[j]);
        }
        im.chunk_core = chunk;
    }
    ctx->store_peak_bytes = std::max(ctx->store_peak_bytes, ctx->store.bytes());
    rc = upload_jobs(ctx, jobs);
    if (rc != MSFM_OK) return rc;
    const dim3 grid(store_grid_x(max_rows), (unsigned)P);
    hipLaunchKernelGGL(st_float_kernel, grid, dim3(256), 0, st, (const StoreJob*)ctx->d_jobs.as<StoreJob>());
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(st_i8_kernel, grid, dim3(256), 0, st, (const StoreJob*)ctx->d_jobs.as<StoreJob>());
    HIPCHK(ctx, hipGetLastError());
    if (!twins.empty()) {
        rc = build_twins(ctx, twins, true);
        if (rc != MSFM_OK) return rc;
    } else {
        HIPCHK(ctx, hipStreamSynchronize(st));   // (the inbox is handed back below; the job table's staging is rewritten by the next caller)
    }
    hc.lap("store: build kernels + twins");
    for (size_t j = 0; j < P; ++j) {
        Image& im = ctx->images[(size_t)ids[j]];
        im.pending = false;
        im.inbox = nullptr;
        ctx->inbox.drop(im.inbox_chunk);
    }
    hc.lap("store: inbox handed back");
    restore.done = true;
    return MSFM_OK;
}

// twins built before a later upload raised the context's level (msfm_q8.hip.h): rebuilt from the resident fp32 rows
int rebuild_stale_twins(msfm_ctx* ctx, const int32_t* ids, int n_ids) {
    if (!ctx->q8_route || ctx->prefilter != 1) return MSFM_OK;
    std::vector<int> stale;
    for (int k = 0; k < n_ids; ++k) {
        const int id = ids[k];
        if (id < 0 || id >= kSlots) continue;
        Image& im = ctx->images[(size_t)id];
        if (im.q8 && im.q8_level != ctx->q8_level) {
            im.q8_level = ctx->q8_level;   // (listed once)
            stale.push_back(id);
        }
    }
    return build_twins(ctx, stale, false);
}

// The float forms of byte images (`wide`: a pair with a float image, the kNN-level API, ratio > 0.95, the fp16-only and brute-force
// routes) and the panels of the brute-force kernel are derived from what is resident the first time a batch needs them.
int ensure_forms(msfm_ctx* ctx, const std::vector<int>& wide_ids, const std::vector<int>& panel_ids) {
    hipStream_t st = store_stream(ctx);
    std::vector<StoreJob> jobs;
    std::vector<int> w;
    size_t total = 0;
    int max_rows = 1;
    for (int id : wide_ids) {
        Image& im = ctx->images[(size_t)id];
        if (im.n <= 0 || im.rawp || !im.i8) continue;
        if (std::find(w.begin(), w.end(), id) != w.end()) continue;
        w.push_back(id);
        const size_t npad = (size_t)im.nalloc * kBM;
        total += al256((size_t)im.n * kDim * 4) + al256(npad * kPfRowBytes) + al256(npad * 4);
    }
    if (!w.empty()) {
        HIPCHK(ctx, ctx->store.reserve(total, 0));
        for (int id : w) {
            Image& im = ctx->images[(size_t)id];
            const size_t npad = (size_t)im.nalloc * kBM;
            int c0 = -1, c1 = -1, c2 = -1;
            im.rawp = static_cast<float*>(ctx->store.take((size_t)im.n * kDim * 4, &c0));
            im.h16 = static_cast<_Float16*>(ctx->store.take(npad * kPfRowBytes, &c1));
            im.nrm = static_cast<float*>(ctx->store.take(npad * 4, &c2));
            ctx->store.chunks[(size_t)c0].live -= 2;
            im.chunk_wide = c0;
            StoreJob J = {};
            J.src = im.i8;
            J.src_kind = kSrcI8Rows;
            J.n = im.n;
            J.npad = (int)npad;
            J.nalloc = im.nalloc;
            J.rawp = im.rawp;
            J.h16 = im.h16;
            J.nrm = im.nrm;
            J.c = im.pf_safe ? im.c : 0.f;
            jobs.push_back(J);
            max_rows = std::max(max_rows, J.npad);
        }
        int rc = upload_jobs(ctx, jobs);
        if (rc != MSFM_OK) return rc;
        hipLaunchKernelGGL(st_float_kernel, dim3(store_grid_x(max_rows), (unsigned)jobs.size()), dim3(256), 0, st,
                           (const StoreJob*)ctx->d_jobs.as<StoreJob>());
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
        ctx->store_peak_bytes = std::max(ctx->store_peak_bytes, ctx->store.bytes());
    }
    jobs.clear();
    std::vector<int> pn;
    total = 0;
    max_rows = 1;
    for (int id : panel_ids) {
        Image& im = ctx->images[(size_t)id];
        if (im.n <= 0 || (im.panel && im.panel_order == ctx->order)) continue;
        if (std::find(pn.begin(), pn.end(), id) != pn.end()) continue;
        pn.push_back(id);
        if (!im.panel) total += al256((size_t)im.nalloc * kPanelFloats * 4);
    }
    if (!pn.empty()) {
        if (total) HIPCHK(ctx, ctx->store.reserve(total, 0));
        for (int id : pn) {
            Image& im = ctx->images[(size_t)id];
            if (!im.panel) im.panel = static_cast<float*>(ctx->store.take((size_t)im.nalloc * kPanelFloats * 4, &im.chunk_panel));
            im.panel_order = ctx->order;
            StoreJob J = {};
            J.src = im.rawp ? (const void*)im.rawp : (const void*)im.i8;
            J.src_kind = im.rawp ? kSrcRawp : kSrcI8Rows;
            J.n = im.n;
            J.npad = im.nalloc * kBM;
            J.nalloc = im.nalloc;
            J.panel = im.panel;
            jobs.push_back(J);
            max_rows = std::max(max_rows, J.npad);
        }
        int rc = upload_jobs(ctx, jobs);
        if (rc != MSFM_OK) return rc;
        const dim3 grid((unsigned)std::min(256, std::max(1, max_rows / 16)), (unsigned)jobs.size());
        const StoreJob* dj = ctx->d_jobs.as<StoreJob>();
        if (ctx->order == MSFM_ORDER_SSE4X4) hipLaunchKernelGGL(st_panel_kernel<0>, grid, dim3(256), 0, st, dj);
        else if (ctx->order == MSFM_ORDER_AVX2_FMA) hipLaunchKernelGGL(st_panel_kernel<1>, grid, dim3(256), 0, st, dj);
        else hipLaunchKernelGGL(st_panel_kernel<3>, grid, dim3(256), 0, st, dj);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
        ctx->store_peak_bytes = std::max(ctx->store_peak_bytes, ctx->store.bytes());
    }
    return MSFM_OK;
}

}  // namespace

// =========================================================================================
// C ABI of the store
// =========================================================================================
extern "C" {

int msfm_upload_image(msfm_ctx* ctx, int image_id, const void* desc, int n, int dim, int dtype) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    if (ctx->series_open) return fail(ctx, MSFM_E_STATE, "the store cannot change while a streaming series (msfm_match_pairs_begin .. _next) is open");
    if (image_id < 0 || image_id >= kSlots) return fail(ctx, MSFM_E_INVALID, "image id out of range");
    if (n < 0 || dim != MSFM_DIM) return fail(ctx, MSFM_E_INVALID, "descriptors must be n x 128");
    if (n >= (1 << 18)) return fail(ctx, MSFM_E_INVALID, "more than 2^18 - 1 rows (BFMatcher packs the train index in 18 bits)");
    if (dtype != MSFM_DTYPE_F32 && dtype != MSFM_DTYPE_U8) return fail(ctx, MSFM_E_INVALID, "dtype must be F32 or U8");
    if (n > 0 && !desc) return fail(ctx, MSFM_E_INVALID, "null descriptor pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void* inbox = nullptr;
    const int kind = dtype == MSFM_DTYPE_U8 ? kSrcU8 : kSrcF32;
    int rc = begin_pending(ctx, image_id, n, kind, dtype == MSFM_DTYPE_U8, 0, &inbox);
    if (rc != MSFM_OK || n == 0) return rc;
    const char* src = static_cast<const char*>(desc);
    return stage_h2d(ctx, inbox, (size_t)n * kDim * (kind == kSrcU8 ? 1 : 4), [src, ctx](char* dst, size_t off, size_t piece) {
        if (ctx->copy_helper) ctx->copier.copy(dst, src + off, piece);
        else std::memcpy(dst, src + off, piece);
    });
    MSFM_API_END
}

int msfm_finalize_store(msfm_ctx* ctx) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    return finalize_store(ctx);
    MSFM_API_END
}

int msfm_store_info(const msfm_ctx* ctx, int64_t* out_device_bytes, int64_t* out_rows, int64_t* out_pending_images) {
    MSFM_API_BEGIN(nullptr)
    if (!ctx) return MSFM_E_INVALID;
    long long rows = 0, pend = 0;
    for (const Image& im : ctx->images) {
        if (im.n > 0) rows += im.n;
        if (im.pending) pend += 1;
    }
    if (out_device_bytes) *out_device_bytes = (int64_t)ctx->store.bytes();
    if (out_rows) *out_rows = rows;
    if (out_pending_images) *out_pending_images = pend;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_subset_image(msfm_ctx* ctx, int src_image_id, int dst_image_id, const int32_t* rows, int count) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    if (ctx->series_open) return fail(ctx, MSFM_E_STATE, "the store cannot change while a streaming series (msfm_match_pairs_begin .. _next) is open");
    if (src_image_id < 0 || src_image_id >= kSlots || dst_image_id < 0 || dst_image_id >= kSlots || src_image_id == dst_image_id)
        return fail(ctx, MSFM_E_INVALID, "bad image ids for msfm_subset_image");
    if (count < 0 || (count > 0 && !rows)) return fail(ctx, MSFM_E_INVALID, "bad row list");
    if (ctx->images[(size_t)src_image_id].n < 0) return fail(ctx, MSFM_E_NOIMAGE, "image not uploaded: " + std::to_string(src_image_id));
    for (int i = 0; i < count; ++i)
        if (rows[i] < 0 || rows[i] >= ctx->images[(size_t)src_image_id].n) return fail(ctx, MSFM_E_INVALID, "row index out of range in msfm_subset_image");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->images[(size_t)src_image_id].pending) {   // the rows are gathered from the BUILT source
        const int rc = finalize_store(ctx);
        if (rc != MSFM_OK) return rc;
    }
    const Image src = ctx->images[(size_t)src_image_id];   // (a copy: begin_pending may move nothing, but dst is another slot anyway)
    const bool as_u8 = src.is_u8;                          // rows of a byte image are bytes
    void* inbox = nullptr;
    const size_t row_bytes = ((size_t)count * kDim * (as_u8 ? 1 : 4) + 255) & ~(size_t)255;
    int rc = begin_pending(ctx, dst_image_id, count, as_u8 ? kSrcU8 : kSrcF32, src.from_u8, (size_t)count * 4 + 256, &inbox);
    if (rc != MSFM_OK || count == 0) return rc;
    int* d_idx = reinterpret_cast<int*>(static_cast<char*>(inbox) + row_bytes);
    rc = stage_h2d(ctx, d_idx, (size_t)count * 4, [rows](char* dst, size_t off, size_t piece) { std::memcpy(dst, reinterpret_cast<const char*>(rows) + off, piece); });
    if (rc != MSFM_OK) return rc;
    StoreJob J = {};
    J.src = src.rawp ? (const void*)src.rawp : (const void*)src.i8;
    J.src_kind = src.rawp ? kSrcRawp : kSrcI8Rows;
    J.n = src.n;
    hipLaunchKernelGGL(st_gather_rows_kernel, dim3((unsigned)std::min(1024, (count * kDim + 255) / 256)), dim3(256), 0, store_stream(ctx), J,
                       (const int*)d_idx, inbox, count, as_u8 ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    return MSFM_OK;
    MSFM_API_END
}

int msfm_image_rows(const msfm_ctx* ctx, int image_id, int* out_n) {
    MSFM_API_BEGIN(nullptr)
    if (!ctx || !out_n) return MSFM_E_INVALID;
    if (image_id < 0 || image_id >= kSlots) return MSFM_E_INVALID;
    if (ctx->images[(size_t)image_id].n < 0) return MSFM_E_NOIMAGE;
    *out_n = ctx->images[(size_t)image_id].n;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_clear_images(msfm_ctx* ctx) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    if (ctx->series_open) return fail(ctx, MSFM_E_STATE, "the store cannot change while a streaming series (msfm_match_pairs_begin .. _next) is open");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (Scratch& sc : ctx->sc)
        if (sc.stream) HIPCHK(ctx, hipStreamSynchronize(sc.stream));
    for (auto& im : ctx->images) im = Image{};
    ctx->pending.clear();
    ctx->inbox_waiting = 0;
    ctx->store.release_all();
    ctx->inbox.release_all();
    ctx->q8_level = 0.f;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_upload_keypoints(msfm_ctx* ctx, int image_id, const float* kpts, int n, int stride_floats) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    if (ctx->series_open) return fail(ctx, MSFM_E_STATE, "the store cannot change while a streaming series (msfm_match_pairs_begin .. _next) is open");
    if (image_id < 0 || image_id >= kSlots) return fail(ctx, MSFM_E_INVALID, "image id out of range");
    if (n < 0 || (n > 0 && !kpts) || stride_floats < 2) return fail(ctx, MSFM_E_INVALID, "bad keypoint array");
    Image& im = ctx->images[(size_t)image_id];
    if (im.n < 0) return fail(ctx, MSFM_E_NOIMAGE, "msfm_upload_keypoints before msfm_upload_image for image " + std::to_string(image_id));
    if (n < im.n) return fail(ctx, MSFM_E_INVALID, "fewer keypoints than descriptor rows");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->store.drop(im.chunk_kp);
    im.kxy = nullptr;
    im.nk = -1;
    const size_t bytes = (size_t)std::max(n, 1) * sizeof(float2);
    HIPCHK(ctx, ctx->store.reserve(bytes, (size_t)1 << 20));
    im.kxy = static_cast<float2*>(ctx->store.take(bytes, &im.chunk_kp));
    if (!im.kxy) return fail(ctx, MSFM_E_DEVICE, "keypoint allocation failed");
    const int rc = stage_h2d(ctx, im.kxy, (size_t)n * sizeof(float2), [kpts, stride_floats](char* dst, size_t off, size_t piece) {
        float2* o = reinterpret_cast<float2*>(dst);
        const size_t i0 = off / sizeof(float2), cnt = piece / sizeof(float2);
        for (size_t i = 0; i < cnt; ++i) o[i] = make_float2(kpts[(i0 + i) * (size_t)stride_floats], kpts[(i0 + i) * (size_t)stride_floats + 1]);
    });
    if (rc != MSFM_OK) return rc;
    im.nk = n;
    return MSFM_OK;
    MSFM_API_END
}

int msfm_set_accum_order(msfm_ctx* ctx, int order) {
    MSFM_API_BEGIN(ctx)
    if (!ctx) return MSFM_E_INVALID;
    if (order != MSFM_ORDER_SSE4X4 && order != MSFM_ORDER_AVX2_FMA && order != MSFM_ORDER_AVX512_FMA)
        return fail(ctx, MSFM_E_INVALID, "unknown accumulation order");
    // (the panels store dimensions in accumulation order: those of another order are re-laid the next time the brute-force route
    // asks for them -- ensure_forms; nothing else in the store depends on the order)
    ctx->order = order;
    return MSFM_OK;
    MSFM_API_END
}

}  // extern "C"
