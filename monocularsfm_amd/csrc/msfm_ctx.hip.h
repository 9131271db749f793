// msfm_ctx.hip.h -- host-side state of the C ABI (include/msfm_match.h): device / page-locked buffers, the descriptor store's
// images, the scratch sets of the sub-batches in flight, the context.  Included by msfm_match.hip only.
#pragma once
namespace {

// MSFM_DEBUG_TIMING=1: what the allocations of a call cost the host (printed at the end of every matching call)
struct AllocClock {
    double dev_ms = 0, pin_ms = 0, free_ms = 0;
    long long dev_bytes = 0, pin_bytes = 0;
    int dev_n = 0, pin_n = 0, free_n = 0;
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
// (per THREAD: a context is driven by one thread at a time and host/FeatureMatching.cpp runs one thread per context -- a process-wide
// clock was a data race between them and mixed their figures)
inline AllocClock& alloc_clock() {
    static thread_local AllocClock c;
    return c;
}
inline hipError_t timed_malloc(void** p, size_t bytes) {
    const double t0 = AllocClock::now();
    const hipError_t e = hipMalloc(p, bytes);
    AllocClock& c = alloc_clock();
    c.dev_ms += AllocClock::now() - t0;
    c.dev_bytes += (long long)bytes;
    c.dev_n += 1;
    return e;
}
inline void timed_free(void* p) {
    const double t0 = AllocClock::now();
    (void)hipFree(p);
    AllocClock& c = alloc_clock();
    c.free_ms += AllocClock::now() - t0;
    c.free_n += 1;
}
inline hipError_t timed_host_malloc(void** p, size_t bytes) {
    const double t0 = AllocClock::now();
    const hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    AllocClock& c = alloc_clock();
    c.pin_ms += AllocClock::now() - t0;
    c.pin_bytes += (long long)bytes;
    c.pin_n += 1;
    return e;
}

// Device buffers REPLACED by larger ones while a matching call has sub-batches in flight wait here for a moment at which the call has
// drained its streams anyway (a re-run, the end of the call): hipFree waits for the whole device, and a sub-batch that replaces buffers
// while its predecessor's sweep 1 is running -- the first call of a job in a process: the scratch set still has the sizes of the call
// before -- stalled its own launches behind that sweep (11 ms of the ComputeMatches executable's matching call, profiles/r05_cli_cold_call.txt).
// Kernels and copies queued earlier may still read the old buffer: it stays valid until the flush.
struct DeferredFrees {
    std::vector<void*> dead, dead_host;   // device memory; page-locked host memory (hipHostFree waits for the device as well)
    bool empty() const { return dead.empty() && dead_host.empty(); }
    void flush() {   // (the caller has synchronised every stream that could touch them)
        for (void* q : dead) timed_free(q);
        for (void* q : dead_host) (void)hipHostFree(q);
        dead.clear();
        dead_host.clear();
    }
};
inline thread_local DeferredFrees* t_deferred_frees = nullptr;   // set by MatchJob while it launches (a context is driven by one thread at a time)
struct DeferFreesScope {
    DeferredFrees* prev;
    explicit DeferFreesScope(DeferredFrees* d) : prev(t_deferred_frees) { t_deferred_frees = d; }
    ~DeferFreesScope() { t_deferred_frees = prev; }
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p && std::getenv("MSFM_DEBUG_TIMING")) std::fprintf(stderr, "[msfm alloc] regrow %p: %zu -> %zu bytes\n", (void*)this, cap, bytes);
        if (p && t_deferred_frees) {
            t_deferred_frees->dead.push_back(p);
        } else if (p) {
            // kernels queued earlier may read the old buffer: wait for the device EXPLICITLY (rounds 3 / 4 leaned on hipFree doing so
            // implicitly).  Rare by construction: buffers are kept while they hold the prediction + 1/8 (msfm_batch.hip.h), the
            // result lists never re-grow (GrowPinned, OutSeg).
            (void)hipDeviceSynchronize();
            timed_free(p);
        }
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = timed_malloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    // grow and keep the first `keep` bytes (the match lists of a call accumulate over its sub-batches)
    hipError_t ensure_keep(size_t bytes, size_t keep, hipStream_t stream, size_t hint = 0) {
        if (bytes <= cap) return hipSuccess;
        const size_t want = std::max(bytes + bytes / 2 + 4096, hint);
        void* q = nullptr;
        hipError_t e = timed_malloc(&q, want);
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

// Small page-locked buffers (the staging of a sub-batch's tables, the words read back at its end, the store's job tables) are pieces of
// ONE block per context: hipHostMalloc costs ~1 ms per call whatever the size, and a fresh process's first job made fifteen of them
// (12 - 17 ms of the ComputeMatches executable's 0.4 s; profiles/r05_cli_cold_call.txt).  A piece is never handed back (a buffer that
// re-grows takes a new one); what does not fit gets its own allocation as before.
struct PinnedPool {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    bool failed = false;
    static constexpr size_t kBytes = (size_t)16 << 20, kLargestPiece = (size_t)4 << 20;
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (failed || bytes > kLargestPiece) return nullptr;
        if (!base) {
            void* q = nullptr;
            if (timed_host_malloc(&q, kBytes) != hipSuccess) {
                failed = true;
                return nullptr;
            }
            base = static_cast<char*>(q);
            cap = kBytes;
        }
        if (used + bytes > cap) return nullptr;
        void* p = base + used;
        used += bytes;
        return p;
    }
    void release() {
        if (base) (void)hipHostFree(base);
        base = nullptr;
        cap = used = 0;
    }
};

// page-locked host memory, grow-only and content-preserving (result lists of a call accumulate over its sub-batches)
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    PinnedPool* pool = nullptr;   // set once by msfm_create for the small buffers
    bool pooled = false;          // p is a piece of the pool: not freed on its own
    hipError_t ensure(size_t bytes, size_t keep, size_t hint = 0) {
        if (bytes <= cap) return hipSuccess;
        size_t want = std::max(bytes + bytes / 2 + 4096, hint);
        void* q = pool ? pool->take(want) : nullptr;
        const bool from_pool = q != nullptr;
        if (!q) {
            const hipError_t e = timed_host_malloc(&q, want);
            if (e != hipSuccess) return e;
        }
        if (p && keep) std::memcpy(q, p, keep);
        if (p && !pooled) {
            if (t_deferred_frees) t_deferred_frees->dead_host.push_back(p);
            else (void)hipHostFree(p);
        }
        p = q;
        cap = want;
        pooled = from_pool;
        return hipSuccess;
    }
    void release() {
        if (p && !pooled) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        pooled = false;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// The call-wide result lists on the host: ONE contiguous range of address space, page-locked piece by piece as the lists grow.
// Round 4 kept them in one hipHostMalloc block, re-allocated "once or twice" per call: page-locking costs 0.2 ms per MiB on this part
// (profiles/r05_ubench_malloc.txt) -- 0.9 s for config 4's 4.3 GB, all of it with the GPU idle behind the two sub-batches in flight
// (the first call of a process: 8.9 s against 8.0 s warm).  Now the range is reserved up front (address space only: an upper bound of
// the job's matches) and a completed sub-batch page-locks just the 32-MiB pieces its lists reach into -- a few ms, while the next
// sub-batches run.  The range never moves: msfm_view_matches' pointers stay valid, nothing waits for a copy of the lists.
struct GrowPinned {
    char* base = nullptr;
    size_t reserved = 0, pinned = 0;   // [0, pinned) is page-locked
    static constexpr size_t kPiece = (size_t)32 << 20;
    // (pieces of 1, 1, 2, 4, 8, 16, 32, 32, ... MiB: msfm_pinned_piece_end, msfm_hostutil.h -- the lists of a small call, the pre-emptive
    // filter's few KB, paid 2 x 6.4 ms for a 32-MiB piece each)
    static size_t piece_end(size_t off) { return (size_t)msfm_pinned_piece_end(off); }
    hipError_t reserve(size_t bytes) {
        bytes = (bytes + kPiece - 1) / kPiece * kPiece;
        if (bytes <= reserved) return hipSuccess;
        release();
        // (the bound can be far beyond what the job will produce -- min(n1, n2) per pair -- and beyond what the system lets one mapping
        // be: take what it gives; a job that outgrows the range is told to use the streaming form)
        void* p = MAP_FAILED;
        for (; bytes >= kPiece; bytes = (bytes / 2 + kPiece - 1) / kPiece * kPiece) {
            p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (p != MAP_FAILED || bytes == kPiece) break;
        }
        if (p == MAP_FAILED) return hipErrorOutOfMemory;
        base = static_cast<char*>(p);
        reserved = bytes;
        pinned = 0;
        return hipSuccess;
    }
    hipError_t ensure_pinned(size_t upto) {
        if (upto > reserved) return hipErrorOutOfMemory;
        while (pinned < upto) {
            const size_t piece = piece_end(pinned) - pinned;
            const double t0 = AllocClock::now();
            const hipError_t e = hipHostRegister(base + pinned, piece, hipHostRegisterDefault);
            AllocClock& c = alloc_clock();
            c.pin_ms += AllocClock::now() - t0;
            c.pin_bytes += (long long)piece;
            c.pin_n += 1;
            if (e != hipSuccess) return e;
            pinned += piece;
        }
        return hipSuccess;
    }
    // device -> this range at byte offset `off`, asynchronous: one copy per registered piece it touches (a copy may not straddle two
    // registrations)
    hipError_t copy_in(size_t off, const void* dev, size_t bytes, hipStream_t stream) {
        const char* s = static_cast<const char*>(dev);
        while (bytes) {
            const size_t n = std::min(bytes, piece_end(off) - off);
            const hipError_t e = hipMemcpyAsync(base + off, s, n, hipMemcpyDeviceToHost, stream);
            if (e != hipSuccess) return e;
            off += n;
            s += n;
            bytes -= n;
        }
        return hipSuccess;
    }
    void release() {
        for (size_t off = 0; off < pinned; off = piece_end(off)) (void)hipHostUnregister(base + off);
        if (base) munmap(base, reserved);
        base = nullptr;
        reserved = pinned = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(base); }
};

// The call-wide result lists on the device (msfm_fetch_matches_device): SEGMENTS, a sub-batch's lists whole in one of them, none ever
// moved or freed during a call (round 4: one buffer, grown by allocate + copy + hipFree behind a drain of every stream -- the hipFree
// alone waited 20-70 ms for the sub-batches in flight).  Kept between calls.
struct OutSeg {
    DevBuf qt, d;
    size_t cap = 0, first = 0, count = 0;   // entries; `first` = index of its first match in the call's lists
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

// A second core for the host side of an upload: memcpy into page-locked memory runs at 32 GB/s on one core of this box and at 59 GB/s on
// two (profiles/r05_ubench_upload.txt) -- and it, not the DMA behind it (38-46 GB/s), is what a bulk load waits for.  The helper takes the
// upper half of every piece; between the uploads of a burst it spins (a wake-up through the condition variable costs more than the copy
// of a whole image), after ~100 us without work it sleeps.
struct CopyHelper {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<unsigned> posted{0}, done{0};
    std::atomic<bool> sleeping{false}, stop{false};
    char* dst = nullptr;
    const char* src = nullptr;
    size_t bytes = 0;
    void run() {
        unsigned seen = 0;
        for (;;) {
            int spins = 0;
            while (posted.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_relaxed)) {
                if (++spins < 40000) {
                    __builtin_ia32_pause();
                    continue;
                }
                std::unique_lock<std::mutex> lk(mu);
                sleeping.store(true);
                cv.wait(lk, [&] { return posted.load() != seen || stop.load(); });
                sleeping.store(false);
                spins = 0;
            }
            if (stop.load()) return;
            seen = posted.load(std::memory_order_acquire);
            std::memcpy(dst, src, bytes);
            done.store(seen, std::memory_order_release);
        }
    }
    // dst[0 .. n) = src[0 .. n), the upper half on the helper's core
    void copy(char* d, const char* s, size_t n) {
        if (n < ((size_t)256 << 10)) {
            std::memcpy(d, s, n);
            return;
        }
        if (!th.joinable()) th = std::thread([this] { run(); });
        const size_t half = (n / 2) & ~(size_t)4095;
        dst = d + half;
        src = s + half;
        bytes = n - half;
        const unsigned ticket = posted.fetch_add(1, std::memory_order_release) + 1;
        if (sleeping.load()) {
            std::lock_guard<std::mutex> lk(mu);
            cv.notify_one();
        }
        std::memcpy(d, s, half);
        while (done.load(std::memory_order_acquire) != ticket) __builtin_ia32_pause();
    }
    ~CopyHelper() {
        if (th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop.store(true);
                cv.notify_one();
            }
            th.join();
        }
    }
};

// One image of the descriptor store (msfm_store.hip.h says what is resident and what is derived on demand).
struct Image {
    int n = -1;  // -1: not uploaded
    int nblk = 0;    // 128-row blocks holding data
    int nalloc = 0;  // allocated blocks (a multiple of four: one sweep work item = 512 A rows); padding rows are zero-filled
    // an upload finalize_store has not built yet: the caller's rows wait in the context's inbox (device memory)
    bool pending = false;
    const void* inbox = nullptr;
    int inbox_chunk = -1;
    int inbox_kind = kSrcF32;   // kSrcF32 / kSrcU8 (row-major)
    bool no_twin = false;       // an MSFM_DTYPE_U8 upload or rows of a byte image: never twinned (route Q is for float stores)
    // the store chunks (StoreArena) the image's arrays live in; -1: none
    int chunk_core = -1, chunk_wide = -1, chunk_panel = -1, chunk_kp = -1;
    // brute-force route only, built on demand: k-major fp32 panels in the accumulation order `panel_order`
    float* panel = nullptr;
    int panel_order = -1;
    // float forms -- the image's core if it is a float image, derived on demand for a byte image: fp32 rows permuted for the exact
    // re-check (PairDesc::a_rawp), fp16 operand rows of 272 B (128 halfs + the norm quadruple of the ninth MFMA k-step), row norms
    // (+inf padded), maxima
    float* rawp = nullptr;
    _Float16* h16 = nullptr;
    float* nrm = nullptr;
    float c = 1.f;            // scale of the quadruples (power of two)
    float nrm_max = 0.f, abs_max = 0.f;
    bool pf_safe = false;
    // byte stores (MSFM_DTYPE_U8 uploads, their subsets, float uploads holding only integers 0..255): signed operand rows of 176 B
    // (kI8RowBytes: 128 operand bytes, 16 digits, 16 constants, padding) for the integer matrix cores, the float "norms"
    // 2 floor(|x - 128|^2 / 2) and the exact |x - 128|^2 (msfm_sweep_i8.hip.h)
    bool is_u8 = false;
    bool from_u8 = false;     // integer values 0..255: S is an exact integer under any accumulation order
    signed char* i8 = nullptr;
    float* nrm_i8 = nullptr;
    int* n2_i8 = nullptr;
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

// Device memory of the store: chunks handed out by a bump pointer, counted per chunk, freed when their last image goes.  One
// finalize_store call places all its images in ONE chunk (one hipMalloc instead of up to eleven per image; a hipFree costs ~90 us on
// this part, profiles/r05_ubench_upload.txt: freeing a 128-image store took 129 ms).
struct StoreChunk {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    int live = 0;
};
struct StoreArena {
    std::vector<StoreChunk> chunks;
    int cur = -1;
    bool recycle = false;   // keep emptied chunks for the next taker instead of freeing them (the inbox: the same few chunks wave after wave)
    // room for `bytes` more in the current chunk, or a new current chunk of max(bytes, min_chunk)
    hipError_t reserve(size_t bytes, size_t min_chunk) {
        bytes += 256;
        if (cur >= 0 && chunks[(size_t)cur].base && chunks[(size_t)cur].used + bytes <= chunks[(size_t)cur].cap) return hipSuccess;
        if (recycle)
            for (size_t i = 0; i < chunks.size(); ++i)
                if (chunks[i].base && chunks[i].live == 0 && chunks[i].cap >= bytes) {
                    chunks[i].used = 0;
                    cur = (int)i;
                    return hipSuccess;
                }
        // a chunk with enough room behind what it holds becomes the current one again: an incremental caller (upload, keypoints, match,
        // image after image) alternates exact-size image chunks with small keypoint requests -- without this every cycle abandoned a
        // partly used 1-MiB keypoint chunk for a fresh one
        for (size_t i = 0; i < chunks.size(); ++i)
            if (chunks[i].base && chunks[i].used + bytes <= chunks[i].cap) {
                cur = (int)i;
                return hipSuccess;
            }
        if (cur >= 0 && chunks[(size_t)cur].live == 0) free_chunk(cur);   // (an empty current chunk that is too small)
        int slot = -1;
        for (size_t i = 0; i < chunks.size(); ++i)
            if (!chunks[i].base) slot = (int)i;
        if (slot < 0) {
            chunks.push_back(StoreChunk{});
            slot = (int)chunks.size() - 1;
        }
        StoreChunk& c = chunks[(size_t)slot];
        const size_t want = std::max(bytes, min_chunk);
        hipError_t e = timed_malloc((void**)&c.base, want);
        if (e != hipSuccess) {
            c.base = nullptr;
            return e;
        }
        c.cap = want;
        c.used = 0;
        c.live = 0;
        cur = slot;
        return hipSuccess;
    }
    // `bytes` from the current chunk (256-byte aligned; reserve() made the room); the caller holds one reference on *chunk per call
    void* take(size_t bytes, int* chunk) {
        StoreChunk& c = chunks[(size_t)cur];
        const size_t at = (c.used + 255) & ~(size_t)255;
        if (at + bytes > c.cap) return nullptr;
        c.used = at + bytes;
        c.live += 1;
        *chunk = cur;
        return c.base + at;
    }
    void free_chunk(int i) {
        StoreChunk& c = chunks[(size_t)i];
        if (c.base) (void)hipFree(c.base);
        c = StoreChunk{};
        if (cur == i) cur = -1;
    }
    void drop(int& chunk) {
        if (chunk < 0) return;
        StoreChunk& c = chunks[(size_t)chunk];
        if (--c.live <= 0) {
            if (chunk == cur || recycle) c.used = 0, c.live = 0;   // the current chunk is kept for the next taker
            else free_chunk(chunk);
        }
        chunk = -1;
    }
    size_t bytes() const {
        size_t s = 0;
        for (const StoreChunk& c : chunks) s += c.cap;
        return s;
    }
    void release_all() {
        for (size_t i = 0; i < chunks.size(); ++i) free_chunk((int)i);
        chunks.clear();
        cur = -1;
    }
};

constexpr int kSlots = 2 * MSFM_MAX_IMAGES + 2;  // ids >= MSFM_MAX_IMAGES: auxiliary (top-scale subsets, two operator-level scratch slots)
// Sub-batches of msfm_match_pairs are bounded by a pair count and by the device scratch of ALL scratch sets in flight together
// (msfm_pair_scratch_bytes per pair, msfm_hostutil.h): at most kDefaultScratchBytes, and never more than a quarter of what the
// device has free when the call starts (hipMemGetInfo + what the sets already hold) -- a 9-second job gains ~2.5 % from 7
// instead of 19 sub-batches per 80 000 pairs, but every GiB of scratch costs ~10 ms the first time it is allocated
// (profiles/r04_scratch_ab.txt).
constexpr long long kDefaultScratchBytes = (long long)64 << 30;   // (round 5: a pair on the integer route is charged 0.7 of round 4's estimate; the sets hold ~50 GiB under this limit)
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
// copy-out) runs on one stream, the sweeps of sub-batches k + 1 and k + 2 are already queued on the others (MatchJob, msfm_job.hip.h).
// Buffers grow on demand (DevBuf::ensure = an explicit hipDeviceSynchronize + hipFree + hipMalloc: a growth inside issue()
// serialises the pipeline once -- in the first call of a job shape, and whenever a later sub-batch is more than 1/8 larger than
// any before -- and the explicit wait is what makes replacing a buffer safe that kernels queued earlier still read; steady
// state allocates nothing).
struct PfPending {                // what the end-of-batch synchronisation has to look at
    bool active = false, compact = false, i8 = false, q8 = false;
    size_t n_lists = 0, P = 0;
    long long rows_cap = 0, cand_cap = 0, items_cap = 0;
    long long rows_ub = 0;            // the rows this sub-batch could compact at most: what its needs are relative to (the next prediction)
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
    PinnedBuf h_sub_qt, h_sub_d;      // streaming form: the same lists in page-locked host memory (msfm_match_pairs_next)
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
    void release_device() { for_each_buf([](DevBuf& b, void*) { b.release(); }, nullptr); }   // (the page-locked host side stays)
    void release_all() {
        for_each_buf([](DevBuf& b, void*) { b.release(); }, nullptr);
        for (PinnedBuf& h : h_up) h.release();
        h_sub_qt.release();
        h_sub_d.release();
        h_summary.release();
        h_tail.release();
    }
};

struct MatchJob;   // msfm_job.hip.h

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
    DevBuf d_zero_row;                // the all-zero operand row
    // ---- descriptor store (msfm_store_host.hip.h)
    StoreArena store;                 // the images' arrays
    StoreArena inbox;                 // uploads waiting for finalize_store (row-major, as the caller handed them over)
    std::vector<int> pending;         // ids uploaded since the last finalize_store
    size_t inbox_waiting = 0;         // bytes of those uploads
    bool store_async = false;         // copies / kernels may be in flight on the store's stream (the other streams of a matching call do not wait for it)
    PinnedBuf up_ring;                // page-locked staging of the uploads: kUpSlots slots of kUpSlotBytes
    CopyHelper copier;                // second core of the host copy (MSFM_UPLOAD_THREADS=1: off)
    bool copy_helper = true;
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    bool up_ev_recorded[2] = {false, false};
    unsigned up_seq = 0;
    DevBuf d_jobs, d_store_maxima;    // finalize_store's job table and its per-job maxima (two sets of 8 words per job)
    PinnedBuf h_jobs, h_store_maxima;
    PinnedPool pinned_pool;           // the small page-locked buffers of the context and its scratch sets
    hipEvent_t jobs_ev = nullptr;     // behind the last copy out of h_jobs
    size_t store_peak_bytes = 0;      // largest store.bytes() seen (msfm_store_info)
    std::vector<OutSeg> out_segs;     // the match lists of the whole call on the device (msfm_fetch_matches_device)
    size_t out_used = 0;              // segments holding lists of the current call
    int fix_cap = 1 << 16;            // entries of the sqrt-space tie queue; grows on overflow (the sub-batch is re-run)
    // sub-batch limits of msfm_match_pairs (msfm_set_limits / MSFM_MAX_PAIRS_PER_BATCH / MSFM_SCRATCH_MIB)
    int max_pairs_per_batch = kDefaultMaxPairsPerBatch;
    long long scratch_bytes = 0;      // msfm_set_limits / MSFM_SCRATCH_MIB: total for the scratch sets in flight; 0 = automatic (above)
    int last_hip_error = 0;           // hipError_t of the last failed HIPCHK (0: none since the matching call began)
    long long budget_cached = 0;      // the automatic scratch budget, derived from hipMemGetInfo once per (store size, limit)
    size_t budget_for_store = 0;
    long long budget_for_limit = 0;
    long long issue_seq = 0;          // sub-batches issued so far
    double pipeline_taper = kDefaultTaper;   // size of a call's last part relative to the average part (MSFM_PIPELINE_TAPER; 1: equal parts)
    int in_flight = kInFlight;        // scratch sets used (MSFM_IN_FLIGHT=1 at msfm_create: no sub-batch overlap, for A/B measurements)
    int pipeline = kDefaultPipeline;  // sub-batches a large call is cut into at least, so that tails overlap sweeps (1: off)
    int prefilter = 1;                // 0: brute force only; 1: MFMA prefilter, integer matrix cores for byte stores; 2: fp16 MFMA only
    int byte_detect = 1;              // a float upload holding only integers in [0, 255] is a byte store (MSFM_BYTE_DETECT=0: off)
    float q8_level = 0.f;             // m: largest value of the twinned images so far, rounded up to a multiple of 1/16 (msfm_q8.hip.h)
    int q8_direct = 1;                // thresholds for sweep 2 straight from the twins' sweep when they are fine enough (MSFM_Q8_DIRECT=0: never, 2: always)
    int q8_route = 1;                 // float images in [0, 1] get byte twins and their first sweep on the integer cores (MSFM_Q8=0: off)
    DeferredFrees deferred;           // device buffers replaced while sub-batches were in flight: freed when the call has drained
    long long cmp_rows_hint = 0;      // compacted rows the previous batch needed (sizes the next batch's buffers) ...
    long long hint_rows_ub = 0;       // ... of the rows it could compact at most: the prediction scales with that, and is void beyond a factor 2
    long long items_hint = 0, cand_hint = 0;   // likewise: work items, candidate-list capacity

    // results of the last msfm_match_pairs call
    bool have_results = false;
    std::vector<int64_t> res_offsets;
    std::vector<int32_t> res_sens;   // per pair: rows / columns without an order-invariance certificate
    GrowPinned res_qt, res_dist;  // (q, t) int32 pairs and distances of res_count matches
    size_t res_count = 0;

    msfm_profile prof = {};
    std::vector<hipEvent_t> ev_pool;
    MatchJob* job = nullptr;          // the matching call in progress (one at a time; the streaming form keeps it between calls)
    bool series_open = false;         // a streaming series (msfm_match_pairs_begin .. _next) has sub-batches in flight: the store must not change
};

#define SC (*ctx->cur)

namespace {

int fail(msfm_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}
// which HIP error the last failed HIPCHK saw: a matching call looks at it to tell "the device is out of memory" (it shrinks its
// sub-batches and tries again, msfm_job.hip.h) from every other device error
inline void note_hip_error(msfm_ctx* ctx, hipError_t e) {
    if (ctx) ctx->last_hip_error = (int)e;
}
inline void note_hip_error(const msfm_ctx*, hipError_t) {}

// The exception barrier of the C ABI (msfm_guard.h): every `extern "C"` entry point runs its body between these two.  An exception leaves
// the context usable: after_api_exception (msfm_match.hip) records the text, drains the streams and closes an open series.
void after_api_exception(msfm_ctx* ctx, const char* text) noexcept;
#define MSFM_API_BEGIN(ctxp) \
    return msfm_guard([&](int, const char* t__) noexcept { after_api_exception(ctxp, t__); }, [&]() -> int {
#define MSFM_API_END \
    });

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            note_hip_error(ctx, e__);                                                         \
            return fail(ctx, MSFM_E_DEVICE,                                                   \
                        std::string(#call) + ": " + hipGetErrorString(e__));                  \
        }                                                                                     \
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

}  // namespace
