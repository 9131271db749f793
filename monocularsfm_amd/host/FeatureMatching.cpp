// FeatureMatching.cpp -- see FeatureMatching.h.  Pair order, batch boundaries, transactions, resume-by-row and stdout lines follow the
// reference's src/Feature/FeatureMatching.cpp; the per-pair arithmetic runs on the GPU through include/msfm_match.h.
//
// Shape of a run (round 6: the executable scales with the device count instead of adding its SQLite time to the GPU time):
//
//   calling thread (owns the SQLite handle)         device threads, one per GPU
//   ------------------------------------------      ---------------------------------------------------------------
//   bulk load, exist-check of every pair
//   [pre-emptive filter of every pair: device threads, streaming series on the 100-row subsets]
//   deal the pairs to do into blocks, round-robin    msfm_match_pairs_begin(all pairs of this device)
//   for every reference group, in order:             loop: msfm_match_pairs_next -> lay the chunk's lists out as stored rows
//       wait for the results of its pairs   <------        (column swap, [host RANSAC], emission options) -> bounded queue
//       BEGIN; stdout lines; WriteMatches; END        (two more sub-batches stay in flight on the GPU meanwhile)
//
// The reference emits inside its pair loop too (FeatureMatching.cpp:13, 63-72); what is printed and written, and in which
// transactions, is unchanged -- tests/test_cli_gpu.py compares stdout, rows and the transaction trace.
#include "FeatureMatching.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <thread>

#include "GeometricVerification.h"
#include "MatchEmission.h"
#include "Pipeline.h"
#include "Timer.h"

namespace MonocularSfM {

namespace {
// MSFM_CLI_TIMING=1: wall-clock per phase on stderr when the matcher closes.  `device` and `layout` are summed over the device
// threads and run BESIDE the calling thread's phases: the calling thread's own time is exist + read + preemptive + wait + emit.
struct PhaseClock {
    double exist = 0, read_desc = 0, device = 0, layout = 0, read_kp = 0, emit = 0, wait = 0, preemptive = 0, open_dev = 0, close_dev = 0, run = 0;
    long long pairs = 0, matches = 0;
    int devices = 1;
    bool on = std::getenv("MSFM_CLI_TIMING") != nullptr;
    void Report() const {
        if (!on || exist + read_desc + device + emit == 0) return;
        std::fprintf(stderr, "[msfm timing] exist-check %.3f s | read descriptors + upload %.3f s | device match + fetch %.3f s | "
                             "pre-emptive filter %.3f s | read keypoints %.3f s | verification %.3f s | stdout + WriteMatches %.3f s | "
                             "open database + device %.3f s | close %.3f s\n",
                     exist, read_desc, device / devices, preemptive, read_kp, layout / devices, emit, open_dev, close_dev);
        // the pipeline's own figures: what bounds the run is the larger of the device threads' time and the calling thread's
        std::fprintf(stderr, "[msfm pipeline] devices %d | pairs matched %lld | matches written %lld | matching phase wall %.3f s | device threads: in "
                             "msfm_match_pairs_next %.3f s + row layout %.3f s (mean per device) | calling thread: waiting for results %.3f s, "
                             "stdout + WriteMatches %.3f s -> bound by %s\n",
                     devices, pairs, matches, run, device / devices, layout / devices, wait, emit, wait > emit ? "the devices" : "emission");
    }
} g_clock;
struct Lap {
    double* acc;
    Timer t;
    explicit Lap(double* a) : acc(a) { t.Start(); }
    ~Lap() { *acc += t.ElapsedSeconds(); }
};
[[noreturn]] void Die(msfm_ctx* ctx, const char* what, int rc) {
    std::fprintf(stderr, "ComputeMatches: %s failed (status %d): %s\n", what, rc, ctx ? msfm_last_error(ctx) : "");
    std::exit(EXIT_FAILURE);
}
#define MSFM_CALL(ctx, expr)                       \
    do {                                           \
        const int rc__ = (expr);                   \
        if (rc__ != MSFM_OK) Die(ctx, #expr, rc__); \
    } while (0)

// Pairs per block of the round-robin deal (cost-balanced: a block ends at the pair nearest to its share of sum n1 * n2).  Small blocks
// keep the devices in step with the emitter -- the series of a device runs across its blocks without a gap, so the block size does not
// cost device efficiency.  MSFM_SUPER_BATCH_PAIRS (the name of round 5's knob) overrides it; tests cross the boundaries with tiny values.
size_t BlockPairs() {
    static const size_t v = [] {
        const char* e = std::getenv("MSFM_SUPER_BATCH_PAIRS");
        const long long x = e ? std::atoll(e) : 0;
        return x > 0 ? (size_t)x : (size_t)1024;
    }();
    return v;
}

// One device's share of a run and the thread that works it off.
struct DeviceRun {
    msfm_ctx* ctx = nullptr;
    std::vector<int32_t> pairs;          // 2 * n: (id1, id2) in the order this device computes them
    std::unique_ptr<ChunkQueue> queue;
    std::thread thread;
    int status = MSFM_OK;
    std::string error;
    double in_next = 0, in_layout = 0;   // seconds inside msfm_match_pairs_next / laying the rows out
    // emitter side
    std::unique_ptr<ResultChunk> cur;
    size_t consumed = 0;
};

}  // namespace

FeatureMatcher::FeatureMatcher(const std::string& database_path, const int& max_num_matches,
                               const double& max_distance, const double& distance_ratio, const bool& cross_check)
    : database_path_(database_path),
      max_num_matches_(max_num_matches),
      max_distance_(max_distance),
      distance_ratio_(distance_ratio),
      cross_check_(cross_check) {
    // MSFM_GEOMETRIC_VERIFICATION: unset / "1" = on the device (default), "host" = host twin, "0" = off
    const char* gv = std::getenv("MSFM_GEOMETRIC_VERIFICATION");
    if (gv && gv[0] == '0') geometric_verification_ = false;
    if (gv && std::string(gv) == "host") verification_on_host_ = true;
}

FeatureMatcher::~FeatureMatcher() { CloseDatabaseAndDevice(); }

void FeatureMatcher::OpenDatabaseAndDevice() {
    Lap lap_open(&g_clock.open_dev);
    database_ = new Database();
    database_->Open(database_path_);
    if (devices_.empty()) {
        std::vector<int> devs;
        const char* list = std::getenv("MSFM_DEVICES");
        if (list && std::string(list) == "all") {   // every gfx950 device of the node: one context and one host thread each
            const int n = msfm_device_count();
            for (int d = 0; d < n; ++d) devs.push_back(d);
        } else if (list) {
            for (const char* c = list; *c;) {
                char* end = nullptr;
                const long v = std::strtol(c, &end, 10);
                if (end == c) break;
                devs.push_back((int)v);
                c = (*end == ',') ? end + 1 : end;
            }
        }
        if (devs.empty()) {
            int dev = 0;
            if (const char* d = std::getenv("MSFM_DEVICE")) dev = std::atoi(d);
            devs.push_back(dev);
        }
        // (contexts of different GPUs come up in parallel: the first stream + first allocation of a device cost ~30 ms each)
        devices_.assign(devs.size(), Device());
        std::vector<int> status(devs.size(), MSFM_OK);
        std::vector<std::thread> starters;
        auto create = [&](size_t g) {
            msfm_ctx* c = nullptr;
            status[g] = msfm_create(devs[g], &c);
            if (status[g] == MSFM_OK)
                if (const char* o = std::getenv("MSFM_ACCUM_ORDER")) status[g] = msfm_set_accum_order(c, std::atoi(o));
            // An ordinal listed k times (the tests' way to run the fan-out on a one-GPU box): every context would size its scratch
            // for a quarter of the device's free memory on its own -- four of them on one MI355X ran out of memory at config-4 scale.
            // They share the default budget instead (MSFM_SCRATCH_MIB still overrides: it is read at msfm_create, this only applies without it).
            const long long sharing = (long long)std::count(devs.begin(), devs.end(), devs[g]);
            if (status[g] == MSFM_OK && sharing > 1 && !std::getenv("MSFM_SCRATCH_MIB"))
                status[g] = msfm_set_limits(c, std::getenv("MSFM_MAX_PAIRS_PER_BATCH") ? std::atoi(std::getenv("MSFM_MAX_PAIRS_PER_BATCH")) : 0,
                                            ((int64_t)48 << 30) / sharing);
            devices_[g].ctx = c;
        };
        create(0);   // (the runtime's own start-up happens once, on this thread)
        for (size_t g = 1; g < devs.size(); ++g) starters.emplace_back(create, g);
        for (auto& t : starters) t.join();
        for (size_t g = 0; g < devs.size(); ++g)
            if (status[g] != MSFM_OK || !devices_[g].ctx) {
                std::fprintf(stderr, "ComputeMatches: no usable gfx950 GPU at ordinal %d (msfm_create status %d); there is no CPU fallback\n", devs[g], status[g]);
                std::exit(EXIT_FAILURE);
            }
        ctx_ = devices_[0].ctx;
    }
}

void FeatureMatcher::CloseDatabaseAndDevice() {
    {
    Lap lap_close(&g_clock.close_dev);
    if (database_) {
        database_->Close();
        delete database_;
        database_ = nullptr;
    }
    // (the contexts of different GPUs go down in parallel: ~30 ms each, ~150 hipFree)
    std::vector<std::thread> closers;
    for (size_t g = 1; g < devices_.size(); ++g) closers.emplace_back([c = devices_[g].ctx] { msfm_destroy(c); });
    if (!devices_.empty()) msfm_destroy(devices_[0].ctx);
    for (auto& t : closers) t.join();
    devices_.clear();
    ctx_ = nullptr;
    descriptor_cache_.clear();
    keypoints_cache_.clear();
    }
    g_clock.Report();
    g_clock = PhaseClock();
}

const std::vector<KeyPoint>& FeatureMatcher::KeyPointsOf(image_t image_id) {
    auto it = keypoints_cache_.find(image_id);
    if (it == keypoints_cache_.end()) it = keypoints_cache_.emplace(image_id, database_->ReadKeyPoints(image_id)).first;
    return it->second;
}

// Descriptors (and, for the device's verification, keypoints) of one image on one device; calling thread only (SQLite)
void FeatureMatcher::EnsureResidentOn(size_t g, image_t image_id) {
    Device& dev = devices_[g];
    if (dev.resident.count(image_id)) return;
    const Descriptors* d = nullptr;
    Descriptors read;
    auto cached = descriptor_cache_.find(image_id);
    if (cached != descriptor_cache_.end()) {
        d = &cached->second;
    } else {
        read = database_->ReadDescriptors(image_id);
        d = devices_.size() > 1 ? &(descriptor_cache_[image_id] = std::move(read)) : &read;
    }
    MSFM_CALL(dev.ctx, msfm_upload_image(dev.ctx, image_id, d->data.data(), d->rows, d->rows ? d->cols : MSFM_DIM, MSFM_DTYPE_F32));
    if (geometric_verification_ && !verification_on_host_) {
        // the device verifies: it needs the keypoint coordinates next to the descriptors
        const std::vector<KeyPoint>& kpts = KeyPointsOf(image_id);
        static_assert(sizeof(KeyPoint) == 16, "KeyPoint must be 4 packed floats");
        MSFM_CALL(dev.ctx, msfm_upload_keypoints(dev.ctx, image_id, reinterpret_cast<const float*>(kpts.data()), (int)kpts.size(), 4));
    }
    dev.resident.insert(image_id);
}

void FeatureMatcher::EnsureResident(image_t image_id) {
    for (size_t g = 0; g < devices_.size(); ++g) EnsureResidentOn(g, image_id);
}

void FeatureMatcher::PreloadAllImages() {
    if (const char* e = std::getenv("MSFM_BULK_LOAD"))
        if (e[0] == '0') return;
    Lap l(&g_clock.read_desc);
    // (several devices: the copies of a row into the devices' page-locked rings run side by side -- one after the other they were
    // G x the load time of one device, as much as the matching itself takes on 8 GPUs)
    DeviceCrew crew(devices_.size());
    struct Sink {
        FeatureMatcher* self;
        bool keypoints;
        DeviceCrew* crew;
    } sink{this, false, &crew};
    // visitors run on this thread while SQLite holds the row: upload from its buffer, keep nothing
    auto visit = [](void* user, image_t id, const void* data, size_t rows, size_t cols, size_t elem) {
        Sink* s = static_cast<Sink*>(user);
        FeatureMatcher* m = s->self;
        if (id < 0 || id >= MSFM_MAX_IMAGES) return;
        if (!s->keypoints) {
            if (m->devices_[0].resident.count(id)) return;
            std::vector<int> status(m->devices_.size(), MSFM_OK);
            s->crew->Run([&](size_t g) {
                Device& dev = m->devices_[g];
                status[g] = msfm_upload_image(dev.ctx, id, data, (int)rows, rows ? (int)cols : MSFM_DIM, elem == 1 ? MSFM_DTYPE_U8 : MSFM_DTYPE_F32);
                if (status[g] == MSFM_OK) dev.resident.insert(id);
            });
            for (size_t g = 0; g < status.size(); ++g)
                if (status[g] != MSFM_OK) Die(m->devices_[g].ctx, "msfm_upload_image (bulk load)", status[g]);
        } else {
            // Database::ReadKeyPoints asserts cols == 4 (x, y, size, angle); a narrower blob would be over-read below
            if (rows > 0 && cols != 4) {
                std::cerr << "ERROR: keypoints of image " << id << " have " << cols << " columns, expected 4" << std::endl;
                std::exit(EXIT_FAILURE);
            }
            std::vector<KeyPoint>& kps = m->keypoints_cache_[id];
            kps.resize(rows);
            if (rows) std::memcpy(kps.data(), data, rows * sizeof(KeyPoint));
            if (m->geometric_verification_ && !m->verification_on_host_ && m->devices_[0].resident.count(id)) {
                std::vector<int> status(m->devices_.size(), MSFM_OK);
                s->crew->Run([&](size_t g) { status[g] = msfm_upload_keypoints(m->devices_[g].ctx, id, static_cast<const float*>(data), (int)rows, 4); });
                for (size_t g = 0; g < status.size(); ++g)
                    if (status[g] != MSFM_OK) Die(m->devices_[g].ctx, "msfm_upload_keypoints (bulk load)", status[g]);
            }
        }
    };
    static_assert(sizeof(KeyPoint) == 16, "KeyPoint must be 4 packed floats");
    // The byte side table replaces the float descriptors ONLY on request (MSFM_USE_DESCRIPTORS_U8=1): the reference always
    // stores L1-root normalised floats, raw bytes give different distances, and a stale or foreign `descriptors_u8`
    // table must not change the results silently.
    const char* use_u8 = std::getenv("MSFM_USE_DESCRIPTORS_U8");
    if (use_u8 && use_u8[0] == '1' && database_->HasDescriptorsU8()) {
        std::cout << "Using byte descriptors of the descriptors_u8 side table (MSFM_USE_DESCRIPTORS_U8=1)" << std::endl;
        database_->VisitAllDescriptorsU8(visit, &sink);
    }
    database_->VisitAllDescriptors(visit, &sink);   // images the side table does not cover
    sink.keypoints = true;
    database_->VisitAllKeyPoints(visit, &sink);
    // the uploads above only copied: build the stores now (classification, one allocation, layout kernels), inside this phase's clock --
    // every device builds its own copy, all at once
    std::vector<int> status(devices_.size(), MSFM_OK);
    crew.Run([&](size_t g) { status[g] = msfm_finalize_store(devices_[g].ctx); });
    for (size_t g = 0; g < devices_.size(); ++g)
        if (status[g] != MSFM_OK) Die(devices_[g].ctx, "msfm_finalize_store", status[g]);
    bulk_loaded_ = true;
}

void FeatureMatcher::MatchImagePairs(const std::vector<std::pair<image_t, image_t>>& image_pairs) {
    MatchImagePairGroups({image_pairs});
}

// The reference calls MatchImagePairs once per group of <= 100 pairs: one transaction, and per pair either the
// "Existing, Continue!" line or match -> filter -> verify -> stdout -> WriteMatches (src/Feature/FeatureMatching.cpp:10-73).
// Here the pairs of ALL groups that still need a row are dealt to the devices up front; the groups are then EMITTED one by one,
// pair by pair, in the reference's order, each as soon as the devices have delivered its pairs -- stdout, the rows and the
// transaction boundaries are the reference's, while each device sees one uninterrupted series instead of 100 pairs at a time.
void FeatureMatcher::MatchImagePairGroups(const std::vector<std::vector<std::pair<image_t, image_t>>>& groups) {
    // ---- which pairs have a row already (the reference asks per pair, FeatureMatching.cpp:21-25)
    std::vector<std::vector<int>> slot(groups.size());   // index into `todo`, or -1: a row exists (or an earlier pair of this call writes it)
    std::vector<int32_t> todo;
    {
        Lap lap(&g_clock.exist);
        size_t total = 0;
        for (const auto& g : groups) total += g.size();
        // a large job reads the index of the matches table once instead of asking for every pair (2.3 us each, one by one)
        // (MSFM_EXIST_SWEEP_MIN: the pair count from which the sweep is used -- the tests set 0 / a huge value to run both ways)
        static const size_t sweep_min = [] {
            const char* e = std::getenv("MSFM_EXIST_SWEEP_MIN");
            return e ? (size_t)std::atoll(e) : (size_t)2048;
        }();
        const bool sweep = total >= sweep_min;
        std::vector<image_pair_t> have;
        if (sweep) have = database_->ReadAllMatchPairIds();
        std::set<std::pair<image_t, image_t>> scheduled;
        for (size_t g = 0; g < groups.size(); ++g) {
            slot[g].resize(groups[g].size());
            for (size_t k = 0; k < groups[g].size(); ++k) {
                const image_t image_id1 = groups[g][k].first, image_id2 = groups[g][k].second;
                const std::pair<image_t, image_t> key(std::min(image_id1, image_id2), std::max(image_id1, image_id2));
                const bool exists = sweep ? std::binary_search(have.begin(), have.end(), Database::ImagePairToPairId(image_id1, image_id2))
                                          : database_->ExistMatches(image_id1, image_id2);
                if (exists || !scheduled.insert(key).second) {
                    slot[g][k] = -1;
                    continue;
                }
                slot[g][k] = (int)(todo.size() / 2);
                todo.push_back(image_id1);
                todo.push_back(image_id2);
            }
        }
    }
    const size_t P = todo.size() / 2;
    const size_t G = devices_.size();
    const bool verify_on_device = geometric_verification_ && !verification_on_host_;
    const bool host_verify = geometric_verification_ && verification_on_host_;
    static const EmissionOptions emission = EmissionOptions::FromEnvironment();

    // ---- everything the device threads will touch is made resident / read now, on this thread (SQLite)
    if (P > 0) {
        std::vector<int32_t> ids(todo);
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        {
            Lap l(&g_clock.read_desc);
            for (int32_t id : ids) EnsureResident(id);
        }
        if (host_verify) {
            Lap l(&g_clock.read_kp);
            for (int32_t id : ids) (void)KeyPointsOf(id);
        }
    }

    // ---- the deal: blocks of ~BlockPairs() pairs and equal cost sum n1 * n2, block b to device b mod G
    // The order the pairs are COMPUTED and their rows WRITTEN in: the reference's (default), or -- MSFM_EMIT_ORDER=pair_id -- ascending
    // pair_id.  The matchers enumerate (i, j < i) with i outermost while the row key is kMaxNumImages * j + i (Database.cpp:656-667):
    // the reference's order inserts into a different one of ~N key bands every time, each insert lands between two full leaves
    // (a 3-KB row fills a 4-KB page) and SQLite rebalances three sibling pages for it -- 16 us per row on the GPU box, 3.4 x the cost
    // of the same rows in key order, which SQLite appends (tools/emit_bench.cpp, profiles/r06_emission_study.txt).  In key order the
    // stdout lines are still the reference's, in the reference's order, a group's as soon as its rows and those of every group before
    // it are in; transactions hold 100 rows.
    static const bool key_order = [] {
        const char* e = std::getenv("MSFM_EMIT_ORDER");
        return e && std::string(e) == "pair_id";
    }();
    std::vector<int> work_order(P);
    for (size_t p = 0; p < P; ++p) work_order[p] = (int)p;
    if (key_order)
        std::sort(work_order.begin(), work_order.end(), [&](int a, int b) {
            return Database::ImagePairToPairId(todo[2 * (size_t)a], todo[2 * (size_t)a + 1]) < Database::ImagePairToPairId(todo[2 * (size_t)b], todo[2 * (size_t)b + 1]);
        });
    std::vector<DeviceRun> runs(G);
    std::vector<uint8_t> dev_of(P, 0);   // by position in work_order
    if (P > 0) {
        std::vector<double> cum(P + 1, 0.0);
        for (size_t w = 0; w < P; ++w) {
            const size_t p = (size_t)work_order[w];
            int n1 = 0, n2 = 0;
            (void)msfm_image_rows(ctx_, todo[2 * p], &n1);
            (void)msfm_image_rows(ctx_, todo[2 * p + 1], &n2);
            cum[w + 1] = cum[w] + (double)n1 * n2 + 1.0;
        }
        const std::vector<size_t> ends = DealBlockEnds(cum, G, BlockPairs());
        size_t begin = 0;
        for (size_t b = 0; b < ends.size(); ++b) {
            DeviceRun& r = runs[b % G];
            for (size_t w = begin; w < ends[b]; ++w) {
                const size_t p = (size_t)work_order[w];
                dev_of[w] = (uint8_t)(b % G);
                r.pairs.push_back(todo[2 * p]);
                r.pairs.push_back(todo[2 * p + 1]);
            }
            begin = ends[b];
        }
    }
    if (G > 255) Die(nullptr, "more than 255 devices", MSFM_E_INVALID);

    // ---- device threads
    msfm_match_params prm;
    prm.ratio = (float)distance_ratio_;  // ComputeCrossMatches takes `const float distance_ratio`
    prm.cross_check = cross_check_ ? 1 : 0;
    prm.max_distance = max_distance_;
    std::atomic<bool> give_up(false);
    Timer run_timer;
    run_timer.Start();
    // what waits between a device and the emitter: 1 GiB over all devices (emission slower than the GPUs: they wait; never the whole job in memory)
    const size_t queue_cap = ((size_t)1 << 30) / std::max<size_t>(1, G);
    auto work = [&](DeviceRun& r) {
        const size_t n = r.pairs.size() / 2;
        auto fail = [&](const char* what, int rc) {
            r.status = rc;
            r.error = std::string(what) + ": " + msfm_last_error(r.ctx);
        };
        int rc = msfm_match_pairs_begin(r.ctx, r.pairs.data(), (int)n, &prm, verify_on_device ? 1 : 0, nullptr);  // NULL: FilterMatches' constants
        if (rc != MSFM_OK) {
            fail("msfm_match_pairs_begin", rc);
            r.queue->Close();
            return;
        }
        Timer chunk_timer;
        chunk_timer.Start();
        while (!give_up.load(std::memory_order_relaxed)) {
            msfm_chunk ch;
            Timer t;
            t.Start();
            rc = msfm_match_pairs_next(r.ctx, &ch);
            r.in_next += t.ElapsedSeconds();
            if (rc != MSFM_OK) {
                fail("msfm_match_pairs_next", rc);
                break;
            }
            if (ch.n_pairs == 0) break;
            t.Restart();
            std::unique_ptr<ResultChunk> out(new ResultChunk());
            out->first = (size_t)ch.first_pair;
            out->n = (size_t)ch.n_pairs;
            out->offsets.resize(out->n + 1);
            out->rows.resize((size_t)ch.count * 2);
            int64_t at = 0;
            std::vector<DMatch> list, kept;
            for (size_t p = 0; p < out->n; ++p) {
                const image_t id1 = r.pairs[2 * (out->first + p)], id2 = r.pairs[2 * (out->first + p) + 1];
                const int32_t* qt = ch.qt + 2 * ch.offsets[p];
                size_t m = (size_t)(ch.offsets[p + 1] - ch.offsets[p]);
                out->offsets[p] = at;
                const bool swap = Database::SwapImagePair(id1, id2);
                point2D_t* dst = out->rows.data() + 2 * at;
                if (!host_verify && !emission.scene_graph_order && emission.min_num_matches <= 0) {
                    // the stored row straight from the device's list: column 0 = the smaller image id's index
                    if (swap)
                        for (size_t i = 0; i < m; ++i) {
                            dst[2 * i] = qt[2 * i + 1];
                            dst[2 * i + 1] = qt[2 * i];
                        }
                    else if (m)
                        std::memcpy(dst, qt, m * 8);
                } else {
                    // FeatureUtils::FilterMatches by the host twin and / or the emission options: through the DMatch form
                    list.resize(m);
                    for (size_t i = 0; i < m; ++i) {
                        list[i].queryIdx = qt[2 * i];
                        list[i].trainIdx = qt[2 * i + 1];
                        list[i].distance = ch.dist[(size_t)ch.offsets[p] + i];
                    }
                    if (host_verify) {
                        kept.clear();   // (FilterMatches appends, and returns without touching the list for an empty input)
                        FilterMatches(keypoints_cache_.at(id1), keypoints_cache_.at(id2), list, &kept);
                        list.swap(kept);
                    }
                    ApplyEmissionOptions(emission, id1, id2, &list);
                    m = list.size();
                    for (size_t i = 0; i < m; ++i) {
                        dst[2 * i + (swap ? 1 : 0)] = list[i].queryIdx;
                        dst[2 * i + (swap ? 0 : 1)] = list[i].trainIdx;
                    }
                }
                at += (int64_t)m;
            }
            out->offsets[out->n] = at;
            out->rows.resize((size_t)at * 2);
            out->seconds_per_pair = chunk_timer.ElapsedSeconds() / (double)out->n;
            chunk_timer.Restart();
            r.in_layout += t.ElapsedSeconds();
            if (!r.queue->Push(std::move(out))) break;
        }
        if (r.status != MSFM_OK || give_up.load()) (void)msfm_match_pairs_end(r.ctx);
        r.queue->Close();
    };
    for (size_t g = 0; g < G; ++g) {
        runs[g].ctx = devices_[g].ctx;
        runs[g].queue.reset(new ChunkQueue(queue_cap));
        if (!runs[g].pairs.empty()) runs[g].thread = std::thread(work, std::ref(runs[g]));
        else runs[g].queue->Close();
    }
    auto stop_all = [&]() {
        give_up.store(true);
        for (DeviceRun& r : runs) r.queue->Abort();
        for (DeviceRun& r : runs)
            if (r.thread.joinable()) r.thread.join();
    };

    // ---- emission
    double waited = 0;
    Timer emit_timer;
    emit_timer.Start();
    std::string out;
    char buf[160];
    long long matches_written = 0;
    static const bool trace_txn = std::getenv("MSFM_TRACE_TRANSACTIONS") != nullptr;  // tests: one line per transaction
    // the lists of the pair at position w of the work order: the next one of its device (a device's pairs are consumed in the order it
    // computes them)
    struct Fetched {
        const point2D_t* rows;
        size_t m;
        double seconds;
    };
    auto fetch = [&](size_t w) -> Fetched {
        DeviceRun& r = runs[dev_of[w]];
        while (!r.cur || r.consumed >= r.cur->first + r.cur->n) {
            Timer wt;
            wt.Start();
            r.cur = r.queue->Pop();
            waited += wt.ElapsedSeconds();
            if (!r.cur) {
                stop_all();
                for (DeviceRun& x : runs)
                    if (x.status != MSFM_OK) {
                        std::fprintf(stderr, "ComputeMatches: %s (status %d)\n", x.error.c_str(), x.status);
                        std::exit(EXIT_FAILURE);
                    }
                Die(r.ctx, "device thread ended before its last pair", MSFM_E_STATE);
            }
        }
        const size_t li = r.consumed - r.cur->first;
        ++r.consumed;
        return Fetched{r.cur->rows.data() + 2 * r.cur->offsets[li], (size_t)(r.cur->offsets[li + 1] - r.cur->offsets[li]), r.cur->seconds_per_pair};
    };
    // one group's stdout lines (and, in the reference's order, its rows): "Existing, Continue!" or the pair's three lines
    std::vector<uint32_t> count_of;    // key order: what the stdout lines need, by todo index
    std::vector<float> seconds_of;
    auto emit_group = [&](size_t g, bool write_rows) {
        out.clear();
        for (size_t k = 0; k < groups[g].size(); ++k) {
            const image_t image_id1 = groups[g][k].first, image_id2 = groups[g][k].second;
            const int p = slot[g][k];
            if (p < 0) {
                std::snprintf(buf, sizeof(buf), "Compute Matches %d - %d Existing, Continue!\n", image_id1, image_id2);
                out += buf;
                continue;
            }
            size_t m;
            double seconds;
            if (write_rows) {
                const Fetched f = fetch((size_t)p);
                database_->WriteMatchesStored(image_id1, image_id2, f.rows, f.m);
                m = f.m;
                seconds = f.seconds;
                matches_written += (long long)m;
            } else {
                m = count_of[(size_t)p];
                seconds = seconds_of[(size_t)p];
            }
            std::snprintf(buf, sizeof(buf), "Compute Matches %d - %d ... \n\t matches num : %zu\n\t ", image_id1, image_id2, m);
            out += buf;
            out += Timer::Format(seconds, "seconds");
            out += "\n";
        }
        std::cout << out << std::flush;
    };
    if (key_order) {
        // rows in ascending pair_id, 100 per transaction; a group's lines are printed -- in the reference's order -- as soon as its rows and
        // those of every group before it are in (brute-force enumeration: row i of the images after key band i - 1)
        count_of.resize(P);
        seconds_of.resize(P);
        std::vector<uint32_t> remaining(groups.size(), 0);
        std::vector<uint32_t> group_of(P, 0);
        for (size_t g = 0; g < groups.size(); ++g)
            for (size_t k = 0; k < groups[g].size(); ++k)
                if (slot[g][k] >= 0) {
                    remaining[g] += 1;
                    group_of[(size_t)slot[g][k]] = (uint32_t)g;
                }
        size_t cursor = 0;
        auto print_ready = [&]() {
            while (cursor < groups.size() && remaining[cursor] == 0) emit_group(cursor++, false);
        };
        print_ready();
        for (size_t w0 = 0; w0 < P; w0 += 100) {
            const size_t w1 = std::min(P, w0 + 100);
            database_->BeginTransaction();
            if (trace_txn) std::fprintf(stderr, "[msfm txn] %zu\n", w1 - w0);
            for (size_t w = w0; w < w1; ++w) {
                const size_t p = (size_t)work_order[w];
                const Fetched f = fetch(w);
                database_->WriteMatchesStored(todo[2 * p], todo[2 * p + 1], f.rows, f.m);
                count_of[p] = (uint32_t)f.m;
                seconds_of[p] = (float)f.seconds;
                matches_written += (long long)f.m;
                remaining[group_of[p]] -= 1;
            }
            database_->EndTransaction();
            print_ready();
        }
        print_ready();
    } else {
        // the reference's order: one transaction per group
        for (size_t g = 0; g < groups.size(); ++g) {
            database_->BeginTransaction();
            if (trace_txn) std::fprintf(stderr, "[msfm txn] %zu\n", groups[g].size());
            emit_group(g, true);
            database_->EndTransaction();
        }
    }
    for (DeviceRun& r : runs)
        if (r.thread.joinable()) r.thread.join();
    for (DeviceRun& r : runs) {
        if (r.status != MSFM_OK) {
            std::fprintf(stderr, "ComputeMatches: %s (status %d)\n", r.error.c_str(), r.status);
            std::exit(EXIT_FAILURE);
        }
        g_clock.device += r.in_next;
        g_clock.layout += r.in_layout;
    }
    g_clock.devices = (int)G;
    g_clock.wait += waited;
    g_clock.emit += emit_timer.ElapsedSeconds() - waited;
    g_clock.run += run_timer.ElapsedSeconds();
    g_clock.pairs += (long long)P;
    g_clock.matches += matches_written;
}

void SequentialFeatureMatcher::RunMatching() {
    OpenDatabaseAndDevice();
    PreloadAllImages();
    const std::vector<Database::Image> images = database_->ReadAllImages();
    std::vector<std::vector<std::pair<image_t, image_t>>> groups;
    for (size_t i = 1; i < images.size(); ++i) {
        std::vector<std::pair<image_t, image_t>> image_pairs;
        for (int k = 1; k <= overlap_; ++k) {
            const int j = (int)i - k;
            if (j < 0) break;
            image_pairs.emplace_back((image_t)i, (image_t)j);
        }
        groups.push_back(std::move(image_pairs));
    }
    if (!groups.empty()) MatchImagePairGroups(groups);
    CloseDatabaseAndDevice();
}

void BruteFeatureMatcher::RunMatching() {
    OpenDatabaseAndDevice();
    PreloadAllImages();
    const std::vector<Database::Image> images = database_->ReadAllImages();
    // the reference's groups: a flush every max_pairs_size_ pairs and at the end of every row i
    std::vector<std::vector<std::pair<image_t, image_t>>> groups;
    for (size_t i = 0; i < images.size(); ++i) {
        std::vector<std::pair<image_t, image_t>> image_pairs;
        int cur_pairs_size = 0;
        for (size_t j = 0; j < i; ++j) {
            image_pairs.emplace_back((image_t)i, (image_t)j);
            cur_pairs_size += 1;
            if (cur_pairs_size == max_pairs_size_) {
                groups.push_back(image_pairs);
                image_pairs.clear();
                cur_pairs_size = 0;
            }
        }
        if (cur_pairs_size != 0) groups.push_back(image_pairs);
    }
    if (!groups.empty()) {
        if (is_preemtive_) PreemptivelyFilterGroups(&groups);
        MatchImagePairGroups(groups);
    }
    CloseDatabaseAndDevice();
}

// PreemptivelyFilterImagePairs (src/Feature/FeatureMatching.cpp:148-179) for every group of the run at once
void BruteFeatureMatcher::PreemptivelyFilterGroups(std::vector<std::vector<std::pair<image_t, image_t>>>* groups) {
    std::vector<std::pair<image_t, image_t>> all;
    for (const auto& g : *groups) all.insert(all.end(), g.begin(), g.end());
    std::vector<char> keep;
    {
        Lap l(&g_clock.preemptive);
        keep = PreemptiveKeepFlags(all);
    }
    size_t at = 0;
    for (auto& g : *groups) {
        std::vector<std::pair<image_t, image_t>> filtered;
        for (const auto& image_pair : g)
            if (keep[at++]) filtered.push_back(image_pair);
        g.swap(filtered);
    }
}

// ComputeCrossMatches / ComputeMatches on the two 100-row top-scale subsets, no distance filter (FeatureMatching.cpp:163-172), keep the
// pair iff it yields >= preemtive_min_num_matches_ matches.  Every device takes a contiguous share of the pairs: its own subsets
// (device-side gathers of the resident rows), one streaming series, only the counts come back.
std::vector<char> BruteFeatureMatcher::PreemptiveKeepFlags(const std::vector<std::pair<image_t, image_t>>& image_pairs) {
    std::vector<char> keep(image_pairs.size(), 0);
    if (image_pairs.empty()) return keep;
    const size_t P = image_pairs.size(), G = devices_.size();
    // calling thread (SQLite): the images resident, their keypoints read, the rows of every subset chosen
    std::vector<image_t> ids;
    for (const auto& pr : image_pairs) {
        ids.push_back(pr.first);
        ids.push_back(pr.second);
    }
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    std::map<image_t, std::vector<int32_t>> rows_of;
    for (image_t id : ids) {
        bool need = false;
        for (const Device& dev : devices_) need |= dev.top_scale.count(id) == 0;
        if (!need) continue;
        // the reference re-reads the descriptors here (FeatureMatching.cpp:181-196); they are resident on the
        // device already (or become so now), so only the keypoint scales are needed on the host
        EnsureResident(id);
        const std::vector<KeyPoint>& kpts = KeyPointsOf(id);
        static_assert(sizeof(KeyPoint) == 16, "KeyPoint must be 4 packed floats");
        std::vector<int32_t> idx(kpts.size() + 1);
        int count = 0;
        const int rc = msfm_topscale_select(reinterpret_cast<const float*>(kpts.data()), (int)kpts.size(), preemtive_num_features_, idx.data(), &count);
        if (rc != MSFM_OK) Die(ctx_, "msfm_topscale_select", rc);
        idx.resize((size_t)count);
        rows_of[id] = std::move(idx);
    }
    msfm_match_params prm;
    prm.ratio = (float)distance_ratio_;
    prm.cross_check = cross_check_ ? 1 : 0;
    prm.max_distance = __builtin_huge_val();
    std::vector<int> status(G, MSFM_OK);
    std::vector<std::string> what(G);
    auto share = [&](size_t g) {
        Device& dev = devices_[g];
        const size_t begin = P * g / G, end = P * (g + 1) / G;
        if (begin == end) return;
        auto fail = [&](const char* w, int rc) {
            status[g] = rc;
            what[g] = std::string(w) + ": " + msfm_last_error(dev.ctx);
        };
        std::vector<int32_t> slots;
        slots.reserve(2 * (end - begin));
        for (size_t p = begin; p < end; ++p)
            for (image_t id : {image_pairs[p].first, image_pairs[p].second}) {
                const int slot = MSFM_MAX_IMAGES + id;   // auxiliary store slot of this image's subset
                if (!dev.top_scale.count(id)) {
                    const std::vector<int32_t>& rows = rows_of.at(id);
                    const int rc = msfm_subset_image(dev.ctx, id, slot, rows.data(), (int)rows.size());
                    if (rc != MSFM_OK) return fail("msfm_subset_image", rc);
                    dev.top_scale.insert(id);
                }
                slots.push_back(slot);
            }
        int rc = msfm_match_pairs_begin(dev.ctx, slots.data(), (int)(end - begin), &prm, 0, nullptr);
        if (rc != MSFM_OK) return fail("msfm_match_pairs_begin (pre-emptive filter)", rc);
        while (true) {
            msfm_chunk ch;
            rc = msfm_match_pairs_next(dev.ctx, &ch);
            if (rc != MSFM_OK) return fail("msfm_match_pairs_next (pre-emptive filter)", rc);
            if (ch.n_pairs == 0) break;
            for (int p = 0; p < ch.n_pairs; ++p)
                keep[begin + (size_t)ch.first_pair + (size_t)p] = (ch.offsets[p + 1] - ch.offsets[p] >= preemtive_min_num_matches_) ? 1 : 0;
        }
    };
    std::vector<std::thread> threads;
    for (size_t g = 1; g < G; ++g) threads.emplace_back(share, g);
    share(0);
    for (auto& t : threads) t.join();
    for (size_t g = 0; g < G; ++g)
        if (status[g] != MSFM_OK) {
            std::fprintf(stderr, "ComputeMatches: %s (status %d)\n", what[g].c_str(), status[g]);
            std::exit(EXIT_FAILURE);
        }
    return keep;
}

std::vector<std::pair<image_t, image_t>> BruteFeatureMatcher::PreemptivelyFilterImagePairs(
    std::vector<std::pair<image_t, image_t>> image_pairs) {
    std::vector<std::pair<image_t, image_t>> filtered_image_pairs;
    const std::vector<char> keep = PreemptiveKeepFlags(image_pairs);
    for (size_t p = 0; p < image_pairs.size(); ++p)
        if (keep[p]) filtered_image_pairs.push_back(image_pairs[p]);
    return filtered_image_pairs;
}

int BruteFeatureMatcher::GetTopScaleDescriptors(const image_t& image_id) {
    const int slot = MSFM_MAX_IMAGES + image_id;  // auxiliary store slot of this image's subset
    if (HasTopScaleDescriptorsCache(image_id)) return slot;
    EnsureResident(image_id);
    const std::vector<KeyPoint>& kpts = KeyPointsOf(image_id);
    std::vector<int32_t> idx(kpts.size() + 1);
    int count = 0;
    const int rc = msfm_topscale_select(reinterpret_cast<const float*>(kpts.data()), (int)kpts.size(),
                                        preemtive_num_features_, idx.data(), &count);
    if (rc != MSFM_OK) Die(ctx_, "msfm_topscale_select", rc);
    MSFM_CALL(ctx_, msfm_subset_image(ctx_, image_id, slot, idx.data(), count));
    devices_[0].top_scale.insert(image_id);
    return slot;
}

bool BruteFeatureMatcher::HasTopScaleDescriptorsCache(const image_t& image_id) {
    return !devices_.empty() && devices_[0].top_scale.count(image_id) > 0;
}

}  // namespace MonocularSfM
