// FeatureMatching.cpp -- see FeatureMatching.h.  Control flow follows the reference's
// src/Feature/FeatureMatching.cpp (pair order, batch boundaries, transactions, resume-by-row,
// stdout lines); the per-pair arithmetic runs on the GPU through include/msfm_match.h.
#include "FeatureMatching.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>

#include "GeometricVerification.h"
#include "MatchEmission.h"
#include "Timer.h"

namespace MonocularSfM {

namespace {
// MSFM_CLI_TIMING=1: wall-clock per phase on stderr when the matcher closes
struct PhaseClock {
    double exist = 0, read_desc = 0, device = 0, read_kp = 0, verify = 0, emit = 0, preemptive = 0, open_dev = 0, close_dev = 0;
    bool on = std::getenv("MSFM_CLI_TIMING") != nullptr;
    void Report() const {
        if (!on || exist + read_desc + device + emit == 0) return;
        std::fprintf(stderr, "[msfm timing] exist-check %.3f s | read descriptors + upload %.3f s | device match + fetch %.3f s | "
                             "pre-emptive filter %.3f s | read keypoints %.3f s | verification %.3f s | stdout + WriteMatches %.3f s | "
                             "open database + device %.3f s | close %.3f s\n",
                     exist, read_desc, device, preemptive, read_kp, verify, emit, open_dev, close_dev);
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
    if (!ctx_) {
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
        for (size_t g = 0; g < devs.size(); ++g) {
            msfm_ctx* c = nullptr;
            const int rc = msfm_create(devs[g], &c);
            if (rc != MSFM_OK) {
                std::fprintf(stderr, "ComputeMatches: no usable gfx950 GPU at ordinal %d (msfm_create status %d); there is no CPU fallback\n", devs[g], rc);
                std::exit(EXIT_FAILURE);
            }
            if (const char* o = std::getenv("MSFM_ACCUM_ORDER")) MSFM_CALL(c, msfm_set_accum_order(c, std::atoi(o)));
            if (g == 0) ctx_ = c;
            else extra_ctxs_.push_back(c);
        }
        extra_resident_.assign(extra_ctxs_.size(), std::set<image_t>());
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
    if (ctx_) {
        msfm_destroy(ctx_);
        ctx_ = nullptr;
    }
    for (msfm_ctx* c : extra_ctxs_) msfm_destroy(c);
    extra_ctxs_.clear();
    extra_resident_.clear();
    descriptor_cache_.clear();
    resident_.clear();
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

void FeatureMatcher::EnsureResidentOn(size_t extra_index, image_t image_id) {
    std::set<image_t>& have = extra_resident_[extra_index];
    if (have.count(image_id)) return;
    EnsureResident(image_id);  // fills the host cache
    msfm_ctx* c = extra_ctxs_[extra_index];
    const Descriptors& d = descriptor_cache_.at(image_id);
    MSFM_CALL(c, msfm_upload_image(c, image_id, d.data.data(), d.rows, d.rows ? d.cols : MSFM_DIM, MSFM_DTYPE_F32));
    if (geometric_verification_ && !verification_on_host_) {
        const std::vector<KeyPoint>& kpts = KeyPointsOf(image_id);
        MSFM_CALL(c, msfm_upload_keypoints(c, image_id, reinterpret_cast<const float*>(kpts.data()), (int)kpts.size(), 4));
    }
    have.insert(image_id);
}

void FeatureMatcher::EnsureResident(image_t image_id) {
    if (resident_.count(image_id)) return;
    Descriptors read = database_->ReadDescriptors(image_id);
    const Descriptors& d = extra_ctxs_.empty() ? read : (descriptor_cache_[image_id] = std::move(read));
    MSFM_CALL(ctx_, msfm_upload_image(ctx_, image_id, d.data.data(), d.rows, d.rows ? d.cols : MSFM_DIM, MSFM_DTYPE_F32));
    if (geometric_verification_ && !verification_on_host_) {
        // the device verifies: it needs the keypoint coordinates next to the descriptors
        const std::vector<KeyPoint>& kpts = KeyPointsOf(image_id);
        static_assert(sizeof(KeyPoint) == 16, "KeyPoint must be 4 packed floats");
        MSFM_CALL(ctx_, msfm_upload_keypoints(ctx_, image_id, reinterpret_cast<const float*>(kpts.data()), (int)kpts.size(), 4));
    }
    resident_.insert(image_id);
}

void FeatureMatcher::PreloadAllImages() {
    if (const char* e = std::getenv("MSFM_BULK_LOAD"))
        if (e[0] == '0') return;
    Lap l(&g_clock.read_desc);
    struct Sink {
        FeatureMatcher* self;
        bool keypoints;
    } sink{this, false};
    // visitors run on this thread while SQLite holds the row: upload from its buffer, keep nothing
    auto visit = [](void* user, image_t id, const void* data, size_t rows, size_t cols, size_t elem) {
        Sink* s = static_cast<Sink*>(user);
        FeatureMatcher* m = s->self;
        if (id < 0 || id >= MSFM_MAX_IMAGES) return;
        std::vector<msfm_ctx*> ctxs(1, m->ctx_);
        ctxs.insert(ctxs.end(), m->extra_ctxs_.begin(), m->extra_ctxs_.end());
        if (!s->keypoints) {
            if (m->resident_.count(id)) return;
            for (msfm_ctx* c : ctxs)
                MSFM_CALL(c, msfm_upload_image(c, id, data, (int)rows, rows ? (int)cols : MSFM_DIM, elem == 1 ? MSFM_DTYPE_U8 : MSFM_DTYPE_F32));
            m->resident_.insert(id);
            for (auto& have : m->extra_resident_) have.insert(id);
        } else {
            // Database::ReadKeyPoints asserts cols == 4 (x, y, size, angle); a narrower blob would be over-read below
            if (rows > 0 && cols != 4) {
                std::cerr << "ERROR: keypoints of image " << id << " have " << cols << " columns, expected 4" << std::endl;
                std::exit(EXIT_FAILURE);
            }
            std::vector<KeyPoint>& kps = m->keypoints_cache_[id];
            kps.resize(rows);
            if (rows) std::memcpy(kps.data(), data, rows * sizeof(KeyPoint));
            if (m->geometric_verification_ && !m->verification_on_host_ && m->resident_.count(id))
                for (msfm_ctx* c : ctxs)
                    MSFM_CALL(c, msfm_upload_keypoints(c, id, static_cast<const float*>(data), (int)rows, 4));
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
    // the uploads above only copied: build the store now (classification, one allocation, layout kernels), inside this phase's clock
    MSFM_CALL(ctx_, msfm_finalize_store(ctx_));
    for (msfm_ctx* c : extra_ctxs_) MSFM_CALL(c, msfm_finalize_store(c));
    bulk_loaded_ = true;
}

void FeatureMatcher::MatchImagePairs(const std::vector<std::pair<image_t, image_t>>& image_pairs) {
    MatchImagePairGroups({image_pairs});
}

// The reference calls MatchImagePairs once per group of <= 100 pairs: one transaction, and per pair either
// the "Existing, Continue!" line or match -> filter -> verify -> stdout -> WriteMatches
// (src/Feature/FeatureMatching.cpp:10-73).  Here several groups are computed together -- one batched GPU
// call, one parallel verification pass -- and then EMITTED group by group, pair by pair, in the reference's
// order, so stdout, the rows and the transaction boundaries are the same while the device sees thousands
// of pairs per launch instead of 100.
void FeatureMatcher::MatchImagePairGroups(const std::vector<std::vector<std::pair<image_t, image_t>>>& groups) {
    struct Slot { int todo_index; };  // -1: a row exists (or an earlier group of this call writes it)
    std::vector<std::vector<Slot>> slots(groups.size());
    std::vector<int32_t> todo;
    std::set<std::pair<image_t, image_t>> scheduled;
    {
    Lap lap(&g_clock.exist);
    for (size_t g = 0; g < groups.size(); ++g) {
        slots[g].resize(groups[g].size());
        for (size_t k = 0; k < groups[g].size(); ++k) {
            const image_t image_id1 = groups[g][k].first, image_id2 = groups[g][k].second;
            const std::pair<image_t, image_t> key(std::min(image_id1, image_id2), std::max(image_id1, image_id2));
            if (scheduled.count(key) || database_->ExistMatches(image_id1, image_id2)) {
                slots[g][k].todo_index = -1;
                continue;
            }
            scheduled.insert(key);
            slots[g][k].todo_index = (int)(todo.size() / 2);
            todo.push_back(image_id1);
            todo.push_back(image_id2);
        }
    }
    }
    const int P = (int)(todo.size() / 2);
    std::vector<std::vector<DMatch>> verified((size_t)P);
    std::vector<double> verify_seconds((size_t)P, 0.0);
    double gpu_seconds_per_pair = 0.0;
    if (P > 0) {
        Timer timer;
        timer.Start();
        {
            Lap l(&g_clock.read_desc);
            for (int32_t id : todo) EnsureResident(id);
        }
        msfm_match_params prm;
        prm.ratio = (float)distance_ratio_;  // ComputeCrossMatches takes `const float distance_ratio`
        prm.cross_check = cross_check_ ? 1 : 0;
        prm.max_distance = max_distance_;
        std::vector<int64_t> offs((size_t)P + 1);
        const int32_t* qt = nullptr;   // the library's page-locked result buffers: valid until the next matching call
        const float* dist = nullptr;
        std::vector<int32_t> qt_merged;   // several devices: their lists concatenated in pair order
        std::vector<float> dist_merged;
        const bool verify_on_device = geometric_verification_ && !verification_on_host_;
        // returns the status: device worker threads must not exit() the process while their siblings run
        auto run_on = [&](msfm_ctx* c, const int32_t* pairs, int n, int64_t* out_offs) -> int {
            if (verify_on_device) return msfm_match_pairs_verified(c, pairs, n, &prm, nullptr, out_offs);  // FilterMatches' constants
            return msfm_match_pairs(c, pairs, n, &prm, out_offs);
        };
        if (extra_ctxs_.empty()) {
            Lap l(&g_clock.device);
            MSFM_CALL(ctx_, run_on(ctx_, todo.data(), P, offs.data()));
            MSFM_CALL(ctx_, msfm_view_matches(ctx_, &qt, &dist, nullptr));
        } else {
            // contiguous ranges of equal cost sum n1 * n2, one per device; every device needs the images of its range
            const size_t G = 1 + extra_ctxs_.size();
            std::vector<double> cum((size_t)P + 1, 0.0);
            for (int p = 0; p < P; ++p) {
                int n1 = 0, n2 = 0;
                (void)msfm_image_rows(ctx_, todo[2 * (size_t)p], &n1);
                (void)msfm_image_rows(ctx_, todo[2 * (size_t)p + 1], &n2);
                cum[(size_t)p + 1] = cum[(size_t)p] + (double)n1 * n2 + 1.0;
            }
            std::vector<int> cut(G + 1, P);
            cut[0] = 0;
            for (size_t g = 1; g < G; ++g)
                cut[g] = (int)(std::lower_bound(cum.begin(), cum.end(), cum[(size_t)P] * (double)g / (double)G) - cum.begin());
            for (size_t g = 1; g <= G; ++g) cut[g] = std::max(cut[g], cut[g - 1]);
            {
                Lap l(&g_clock.read_desc);
                for (size_t g = 1; g < G; ++g)
                    for (int p = cut[g]; p < cut[g + 1]; ++p) {
                        EnsureResidentOn(g - 1, todo[2 * (size_t)p]);
                        EnsureResidentOn(g - 1, todo[2 * (size_t)p + 1]);
                    }
            }
            Lap l(&g_clock.device);
            std::vector<std::vector<int64_t>> part_offs(G);
            std::vector<std::thread> workers;
            std::vector<int> status(G, MSFM_OK);
            for (size_t g = 0; g < G; ++g) {
                part_offs[g].assign((size_t)(cut[g + 1] - cut[g]) + 1, 0);
                msfm_ctx* c = g == 0 ? ctx_ : extra_ctxs_[g - 1];
                workers.emplace_back([&, g, c] { status[g] = run_on(c, todo.data() + 2 * (size_t)cut[g], cut[g + 1] - cut[g], part_offs[g].data()); });
            }
            for (auto& w : workers) w.join();
            for (size_t g = 0; g < G; ++g)
                if (status[g] != MSFM_OK) Die(g == 0 ? ctx_ : extra_ctxs_[g - 1], "msfm_match_pairs (device worker)", status[g]);
            int64_t total = 0;
            for (size_t g = 0; g < G; ++g) {
                for (int p = cut[g]; p < cut[g + 1]; ++p) offs[(size_t)p + 1] = total + part_offs[g][(size_t)(p - cut[g]) + 1];
                total += part_offs[g].back();
            }
            qt_merged.resize((size_t)total * 2 + 2);
            dist_merged.resize((size_t)total + 1);
            for (size_t g = 0; g < G; ++g) {
                msfm_ctx* c = g == 0 ? ctx_ : extra_ctxs_[g - 1];
                const int32_t* q = nullptr;
                const float* dd = nullptr;
                int64_t cnt = 0;
                MSFM_CALL(c, msfm_view_matches(c, &q, &dd, &cnt));
                const size_t at = (size_t)offs[(size_t)cut[g]];
                if (cnt > 0) {
                    std::memcpy(qt_merged.data() + 2 * at, q, (size_t)cnt * 8);
                    std::memcpy(dist_merged.data() + at, dd, (size_t)cnt * 4);
                }
            }
            qt = qt_merged.data();
            dist = dist_merged.data();
        }
        gpu_seconds_per_pair = timer.ElapsedSeconds() / P;

        // Geometric verification (FeatureUtils::FilterMatches) already happened on the device, unless the host
        // twin was asked for: then keypoints are read once per image (SQLite handle: this thread only) and
        // the pairs are verified on all host cores.
        const bool host_verify = geometric_verification_ && verification_on_host_;
        if (host_verify) {
            Lap l(&g_clock.read_kp);
            for (int32_t id : todo) (void)KeyPointsOf(id);
        }
        Lap lv(&g_clock.verify);
        auto verify_pair = [&](int p) {
            Timer pair_timer;
            pair_timer.Start();
            const image_t image_id1 = todo[2 * (size_t)p], image_id2 = todo[2 * (size_t)p + 1];
            std::vector<DMatch> prune_matches((size_t)(offs[(size_t)p + 1] - offs[(size_t)p]));
            for (size_t i = 0; i < prune_matches.size(); ++i) {
                const size_t k = (size_t)offs[(size_t)p] + i;
                prune_matches[i].queryIdx = qt[2 * k];
                prune_matches[i].trainIdx = qt[2 * k + 1];
                prune_matches[i].distance = dist[k];
            }
            if (host_verify)
                FilterMatches(keypoints_cache_.at(image_id1), keypoints_cache_.at(image_id2), prune_matches, &verified[(size_t)p]);
            else
                verified[(size_t)p].swap(prune_matches);
            verify_seconds[(size_t)p] = pair_timer.ElapsedSeconds();
        };
        // (the host twin of FilterMatches runs RANSAC per pair: all cores; otherwise this loop only copies the lists -- starting a
        // thread per core of a 256-core host cost 10 ms for 1 ms of copying)
        const int nthreads = host_verify ? std::max(1, std::min<int>(P / 4 + 1, (int)std::thread::hardware_concurrency())) : 1;
        std::atomic<int> next(0);
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t)
            pool.emplace_back([&] { for (int p; (p = next.fetch_add(1)) < P;) verify_pair(p); });
        for (int p; (p = next.fetch_add(1)) < P;) verify_pair(p);
        for (auto& th : pool) th.join();
    }
    // emission: the reference's order, one transaction per group
    Lap le(&g_clock.emit);
    static const EmissionOptions emission = EmissionOptions::FromEnvironment();
    std::string out;
    char buf[160];
    static const bool trace_txn = std::getenv("MSFM_TRACE_TRANSACTIONS") != nullptr;  // tests: one line per transaction
    for (size_t g = 0; g < groups.size(); ++g) {
        database_->BeginTransaction();
        if (trace_txn) std::fprintf(stderr, "[msfm txn] %zu\n", groups[g].size());
        out.clear();
        for (size_t k = 0; k < groups[g].size(); ++k) {
            const image_t image_id1 = groups[g][k].first, image_id2 = groups[g][k].second;
            const int p = slots[g][k].todo_index;
            if (p < 0) {
                std::snprintf(buf, sizeof(buf), "Compute Matches %d - %d Existing, Continue!\n", image_id1, image_id2);
                out += buf;
                continue;
            }
            ApplyEmissionOptions(emission, image_id1, image_id2, &verified[(size_t)p]);
            std::snprintf(buf, sizeof(buf), "Compute Matches %d - %d ... \n\t matches num : %zu\n\t ", image_id1, image_id2,
                          verified[(size_t)p].size());
            out += buf;
            out += Timer::Format(gpu_seconds_per_pair + verify_seconds[(size_t)p], "seconds");
            out += "\n";
            database_->WriteMatches(image_id1, image_id2, verified[(size_t)p]);
        }
        std::cout << out << std::flush;
        database_->EndTransaction();
    }
}

namespace {
// Pairs computed per device call (the reference's groups are <= 100 pairs; they are emitted unchanged afterwards).
// Finished groups reach the database once per super-batch, so this is also what an interrupted run can lose:
// MSFM_SUPER_BATCH_PAIRS lowers it (tests cross the boundary with tiny values).
size_t SuperBatchPairs() {
    static const size_t v = [] {
        const char* e = std::getenv("MSFM_SUPER_BATCH_PAIRS");
        const long long x = e ? std::atoll(e) : 0;
        return x > 0 ? (size_t)x : (size_t)16384;
    }();
    return v;
}
}

void SequentialFeatureMatcher::RunMatching() {
    OpenDatabaseAndDevice();
    PreloadAllImages();
    const std::vector<Database::Image> images = database_->ReadAllImages();
    std::vector<std::vector<std::pair<image_t, image_t>>> groups;
    size_t pending = 0;
    for (size_t i = 1; i < images.size(); ++i) {
        std::vector<std::pair<image_t, image_t>> image_pairs;
        for (int k = 1; k <= overlap_; ++k) {
            const int j = (int)i - k;
            if (j < 0) break;
            image_pairs.emplace_back((image_t)i, (image_t)j);
        }
        pending += image_pairs.size();
        groups.push_back(std::move(image_pairs));
        if (pending >= SuperBatchPairs()) {
            MatchImagePairGroups(groups);
            groups.clear();
            pending = 0;
        }
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
    size_t pending = 0;
    auto flush = [&]() {
        if (groups.empty()) return;
        if (is_preemtive_) PreemptivelyFilterGroups(&groups);
        MatchImagePairGroups(groups);
        groups.clear();
        pending = 0;
    };
    for (size_t i = 0; i < images.size(); ++i) {
        std::vector<std::pair<image_t, image_t>> image_pairs;
        int cur_pairs_size = 0;
        for (size_t j = 0; j < i; ++j) {
            image_pairs.emplace_back((image_t)i, (image_t)j);
            cur_pairs_size += 1;
            if (cur_pairs_size == max_pairs_size_) {
                pending += image_pairs.size();
                groups.push_back(image_pairs);
                image_pairs.clear();
                cur_pairs_size = 0;
            }
        }
        if (cur_pairs_size != 0) {
            pending += image_pairs.size();
            groups.push_back(image_pairs);
        }
        if (pending >= SuperBatchPairs()) flush();
    }
    flush();
    CloseDatabaseAndDevice();
}

// PreemptivelyFilterImagePairs (src/Feature/FeatureMatching.cpp:148-179) for every group of a super-batch in
// one device call
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

std::vector<char> BruteFeatureMatcher::PreemptiveKeepFlags(const std::vector<std::pair<image_t, image_t>>& image_pairs) {
    std::vector<char> keep(image_pairs.size(), 0);
    if (image_pairs.empty()) return keep;
    std::vector<int32_t> slots;
    for (const auto& image_pair : image_pairs) {
        slots.push_back(GetTopScaleDescriptors(image_pair.first));
        slots.push_back(GetTopScaleDescriptors(image_pair.second));
    }
    // ComputeCrossMatches / ComputeMatches on the two 100-row subsets, no distance filter here
    // (src/Feature/FeatureMatching.cpp:163-172)
    msfm_match_params prm;
    prm.ratio = (float)distance_ratio_;
    prm.cross_check = cross_check_ ? 1 : 0;
    prm.max_distance = __builtin_huge_val();
    const int P = (int)image_pairs.size();
    std::vector<int64_t> offs((size_t)P + 1);
    MSFM_CALL(ctx_, msfm_match_pairs(ctx_, slots.data(), P, &prm, offs.data()));
    for (int p = 0; p < P; ++p) keep[(size_t)p] = (offs[(size_t)p + 1] - offs[(size_t)p] >= preemtive_min_num_matches_) ? 1 : 0;
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
    // the reference re-reads the descriptors here (FeatureMatching.cpp:181-196); they are resident on the
    // device already (or become so now), so only the keypoint scales are needed on the host
    EnsureResident(image_id);
    const std::vector<KeyPoint>& kpts = KeyPointsOf(image_id);
    static_assert(sizeof(KeyPoint) == 16, "KeyPoint must be 4 packed floats");
    std::vector<int32_t> idx(kpts.size() + 1);
    int count = 0;
    const int rc = msfm_topscale_select(reinterpret_cast<const float*>(kpts.data()), (int)kpts.size(),
                                        preemtive_num_features_, idx.data(), &count);
    if (rc != MSFM_OK) Die(ctx_, "msfm_topscale_select", rc);
    MSFM_CALL(ctx_, msfm_subset_image(ctx_, image_id, slot, idx.data(), count));
    top_scale_descriptors_cache_.insert(image_id);
    return slot;
}

bool BruteFeatureMatcher::HasTopScaleDescriptorsCache(const image_t& image_id) {
    return top_scale_descriptors_cache_.count(image_id) > 0;
}

}  // namespace MonocularSfM
