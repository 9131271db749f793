// msfm_match.hip -- host side of the C ABI declared in include/msfm_match.h.
//
// Owns the device-resident descriptor store, schedules image pairs onto the gfx950 kernels in
// msfm_kernels.hip.h and returns match lists.  Mirrors what FeatureMatcher::MatchImagePairs
// (src/Feature/FeatureMatching.cpp:10-73 of the reference) does between its two
// Database::ReadDescriptors calls and FeatureUtils::FilterMatches, without the per-pair
// descriptor re-read.  No CPU fallback: every entry point that needs the GPU fails loudly
// when there is none.
#include "msfm_match.h"
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
    hipError_t ensure_keep(size_t bytes, size_t keep, hipStream_t stream) {
        if (bytes <= cap) return hipSuccess;
        const size_t want = bytes + bytes / 2 + 4096;
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
    hipError_t ensure(size_t bytes, size_t keep) {
        if (bytes <= cap) return hipSuccess;
        size_t want = bytes + bytes / 2 + 4096;
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

struct Image {
    int n = -1;  // -1: not uploaded
    int nblk = 0;    // 128-row blocks holding data
    int nalloc = 0;  // allocated blocks (even: the prefilter walks 256-row A blocks); padding is zero-filled
    float* panel = nullptr;
    float* raw = nullptr;
    // prefilter operands: fp16 rows of 272 B (128 halfs + the norm quadruple of the ninth MFMA k-step), row norms
    // (+inf padded), maxima
    _Float16* h16 = nullptr;
    float* nrm = nullptr;
    float c = 1.f;            // scale of the quadruples (power of two)
    float nrm_max = 0.f, abs_max = 0.f;
    bool pf_safe = false;
    // keypoint coordinates (x, y) for the geometric verification; nk = -1: not uploaded
    float2* kxy = nullptr;
    int nk = -1;
};

void free_image(Image& im) {
    if (im.panel) (void)hipFree(im.panel);
    if (im.raw) (void)hipFree(im.raw);
    if (im.h16) (void)hipFree(im.h16);
    if (im.nrm) (void)hipFree(im.nrm);
    if (im.kxy) (void)hipFree(im.kxy);
    im = Image{};
}

constexpr int kSlots = 2 * MSFM_MAX_IMAGES + 2;  // ids >= MSFM_MAX_IMAGES: auxiliary (top-scale subsets, two operator-level scratch slots)
// sub-batches of msfm_match_pairs are bounded by the partial-result scratch (4-byte units: ~48 GiB of the 288 GB) and a pair count
constexpr long long kDefaultScratchElems = (long long)12 << 30;
constexpr int kDefaultMaxPairsPerBatch = 16384;

}  // namespace

struct msfm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int order = MSFM_ORDER_SSE4X4;
    int cu_count = 0, clock_mhz = 0;
    char dev_name[256] = {0};
    std::vector<Image> images;
    std::string err;

    DevBuf d_pairs, d_items, d_stage;
    DevBuf d_rp_s0, d_rp_i0, d_rp_s1, d_cp_s0, d_cp_i0, d_cp_s1;
    DevBuf d_k_i0, d_k_d0, d_k_d1;
    DevBuf d_st_qt, d_st_d, d_counts, d_offsets, d_out_qt, d_out_d;
    DevBuf d_fix_count, d_fix_list;
    int fix_cap = 1 << 16;            // entries of the sqrt-space tie queue; grows on overflow (the sub-batch is re-run)
    int fix_cap_eff = 0;              // capacity handed to the kernels of the current batch (0: ties need no fix-up)
    // sub-batch limits of msfm_match_pairs (msfm_set_limits / MSFM_MAX_PAIRS_PER_BATCH / MSFM_SCRATCH_MIB)
    int max_pairs_per_batch = kDefaultMaxPairsPerBatch;
    long long scratch_elems = kDefaultScratchElems;
    // prefilter path
    int prefilter = 1;
    DevBuf d_pf, d_tu, d_tv, d_cand, d_cand_s, d_cand_count, d_best, d_second, d_maxima;
    DevBuf d_live_cnt, d_cmp_h, d_cmp_tu, d_live_idx, d_row_pair, d_cand_pair, d_active, d_vpairs, d_vpf, d_vitems, d_jobs, d_lists;
    // geometric verification
    DevBuf d_vf_pairs, d_vf_x1, d_vf_y1, d_vf_x2, d_vf_y2, d_vf_hyp, d_vf_best_it, d_vf_best_count, d_vf_flags,
        d_st2_qt, d_st2_d, d_counts2;

    // results of the last msfm_match_pairs call
    bool have_results = false;
    std::vector<int64_t> res_offsets;
    PinnedBuf res_qt, res_dist;  // (q, t) int32 pairs and distances of res_count matches
    size_t res_count = 0;

    msfm_profile prof = {};
    std::vector<hipEvent_t> ev_pool;
};

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
            hipError_t e__ = hipStreamSynchronize((ctx)->stream);                 \
            std::fprintf(stderr, " %s\n", hipGetErrorString(e__));                \
        }                                                                         \
    } while (0)

struct Batch {
    std::vector<PairDesc> pairs;
    std::vector<PfPair> pf;
    std::vector<WorkItem> items;
    long long rp_elems = 0, cp_elems = 0, kf_elems = 0, kr_elems = 0, out_elems = 0, cand_elems = 0;
    int max_npad = 0;
    int64_t desc_pairs = 0;
    int64_t algo_bytes = 0;
};

// Work items of the pairs whose `path` matches.  Items of one pair are contiguous; the list is then
// interleaved over the 8 XCDs (workgroup b runs on XCD b % 8) so that the workgroups streaming the
// same B panels share one L2.
void build_items(Batch& b, int path) {
    std::vector<WorkItem> lin;
    for (size_t p = 0; p < b.pairs.size(); ++p) {
        PairDesc& pd = b.pairs[p];
        if (!pd.valid || pd.path != path) continue;
        for (int r = 0; r < pd.ranges; ++r) {
            const int t0 = (int)((long long)pd.b_tiles * r / pd.ranges);
            const int t1 = (int)((long long)pd.b_tiles * (r + 1) / pd.ranges);
            const int nab = path == 1 ? pd.a_blocks256 : pd.a_blocks;
            for (int ab = 0; ab < nab; ++ab) {
                WorkItem w = {};
                w.pair = (int)p;
                w.a_blk = ab;
                w.bt_begin = t0;
                w.bt_end = t1;
                w.range = r;
                lin.push_back(w);
            }
        }
    }
    const size_t n = lin.size();
    const size_t per = (n + 7) / 8;
    b.items.assign(per * 8, WorkItem{-1, 0, 0, 0, 0, {0, 0, 0}});
    for (size_t k = 0; k < n; ++k) {
        const size_t x = k / per, j = k % per;
        b.items[j * 8 + x] = lin[k];
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

int upload_pairs(msfm_ctx* ctx, Batch& b) {
    const size_t P = b.pairs.size();
    HIPCHK(ctx, ctx->d_pairs.ensure(P * sizeof(PairDesc)));
    HIPCHK(ctx, ctx->d_pf.ensure(P * sizeof(PfPair)));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_pairs.p, b.pairs.data(), P * sizeof(PairDesc), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_pf.p, b.pf.data(), P * sizeof(PfPair), hipMemcpyHostToDevice, ctx->stream));
    return MSFM_OK;
}

int upload_items(msfm_ctx* ctx, Batch& b) {
    HIPCHK(ctx, ctx->d_items.ensure(std::max<size_t>(1, b.items.size()) * sizeof(WorkItem)));
    if (!b.items.empty())
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_items.p, b.items.data(), b.items.size() * sizeof(WorkItem), hipMemcpyHostToDevice, ctx->stream));
    return MSFM_OK;
}

// XCD interleave of a linear item list (see build_items)
std::vector<WorkItem> interleave_items(const std::vector<WorkItem>& lin) {
    const size_t n = lin.size();
    const size_t per = (n + 7) / 8;
    std::vector<WorkItem> out(per * 8, WorkItem{-1, 0, 0, 0, 0, {0, 0, 0}});
    for (size_t k = 0; k < n; ++k) out[(k % per) * 8 + k / per] = lin[k];
    return out;
}

// ---- host-side plan of sweep 2 ----------------------------------------------------------------
struct VMember { int pair, dir, cnt; long long row; };                                   // one (pair, direction): a slice of a group's rows
struct VGroup { const _Float16* b_h; int dir; int first, count; long long row0, rows; };  // the pairs streaming one image in one direction
struct Sweep2Plan {
    std::vector<VMember> members;
    std::vector<VGroup> groups;
    std::vector<PairDesc> vpairs;   // per group: the compacted "pair" (A = concatenated live rows, B = the streamed image)
    std::vector<PfPair> vpf;
    std::vector<GatherJob> jobs;    // per member
    std::vector<CandList> lists;    // [0, P): dense lists of the pairs; [P, P+V): one per group
    std::vector<WorkItem> ditems, vitems;
    long long cand_elems = 0;
};

// Which candidate lists exist, where the compacted rows go, which work items sweep them.  `compact[p]`: pair p
// is swept through its compacted live rows (live[2p], live[2p+1] of them); the others get the dense sweep.
int plan_sweep2(msfm_ctx* ctx, Batch& b, const std::vector<char>& compact, const std::vector<int>& live, Sweep2Plan& plan) {
    const size_t P = b.pairs.size();
    // Compacted sweeps are GROUPED: the live rows of every pair that streams the same image in the same
    // direction are concatenated into one dense matrix (blocks are then full except for one tail per group;
    // a pair's own ~300 live rows would fill its last 256-row block to a fifth).
    std::vector<VMember>& members = plan.members;
    std::vector<VGroup>& groups = plan.groups;
    long long cmp_rows = 0;
    long long& cand_elems = plan.cand_elems;
    cand_elems = 0;
    {
        std::map<std::pair<const void*, int>, std::vector<VMember>> by_key;
        std::vector<std::pair<const void*, int>> key_order;
        for (size_t p = 0; p < P; ++p) {
            PairDesc& pd = b.pairs[p];
            PfPair& pp = b.pf[p];
            if (!pd.valid || !pp.use) continue;
            if (!compact[p]) {
                pp.cand_off = cand_elems;
                pp.cand_cap = 8 * (pd.n1 + pd.n2) + 1024;
                cand_elems += pp.cand_cap;
                continue;
            }
            for (int dir = 0; dir < 2; ++dir) {
                const int cnt = live[2 * p + dir];
                if (cnt == 0) continue;
                const std::pair<const void*, int> key(dir ? (const void*)pp.a_h : (const void*)pp.b_h, dir);
                auto it = by_key.find(key);
                if (it == by_key.end()) {
                    key_order.push_back(key);
                    it = by_key.emplace(key, std::vector<VMember>()).first;
                }
                it->second.push_back(VMember{(int)p, dir, cnt, 0});
            }
        }
        for (const auto& key : key_order) {
            std::vector<VMember>& ms = by_key[key];
            VGroup g{(const _Float16*)key.first, key.second, (int)members.size(), (int)ms.size(), cmp_rows, 0};
            for (VMember& m : ms) {
                m.row = cmp_rows + g.rows;
                g.rows += m.cnt;
                members.push_back(m);
            }
            cmp_rows += (g.rows + kPfWgRows - 1) / kPfWgRows * kPfWgRows;
            groups.push_back(g);
        }
    }
    const size_t V = groups.size();
    HIPCHK(ctx, ctx->d_cmp_h.ensure(std::max<long long>(1, cmp_rows) * kPfRowBytes));
    HIPCHK(ctx, ctx->d_cmp_tu.ensure(std::max<long long>(1, cmp_rows) * 4));
    HIPCHK(ctx, ctx->d_live_idx.ensure(std::max<long long>(1, cmp_rows) * 4));
    HIPCHK(ctx, ctx->d_row_pair.ensure(std::max<long long>(1, cmp_rows) * 4));
    std::vector<PairDesc>& vpairs = plan.vpairs;
    std::vector<PfPair>& vpf = plan.vpf;
    std::vector<GatherJob>& jobs = plan.jobs;
    std::vector<CandList>& lists = plan.lists;
    vpairs.assign(V, PairDesc{});
    vpf.assign(V, PfPair{});
    jobs.assign(members.size(), GatherJob{});
    lists.assign(P + V, CandList{});
    long long v_ablocks = 0;
    for (size_t v = 0; v < V; ++v) v_ablocks += (groups[v].rows + kPfWgRows - 1) / kPfWgRows;
    for (size_t p = 0; p < P; ++p) {
        const PfPair& pp = b.pf[p];
        lists[p] = CandList{(int)p, 0, pp.cand_off, (b.pairs[p].valid && pp.use && !compact[p]) ? pp.cand_cap : 0, 0, nullptr, nullptr};
    }
    std::vector<WorkItem> vlin;
    for (size_t v = 0; v < V; ++v) {
        const VGroup& g = groups[v];
        const VMember& m0 = members[(size_t)g.first];
        const PairDesc& pd = b.pairs[m0.pair];   // every member streams the same image: take its description from the first
        const PfPair& pp = b.pf[m0.pair];
        PairDesc& vd = vpairs[v];
        PfPair& vp = vpf[v];
        vd = PairDesc{};
        vd.n1 = (int)g.rows;
        vd.n2 = g.dir ? pd.n1 : pd.n2;
        vd.a_blocks256 = (int)((g.rows + kPfWgRows - 1) / kPfWgRows);
        vd.b_tiles = g.dir ? pd.a_blocks : pd.b_tiles;
        vd.n1pad = vd.a_blocks256 * kPfWgRows;
        vd.n2pad = g.dir ? pd.n1pad : pd.n2pad;
        vd.valid = 1;
        vd.path = 1;
        vd.ranges = 1;
        if (v_ablocks < 8LL * ctx->cu_count)
            vd.ranges = (int)std::max<long long>(1, std::min<long long>((8LL * ctx->cu_count + v_ablocks - 1) / v_ablocks, vd.b_tiles));
        vp = PfPair{};
        vp.a_h = ctx->d_cmp_h.as<_Float16>() + (size_t)g.row0 * kPfRowHalfs;
        vp.b_h = g.dir ? pp.a_h : pp.b_h;
        vp.b_nrm = g.dir ? pp.a_nrm : pp.b_nrm;
        vp.b_c = g.dir ? pp.a_c : pp.b_c;
        vp.a_c = g.dir ? pp.b_c : pp.a_c;
        vp.tu_off = g.row0;
        vp.cand_off = cand_elems;
        vp.cand_cap = (int)std::min<long long>(8 * g.rows + 1024, 1LL << 30);
        vp.use = 1;
        cand_elems += vp.cand_cap;
        for (int k = 0; k < g.count; ++k) {
            const VMember& m = members[(size_t)(g.first + k)];
            const PairDesc& mpd = b.pairs[m.pair];
            const PfPair& mpp = b.pf[m.pair];
            const bool last = k + 1 == g.count;
            jobs[(size_t)(g.first + k)] = GatherJob{m.dir ? mpp.b_h : mpp.a_h, m.dir ? mpp.b_nrm : mpp.a_nrm,
                                                    m.dir ? mpp.tv_off : mpp.tu_off, m.row,
                                                    last ? g.row0 + (long long)vd.n1pad : m.row + m.cnt,
                                                    m.dir ? mpd.n2 : mpd.n1, m.pair};
        }
        lists[P + v] = CandList{-1, 1 + g.dir, vp.cand_off, vp.cand_cap, 0, ctx->d_live_idx.as<int>() + g.row0,
                                ctx->d_row_pair.as<int>() + g.row0};
        for (int r = 0; r < vd.ranges; ++r) {
            const int t0 = (int)((long long)vd.b_tiles * r / vd.ranges), t1 = (int)((long long)vd.b_tiles * (r + 1) / vd.ranges);
            for (int ab = 0; ab < vd.a_blocks256; ++ab) vlin.push_back(WorkItem{(int)v, ab, t0, t1, r, {0, 0, 0}});
        }
        ctx->prof.sweep2_descriptor_pairs += (int64_t)vd.n1pad * vd.n2;
    }
    // dense items: the sweep-1 list minus the compacted pairs
    std::vector<WorkItem> dlin;
    for (const WorkItem& w : b.items)
        if (w.pair >= 0 && !compact[w.pair]) dlin.push_back(w);
    for (size_t p = 0; p < P; ++p)
        if (b.pairs[p].valid && b.pf[p].use && !compact[p]) ctx->prof.sweep2_descriptor_pairs += (int64_t)b.pairs[p].n1pad * b.pairs[p].n2;
    plan.ditems = dlin.empty() ? dlin : interleave_items(dlin);
    plan.vitems = vlin.empty() ? vlin : interleave_items(vlin);
    return MSFM_OK;
}


// MFMA prefilter + exact re-check for the pairs on path 1.  On return pairs whose candidate list
// overflowed have been moved to path 0.
//   sweep 1 (sweep_kernel<1>): S~ minima per row / column -> thresholds (with pruning for match lists)
//   sweep 2: dense pairs re-sweep everything (sweep_kernel<2>); pairs where pruning left few live rows
//            and columns sweep only those, compacted per direction (sweep_kernel<3>)
//   exact pinned-order S of the candidates, 64-bit atomicMin reduce, finalize
int run_prefilter(msfm_ctx* ctx, Batch& b, size_t ev_base, PruneParams prune) {
    const size_t P = b.pairs.size();
    HostClock hc;
    assign_partials(b, 1, 8 * ctx->cu_count);
    for (size_t p = 0; p < P; ++p) {
        b.pf[p].tu_off = b.pairs[p].kf_off;
        b.pf[p].tv_off = b.pairs[p].kr_off;  // same combined index space as the kNN arrays
        b.pf[p].cand_off = 0;
        b.pf[p].cand_cap = 0;
    }
    build_items(b, 1);
    if (b.items.empty()) return MSFM_OK;
    const long long kn = std::max<long long>(1, b.kf_elems + b.kr_elems);
    HIPCHK(ctx, ctx->d_rp_s0.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, ctx->d_rp_s1.ensure(std::max<long long>(1, b.rp_elems) * 4));
    // column partials of sweep 1: one float4 (four row-class maxima) per 512-row A block and column
    HIPCHK(ctx, ctx->d_cp_s0.ensure(std::max<long long>(1, b.cp_elems) * 16));
    HIPCHK(ctx, ctx->d_tu.ensure(kn * 4));
    HIPCHK(ctx, ctx->d_best.ensure(kn * 8));
    HIPCHK(ctx, ctx->d_second.ensure(kn * 8));
    HIPCHK(ctx, ctx->d_live_cnt.ensure(2 * P * 4));
    int rc = upload_pairs(ctx, b);
    if (rc != MSFM_OK) return rc;
    rc = upload_items(ctx, b);
    if (rc != MSFM_OK) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_best.p, 0xff, kn * 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_second.p, 0xff, kn * 8, ctx->stream));

    hipEvent_t e0 = get_event(ctx, ev_base), e1 = get_event(ctx, ev_base + 1);
    hipEvent_t e2 = get_event(ctx, ev_base + 2), e3 = get_event(ctx, ev_base + 3);
    if (!e0 || !e1 || !e2 || !e3) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
    const dim3 block(kPfThreads);
    const PairDesc* dp = ctx->d_pairs.as<PairDesc>();
    const PfPair* dpf = ctx->d_pf.as<PfPair>();
    float* tuv = ctx->d_tu.as<float>();  // rows at kf offsets, columns at kr offsets (one buffer)
    hc.lap("sweep-1 setup + uploads");
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(sweep_kernel<1>, dim3((unsigned)b.items.size()), block, kPfLdsBytes, ctx->stream, dp, dpf,
                       ctx->d_items.as<WorkItem>(), ctx->d_rp_s0.as<float>(), ctx->d_rp_s1.as<float>(),
                       ctx->d_cp_s0.as<float>(), (float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (int2*)nullptr, (unsigned long long*)nullptr);
    HIPCHK(ctx, hipGetLastError());
    DBGSYNC(ctx, "sweep_kernel<1>");
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    ctx->prof.approx_kernel_launches += 1;
    const dim3 mgrid((unsigned)((b.max_npad + 255) / 256), (unsigned)P);
    hipLaunchKernelGGL(pf_thresholds_kernel, mgrid, dim3(256), 0, ctx->stream, dp, dpf, ctx->d_rp_s0.as<float>(),
                       ctx->d_rp_s1.as<float>(), ctx->d_cp_s0.as<float>(), (const float*)nullptr, tuv, tuv, prune);
    HIPCHK(ctx, hipGetLastError());
    DBGSYNC(ctx, "pf_thresholds_kernel");
#ifdef MSFM_SWEEP_PROBE
    if (std::getenv("MSFM_DUMP_T")) {   // diagnostic build: thresholds and column class maxima of pair 0
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const PairDesc& pd = b.pairs[0];
        std::vector<float> t((size_t)pd.n1pad + pd.n2pad), cp((size_t)pd.a_blocks256 * pd.n2pad * 4);
        HIPCHK(ctx, hipMemcpy(t.data(), tuv + b.pf[0].tu_off, (size_t)pd.n1pad * 4, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(t.data() + pd.n1pad, tuv + b.pf[0].tv_off, (size_t)pd.n2pad * 4, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(cp.data(), ctx->d_cp_s0.as<float>() + pd.cp_off * 4, cp.size() * 4, hipMemcpyDeviceToHost));
        auto stat = [&](const char* what, const float* v, int n) {
            int ninf = 0; double sum = 0; int cnt = 0;
            for (int i = 0; i < n; ++i) { if (std::isinf(v[i])) ++ninf; else { sum += v[i]; ++cnt; } }
            std::fprintf(stderr, "[dump] %s: n %d, inf %d, mean finite %.4f\n", what, n, ninf, cnt ? sum / cnt : 0.0);
        };
        stat("T rows", t.data(), pd.n1);
        stat("T cols", t.data() + pd.n1pad, pd.n2);
        for (int e = 0; e < 3 && e < pd.n2; ++e)
            std::fprintf(stderr, "[dump] col %d classes (S-space): %.4f %.4f %.4f %.4f\n", e, -2 * cp[4 * e], -2 * cp[4 * e + 1], -2 * cp[4 * e + 2], -2 * cp[4 * e + 3]);
    }
#endif

    // ---- which pairs are worth compacting: needs the live counts on the host -------------------
    std::vector<int> live(2 * P, 0);
    std::vector<char> compact(P, 0);
    if (prune.prune) {
        hipLaunchKernelGGL(pf_count_live_kernel, dim3((unsigned)(2 * P)), dim3(256), 0, ctx->stream, dp, dpf,
                           (const float*)tuv, ctx->d_live_cnt.as<int>());
        HIPCHK(ctx, hipGetLastError());
    DBGSYNC(ctx, "pf_count_live_kernel");
        HIPCHK(ctx, hipMemcpyAsync(live.data(), ctx->d_live_cnt.p, 2 * P * 4, hipMemcpyDeviceToHost, ctx->stream));
        hc.lap("launch sweep 1 .. count");
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        hc.lap("wait for live counts (GPU)");
        for (size_t p = 0; p < P; ++p) {
            const PairDesc& pd = b.pairs[p];
            if (!pd.valid || !b.pf[p].use) continue;
            // both compact sweeps together must be clearly cheaper than the one dense sweep
            const long long dense_cost = (long long)pd.a_blocks256 * pd.b_tiles;
            const long long cmp_cost = (long long)((live[2 * p] + kPfWgRows - 1) / kPfWgRows) * pd.b_tiles +
                                       (long long)((live[2 * p + 1] + kPfWgRows - 1) / kPfWgRows) * pd.a_blocks;
            compact[p] = (2 * cmp_cost <= dense_cost) ? 1 : 0;
        }
    }

    // ---- sweep-2 descriptors -------------------------------------------------------------------
    Sweep2Plan plan;
    rc = plan_sweep2(ctx, b, compact, live, plan);
    if (rc != MSFM_OK) return rc;
    const size_t V = plan.groups.size();
    const std::vector<VMember>& members = plan.members;
    const std::vector<VGroup>& groups = plan.groups;
    const std::vector<PairDesc>& vpairs = plan.vpairs;
    const std::vector<PfPair>& vpf = plan.vpf;
    const std::vector<GatherJob>& jobs = plan.jobs;
    const std::vector<CandList>& lists = plan.lists;
    const std::vector<WorkItem>&ditems = plan.ditems, &vitems = plan.vitems;
    const long long cand_elems = plan.cand_elems;
    HIPCHK(ctx, ctx->d_cand.ensure(std::max<long long>(1, cand_elems) * sizeof(int2)));
    HIPCHK(ctx, ctx->d_cand_s.ensure(std::max<long long>(1, cand_elems) * 4));
    HIPCHK(ctx, ctx->d_cand_pair.ensure(std::max<long long>(1, cand_elems) * 4));
    HIPCHK(ctx, ctx->d_cand_count.ensure((P + V) * 8));
    HIPCHK(ctx, ctx->d_lists.ensure((P + V) * sizeof(CandList)));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_cand_count.p, 0, (P + V) * 8, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_lists.p, lists.data(), (P + V) * sizeof(CandList), hipMemcpyHostToDevice, ctx->stream));
    std::vector<int> active;  // the exact / reduce kernels only visit lists that can hold candidates
    for (size_t l = 0; l < P + V; ++l)
        if (lists[l].cap > 0) active.push_back((int)l);
    HIPCHK(ctx, ctx->d_active.ensure(std::max<size_t>(1, active.size()) * 4));
    if (!active.empty())
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_active.p, active.data(), active.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_pf.p, b.pf.data(), P * sizeof(PfPair), hipMemcpyHostToDevice, ctx->stream));  // cand_off / cap
    if (!ditems.empty()) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_items.p, ditems.data(), ditems.size() * sizeof(WorkItem), hipMemcpyHostToDevice, ctx->stream));
    }
    if (V > 0) {
        HIPCHK(ctx, ctx->d_vpairs.ensure(V * sizeof(PairDesc)));
        HIPCHK(ctx, ctx->d_vpf.ensure(V * sizeof(PfPair)));
        HIPCHK(ctx, ctx->d_jobs.ensure(jobs.size() * sizeof(GatherJob)));
        HIPCHK(ctx, ctx->d_vitems.ensure(vitems.size() * sizeof(WorkItem)));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_vpairs.p, vpairs.data(), V * sizeof(PairDesc), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_vpf.p, vpf.data(), V * sizeof(PfPair), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_jobs.p, jobs.data(), jobs.size() * sizeof(GatherJob), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_vitems.p, vitems.data(), vitems.size() * sizeof(WorkItem), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(pf_gather_live_kernel, dim3((unsigned)jobs.size()), dim3(256), 0, ctx->stream, ctx->d_jobs.as<GatherJob>(),
                           (const float*)tuv, ctx->d_live_idx.as<int>(), ctx->d_row_pair.as<int>(), ctx->d_cmp_tu.as<float>(),
                           ctx->d_cmp_h.as<_Float16>());
        HIPCHK(ctx, hipGetLastError());
    DBGSYNC(ctx, "pf_gather_live_kernel");
    }
    hc.lap("sweep-2 descriptors + uploads");
    HIPCHK(ctx, hipEventRecord(e2, ctx->stream));
    if (!ditems.empty()) {
        hipLaunchKernelGGL(sweep_kernel<2>, dim3((unsigned)ditems.size()), block, kPfLdsBytes, ctx->stream, dp, dpf,
                           ctx->d_items.as<WorkItem>(), (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (const float*)tuv, (const float*)tuv, ctx->d_cand.as<int2>(), ctx->d_cand_count.as<unsigned long long>());
        HIPCHK(ctx, hipGetLastError());
    DBGSYNC(ctx, "sweep_kernel<2>");
        ctx->prof.sweep2_launches += 1;
    }
    if (V > 0) {
        hipLaunchKernelGGL(sweep_kernel<3>, dim3((unsigned)vitems.size()), block, kPfLdsBytes, ctx->stream,
                           ctx->d_vpairs.as<PairDesc>(), ctx->d_vpf.as<PfPair>(), ctx->d_vitems.as<WorkItem>(), (float*)nullptr,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, ctx->d_cmp_tu.as<float>(), (const float*)nullptr,
                           ctx->d_cand.as<int2>(), ctx->d_cand_count.as<unsigned long long>() + P);
        HIPCHK(ctx, hipGetLastError());
    DBGSYNC(ctx, "sweep_kernel<3>");
        ctx->prof.sweep2_launches += 1;
    }
    HIPCHK(ctx, hipEventRecord(e3, ctx->stream));

    if (!active.empty()) {
        const CandList* dl = ctx->d_lists.as<CandList>();
        const dim3 cgrid(64, (unsigned)std::max<size_t>(1, active.size()));
        const int* dact = ctx->d_active.as<int>();
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL(pf_exact_candidates_kernel<0>, cgrid, dim3(256), 0, ctx->stream, dp, dl, dact, (const unsigned long long*)ctx->d_cand_count.as<unsigned long long>(),
                               ctx->d_cand.as<int2>(), ctx->d_cand_s.as<float>(), ctx->d_cand_pair.as<int>());
        else
            hipLaunchKernelGGL(pf_exact_candidates_kernel<1>, cgrid, dim3(256), 0, ctx->stream, dp, dl, dact, (const unsigned long long*)ctx->d_cand_count.as<unsigned long long>(),
                               ctx->d_cand.as<int2>(), ctx->d_cand_s.as<float>(), ctx->d_cand_pair.as<int>());
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "pf_exact_candidates_kernel<1>");
        const dim3 rgrid(16, (unsigned)std::max<size_t>(1, active.size()));
        hipLaunchKernelGGL(pf_reduce_best_kernel, rgrid, dim3(256), 0, ctx->stream, dp, dl, dact, (const unsigned long long*)ctx->d_cand_count.as<unsigned long long>(),
                           ctx->d_cand.as<int2>(), ctx->d_cand_s.as<float>(), (const int*)ctx->d_cand_pair.as<int>(),
                           ctx->d_best.as<unsigned long long>());
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "pf_reduce_best_kernel");
        hipLaunchKernelGGL(pf_reduce_second_kernel, rgrid, dim3(256), 0, ctx->stream, dp, dl, dact, (const unsigned long long*)ctx->d_cand_count.as<unsigned long long>(),
                           ctx->d_cand.as<int2>(), ctx->d_cand_s.as<float>(), (const int*)ctx->d_cand_pair.as<int>(),
                           ctx->d_best.as<unsigned long long>(), ctx->d_second.as<unsigned long long>());
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "pf_reduce_second_kernel");
    }
    hipLaunchKernelGGL(pf_finalize_kernel, mgrid, dim3(256), 0, ctx->stream, dp, dpf, (const float*)tuv, ctx->d_best.as<unsigned long long>(),
                       ctx->d_second.as<unsigned long long>(), ctx->d_k_i0.as<int>(), ctx->d_k_d0.as<float>(),
                       ctx->d_k_d1.as<float>(), ctx->d_fix_count.as<int>(), ctx->d_fix_list.as<int4>(), ctx->fix_cap_eff);
    HIPCHK(ctx, hipGetLastError());
    DBGSYNC(ctx, "pf_finalize_kernel");

    // candidate-list overflow -> brute-force exact path for that pair
    std::vector<unsigned long long> counts(P + V);
    HIPCHK(ctx, hipMemcpyAsync(counts.data(), ctx->d_cand_count.p, (P + V) * 8, hipMemcpyDeviceToHost, ctx->stream));
#ifdef MSFM_SWEEP_PROBE
    if (std::getenv("MSFM_DUMP_T")) {   // diagnostic build: the raw candidate records of list 0
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        unsigned long long n0 = 0;
        HIPCHK(ctx, hipMemcpy(&n0, ctx->d_cand_count.p, 8, hipMemcpyDeviceToHost));
        const int cap0 = lists[0].cap, take = (int)std::min<unsigned long long>(n0, (unsigned long long)cap0);
        std::vector<int2> c((size_t)std::max(take, 1));
        if (take) HIPCHK(ctx, hipMemcpy(c.data(), ctx->d_cand.as<int2>() + lists[0].off, (size_t)take * 8, hipMemcpyDeviceToHost));
        std::map<std::pair<int, int>, int> seen;
        int colhist[8] = {0}, rowhist[8] = {0};
        for (int i = 0; i < take; ++i) { seen[{c[i].x, c[i].y}]++; colhist[(c[i].y >> 3) & 7]++; rowhist[(c[i].x >> 3) & 7]++; }
        std::fprintf(stderr, "[dump] list 0: count %llu cap %d distinct %zu | cols by 8: %d %d %d %d %d %d %d %d | rows by 8: %d %d %d %d %d %d %d %d\n", n0, cap0,
                     seen.size(), colhist[0], colhist[1], colhist[2], colhist[3], colhist[4], colhist[5], colhist[6], colhist[7],
                     rowhist[0], rowhist[1], rowhist[2], rowhist[3], rowhist[4], rowhist[5], rowhist[6], rowhist[7]);
    }
#endif
    hc.lap("launch sweep 2 .. finalize");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    hc.lap("wait for candidate counts (GPU)");
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    ctx->prof.approx_kernel_ms += ms;
#ifdef MSFM_SWEEP_PROBE
    {   // diagnostic build: average cycles per tile and wave of the four loop segments of sweep 1
        unsigned long long pr[kPfWaves][8];
        HIPCHK(ctx, hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_sweep_probe), sizeof(pr)));
        for (int w = 0; w < kPfWaves; ++w) {
            const double n = (double)std::max<unsigned long long>(1, pr[w][4]);
            std::fprintf(stderr, "[sweep probe] wave %d: MFMA %.0f | wait+barrier %.0f | EPI %.0f | wait+barrier %.0f cycles per tile (%.0f tiles), sweep 1 %.3f ms\n",
                         w, pr[w][0] / n, pr[w][1] / n, pr[w][2] / n, pr[w][3] / n, n, ms);
        }
        std::memset(pr, 0, sizeof(pr));
        HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_probe), pr, sizeof(pr)));
    }
#endif
    HIPCHK(ctx, hipEventElapsedTime(&ms, e2, e3));
    ctx->prof.sweep2_ms += ms;
    std::vector<char> overflow(P, 0);
    for (size_t l = 0; l < P + V; ++l) {
        if (lists[l].cap == 0) continue;
        if (counts[l] <= (unsigned long long)lists[l].cap) {
            ctx->prof.candidates += counts[l];
            continue;
        }
        if (l < P) overflow[l] = 1;
        else  // a group's list: every pair that has rows in it
            for (int k = 0; k < groups[l - P].count; ++k) overflow[(size_t)members[(size_t)(groups[l - P].first + k)].pair] = 1;
    }
    for (size_t p = 0; p < P; ++p) {
        if (!b.pairs[p].valid || !b.pf[p].use) continue;
        if (overflow[p]) {
            b.pf[p].use = 0;
            b.pairs[p].path = 0;
            ctx->prof.fallback_pairs += 1;
        } else {
            ctx->prof.prefilter_pairs += 1;
            ctx->prof.prefilter_descriptor_pairs += (int64_t)b.pairs[p].n1 * b.pairs[p].n2;
            if (compact[p]) ctx->prof.compacted_pairs += 1;
        }
    }
    return MSFM_OK;
}

// brute-force exact distance kernel + merge for the pairs on path 0
int run_exact(msfm_ctx* ctx, Batch& b, size_t ev_base) {
    const size_t P = b.pairs.size();
    assign_partials(b, 0, 4 * ctx->cu_count);
    build_items(b, 0);
    if (b.items.empty()) return MSFM_OK;
    HIPCHK(ctx, ctx->d_rp_s0.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, ctx->d_rp_i0.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, ctx->d_rp_s1.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, ctx->d_cp_s0.ensure(std::max<long long>(1, b.cp_elems) * 4));
    HIPCHK(ctx, ctx->d_cp_i0.ensure(std::max<long long>(1, b.cp_elems) * 4));
    HIPCHK(ctx, ctx->d_cp_s1.ensure(std::max<long long>(1, b.cp_elems) * 4));
    int rc = upload_pairs(ctx, b);
    if (rc != MSFM_OK) return rc;
    rc = upload_items(ctx, b);
    if (rc != MSFM_OK) return rc;
    hipEvent_t e0 = get_event(ctx, ev_base), e1 = get_event(ctx, ev_base + 1);
    if (!e0 || !e1) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    const dim3 grid((unsigned)b.items.size()), block(kThreads);
    if (ctx->order == MSFM_ORDER_SSE4X4)
        hipLaunchKernelGGL(dist_top2_kernel<0>, grid, block, kLdsBytes, ctx->stream,
                           ctx->d_pairs.as<PairDesc>(), ctx->d_items.as<WorkItem>(),
                           ctx->d_rp_s0.as<float>(), ctx->d_rp_i0.as<int>(), ctx->d_rp_s1.as<float>(),
                           ctx->d_cp_s0.as<float>(), ctx->d_cp_i0.as<int>(), ctx->d_cp_s1.as<float>());
    else
        hipLaunchKernelGGL(dist_top2_kernel<1>, grid, block, kLdsBytes, ctx->stream,
                           ctx->d_pairs.as<PairDesc>(), ctx->d_items.as<WorkItem>(),
                           ctx->d_rp_s0.as<float>(), ctx->d_rp_i0.as<int>(), ctx->d_rp_s1.as<float>(),
                           ctx->d_cp_s0.as<float>(), ctx->d_cp_i0.as<int>(), ctx->d_cp_s1.as<float>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    ctx->prof.dist_kernel_launches += 1;
    for (auto& pd : b.pairs)
        if (pd.valid && pd.path == 0) ctx->prof.exact_descriptor_pairs += (int64_t)pd.n1 * pd.n2;
    const dim3 mgrid((unsigned)((b.max_npad + 255) / 256), (unsigned)P);
    hipLaunchKernelGGL(merge_knn_kernel, mgrid, dim3(256), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(),
                       ctx->d_rp_s0.as<float>(), ctx->d_rp_i0.as<int>(), ctx->d_rp_s1.as<float>(),
                       ctx->d_cp_s0.as<float>(), ctx->d_cp_i0.as<int>(), ctx->d_cp_s1.as<float>(),
                       ctx->d_k_i0.as<int>(), ctx->d_k_d0.as<float>(), ctx->d_k_d1.as<float>(),
                       ctx->d_fix_count.as<int>(), ctx->d_fix_list.as<int4>(), ctx->fix_cap_eff);
    HIPCHK(ctx, hipGetLastError());
    return MSFM_OK;
}

// kNN-2 of both directions for every pair of the batch (device arrays left in the ctx buffers):
// prefilter path where eligible, brute-force exact path for the rest, then the sqrt-space tie fix-up
//   need_fix: the caller can observe WHICH index a sqrt-space tie resolves to (knnMatch-level API, ratio > 1).
//   For match lists with ratio <= 1 a row with d0 == d1 fails `d0 < ratio * d1` in both directions, so its
//   index never reaches a list: the queue is not filled and nothing is re-scanned.
int run_knn(msfm_ctx* ctx, Batch& b, size_t ev_base, bool* exact_launched, PruneParams prune, bool need_fix) {
    assign_common(b);
    ctx->fix_cap_eff = need_fix ? ctx->fix_cap : 0;
    const long long kn = std::max<long long>(1, b.kf_elems + b.kr_elems);
    HIPCHK(ctx, ctx->d_k_i0.ensure(kn * 4));
    HIPCHK(ctx, ctx->d_k_d0.ensure(kn * 4));
    HIPCHK(ctx, ctx->d_k_d1.ensure(kn * 4));
    HIPCHK(ctx, ctx->d_fix_count.ensure(4));
    HIPCHK(ctx, ctx->d_fix_list.ensure((size_t)ctx->fix_cap * sizeof(int4)));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_fix_count.p, 0, 4, ctx->stream));
    bool any_pf = false, any_exact = false;
    for (auto& pd : b.pairs) any_pf |= (pd.valid && pd.path == 1);
    int rc;
    if (any_pf) {
        rc = run_prefilter(ctx, b, ev_base + 2, prune);  // events ev_base+2 .. ev_base+5
        if (rc != MSFM_OK) return rc;
    }
    for (auto& pd : b.pairs) any_exact |= (pd.valid && pd.path == 0);
    *exact_launched = false;
    if (any_exact) {
        rc = run_exact(ctx, b, ev_base);
        if (rc != MSFM_OK) return rc;
        *exact_launched = !b.items.empty();
    } else if (!any_pf) {
        rc = upload_pairs(ctx, b);  // later kernels still read the (all-invalid) pair table
        if (rc != MSFM_OK) return rc;
    }
    if (any_pf || any_exact) {
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL(tie_fixup_kernel<0>, dim3(256), dim3(64), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(),
                               ctx->d_fix_count.as<int>(), ctx->d_fix_list.as<int4>(), ctx->fix_cap_eff,
                               ctx->d_k_i0.as<int>(), ctx->d_k_d0.as<float>());
        else
            hipLaunchKernelGGL(tie_fixup_kernel<1>, dim3(256), dim3(64), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(),
                               ctx->d_fix_count.as<int>(), ctx->d_fix_list.as<int4>(), ctx->fix_cap_eff,
                               ctx->d_k_i0.as<int>(), ctx->d_k_d0.as<float>());
        HIPCHK(ctx, hipGetLastError());
    }
    ctx->prof.descriptor_pairs += b.desc_pairs;
    ctx->prof.dist_algo_bytes += b.algo_bytes;
    return MSFM_OK;
}

// Synchronises the stream.  *retry = true: more tied rows than the queue holds -- the queue has been grown to
// fit, the caller re-runs the batch (rare: duplicate descriptors on the brute-force path with ratio > 1 or
// through the knnMatch-level API).
int check_fix_overflow(msfm_ctx* ctx, bool* retry) {
    int nfix = 0;
    *retry = false;
    HIPCHK(ctx, hipMemcpyAsync(&nfix, ctx->d_fix_count.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->fix_cap_eff > 0 && nfix > ctx->fix_cap_eff) {
        ctx->fix_cap = nfix + nfix / 8 + 1024;
        ctx->prof.tie_queue_regrows += 1;
        *retry = true;
        return MSFM_OK;
    }
    ctx->prof.tie_rows += nfix;
    return MSFM_OK;
}

int accumulate_kernel_time(msfm_ctx* ctx, size_t ev_base, bool launched) {
    if (!launched) return MSFM_OK;
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[ev_base], ctx->ev_pool[ev_base + 1]));
    ctx->prof.dist_kernel_ms += ms;
    return MSFM_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char* msfm_version(void) { return "msfm-match 0.1 (gfx950)"; }

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
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return MSFM_E_DEVICE;
    }
    // the distance kernel needs 108 KiB of dynamic LDS
    hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(dist_top2_kernel<0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(dist_top2_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e0 != hipSuccess || e1 != hipSuccess) {
        std::fprintf(stderr, "msfm_create: cannot reserve %d bytes of LDS: %s\n", kLdsBytes,
                     hipGetErrorString(e0 != hipSuccess ? e0 : e1));
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return MSFM_E_DEVICE;
    }
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kPfLdsBytes);
    hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kPfLdsBytes);
    hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kPfLdsBytes);
    if (e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
        std::fprintf(stderr, "msfm_create: cannot reserve %d bytes of LDS for the prefilter kernels\n", kPfLdsBytes);
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return MSFM_E_DEVICE;
    }
    if (const char* e = std::getenv("MSFM_PREFILTER")) ctx->prefilter = (e[0] != '0');
    if (const char* e = std::getenv("MSFM_MAX_PAIRS_PER_BATCH"))
        if (std::atoi(e) > 0) ctx->max_pairs_per_batch = std::atoi(e);
    if (const char* e = std::getenv("MSFM_SCRATCH_MIB"))
        if (std::atoll(e) > 0) ctx->scratch_elems = std::atoll(e) * (1 << 20) / 4;
    *out_ctx = ctx;
    return MSFM_OK;
}

void msfm_destroy(msfm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& im : ctx->images) free_image(im);
    DevBuf* bufs[] = {&ctx->d_pairs, &ctx->d_items, &ctx->d_stage, &ctx->d_rp_s0, &ctx->d_rp_i0, &ctx->d_rp_s1,
                      &ctx->d_cp_s0, &ctx->d_cp_i0, &ctx->d_cp_s1, &ctx->d_k_i0, &ctx->d_k_d0, &ctx->d_k_d1,
                      &ctx->d_st_qt, &ctx->d_st_d, &ctx->d_counts, &ctx->d_offsets, &ctx->d_out_qt,
                      &ctx->d_out_d, &ctx->d_fix_count, &ctx->d_fix_list, &ctx->d_pf, &ctx->d_tu, &ctx->d_tv,
                      &ctx->d_cand, &ctx->d_cand_s, &ctx->d_cand_count, &ctx->d_best, &ctx->d_second, &ctx->d_maxima,
                      &ctx->d_live_cnt, &ctx->d_cmp_h, &ctx->d_cmp_tu, &ctx->d_live_idx, &ctx->d_vpairs, &ctx->d_vpf,
                      &ctx->d_vitems, &ctx->d_jobs, &ctx->d_lists, &ctx->d_row_pair, &ctx->d_cand_pair, &ctx->d_active, &ctx->d_vf_pairs, &ctx->d_vf_x1, &ctx->d_vf_y1,
                      &ctx->d_vf_x2, &ctx->d_vf_y2, &ctx->d_vf_hyp, &ctx->d_vf_best_it, &ctx->d_vf_best_count,
                      &ctx->d_vf_flags, &ctx->d_st2_qt, &ctx->d_st2_d, &ctx->d_counts2};
    for (DevBuf* b : bufs) b->release();
    ctx->res_qt.release();
    ctx->res_dist.release();
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
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
    if (order != MSFM_ORDER_SSE4X4 && order != MSFM_ORDER_AVX2_FMA) return fail(ctx, MSFM_E_INVALID, "unknown accumulation order");
    if (order == ctx->order) return MSFM_OK;
    // the panel layout stores dimensions in accumulation order: re-lay every resident image
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->order = order;
    for (auto& im : ctx->images) {
        if (im.n <= 0) continue;
        const int blocks = std::min(4096, im.nalloc * 16);
        if (order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL((layout_kernel<0, float>), dim3(blocks), dim3(256), 0, ctx->stream, im.raw, (float*)nullptr, im.panel, im.n, im.nalloc);
        else
            hipLaunchKernelGGL((layout_kernel<1, float>), dim3(blocks), dim3(256), 0, ctx->stream, im.raw, (float*)nullptr, im.panel, im.n, im.nalloc);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MSFM_OK;
}

int msfm_set_prefilter(msfm_ctx* ctx, int enable) {
    if (!ctx) return MSFM_E_INVALID;
    ctx->prefilter = enable ? 1 : 0;
    return MSFM_OK;
}

int msfm_set_limits(msfm_ctx* ctx, int max_pairs_per_batch, int64_t scratch_bytes) {
    if (!ctx) return MSFM_E_INVALID;
    ctx->max_pairs_per_batch = max_pairs_per_batch > 0 ? max_pairs_per_batch : kDefaultMaxPairsPerBatch;
    ctx->scratch_elems = scratch_bytes > 0 ? std::max<long long>(1, scratch_bytes / 4) : kDefaultScratchElems;
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
    return MSFM_OK;
}

// everything derived from the row-major fp32 copy `im.raw` (src8 != nullptr: u8 rows still to be widened
// into im.raw by the layout kernel): panels in accumulation order, prefilter operands
static int build_image(msfm_ctx* ctx, Image& im, const unsigned char* src8) {
    const int n = im.n;
    const int blocks = std::min(4096, im.nalloc * 16);
    if (!src8) {
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL((layout_kernel<0, float>), dim3(blocks), dim3(256), 0, ctx->stream, im.raw, (float*)nullptr, im.panel, n, im.nalloc);
        else
            hipLaunchKernelGGL((layout_kernel<1, float>), dim3(blocks), dim3(256), 0, ctx->stream, im.raw, (float*)nullptr, im.panel, n, im.nalloc);
    } else {
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL((layout_kernel<0, unsigned char>), dim3(blocks), dim3(256), 0, ctx->stream, src8, im.raw, im.panel, n, im.nalloc);
        else
            hipLaunchKernelGGL((layout_kernel<1, unsigned char>), dim3(blocks), dim3(256), 0, ctx->stream, src8, im.raw, im.panel, n, im.nalloc);
    }
    HIPCHK(ctx, hipGetLastError());
    // prefilter operands (order-independent): fp16 swizzled blocks, norms, maxima
    const int npad = im.nalloc * kBM;
    HIPCHK(ctx, hipMalloc((void**)&im.h16, (size_t)npad * kPfRowBytes));
    HIPCHK(ctx, hipMalloc((void**)&im.nrm, (size_t)npad * 4));
    HIPCHK(ctx, ctx->d_maxima.ensure(8));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_maxima.p, 0, 8, ctx->stream));
    hipLaunchKernelGGL(pf_prepare_kernel, dim3(std::min(2048, (npad * 16 + 255) / 256)), dim3(256), 0, ctx->stream,
                       im.raw, im.h16, im.nrm, ctx->d_maxima.as<unsigned>(), n, npad);
    HIPCHK(ctx, hipGetLastError());
    unsigned mx[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(mx, ctx->d_maxima.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    // the caller may free/reuse its buffer (and we reuse d_stage) as soon as we return
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(&im.nrm_max, &mx[0], 4);
    std::memcpy(&im.abs_max, &mx[1], 4);
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
            hipLaunchKernelGGL(pf_ext_kernel, dim3((npad + 255) / 256), dim3(256), 0, ctx->stream, im.nrm, im.h16, npad, im.c);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
        HIPCHK(ctx, hipMemcpyAsync(im.raw, desc, (size_t)n * kDim * 4, hipMemcpyHostToDevice, ctx->stream));
        return build_image(ctx, im, nullptr);
    }
    HIPCHK(ctx, ctx->d_stage.ensure((size_t)n * kDim));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage.p, desc, (size_t)n * kDim, hipMemcpyHostToDevice, ctx->stream));
    return build_image(ctx, im, ctx->d_stage.as<unsigned char>());
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
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage.p, rows, (size_t)count * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(subset_rows_kernel, dim3(std::min(1024, (count * kDim + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float*)ctx->images[src_image_id].raw, (const int*)ctx->d_stage.as<int>(), im.raw, count);
    HIPCHK(ctx, hipGetLastError());
    return build_image(ctx, im, nullptr);
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
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& im : ctx->images) free_image(im);
    return MSFM_OK;
}

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
    ctx->have_results = false;
    ctx->res_offsets.assign((size_t)n_pairs + 1, 0);
    ctx->res_count = 0;
    ctx->prof = msfm_profile{};

    hipEvent_t ev_begin = get_event(ctx, 0), ev_end = get_event(ctx, 1);
    if (!ev_begin || !ev_end) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
    HIPCHK(ctx, hipEventRecord(ev_begin, ctx->stream));

    // sub-batches bounded by the partial-result scratch (12 B per partial entry) and a pair count
    const long long kScratchElems = ctx->scratch_elems;
    const int kMaxPairsPerBatch = ctx->max_pairs_per_batch;
    // a tie in sqrt space can only surface in a match list when a row with d0 == d1 can pass the ratio test
    const bool need_fix = !(prm.ratio <= 1.f);
    size_t ev_next = 2;
    int begin = 0;
    while (begin < n_pairs) {
      int end = begin;
      for (int attempt = 0;; ++attempt) {   // a sub-batch is re-run when its tie queue was too small (grown by then)
        Batch b;
        long long est = 0;
        end = begin;
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
            // partial-result scratch of the larger of the two paths (prefilter: 2x slots + candidates)
            const long long need = pd.valid ? ((long long)pd.n1pad + 2 * (long long)pd.a_blocks * pd.n2pad +
                                               3 * (8LL * (pd.n1 + pd.n2) + 1024)) : 0;
            if (end > begin && est + need > kScratchElems) break;
            est += need;
            b.pairs.push_back(pd);
            b.pf.push_back(pp);
            ++end;
        }
        const size_t P = b.pairs.size();
        const size_t ev_base = ev_next;
        if (attempt == 0) ev_next += 8;
        bool exact_launched = false;
        int rc = run_knn(ctx, b, ev_base, &exact_launched, prune, need_fix);
        if (rc != MSFM_OK) return rc;

        HIPCHK(ctx, ctx->d_st_qt.ensure(std::max<long long>(1, b.out_elems) * sizeof(int2)));
        HIPCHK(ctx, ctx->d_st_d.ensure(std::max<long long>(1, b.out_elems) * 4));
        // the lists of the whole call stay on the device too (msfm_fetch_matches_device): this sub-batch appends
        // at most out_elems matches behind the res_count already there
        const size_t base = ctx->res_count;
        HIPCHK(ctx, ctx->d_out_qt.ensure_keep((base + (size_t)std::max<long long>(1, b.out_elems)) * sizeof(int2), base * sizeof(int2), ctx->stream));
        HIPCHK(ctx, ctx->d_out_d.ensure_keep((base + (size_t)std::max<long long>(1, b.out_elems)) * 4, base * 4, ctx->stream));
        HIPCHK(ctx, ctx->d_counts.ensure(P * 4));
        HIPCHK(ctx, ctx->d_offsets.ensure((P + 1) * 8));
        EpiParams ep = {prm.ratio, prm.cross_check, prm.max_distance};
        hipLaunchKernelGGL(epilogue_kernel, dim3((unsigned)P), dim3(256), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(), ep,
                           ctx->d_k_i0.as<int>(), ctx->d_k_d0.as<float>(), ctx->d_k_d1.as<float>(),
                           ctx->d_st_qt.as<int2>(), ctx->d_st_d.as<float>(), ctx->d_counts.as<int>());
        HIPCHK(ctx, hipGetLastError());
        const int* d_counts = ctx->d_counts.as<int>();
        const int2* d_st_qt = ctx->d_st_qt.as<int2>();
        const float* d_st_d = ctx->d_st_d.as<float>();
        if (verify) {
            // FeatureUtils::FilterMatches on the staged lists: all hypotheses of all pairs at once
            VerifyParams vprm = {verify->threshold * verify->threshold, verify->confidence, verify->max_iters, 0, verify->seed};
            const long long oe = std::max<long long>(1, b.out_elems);
            std::vector<VerifyPair> vpairs(P);
            for (size_t p = 0; p < P; ++p)
                vpairs[p] = VerifyPair{ctx->images[pairs[2 * (begin + (int)p)]].kxy, ctx->images[pairs[2 * (begin + (int)p) + 1]].kxy};
            HIPCHK(ctx, ctx->d_vf_pairs.ensure(P * sizeof(VerifyPair)));
            HIPCHK(ctx, ctx->d_vf_x1.ensure(oe * 4));
            HIPCHK(ctx, ctx->d_vf_y1.ensure(oe * 4));
            HIPCHK(ctx, ctx->d_vf_x2.ensure(oe * 4));
            HIPCHK(ctx, ctx->d_vf_y2.ensure(oe * 4));
            HIPCHK(ctx, ctx->d_vf_flags.ensure(oe));
            HIPCHK(ctx, ctx->d_vf_hyp.ensure(P * (size_t)vprm.max_iters * 4));
            HIPCHK(ctx, ctx->d_vf_best_it.ensure(P * 4));
            HIPCHK(ctx, ctx->d_vf_best_count.ensure(P * 4));
            HIPCHK(ctx, ctx->d_st2_qt.ensure(oe * sizeof(int2)));
            HIPCHK(ctx, ctx->d_st2_d.ensure(oe * 4));
            HIPCHK(ctx, ctx->d_counts2.ensure(P * 4));
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_vf_pairs.p, vpairs.data(), P * sizeof(VerifyPair), hipMemcpyHostToDevice, ctx->stream));
            hipEvent_t v0 = get_event(ctx, ev_base + 6), v1 = get_event(ctx, ev_base + 7);
            if (!v0 || !v1) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
            HIPCHK(ctx, hipEventRecord(v0, ctx->stream));
            const PairDesc* dp = ctx->d_pairs.as<PairDesc>();
            float *x1 = ctx->d_vf_x1.as<float>(), *y1 = ctx->d_vf_y1.as<float>(), *x2 = ctx->d_vf_x2.as<float>(), *y2 = ctx->d_vf_y2.as<float>();
            hipLaunchKernelGGL(vf_points_kernel, dim3((unsigned)P), dim3(256), 0, ctx->stream, dp, ctx->d_vf_pairs.as<VerifyPair>(),
                               d_counts, d_st_qt, x1, y1, x2, y2);
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(vf_hypotheses_kernel, dim3((unsigned)((vprm.max_iters + 255) / 256), (unsigned)P), dim3(256), 0, ctx->stream,
                               dp, d_counts, (const float*)x1, (const float*)y1, (const float*)x2, (const float*)y2,
                               ctx->d_vf_hyp.as<int>(), vprm);
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(vf_select_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, ctx->stream, d_counts,
                               (const int*)ctx->d_vf_hyp.as<int>(), (int)P, vprm, ctx->d_vf_best_it.as<int>(), ctx->d_vf_best_count.as<int>());
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(vf_mask_compact_kernel, dim3((unsigned)P), dim3(256), 0, ctx->stream, dp, d_counts, d_st_qt, d_st_d,
                               (const float*)x1, (const float*)y1, (const float*)x2, (const float*)y2,
                               (const int*)ctx->d_vf_best_it.as<int>(), (const int*)ctx->d_vf_best_count.as<int>(),
                               ctx->d_vf_flags.as<unsigned char>(), vprm, ctx->d_st2_qt.as<int2>(), ctx->d_st2_d.as<float>(),
                               ctx->d_counts2.as<int>());
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipEventRecord(v1, ctx->stream));
            d_counts = ctx->d_counts2.as<int>();
            d_st_qt = ctx->d_st2_qt.as<int2>();
            d_st_d = ctx->d_st2_d.as<float>();
        }
        hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(256), 0, ctx->stream, d_counts,
                           ctx->d_offsets.as<long long>(), (int)P);
        HIPCHK(ctx, hipGetLastError());
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)P), dim3(256), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(),
                           d_counts, ctx->d_offsets.as<long long>(), d_st_qt,
                           d_st_d, ctx->d_out_qt.as<int2>() + base, ctx->d_out_d.as<float>() + base);
        HIPCHK(ctx, hipGetLastError());

        std::vector<long long> offs(P + 1);
        HIPCHK(ctx, hipMemcpyAsync(offs.data(), ctx->d_offsets.p, (P + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        bool retry = false;
        rc = check_fix_overflow(ctx, &retry);  // synchronises the stream
        if (rc != MSFM_OK) return rc;
        if (retry && attempt < 4) continue;
        if (retry) return fail(ctx, MSFM_E_DEVICE, "tie fix-up queue kept overflowing");
        const long long total = offs[P];
        HIPCHK(ctx, ctx->res_qt.ensure((base + (size_t)total + 1) * 8, base * 8));
        HIPCHK(ctx, ctx->res_dist.ensure((base + (size_t)total + 1) * 4, base * 4));
        ctx->res_count = base + (size_t)total;
        if (total > 0) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->res_qt.as<int32_t>() + 2 * base, ctx->d_out_qt.as<int2>() + base, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->res_dist.as<float>() + base, ctx->d_out_d.as<float>() + base, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        for (size_t p = 0; p < P; ++p) ctx->res_offsets[(size_t)begin + p + 1] = (int64_t)base + offs[p + 1];
        rc = accumulate_kernel_time(ctx, ev_base, exact_launched);
        if (rc != MSFM_OK) return rc;
        if (verify) {
            float vms = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&vms, ctx->ev_pool[ev_base + 6], ctx->ev_pool[ev_base + 7]));
            ctx->prof.verify_ms += vms;
        }
        ctx->prof.sub_batches += 1;
        break;
      }
      begin = end;
    }
    HIPCHK(ctx, hipEventRecord(ev_end, ctx->stream));
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
    return match_pairs_impl(ctx, pairs, n_pairs, params, nullptr, out_offsets);
}

int msfm_match_pairs_verified(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params,
                              const msfm_verify_params* verify, int64_t* out_offsets) {
    msfm_verify_params v = {3.0, 0.99, 1000, 0x5eed5eedULL};  // FeatureUtils.cpp:196: FM_RANSAC, 3.0, 0.99; OpenCV's maxIters
    if (verify) v = *verify;
    if (!(v.threshold >= 0.0) || !(v.confidence > 0.0) || !(v.confidence < 1.0) || v.max_iters < 1 || v.max_iters > (1 << 16))
        return fail(ctx, MSFM_E_INVALID, "bad verification parameters");
    return match_pairs_impl(ctx, pairs, n_pairs, params, &v, out_offsets);
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
    HIPCHK(ctx, hipMemcpyAsync(im.kxy, xy.data(), xy.size() * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
        HIPCHK(ctx, hipMemcpyAsync(d_out_qt, ctx->d_out_qt.p, ctx->res_count * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_out_dist && ctx->res_count)
        HIPCHK(ctx, hipMemcpyAsync(d_out_dist, ctx->d_out_d.p, ctx->res_count * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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

int msfm_knn2_pair(msfm_ctx* ctx, int id1, int id2, int32_t* fwd_idx0, float* fwd_d0, float* fwd_d1,
                   int32_t* rev_idx0, float* rev_d0, float* rev_d1) {
    if (!ctx) return MSFM_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->prof = msfm_profile{};
    Batch b;
    PairDesc pd;
    PfPair pp;
    int rc = fill_pair(ctx, id1, id2, pd, pp);
    if (rc != MSFM_OK) return rc;
    b.pairs.push_back(pd);
    b.pf.push_back(pp);
    bool exact_launched = false;
    int regrows = 0;
    for (int attempt = 0;; ++attempt) {
        b.items.clear();
        b.rp_elems = b.cp_elems = b.kf_elems = b.kr_elems = b.out_elems = b.cand_elems = 0;
        b.desc_pairs = b.algo_bytes = 0;
        b.pairs[0].path = b.pf[0].use = pp.use;
        ctx->prof = msfm_profile{};
        rc = run_knn(ctx, b, 2, &exact_launched, PruneParams{0, 0.f, 0.f}, true);  // knnMatch twin: every row keeps its neighbours
        if (rc != MSFM_OK) return rc;
        bool retry = false;
        rc = check_fix_overflow(ctx, &retry);
        if (rc != MSFM_OK) return rc;
        if (!retry) break;
        ++regrows;
        if (attempt >= 4) return fail(ctx, MSFM_E_DEVICE, "tie fix-up queue kept overflowing");
    }
    ctx->prof.tie_queue_regrows = regrows;
    ctx->prof.sub_batches = 1;
    rc = accumulate_kernel_time(ctx, 2, exact_launched);
    if (rc != MSFM_OK) return rc;
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
    const Cp cps[6] = {{fwd_idx0, &ctx->d_k_i0, q.kf_off, n1}, {fwd_d0, &ctx->d_k_d0, q.kf_off, n1},
                       {fwd_d1, &ctx->d_k_d1, q.kf_off, n1},   {rev_idx0, &ctx->d_k_i0, q.kr_off, n2},
                       {rev_d0, &ctx->d_k_d0, q.kr_off, n2},   {rev_d1, &ctx->d_k_d1, q.kr_off, n2}};
    for (const Cp& c : cps)
        if (c.dst && c.n > 0)
            HIPCHK(ctx, hipMemcpyAsync(c.dst, c.src->as<char>() + c.off * 4, (size_t)c.n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
