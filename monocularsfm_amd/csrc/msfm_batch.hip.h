// msfm_batch.hip.h -- every launch of ONE device sub-batch (msfm_match.hip cuts a call into them and keeps several in flight):
// pair tables, work items, the matrix-core prefilter route (sweep 1 -> thresholds -> device-side plan -> sweep 2 -> exact re-check),
// the brute-force route, the words the host reads at the end.  Included by msfm_match.hip only.
#pragma once
namespace {
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
    bool route_i8 = false;           // every prefiltered pair joins two byte images and sweep 2 is the compacted one: both sweeps on the integer cores (prepare_batch_images)
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
    if (a.pending || b.pending) return fail(ctx, MSFM_E_STATE, "image not built (finalize_store has not run)");
    pd.a_panel = a.panel;   // (the pointers are refreshed by prepare_batch_images once the batch's routes are known)
    pd.b_panel = b.panel;
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

// Which forms of its images does this sub-batch read?  Decided here, before anything is launched:
//   * route_i8 -- match lists with ratio <= 0.95 (the compacted sweep 2), every prefiltered pair joins two byte images: both sweeps on the
//     integer matrix cores, the candidates' exact S from the sweep itself (pf_exact_candidates_kernel<4>): only the 176-byte rows are read;
//   * otherwise the float forms (permuted fp32 rows, fp16 operand rows) -- derived on demand for byte images;
//   * brute-force pairs (not fp16-safe, MSFM_PREFILTER=0, a candidate-list overflow) also read the panels.
// Then every pair's pointers are refreshed (fill_pair ran before the forms existed).
int prepare_batch_images(msfm_ctx* ctx, Batch& b, const PruneParams& prune) {
    const size_t P = b.pairs.size();
    const bool compact = prune.prune != 0 && prune.ratio > 0.f && prune.ratio <= 0.95f;
    bool i8 = compact && ctx->prefilter == 1;
    for (size_t p = 0; p < P && i8; ++p)
        if (b.pairs[p].valid && b.pf[p].use) i8 = ctx->images[(size_t)b.id1[p]].is_u8 && ctx->images[(size_t)b.id2[p]].is_u8;
    b.route_i8 = i8;
    std::vector<int> wide, panel;
    for (size_t p = 0; p < P; ++p) {
        if (!b.pairs[p].valid) continue;
        const bool brute = !b.pf[p].use || b.pairs[p].path == 0;
        if (brute || !i8) {
            wide.push_back(b.id1[p]);
            wide.push_back(b.id2[p]);
        }
        if (brute) {
            panel.push_back(b.id1[p]);
            panel.push_back(b.id2[p]);
        }
    }
    auto uniq = [](std::vector<int>& v) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    };
    uniq(wide);
    uniq(panel);
    bool need = false;
    for (int id : wide) need |= ctx->images[(size_t)id].n > 0 && !ctx->images[(size_t)id].rawp;
    for (int id : panel) need |= ctx->images[(size_t)id].n > 0 && (!ctx->images[(size_t)id].panel || ctx->images[(size_t)id].panel_order != ctx->order);
    if (need) {
        const int rc = ensure_forms(ctx, wide, panel);
        if (rc != MSFM_OK) return rc;
    }
    for (size_t p = 0; p < P; ++p) {
        const Image& ia = ctx->images[(size_t)b.id1[p]];
        const Image& ib = ctx->images[(size_t)b.id2[p]];
        PairDesc& pd = b.pairs[p];
        PfPair& pp = b.pf[p];
        pd.a_panel = ia.panel;
        pd.b_panel = ib.panel;
        pd.a_rawp = ia.rawp;
        pd.b_rawp = ib.rawp;
        pp.a_h = ia.h16;
        pp.b_h = ib.h16;
        pp.a_nrm = ia.nrm;
        pp.b_nrm = ib.nrm;
    }
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
// The launch sequence of one sub-batch on the prefilter path, stage by stage (VERDICT r05 hygiene: it was one 490-line function).
// Everything the stages share lives in the struct; each stage only LAUNCHES -- nothing waits for the device.
struct PrefilterLaunch {
    msfm_ctx* ctx;
    Batch& b;
    size_t ev_base;
    PruneParams prune;
    size_t P = 0;
    HostClock hc;
    // routes of the sub-batch (choose_routes)
    bool compact = false, i8 = false, q8 = false, q8_mixed = false, fine_twins = false, q8_direct = false, q8_refine = false;
    std::vector<char> twin;
    std::vector<PfPair> pfq, pf16;
    long long dense_cand = 0;
    // work items and buffers (build_and_upload)
    std::vector<int> item_base16;
    size_t per16 = 0;
    long long kn = 1;
    // launch state (launch_sweep1 onwards)
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    dim3 block, mgrid;
    unsigned sweep_grid = 8;
    const PairDesc* dp = nullptr;
    const PfPair* dpf = nullptr;
    float* tuv = nullptr;
    unsigned* colmask = nullptr;
    size_t n_lists = 0;
    const CandList* dl = nullptr;
    // the device-side plan of the compacted sweep 2 (plan_buffers)
    CompactPlan cp;
    size_t G = 0, M = 0;
    long long rows_cap = 0, cand_cap = 0, items_cap = 0;
    const PlanPair* dpp = nullptr;
    PlanCounts pc = {};
    PlanOut po = {};

    PrefilterLaunch(msfm_ctx* c, Batch& batch, size_t ev, PruneParams pr) : ctx(c), b(batch), ev_base(ev), prune(pr) {}

    // ---- stage 1: which matrix cores sweep what (byte stores, route Q, mixed sub-batches), the pairs' tables of each sweep
    int choose_routes() {
        P = b.pairs.size();
        hc = HostClock();
        SC.pf_pending = PfPending{};
        assign_partials(b, 1, 4 * ctx->cu_count);
        // Sweep 2 on the compacted live rows, or on everything again?  Decided per batch, before any result exists: the Lowe
        // test is what kills rows (~94 % at ratio 0.8 on SIFT-like data); with a ratio near or above 1 almost every row stays
        // alive and the compacted sweep (both directions separately) would multiply up to twice what the dense one does.
        compact = prune.prune != 0 && prune.ratio > 0.f && prune.ratio <= 0.95f;
        // Byte stores: both sweeps on the integer matrix cores (msfm_sweep_i8.hip.h) when every prefiltered pair of the batch
        // joins two byte images (a store is bytes throughout or not at all; a mixed batch takes the fp16 kernels)
        i8 = compact && b.route_i8;   // (prepare_batch_images: the images' float forms may not even exist on this route)
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
        q8 = compact && !i8 && ctx->prefilter == 1 && ctx->q8_route;
        long long q8_rows = 0, q8_pairs = 0, twin_pairs = 0, twin_rows = 0;
        twin.clear();
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
    fine_twins = ctx->q8_direct == 2 || (ctx->q8_direct == 1 && ctx->q8_level <= kQ8DirectMaxLevel);
    q8_mixed = false;   // some pairs join images without twins: THEIR sweep 1 runs on the fp16 cores, the twins' on the integer cores
    if (q8 && twin_pairs < q8_pairs) {
        // (fine twins only: the coarse route's plan A / sweep 1' cover whole sub-batches; and only when a quarter of the work or more has twins)
        q8_mixed = fine_twins && twin_pairs > 0 && 4 * twin_rows >= q8_rows;
        if (!q8_mixed) q8 = false;
    }
    if (q8 && ctx->q8_route < 2 && (twin_pairs == 0 || twin_rows < 2 * 1024 * twin_pairs)) q8 = q8_mixed = false;
    pfq.clear(), pf16.clear();
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
    q8_direct = q8 && fine_twins;
    q8_refine = q8 && !q8_direct;
    SC.pf_pending.q8 = q8_refine;   // (a plan A and a sweep 1' to account for at the end of the batch)
    dense_cand = 0;
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
    return MSFM_OK;
    }

    // ---- stage 2: work items, the buffers of sweep 1 and of the reduce, every clear in one launch, the tables in one copy
    int build_and_upload() {
        item_base16.clear();
        per16 = 0;
        build_items(b, 1);
        if (b.n_items == 0) return MSFM_OK;   // (nothing on this path: run() stops here)
        if (q8_mixed) {   // the fp16 first sweep's own item list: the pairs without twins
            std::vector<char> only(P, 0);
            for (size_t p = 0; p < P; ++p) only[p] = (b.pairs[p].valid && b.pf[p].use && !twin[p]) ? 1 : 0;
            build_items(b, 1, &only, &item_base16, &per16);
    }
    kn = std::max<long long>(1, b.kf_elems + b.kr_elems);
    HIPCHK(ctx, SC.d_rp_s0.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, SC.d_rp_s1.ensure(std::max<long long>(1, b.rp_elems) * 4));
    // column partials of sweep 1: one float2 (the two largest of four row-class maxima) per 512-row A block and column; the integer
    // sweeps pack theirs into 4 bytes (msfm_cp_pack) -- a mixed sub-batch keeps the 8-byte stride for both
    HIPCHK(ctx, SC.d_cp_s0.ensure(std::max<long long>(1, b.cp_elems) * ((i8 || (q8_direct && !q8_mixed)) ? 4 : 8)));   // (coarse twins: q8_scatter_kernel writes float2 entries)
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
    const int rc = upload_pair_tables(ctx, b, 1, q8 ? &pfq : nullptr, fills, q8_mixed ? &pf16 : nullptr, q8_mixed ? &item_base16 : nullptr, per16);
    if (rc != MSFM_OK) return rc;
    return MSFM_OK;
    }

    // ---- stage 3: sweep 1 (ordered behind the other streams' sweeps), for the dense second sweep also the thresholds
    int launch_sweep1() {
        e0 = get_event(ctx, ev_base), e1 = get_event(ctx, ev_base + 1);
        e2 = get_event(ctx, ev_base + 2), e3 = get_event(ctx, ev_base + 3);
        if (!e0 || !e1 || !e2 || !e3) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
        block = dim3(kPfThreads);
        sweep_grid = (unsigned)std::max(8, (ctx->cu_count / 8) * 8);   // one persistent workgroup per CU, a multiple of the 8 XCDs
        dp = SC.d_pairs.as<PairDesc>();
        dpf = SC.d_pf.as<PfPair>();
        tuv = SC.d_tu.as<float>();  // rows at kf offsets, columns at kr offsets (one buffer)
        colmask = SC.d_colmask.as<unsigned>();
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
    mgrid = dim3((unsigned)((b.max_npad + 255) / 256), (unsigned)P);
    if (!compact) {
        hipLaunchKernelGGL(pf_thresholds_kernel, mgrid, dim3(256), 0, SC.stream, dp, dpf, SC.d_rp_s0.as<float>(),
                           SC.d_rp_s1.as<float>(), SC.d_cp_s0.as<float>(), (unsigned*)nullptr, tuv, tuv, prune, PlanCounts{}, 0);
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "pf_thresholds_kernel");
    }
    hc.lap("launch sweep 1 (+ thresholds)");
    return MSFM_OK;
    }

    // ---- stage 4a (match lists): the static tables of the device-side plan of sweep 2 (the GPU is busy with sweep 1 meanwhile), its
    // buffers from the prediction, everything it clears in one launch
    int plan_buffers() {
        // ---- static plan tables (the GPU is busy with sweep 1 meanwhile), buffers from the prediction ----------
        cp = CompactPlan{};
        build_compact_plan(ctx, b, cp, q8_refine);
        G = cp.groups.size(), M = cp.member_pair.size();
        n_lists = G;
        long long max_ranges = 1;
        for (const PlanGroup& g : cp.groups) max_ranges = std::max<long long>(max_ranges, g.ranges);
        // Capacities: msfm_plan_room (msfm_hostutil.h) -- a prediction relative to the rows this sub-batch could compact at most; existing
        // buffers are kept while they hold it.
        const long long ub = q8_refine ? cp.rows_ub_all_bits : cp.rows_ub;
        long long rows_have = (long long)std::min(std::min(SC.d_cmp_tu.cap / 4, SC.d_live_idx.cap / 4), std::min(SC.d_row_pair.cap / 4, SC.d_row_src.cap / 8));
        if (i8) rows_have = std::min<long long>(rows_have, (long long)(SC.d_cmp_n2.cap / 4));
        if (q8_refine) rows_have = std::min<long long>(rows_have, (long long)std::min(SC.d_cmp_s0.cap / 4, SC.d_cmp_s1.cap / 4));
        long long cand_have = (long long)(SC.d_cand.cap / sizeof(int2));
        if (i8) cand_have = std::min<long long>(cand_have, (long long)(SC.d_cand_val.cap / 4));
        const long long items_have = (long long)(SC.d_vitems.cap / sizeof(WorkItem)) / 8 * 8;
        const MsfmPlanRoom room = msfm_plan_room(ub, q8_refine ? cp.rows_ub_all_bits * 5 / 32 : cp.rows_ub * 5 / 16, (long long)G, max_ranges, kPfWgRows,
                                                 ctx->hint_rows_ub, ctx->cmp_rows_hint, ctx->cand_hint, ctx->items_hint, rows_have, cand_have, items_have);
        rows_cap = room.rows, cand_cap = room.cand, items_cap = room.items;
        SC.pf_pending.rows_ub = ub;
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
        dpp = SC.d_ppair.as<PlanPair>();
        pc = PlanCounts{dpp, (const int*)SC.d_member_group.as<int>(), SC.d_cnt.as<int>(), SC.d_gtot.as<int>()};
        HIPCHK(ctx, SC.d_summary_a.ensure(sizeof(PlanSummary)));
        po = PlanOut{};
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
        return MSFM_OK;
    }

    // the plan from the live counts in d_cnt / d_gtot: scan, descriptors + work items, member rows, slot assignment
    int launch_plan(PlanSummary* summary, int norms_only) {
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
    }

    // ---- stage 4b (route Q): live / dead from the twins' sweep; coarse twins: plan A, the fp16 sweep 1' of the live rows, scatter
    int launch_route_q() {
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
            int rc = launch_plan(SC.d_summary_a.as<PlanSummary>(), 1);
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
        return MSFM_OK;
    }

    // ---- stage 4c: thresholds + live counts, the plan, the compacted sweep 2
    int launch_compact_sweep2() {
        // thresholds + live counts per member / group (the plan tables above are uploaded by now; sweep 1 is still running)
        if (!q8_direct)
            hipLaunchKernelGGL(pf_thresholds_kernel, mgrid, dim3(256), 0, SC.stream, dp, dpf, SC.d_rp_s0.as<float>(),
                               SC.d_rp_s1.as<float>(), SC.d_cp_s0.as<float>(), colmask, tuv, tuv, prune, pc, q8 ? 1 : 0);
        else if (q8_mixed)   // (the pairs of the fp16 sweep 1: thresholds and plan counts the usual way; the prune kernel did the twins')
            hipLaunchKernelGGL(pf_thresholds_kernel, mgrid, dim3(256), 0, SC.stream, dp, (const PfPair*)SC.d_pf16.as<PfPair>(), SC.d_rp_s0.as<float>(),
                               SC.d_rp_s1.as<float>(), SC.d_cp_s0.as<float>(), colmask, tuv, tuv, prune, pc, 0);
        HIPCHK(ctx, hipGetLastError());
        DBGSYNC(ctx, "pf_thresholds_kernel");
        const int rc = launch_plan(SC.d_summary.as<PlanSummary>(), 0);
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
        return MSFM_OK;
    }

    // ---- stage 4d: kNN-level API / ratio near 1 -- dense sweep 2: the pairs' own lists, the sweep-1 items again
    int launch_dense_sweep2() {
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
        return MSFM_OK;
    }

    // ---- stage 5: exact S of the candidates, [finalize], overflow bookkeeping; what finish_prefilter reads at the end of the batch
    int launch_exact_and_finalize() {
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

    int run() {
        int rc = choose_routes();
        if (rc != MSFM_OK) return rc;
        rc = build_and_upload();
        if (rc != MSFM_OK || b.n_items == 0) return rc;
        rc = launch_sweep1();
        if (rc != MSFM_OK) return rc;
        if (compact) {
            rc = plan_buffers();
            if (rc == MSFM_OK && q8) rc = launch_route_q();
            if (rc == MSFM_OK) rc = launch_compact_sweep2();
        } else {
            rc = launch_dense_sweep2();
        }
        if (rc != MSFM_OK) return rc;
        return launch_exact_and_finalize();
    }
};

int run_prefilter(msfm_ctx* ctx, Batch& b, size_t ev_base, PruneParams prune) {
    PrefilterLaunch launch(ctx, b, ev_base, prune);
    return launch.run();
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
        std::fprintf(stderr, "[msfm plan] ok %d, items %lld (room for %lld), compacted rows %lld (%lld; of at most %lld), candidate entries %lld (%lld), swept descriptor pairs %lld; sweep 1 %.3f ms, sweep 2 %.3f ms\n",
                     d.ok, d.items_needed, pe.items_cap, d.cmp_rows, pe.rows_cap, pe.rows_ub, d.cand_elems, pe.cand_cap, d.swept_desc_pairs, SC.prof.approx_kernel_ms, ms);
    }
    if (pe.compact) {
        std::memcpy(&sm, hs, sizeof(PlanSummary));
        ctx->cmp_rows_hint = sm.cmp_rows;
        ctx->items_hint = sm.items_needed;
        ctx->cand_hint = sm.cand_elems;
        ctx->hint_rows_ub = pe.rows_ub;
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
    HIPCHK(ctx, SC.d_fix_count.ensure(4));
    HIPCHK(ctx, SC.d_fix_list.ensure((size_t)ctx->fix_cap * sizeof(int4)));
    HIPCHK(ctx, hipMemsetAsync(SC.d_fix_count.p, 0, 4, SC.stream));
    bool any_pf = false, any_exact = false;
    for (auto& pd : b.pairs) any_pf |= (pd.valid && pd.path == 1);
    for (auto& pd : b.pairs) any_exact |= (pd.valid && pd.path == 0);
    // match lists of a batch that is on the matrix-core route throughout, no sqrt-space tie queue: the epilogue reads best / second
    // keys directly (KnnFromKeys); the knnMatch-level API and mixed batches keep the kNN arrays
    SC.keys_epilogue = lists_only && any_pf && !any_exact && SC.fix_cap_eff == 0;
    if (!SC.keys_epilogue) {   // (the final kNN arrays: 12 bytes per padded row and column, only where something reads them)
        HIPCHK(ctx, SC.d_k_i0.ensure(kn * 4));
        HIPCHK(ctx, SC.d_k_d0.ensure(kn * 4));
        HIPCHK(ctx, SC.d_k_d1.ensure(kn * 4));
    }
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

