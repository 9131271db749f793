// msfm_plan.hip.h -- the plan of the compacted sweep 2, built ON THE DEVICE (included by msfm_prefilter.hip.h).
//
// After sweep 1 and pf_thresholds_kernel every row / column of every pair is either dead (it provably cannot yield a
// match: threshold -inf) or alive, and for every live column a bit mask names the 512-row blocks of image 1 that can
// hold a candidate of it.  Sweep 2 only multiplies live rows:
//
//   forward group  (streamed image j):             the live ROWS of image i of every pair (i, j) of the batch, one after
//                                                  the other, against ALL tiles of image j;
//   reverse group  (streamed image i, block bit b): the live COLUMNS (rows of image j) of every pair (i, j) whose mask has
//                                                  bit b, against the tiles of THAT block of image i only.
//
// Which groups exist, which (pair, direction[, bit]) slices ("members") they consist of and in which order is a function
// of the pair list alone: the host builds those tables before the first launch.  How many rows each member contributes
// is only known on the device; round 1 copied the counts to the host, planned there and uploaded the plan -- two stream
// synchronisations and a std::map in the middle of the pipeline.  Now:
//
//   (pf_thresholds_kernel counts the live rows per member and per group while it writes the thresholds)
//   pf_plan_kernel     prefix sums -> row ranges of the groups (512-aligned), the groups' sweep descriptors
//                      (PairDesc / PfPair / CandList), the work-item list, capacity check   one workgroup
//   pf_member_rows_kernel  first compact row of every member (a wave-level scan per group)   one wave per group
//   pf_assign_kernel   every live row gets its slot(s): index, pair, threshold, address of its operand row (the sweep
//                      reads its A fragments through that table: no compacted copy of the rows)   grid = 2 x pairs
//
// and the sweep kernels take the number of work items from device memory.  Buffers are sized from a prediction (the
// previous call's need, or a fraction of the row total); if the plan does not fit, it marks itself invalid, everything
// downstream degenerates to a no-op and the host -- which looks at the summary together with the results, at the one
// synchronisation at the end of the batch -- grows the buffers and runs the batch again.
#pragma once
// (included inside namespace msfm)

struct PlanGroup {            // static description of one compacted sweep (host-built)
    const _Float16* b_h;      // streamed image: fp16 operand rows
    const float* b_nrm;
    float b_c, a_c;           // scales of the streamed image's / the compacted rows' norm quadruples
    int dir;                  // 0: forward (candidate records (k, t)), 1: reverse (records (k, q))
    int bt_begin, bt_end;     // streamed range, in 128-row blocks of the streamed image
    int n2, n2pad, b_tiles;   // rows / padded rows / 128-row blocks of the streamed image
    int first, count;         // members: gmembers[first .. first + count)
    int ranges;               // B-range split of its work items (1 unless the batch is small)
    int b_h0;                 // integer-core route: the streamed image's centre H0
};

struct PlanSummary {
    int ok;                   // 0: a capacity was exceeded, nothing downstream ran
    int n_items;              // work items of the compacted sweep (a multiple of 8: XCD interleave)
    long long cmp_rows;       // rows of the compacted matrices incl. the 512-alignment of the groups
    long long cand_elems;     // candidate-list capacity the plan wants
    long long items_needed;
    long long swept_desc_pairs;  // descriptor pairs the compacted sweep multiplies (padding included)
};

constexpr int kPlanThreads = 512;   // (1024 would halve the registers per thread: the per-XCD totals then spill)
// One workgroup of kPlanThreads threads.  All loops are strided over groups / members; the two exclusive scans (rows and work
// items / candidate capacity over groups) run as chunked scans through LDS.
struct PlanOut {
    PairDesc* vpairs;         // [n_groups] sweep descriptors
    PfPair* vpf;
    CandList* lists;          // [n_groups]
    WorkItem* items;          // [items_cap], pre-filled with pair = -1
    long long* grow0;         // [n_groups] first compact row of the group (-1: invalid plan); pf_member_rows_kernel -> mrow
    PlanSummary* summary;
    const _Float16* const* row_src;  // per compact row: its operand row (filled by pf_assign_kernel)
    const _Float16* zero_row;        // 272 zero bytes
    int* live_idx;            // per compact row: row index in its image
    int* row_pair;            // per compact row: pair of the batch
    long long rows_cap, cand_cap, items_cap;
};

__device__ __forceinline__ long long plan_block_exclusive_scan(long long v, long long* total) {
    // exclusive scan of one value per thread over the workgroup (<= 1024 threads)
    __shared__ long long wsum[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    long long base = 0, tot = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
        if (k < w) base += wsum[k];
        tot += wsum[k];
    }
    *total = tot;
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(kPlanThreads) void pf_plan_kernel(const PlanGroup* __restrict__ groups, int n_groups,
                                                         const int* __restrict__ gtot, PlanOut out) {
    MSFM_TAIL_PRIO();
    const int tid = threadIdx.x, nt = blockDim.x;
    __shared__ long long s_base_rows, s_base_cand, s_base_items[8], s_base_rev[8];
    __shared__ int s_ok;
    if (tid == 0) { s_base_rows = 0; s_base_cand = 0; s_ok = 1; }
    if (tid < 8) s_base_items[tid] = 0;
    __syncthreads();
    // Work items: all items of a group go to XCD (group & 7) -- they stream the same image through that XCD's L2 -- and
    // item j of XCD x sits at list position 8 j + x (workgroup b runs on XCD b % 8 and strides over the list by a
    // multiple of 8).  Groups alternate between long forward sweeps and short reverse ones in creation order, so the
    // residue classes carry comparable work; a contiguous chunk of the list per XCD (as for the uniform sweep-1 items)
    // would give some XCDs all the long items.
    // pass 0: totals (does the plan fit?)   pass 1: write it
    long long tot_rows = 0, tot_cand = 0, tot_swept = 0, tot_items_x[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot_fwd_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long n_items = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            for (int x = 0; x < 8; ++x) n_items = tot_items_x[x] > n_items ? tot_items_x[x] : n_items;
            n_items *= 8;
            __syncthreads();
            if (tid == 0) {
                s_ok = (tot_rows <= out.rows_cap && tot_cand <= out.cand_cap && n_items <= out.items_cap) ? 1 : 0;
                s_base_rows = s_base_cand = 0;
                PlanSummary sm;
                sm.ok = s_ok;
                sm.n_items = s_ok ? (int)n_items : 0;
                sm.cmp_rows = tot_rows;
                sm.cand_elems = tot_cand;
                sm.items_needed = n_items;
                sm.swept_desc_pairs = tot_swept;
                *out.summary = sm;
            }
            if (tid < 8) { s_base_items[tid] = 0; s_base_rev[tid] = tot_fwd_x[tid]; }   // reverse items behind all forward ones
            __syncthreads();
        }
        const int ok = s_ok;
        for (int g0 = 0; g0 < n_groups; g0 += nt) {
            const int g = g0 + tid;
            long long rows = 0;
            PlanGroup G = {};
            if (g < n_groups) {
                G = groups[g];
                rows = gtot[g];   // (summed by pf_count_kernel: a serial loop over up to 127 members here was most of this kernel)
            }
            const long long rows512 = (rows + kPfWgRows - 1) / kPfWgRows * kPfWgRows;
            const long long ablocks = rows512 / kPfWgRows;
            const long long items = ablocks * (g < n_groups ? G.ranges : 0);
            const long long cap = rows > 0 ? (8 * rows + 1024 < (1LL << 30) ? 8 * rows + 1024 : (1LL << 30)) : 0;
            long long tr, tc, ti[8], ti_rev[8], item0 = 0;
            const long long row0 = s_base_rows + plan_block_exclusive_scan(rows512, &tr);
            const long long cand0 = s_base_cand + plan_block_exclusive_scan(cap, &tc);
            // (forward groups -- the long sweeps -- come first in every XCD's list: they are fetched first)
            for (int x = 0; x < 8; ++x) {
                long long tf, tb;
                const long long lf = plan_block_exclusive_scan(((g & 7) == x && G.dir == 0) ? items : 0, &tf);
                const long long lb = plan_block_exclusive_scan(((g & 7) == x && G.dir == 1) ? items : 0, &tb);
                ti[x] = tf;          // forward items of this stride
                ti_rev[x] = tb;
                if ((g & 7) == x) item0 = G.dir == 0 ? s_base_items[x] + lf : s_base_rev[x] + lb;
            }
            if (pass == 0) {
                tot_rows += tr;
                tot_cand += tc;
                for (int x = 0; x < 8; ++x) { tot_items_x[x] += ti[x] + ti_rev[x]; tot_fwd_x[x] += ti[x]; }
                long long swept = (g < n_groups) ? rows512 * (long long)min(G.n2 - min(G.n2, G.bt_begin * kBN), (G.bt_end - G.bt_begin) * kBN) : 0, ts;
                (void)plan_block_exclusive_scan(swept, &ts);
                tot_swept += ts;
            } else if (g < n_groups) {
                // the group's sweep descriptor: A = its compacted rows, B = the streamed image
                PairDesc vd = {};
                vd.n1 = ok ? (int)rows : 0;
                vd.n2 = G.n2;
                vd.a_blocks256 = ok ? (int)ablocks : 0;
                vd.b_tiles = G.b_tiles;
                vd.n1pad = ok ? (int)rows512 : 0;
                vd.n2pad = G.n2pad;
                vd.valid = 1;
                vd.path = 1;
                vd.ranges = G.ranges;
                out.vpairs[g] = vd;
                PfPair vp = {};
                vp.a_h = out.zero_row;
                vp.a_rows = out.row_src + row0;
                vp.b_h = G.b_h;
                vp.b_nrm = G.b_nrm;
                vp.b_c = G.b_c;
                vp.a_c = G.a_c;
                vp.b_h0 = G.b_h0;
                vp.tu_off = row0;
                vp.cand_off = cand0;
                vp.cand_cap = ok ? (int)cap : 0;
                vp.use = (ok && rows > 0) ? 1 : 0;
                out.vpf[g] = vp;
                CandList L = {};
                L.pair = -1;
                L.mode = 1 + G.dir;
                L.off = cand0;
                L.cap = ok ? (int)cap : 0;
                L.live_idx = out.live_idx + row0;
                L.row_pair = out.row_pair + row0;
                out.lists[g] = L;
                out.grow0[g] = ok ? row0 : -1;
                if (ok) {
                    const int nblk = G.bt_end - G.bt_begin;
                    long long k = item0;
                    for (int rg = 0; rg < G.ranges; ++rg) {
                        const int t0 = G.bt_begin + (int)((long long)nblk * rg / G.ranges), t1 = G.bt_begin + (int)((long long)nblk * (rg + 1) / G.ranges);
                        for (int ab = 0; ab < (int)ablocks; ++ab, ++k) {
                            WorkItem w = {};
                            w.pair = g;
                            w.a_blk = ab;
                            w.bt_begin = t0;
                            w.bt_end = t1;
                            w.range = rg;
                            out.items[k * 8 + (g & 7)] = w;
                        }
                    }
                }
            }
            if (tid == 0) {
                s_base_rows += tr;
                s_base_cand += tc;
            }
            if (tid < 8) { s_base_items[tid] += ti[tid]; s_base_rev[tid] += ti_rev[tid]; }
            __syncthreads();
        }
    }
}

// first compact row of every member: the group's first row + the rows of the members before it.  One wave per group.
__global__ void pf_member_rows_kernel(const PlanGroup* __restrict__ groups, int n_groups, const int* __restrict__ gmembers,
                                      const int* __restrict__ cnt, const long long* __restrict__ grow0, long long* __restrict__ mrow) {
    MSFM_TAIL_PRIO();
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= n_groups) return;
    const PlanGroup G = groups[g];
    const long long row0 = grow0[g];
    long long base = 0;
    for (int k0 = 0; k0 < G.count; k0 += 64) {
        const int k = k0 + lane;
        const int m = k < G.count ? gmembers[G.first + k] : -1;
        const int c = m >= 0 ? cnt[m] : 0;
        int inc = c;
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (m >= 0) mrow[m] = row0 >= 0 ? row0 + base + (inc - c) : -1;
        base += __shfl(inc, 63);
    }
}

// every live row takes its slot(s) in the compacted row set: index, pair, threshold (sweep 2 folds (T - |a|^2)/2 into
// the MFMA), and the address of its fp16 operand row.  Every member is filled by exactly one workgroup (the forward
// member by (pair, 0), the reverse members by (pair, 1)), so the fill cursors live in LDS.
__global__ void pf_assign_kernel(const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf, const PlanPair* __restrict__ pp_plan,
                                 const float* __restrict__ tuv, const unsigned* __restrict__ colmask, const long long* __restrict__ mrow,
                                 int* __restrict__ live_idx, int* __restrict__ row_pair, float* __restrict__ cmp_tu,
                                 const _Float16** __restrict__ row_src, int row_halfs /* 136: fp16 rows, 72: byte rows */,
                                 unsigned long long* __restrict__ best, unsigned long long* __restrict__ second) {
    MSFM_TAIL_PRIO();
    const int p = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const PlanPair pl = pp_plan[p];
    if (pl.fwd_member < 0) return;
    const PairDesc pd = pairs[p];
    const PfPair pp = pf[p];
    const int n = dir ? pd.n2 : pd.n1;
    const long long off = dir ? pp.tv_off : pp.tu_off;
    const float* nrm = dir ? pp.b_nrm : pp.a_nrm;
    const _Float16* src = dir ? pp.b_h : pp.a_h;
    __shared__ int cursor[32];
    __shared__ long long base[32];
    if (threadIdx.x < 32) {
        cursor[threadIdx.x] = 0;
        const int bits = dir ? pl.rev_bits : 1;
        base[threadIdx.x] = (int)threadIdx.x < bits ? mrow[(dir ? pl.rev_member0 : pl.fwd_member) + threadIdx.x] : -1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const float t = tuv[off + e];
        if (t == -f_inf()) continue;
        // the reduce kernels only ever touch the slots of live rows / columns (and pf_finalize_kernel reads no others):
        // "no candidate yet" is written here, for the 6 % that are alive, instead of a memset over every slot of the batch
        best[off + e] = ~0ull;
        second[off + e] = ~0ull;
        unsigned m = dir ? colmask[off + e] : 1u;
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            const long long r0 = base[b];
            if (r0 < 0) continue;   // invalid plan
            const long long k = r0 + atomicAdd(&cursor[b], 1);
            live_idx[k] = e;
            row_pair[k] = p;
            cmp_tu[k] = t - nrm[e];
            row_src[k] = src + (size_t)e * row_halfs;
        }
    }
}

// after the candidate kernels: which pairs own a list that overflowed (they are re-run on the brute-force path), how
// many candidates were evaluated
__global__ void pf_overflow_kernel(const CandList* __restrict__ lists, int n_lists, const unsigned long long* __restrict__ cand_count,
                                   const PlanGroup* __restrict__ groups, const int* __restrict__ gmembers, const int* __restrict__ member_pair,
                                   unsigned char* __restrict__ overflow_pair, unsigned long long* __restrict__ totals /* [0] candidates, [1] overflowed lists */) {
    MSFM_TAIL_PRIO();
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < n_lists; l += gridDim.x * blockDim.x) {
        const CandList L = lists[l];
        if (L.cap == 0) continue;
        const unsigned long long c = cand_count[l];
        if (c <= (unsigned long long)L.cap) {
            atomicAdd(&totals[0], c);
            continue;
        }
        atomicAdd(&totals[1], 1ull);
        if (L.mode == 0) overflow_pair[L.pair] = 1;
        else
            for (int k = 0; k < groups[l].count; ++k) overflow_pair[member_pair[gmembers[groups[l].first + k]]] = 1;
    }
}
