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
    const int* b_n2;          // integer-core route: the streamed image's exact row norms n' (CandList::b_n2)
};

struct PlanSummary {
    int ok;                   // 0: a capacity was exceeded, nothing downstream ran
    int n_items;              // work items of the compacted sweep (a multiple of 8: XCD interleave)
    long long cmp_rows;       // rows of the compacted matrices incl. the 512-alignment of the groups
    long long cand_elems;     // candidate-list capacity the plan wants
    long long items_needed;
    long long swept_desc_pairs;  // descriptor pairs the compacted sweep multiplies (padding included)
};

// The plan is made by two kernels that both FIT NEXT TO a sweep-1 workgroup of the other stream (two waves of 226
// registers per SIMD leave 48 of the 512): a 170-register single-workgroup version waited for a CU until that persistent
// sweep had ended -- measured 8.4 ms instead of 0.08 ms, and the whole tail of the sub-batch with it.
//   pf_plan_scan_kernel   one workgroup, one pass: per group its first compact row, its candidate-list base and its position
//                         among the forward / reverse work items of its XCD (prefix sums through LDS), the totals, the verdict;
//   pf_plan_write_kernel  one thread per group: the sweep descriptors (field by field into zeroed arrays) and the work items.
struct PlanOut {
    PairDesc* vpairs;         // [n_groups] sweep descriptors (zeroed by the host)
    PfPair* vpf;
    CandList* lists;          // [n_groups]
    WorkItem* items;          // [items_cap], pre-filled with pair = -1
    long long* grow0;         // [n_groups] first compact row of the group (-1: invalid plan); pf_member_rows_kernel -> mrow
    long long* gpos;          // [n_groups][3] scan -> write: first compact row | candidate-list base | position among the
                              //   forward (dir 0) / reverse (dir 1) items of XCD g & 7
    long long* fwd_items_x;   // [8] forward items per XCD: the reverse items of an XCD sit behind them
    PlanSummary* summary;
    const _Float16* const* row_src;  // per compact row: its operand row (filled by pf_assign_kernel)
    const _Float16* zero_row;        // 272 zero bytes
    int* live_idx;            // per compact row: row index in its image
    int* row_pair;            // per compact row: pair of the batch
    long long rows_cap, cand_cap, items_cap;
    const float* cmp_tu;      // per compact row: T - |a|^2 (pf_assign_kernel); the integer route's exact-S fold reads it back
    const int* cmp_n2;        // per compact row: n' (integer route; pf_assign_kernel)
};

// candidate-list capacity of a group in units of 1024 entries: 8 per compacted row, rounded up, + 1024; at most 2^30 entries
__device__ __forceinline__ int plan_group_cap_units(long long rows) {
    return rows > 0 ? (int)min((rows + 127) / 128 + 1, 1LL << 20) : 0;
}

// inclusive scan over the 64 lanes of the wave (32-bit: rows travel as 512-row blocks, capacities as 1024-entry units)
__device__ __forceinline__ int plan_wave_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// ONE WAVE.  Work items: all items of a group go to XCD (group & 7) -- they stream the same image through that XCD's L2 --
// and item j of XCD x sits at list position 8 j + x (workgroup b runs on XCD b % 8 and strides over the list by a multiple
// of 8).  Groups alternate between long forward sweeps and short reverse ones in creation order, so the residue classes
// carry comparable work; a contiguous chunk of the list per XCD (as for the uniform sweep-1 items) would give some XCDs
// all the long items.  Forward groups -- the long sweeps -- come first in every XCD's list: a group's position is its
// rank among the forward (reverse) items of its XCD; pf_plan_write_kernel puts the reverse ones behind all forward ones.
// Lane l serves the groups g = 64 c + l: its XCD class l & 7 never changes, so the per-class running sums live in the
// lanes themselves and a class-wise prefix sum is three shuffle steps of stride 8.
__global__ __launch_bounds__(64) void pf_plan_scan_kernel(const PlanGroup* __restrict__ groups, int n_groups,
                                                        const int* __restrict__ gtot, PlanOut out) {
    MSFM_TAIL_PRIO();
    const int lane = threadIdx.x;
    long long base_rows = 0, base_cand = 0, tot_swept = 0;   // 512-row blocks / 1024-entry units of the chunks before
    int carry_f = 0, carry_r = 0;   // forward / reverse items of this lane's XCD class in the chunks before
#pragma unroll 1
    for (int g0 = 0; g0 < n_groups; g0 += 64) {
        const int g = g0 + lane;
        const bool in = g < n_groups;
        const long long rows = in ? gtot[g] : 0;   // (summed by pf_thresholds_kernel)
        const int dir = in ? groups[g].dir : 0, ranges = in ? groups[g].ranges : 0;
        const long long rows512 = (rows + kPfWgRows - 1) / kPfWgRows * kPfWgRows;
        const int items = (int)(rows512 / kPfWgRows) * ranges;
        long long swept = 0;
        if (in) {
            const int n2 = groups[g].n2, bt0 = groups[g].bt_begin, bt1 = groups[g].bt_end;
            swept = rows512 * (long long)min(n2 - min(n2, bt0 * kBN), (bt1 - bt0) * kBN);
        }
        const int capu = plan_group_cap_units(rows), ablocks = (int)(rows512 / kPfWgRows);
        tot_swept += swept;   // (lane-local: reduced once, behind the loop)
        const int ir = plan_wave_scan(ablocks), ic = plan_wave_scan(capu);
        int vf = dir == 0 ? items : 0, vr = dir == 1 ? items : 0;
        const int f0 = vf, r0 = vr;
#pragma unroll
        for (int d = 8; d < 64; d <<= 1) {   // inclusive scan within the residue class lane & 7
            const int of = __shfl_up(vf, d), orv = __shfl_up(vr, d);
            if (lane >= d) { vf += of; vr += orv; }
        }
        if (in) {   // (one array, three values per group: one address register pair)
            out.gpos[3 * (long long)g] = (base_rows + (ir - ablocks)) * kPfWgRows;
            out.gpos[3 * (long long)g + 1] = (base_cand + (ic - capu)) * 1024;
            out.gpos[3 * (long long)g + 2] = dir == 0 ? (long long)(carry_f + vf - f0) : (long long)(carry_r + vr - r0);
        }
        base_rows += __shfl(ir, 63);
        base_cand += __shfl(ic, 63);
        carry_f += __shfl(vf, 56 + (lane & 7));   // the class total sits in the class's last lane
        carry_r += __shfl(vr, 56 + (lane & 7));
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) tot_swept += __shfl_xor(tot_swept, d);
    // lanes 0..7 hold the totals of XCD 0..7 (group g = lane in chunk 0: class lane & 7)
    int n_items = carry_f + carry_r;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) n_items = max(n_items, __shfl_xor(n_items, d));
    if (lane < 8) out.fwd_items_x[lane] = carry_f;
    if (lane == 0) {
        const long long n8 = 8LL * n_items;
        PlanSummary sm;
        sm.ok = (base_rows * kPfWgRows <= out.rows_cap && base_cand * 1024 <= out.cand_cap && n8 <= out.items_cap) ? 1 : 0;
        sm.n_items = sm.ok ? (int)n8 : 0;
        sm.cmp_rows = base_rows * kPfWgRows;
        sm.cand_elems = base_cand * 1024;
        sm.items_needed = n8;
        sm.swept_desc_pairs = tot_swept;
        *out.summary = sm;
    }
}

// one thread per group: its sweep descriptor (A = its compacted rows, B = the streamed image), candidate list and work
// items.  The descriptor arrays arrive zeroed: only the fields that are not zero are stored, one at a time (few registers).
__global__ void pf_plan_write_kernel(const PlanGroup* __restrict__ groups, int n_groups, const int* __restrict__ gtot, PlanOut out) {
    MSFM_TAIL_PRIO();
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int ok = out.summary->ok;
    const long long rows = gtot[g];
    const long long rows512 = (rows + kPfWgRows - 1) / kPfWgRows * kPfWgRows;
    const int ablocks = (int)(rows512 / kPfWgRows);
    const long long row0 = out.gpos[3 * (long long)g], cand0 = out.gpos[3 * (long long)g + 1];
    const int cap = ok ? plan_group_cap_units(rows) * 1024 : 0;
    const PlanGroup* G = groups + g;
#define MSFM_PLAN_FENCE() asm volatile("" ::: "memory")   /* loads stay next to their stores: few live registers */
    PairDesc* vd = out.vpairs + g;
    vd->n1 = ok ? (int)rows : 0;
    vd->n2 = G->n2;
    vd->a_blocks256 = ok ? ablocks : 0;
    MSFM_PLAN_FENCE();
    vd->b_tiles = G->b_tiles;
    vd->n1pad = ok ? (int)rows512 : 0;
    vd->n2pad = G->n2pad;
    MSFM_PLAN_FENCE();
    vd->valid = 1;
    vd->path = 1;
    vd->ranges = G->ranges;
    vd->rp_off = row0;   // (sweep 1' of route Q writes its row results per compacted row)
    MSFM_PLAN_FENCE();
    PfPair* vp = out.vpf + g;
    vp->a_h = out.zero_row;
    vp->a_rows = out.row_src + row0;
    MSFM_PLAN_FENCE();
    vp->b_h = G->b_h;
    vp->b_nrm = G->b_nrm;
    MSFM_PLAN_FENCE();
    vp->b_c = G->b_c;
    vp->a_c = G->a_c;
    vp->b_h0 = G->b_h0;
    MSFM_PLAN_FENCE();
    vp->tu_off = row0;
    vp->cand_off = cand0;
    vp->cand_cap = cap;
    vp->use = (ok && rows > 0) ? 1 : 0;
    MSFM_PLAN_FENCE();
    CandList* L = out.lists + g;
    L->pair = -1;
    L->mode = 1 + G->dir;
    L->off = cand0;
    L->cap = cap;
    MSFM_PLAN_FENCE();
    L->live_idx = out.live_idx + row0;
    L->row_pair = out.row_pair + row0;
    MSFM_PLAN_FENCE();
    L->cmp_tu = out.cmp_tu + row0;
    L->cmp_n2 = out.cmp_n2 ? out.cmp_n2 + row0 : nullptr;
    L->b_n2 = G->b_n2;
    MSFM_PLAN_FENCE();
    out.grow0[g] = ok ? row0 : -1;   // (-1: pf_member_rows_kernel / pf_assign_kernel see an invalid plan, nothing is assigned)
    if (!ok) return;
    const int nblk = G->bt_end - G->bt_begin, ranges = G->ranges, bt0 = G->bt_begin;
    long long k = out.gpos[3 * (long long)g + 2] + (G->dir == 1 ? out.fwd_items_x[g & 7] : 0);   // reverse items behind all forward ones
    for (int rg = 0; rg < ranges; ++rg) {
        const int t0 = bt0 + (int)((long long)nblk * rg / ranges), t1 = bt0 + (int)((long long)nblk * (rg + 1) / ranges);
        for (int ab = 0; ab < ablocks; ++ab, ++k) {
            WorkItem* w = out.items + (k * 8 + (g & 7));
            w->pair = g;
            w->a_blk = ab;
            w->bt_begin = t0;
            w->bt_end = t1;
            w->range = rg;
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
// the MFMA), and the address of its fp16 operand row.
// One workgroup per (pair, direction, chunk of 1024 rows): it counts its live rows per member (LDS), reserves that many slots of
// every member with ONE atomicSub on the member's count -- the counts the thresholds / prune kernel left there are exactly the rows
// this kernel finds, they tick down to zero -- and fills them.  Round 3 walked a whole (pair, direction) per workgroup with the fill
// cursors in LDS: five chunks in a row, each a chain of dependent round trips -- 65 000 waves living 80 us each, 79 % of it in
// SQ_WAIT_ANY, 1.3 TB/s.  Every load of a thread is issued before anything depends on one.   grid = (2 * pairs, chunks)
constexpr int kAssignChunk = 1024;
__global__ __launch_bounds__(256) void pf_assign_kernel(
    const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf, const PlanPair* __restrict__ pp_plan,
    const float* __restrict__ tuv, const unsigned* __restrict__ colmask, const long long* __restrict__ mrow, int* __restrict__ cnt,
    int* __restrict__ live_idx, int* __restrict__ row_pair, float* __restrict__ cmp_tu,
    const _Float16** __restrict__ row_src, int row_halfs /* 136: fp16 rows, 88: byte rows */,
    unsigned long long* __restrict__ best, unsigned long long* __restrict__ second,
    int norms_only /* plan A of route Q: the sweep wants -|a|^2 (accumulator = -S~/2), not T - |a|^2 */,
    int* __restrict__ cmp_n2 /* integer route: n' per compact row (null otherwise) */) {
    MSFM_TAIL_PRIO();
    const int p = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const PlanPair pl = pp_plan[p];
    if (pl.fwd_member < 0) return;
    const PairDesc pd = pairs[p];
    const int n = dir ? pd.n2 : pd.n1;
    const int e_begin = blockIdx.y * kAssignChunk;
    if (e_begin >= n) return;
    const PfPair pp = pf[p];
    const long long off = dir ? pp.tv_off : pp.tu_off;
    const float* nrm = dir ? pp.b_nrm : pp.a_nrm;
    const _Float16* src = dir ? pp.b_h : pp.a_h;
    const int* n2 = dir ? pp.b_n2 : pp.a_n2;
    const bool with_n2 = cmp_n2 != nullptr && n2 != nullptr;
    const int bits = dir ? pl.rev_bits : 1, member0 = dir ? pl.rev_member0 : pl.fwd_member;
    __shared__ int hist[32];
    __shared__ long long base[32];
    // ---- all loads of the thread (clamped: the arrays cover the padded rows)
    float t4[4], n4[4];
    unsigned m4[4];
    int x4[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = e_begin + (int)threadIdx.x + j * 256, ec = e < n ? e : n - 1;
        t4[j] = tuv[off + ec];
        m4[j] = dir ? colmask[off + ec] : 1u;
        n4[j] = nrm[ec];
        if (with_n2) x4[j] = n2[ec];
        if (e >= n) t4[j] = -f_inf();
    }
    long long my_row0 = -1;
    if ((int)threadIdx.x < bits) my_row0 = mrow[member0 + threadIdx.x];
    if (threadIdx.x < 32) hist[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (t4[j] == -f_inf()) continue;
        for (unsigned m = m4[j]; m; m &= m - 1) atomicAdd(&hist[__builtin_ctz(m)], 1);
    }
    __syncthreads();
    if ((int)threadIdx.x < bits) {
        const int h = hist[threadIdx.x];
        // (an invalid plan -- capacity exceeded, the batch is re-run -- has no rows to give out)
        base[threadIdx.x] = (h > 0 && my_row0 >= 0) ? my_row0 + (long long)(atomicSub(&cnt[member0 + threadIdx.x], h) - h) : -1;
        hist[threadIdx.x] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t = t4[j];
        if (t == -f_inf()) continue;
        const int e = e_begin + (int)threadIdx.x + j * 256;
        // the reduce kernels only ever touch the slots of live rows / columns: "no candidate yet" is written here, for the
        // 6 % that are alive, instead of a memset over every slot of the batch
        best[off + e] = ~0ull;
        second[off + e] = ~0ull;
        for (unsigned m = m4[j]; m; m &= m - 1) {
            const int b = __builtin_ctz(m);
            const long long r0 = base[b];
            if (r0 < 0) continue;
            const long long k = r0 + atomicAdd(&hist[b], 1);
            live_idx[k] = e;
            row_pair[k] = p;
            cmp_tu[k] = norms_only ? -n4[j] : t - n4[j];
            if (with_n2) cmp_n2[k] = x4[j];
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
