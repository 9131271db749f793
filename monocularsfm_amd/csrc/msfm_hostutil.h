// msfm_hostutil.h -- small pure functions shared by device and host code, compilable on their own (tests/test_hostutil.py builds a
// g++ driver around them): the 4-byte packing of the integer sweeps' column partials, the cost marks of a call's sub-batches, the
// scratch memory one image pair of a sub-batch needs, the fold of a candidate's key into a slot's (best, second).
#pragma once
#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MSFM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define MSFM_HD inline
#endif

// Column partial of the integer sweeps -- per column and 512-row block the two largest accumulators (relative to the item's K:
// -S~/2, <= 1, > -2^23) -- in FOUR bytes instead of the float pipeline's float2: the largest exactly (24 bits; -2^23 = "no real
// row"), and an 8-bit code of how far below it the second largest is, rounded UP: code c stands for
// ((16 + (c & 15)) << (c >> 4)) - 16 (exact up to 15, then 1/16 steps; 255 = "no second").  The second value is only ever used as
// an upper bound of the column's second-smallest S^ -- a smaller accumulator is a larger S^ -- so rounding the gap up keeps every
// bound valid; it halves what sweep 1 writes and the thresholds / prune kernels read (3.3 GB -> 1.65 GB per 8128-pair job).
MSFM_HD int msfm_cp_pack(int hi, bool hi_valid, int lo, bool lo_valid) {
    unsigned c = 255u;
    if (lo_valid) {
        const unsigned x = (unsigned)(hi - lo) + 16u;          // gap + 16 >= 16
        int e = 27 - __builtin_clz(x);                         // floor(log2 x) - 4 >= 0
        unsigned m = (x + (1u << e) - 1u) >> e;                // 16 .. 32, rounded up
        e += (int)(m >> 5);                                    // (m == 32 -> 16 at the next exponent)
        m = (m >> 5) ? 16u : m;
        const unsigned code = ((unsigned)e << 4) | (m - 16u);
        c = code >= 255u ? 255u : code;                        // (e > 15 lands here as well)
    }
    return hi_valid ? (int)(((unsigned)hi << 8) | c) : (int)0x80000000;
}
constexpr int kCpNone = -(1 << 23);                                           // the "no real row" value of the 24-bit field
MSFM_HD int msfm_cp_hi(int code) { return code >> 8; }                        // kCpNone: no real row
MSFM_HD bool msfm_cp_has_second(int code) { return (code & 255) != 255; }
MSFM_HD int msfm_cp_gap(int code) { const int c = code & 255; return ((16 + (c & 15)) << (c >> 4)) - 16; }

// Cumulative-cost marks of the parts a large call is cut into (msfm_set_pipeline): n_sub parts whose size shrinks linearly towards
// the end of the call, the last one to `taper` of the average -- what follows the LAST sweep 1 of a call (that part's thresholds,
// plan, sweep 2, exact re-check, epilogue) has nothing left to hide behind, and it is proportional to the part's size.
// -> marks[0] = 0 < marks[1] < ... < marks[n_sub] = total.
inline std::vector<long long> msfm_pipeline_marks(long long total, long long n_sub, double taper) {
    std::vector<long long> marks;
    if (n_sub < 2) return marks;
    const double last = taper, first = 2.0 - last;
    double acc = 0.0;
    marks.push_back(0);
    for (long long k = 0; k < n_sub; ++k) {
        acc += first - (first - last) * (double)k / (double)(n_sub - 1);
        marks.push_back(k + 1 == n_sub ? total : (long long)((double)total * acc / (double)n_sub));
    }
    return marks;
}

// Device scratch ONE image pair adds to a sub-batch, in bytes -- what MatchJob::build (msfm_job.hip.h) cuts a call by, and
// what the buffers of a scratch set really hold (round 3 charged 2 x a_blocks128 x n2pad 4-byte units per pair whatever the
// route: 2.6 - 3.4 x what the matrix-core route allocates, so a "48 GiB" budget produced 200 sub-batches of config 4 on a
// 58 GiB footprint).  n1pad / n2pad: rows padded to 512; blocks128 / blocks512: 128- / 512-row blocks of image 1.
//   route 1, matrix cores, match lists with ratio <= 0.95 (compacted sweep 2): row partials 2 x 4 B x n1pad | column partials
//     8 B (fp16 route: float2; the integer routes use 4 of them) per 512-row block and column | thresholds, block masks, best /
//     second keys, final kNN arrays: 36 B per padded row and column | staged + compact match lists 24 B per row | the plan of
//     sweep 2 and its candidate lists for ONE SIXTEENTH of the rows alive (measured: 1 - 3.3 %; buffers that turn out too small are
//     re-grown and the sub-batch re-run, this is only the cut): 20 B + 8 candidates x 8 B per compacted row, a column once per
//     512-row block group (at most 32);
//   route 3, INTEGER matrix cores (both images byte stores, match lists with ratio <= 0.95): as route 1 with 4-byte column partials and
//     without the final kNN arrays (the epilogue reads the reduce slots: 24 B per padded row and column), + 4 B per candidate and
//     compacted row for the exact-S hand-over of sweep 2;
//   route 4, route 1 for float stores whose byte twins are coarse (values beyond 0.625): one EIGHTH of the rows instead of a sixteenth;
//   route 2, matrix cores, dense sweep 2 (kNN-level API, ratio > 0.95): candidate lists of 16 entries per row and column instead;
//   route 0, brute force: three 4-byte row partials per padded row, three per 128-row block and column.
// (A mixed sub-batch -- byte pairs next to float pairs -- allocates by route 1's sizes for all of them: this is the cut, not a cap.)
inline long long msfm_pair_scratch_bytes(int n1, int n2, int n1pad, int n2pad, int blocks128, int blocks512, int route) {
    if (n1 <= 0 || n2 <= 0) return 0;
    const long long slots = (long long)n1pad + n2pad;
    const long long common = (route == 3 ? 24LL : 36LL) * slots + 24LL * n1 + 1024;
    if (route == 0) return common + 12LL * n1pad + 12LL * (long long)blocks128 * n2pad;
    const long long partials = 8LL * n1pad + (route == 3 ? 4LL : 8LL) * (long long)blocks512 * n2pad;
    if (route == 2) return common + partials + 128LL * ((long long)n1 + n2) + 16384;
    const long long bits = blocks512 < 32 ? blocks512 : 32;
    // (route 4 = route 1 with COARSE byte twins, msfm_q8.hip.h: plan A holds every live column once per block group and the prune kernel
    // only knows live / dead -- twice the compacted rows of the direct route)
    const long long cmp_rows = ((long long)n1 + (long long)n2 * bits) / (route == 4 ? 8 : 16) + 1024;
    return common + partials + (route == 3 ? 120LL : 84LL) * cmp_rows;
}

// Room for the device-side plan of sweep 2 (compacted rows, candidate entries, work items) of a sub-batch that could compact `ub` rows at
// most, in `groups` groups of at most `max_ranges` ranges -- MatchJob sizes the buffers BEFORE the plan exists.  The prediction is what
// the previous sub-batch needed (hint_*), scaled by the ratio of the two upper bounds and void beyond a factor two: the parts of a call and
// the calls of a repeated job are alike, a call of another kind (the pre-emptive filter's 100-row subsets before the full images) says
// nothing.  Without one: `prior_16ths` / 16 of the rows (5 on the direct routes, 2.5 -> the caller passes rows_ub_all_bits * 5 / 32 as
// `ub_prior`), and all rows for a small sub-batch (up to 2 M rows = 0.25 GB of plan buffers: on 100-row subsets more than half stay alive).
// A buffer that EXISTS (have_*) is kept as long as it holds the prediction + 1/8, a fresh one gets 1.5 x the prediction: growing means
// replacing the buffer.  (Round 5, first half: absolute hints -- the buffers kept from the small call counted as large enough for the
// large call's first sub-batch, whose plan then overflowed: profiles/r05_cli_cold_call.txt.)
struct MsfmPlanRoom { long long rows, cand, items; bool hinted; };
inline MsfmPlanRoom msfm_plan_room(long long ub, long long ub_prior, long long groups, long long max_ranges, int wg_rows,
                                   long long hint_ub, long long hint_rows, long long hint_cand, long long hint_items,
                                   long long have_rows, long long have_cand, long long have_items) {
    auto mx = [](long long a, long long b) { return a > b ? a : b; };
    const long long slack = (long long)wg_rows * groups + wg_rows;
    const bool hinted = hint_rows > 0 && hint_ub > 0 && ub <= 2 * hint_ub && 2 * ub >= hint_ub;
    const double s = hinted ? (double)ub / (double)hint_ub : 0.0;
    const long long h_rows = (long long)((double)hint_rows * s), h_cand = (long long)((double)hint_cand * s), h_items = (long long)((double)hint_items * s);
    const long long prior = mx(ub_prior, ub < (1LL << 21) ? ub : (1LL << 21));
    const long long rows_min = (hinted ? h_rows + h_rows / 8 : prior) + slack;
    MsfmPlanRoom r;
    r.hinted = hinted;
    r.rows = have_rows >= rows_min ? have_rows : mx(h_rows + h_rows / 2, prior) + slack;
    // (per group: 8 entries per compacted row rounded up to 1024, + 1024 -- plan_group_cap_units)
    const long long cand_min = mx(8 * rows_min + 2048 * groups, h_cand);
    r.cand = have_cand >= cand_min ? have_cand : mx(8 * r.rows + 2048 * groups, h_cand);
    // (the list holds 8 x the longest per-XCD sub-list: twice the balanced size leaves room for skew)
    const long long items_min = mx(2 * ((rows_min / wg_rows + groups) * max_ranges + 64) / 8 * 8, (h_items + 64) / 8 * 8);
    r.items = have_items >= items_min ? have_items : mx(2 * ((r.rows / wg_rows + groups) * max_ranges + 64) / 8 * 8, (h_items + 64) / 8 * 8);
    return r;
}

// Where the page-locked piece of a call's result lists that holds byte `off` ends (GrowPinned, msfm_ctx.hip.h): the first 32 MiB are cut
// 1, 1, 2, 4, 8, 16 MiB, then 32-MiB pieces -- page-locking costs 0.2 ms per MiB, and a small call's lists should cost a small piece.
inline unsigned long long msfm_pinned_piece_end(unsigned long long off) {
    const unsigned long long piece = 32ull << 20;
    if (off >= piece) return (off / piece + 1) * piece;
    unsigned long long end = 1ull << 20;
    while (end <= off) end *= 2;
    return end;
}

// One key into a slot's (best, second) -- the reduction of the exact re-check (pf_exact_candidates_kernel).  best ends as the
// smallest key the slot ever saw, second as the second smallest DISTINCT key: every key but the final minimum loses exactly once
// against `best` (when it arrives, or when a smaller one displaces it) and is then offered to `second`; a key arriving twice meets
// itself and is dropped.  seen_best / seen_second: what a plain load saw in the two words some time BEFORE (any earlier state: the
// words only ever decrease, so a stale look is >= the truth): a key that does not beat what was seen cannot change the word and
// skips the atomic -- most candidates of a row do.  amin(word, key) = atomic minimum returning the old value.  ~0 = "nothing yet".
template <class AtomicMin>
MSFM_HD void msfm_fold_key(unsigned long long* best, unsigned long long* second, unsigned long long key, unsigned long long seen_best,
                           unsigned long long seen_second, AtomicMin amin) {
    if (key == seen_best) return;
    unsigned long long loser = key;
    if (key < seen_best) {
        const unsigned long long old = amin(best, key);
        if (old == key) return;
        loser = old > key ? old : key;
        if (loser == ~0ull) return;
    }
    if (loser < seen_second) (void)amin(second, loser);
}
