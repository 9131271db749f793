// msfm_prefilter.hip.h -- MFMA prefilter + exact re-check: the fast way to the SAME bits.
//
// The exact-order kernel (dist_top2_kernel) spends 384 un-fusable VALU ops on every one of the
// n1*n2 descriptor pairs although only the two nearest neighbours per row/column matter.  This path
// finds a small, PROVABLY sufficient candidate set with the matrix cores and evaluates the pinned
// fp32 order only on it:
//
//   sweep 1  sweep_kernel<1>   S~ = |a|^2 + |b|^2 - 2 a~.b~ with fp16 operands on
//                              v_mfma_f32_32x32x16_f16 -- the norms ride in a ninth k-step (below), so the
//                              accumulator IS -S~/2; per row / column the two smallest S~ values.
//   thresholds_kernel          T = S~(2) + 2*eps, eps = rigorous bound on |S~ - S_exact| (below); rows /
//                              columns that provably cannot match get T = -inf (match lists only).
//   sweep 2  sweep_kernel<3>   live rows of one image (compacted) against the other image, per direction:
//                              the row threshold rides in the ninth k-step too, a hit is accumulator >= 0;
//            sweep_kernel<2>   dense variant (nothing pruned): S~ <= T_row[q] or S~ <= T_col[t];
//                              hits are appended to the pair's candidate list (a few per row).
//   exact_candidates_kernel    S_exact in the pinned accumulation order for the candidates only.
//   reduce + finalize          per row / column the best (atomicMin inside exact_candidates_kernel) and second best
//                              (S_exact, index) among the candidates -> the same kNN arrays merge_knn_kernel produces.
//
// Why the candidates suffice.  Let |S~ - S_exact| <= eps for every element of a row.  The two
// elements with the smallest S~ have S_exact <= S~(2) + eps, so the true second-smallest S_exact is
// <= S~(2) + eps.  A non-candidate has S~ > S~(2) + 2 eps, hence S_exact > S~(2) + eps: strictly
// worse than the true second neighbour, so it can neither enter the top-2 nor tie with it.
//
// eps.  a~ = fl16(a) (round to nearest): |a~_c - a_c| <= max(2^-11 |a_c|, 2^-25).  MFMA products of
// two fp16 values are exact in fp32; the fp32 accumulation of 128 terms errs by at most
// c_acc * sum|a~ b~| (we allow c_acc = 2^-13, an order of magnitude above 128 roundings).  With
// Cauchy-Schwarz:  |a~.b~ - a.b| <= (2^-10 + 2^-13 + ...) sqrt(na nb) + 2^-25 sqrt(128) (sqrt(na)+sqrt(nb)).
// Norms (fp32 sums of 128 squares), the three fp32 ops forming S~, and the distance of the pinned
// fp32 order from the real-number value add < 1e-4 (na + nb).  Using sqrt(na nb) <= (na+nb)/2:
//     eps(q,t) <= kEpsRel (na + nb) + kEpsAbs (sqrt(na) + sqrt(nb)),  kEpsRel = 1.5e-3, kEpsAbs = 1e-6
// (the derivation gives 1.2e-3 / 8e-7).  Per row we use nb -> max_t nb.  Images with |value| > 6e4
// (fp16 overflow) or non-finite norms are not prefiltered; a pair whose candidate list overflows
// falls back to the brute-force exact kernel.  tests/test_gpu_prefilter.py checks the bound
// empirically and the end results bit-for-bit.
//
// Norms in the MFMA.  Per image c = 2^k with max|row|^2 / 2 / c in (2^11, 2^12]; each row stores the fp16
// quadruple e = [h_hi, h_lo, c, c] with h = |row|^2 / 2 / c split into two fp16 (hi + lo: 2^-21 relative).
// The A side (built in registers per work item) is [-c, -c, x_hi, x_lo] with x = X / c, X = -|a|^2/2 (sweep
// 1, dense sweep 2) or X = (T_row - |a|^2)/2 (compacted sweep 2), so the ninth k-step adds -|b|^2/2 + X and
// the accumulator is -S~/2 resp. -(S~ - T_row)/2.  Extra error: 2^-21 relative on the norms (inside the
// 1e-4 budget above) plus, should the MFMA flush fp16 subnormals, <= 4 * 2^-14 c absolute on the
// accumulator -> eps_norm = 2^-11 c is added to eps.  Pairs whose |a|^2 maximum exceeds 8x the |b|^2
// maximum (x would leave the fp16 range) take the brute-force path.
#pragma once
#include "msfm_kernels.hip.h"

namespace msfm {

constexpr float kEpsRel = 1.5e-3f;
constexpr float kEpsAbs = 1.0e-6f;
constexpr float kF16Safe = 6.0e4f;

constexpr int kPfRB = 2;                  // 32-row MFMA blocks per wave (their A fragments stay in registers)
constexpr int kPfWaveRows = 32 * kPfRB;   // 64
constexpr int kPfWaves = 8;               // two per SIMD: waves w and w + 4 share a SIMD and alternate roles (msfm_sweep.hip.h)
constexpr int kPfWgRows = kPfWaves * kPfWaveRows;  // 512 A rows per workgroup / work item
constexpr int kPfThreads = 64 * kPfWaves;
constexpr int kPfBT = 64;                 // B rows per tile
// One fp16 descriptor row of the prefilter operand = 17 granules of 16 B: 16 data granules (128 halfs) + the norm
// quadruple [h_hi, h_lo, c, c, 0, 0, 0, 0].  The odd granule count is the LDS bank swizzle: row r starts at bank
// (68 r) mod 64 = (4 r) mod 64, so the 16 rows a ds_read_b128 lane group touches at one k-step cover all 64 banks, and a
// k-step is a CONSTANT byte offset from the row (no XOR on the address: the operand reads are base + immediate).
constexpr int kPfRowHalfs = 136;
constexpr int kPfRowBytes = kPfRowHalfs * 2;    // 272
constexpr int kPfTileBytes = kPfBT * kPfRowBytes;  // 17408 B = 17 LDS-DMA pieces of 1 KiB
constexpr int kPfTilePieces = kPfTileBytes / 1024;
constexpr int kPfRing = 4;                // B tiles in LDS: DMA runs three tiles ahead
constexpr int kPfColClasses = 4;          // column partials: maxima over 4 disjoint row classes per 512-row block
constexpr int kPfCandBuf = 256;           // per-wave LDS candidate buffer (sweep 2), int2 entries
// B ring 68 KiB | per-wave column-threshold rings 8 KiB (dense sweep 2) | per-wave candidate buffers 16 KiB (sweep 2) |
// column class maxima 2 tiles x 64 x 4 floats = 2 KiB: one workgroup per CU
constexpr int kPfLdsBytes = kPfRing * kPfTileBytes + kPfWaves * kPfRing * 64 * 4 + kPfWaves * kPfCandBuf * 8 +
                            2 * kPfBT * kPfColClasses * 4;
static_assert(kPfTileBytes % 1024 == 0 && kPfTilePieces == 17, "tile = 17 DMA pieces");

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct PfPair {            // per-pair extras of the prefilter path (parallel to PairDesc)
    const _Float16* a_h;   // fp16 operand rows [npad][136 halfs] (quadruple in the 17th granule); compacted sweep: a zero row
    const _Float16* b_h;
    const _Float16* const* a_rows;  // compacted sweep only: per compacted row the address of its operand row (null: none)
    const float* a_nrm;    // |row|^2, +inf on padding rows
    const float* b_nrm;
    float a_c, b_c;        // the images' scales c (powers of two)
    float a_nrm_max, b_nrm_max;
    long long tu_off;      // row thresholds T (S-space) [n1pad]; compacted sweep: T - |a|^2 per live row
    long long tv_off;      // column thresholds T (S-space) [n2pad]
    long long cand_off;    // candidate list base
    int cand_cap;
    int use;               // 1: prefiltered; 0: not safe -> exact brute force
    int i8;                // 1: byte images on the integer matrix cores: a_h / b_h are 176-byte rows, the norms are 2 h
                           //    (msfm_sweep_i8.hip.h), eps = 2
    int a_h0, b_h0;        // i8: the images' centres H0 (the digit k-step of a row carries H0 - h)
    int pad;
    const float* a_err;    // route Q (msfm_q8.hip.h), twin pairs only: per-row quantisation error norms of the byte twins;
    const float* b_err;    //   the images' maxima of them travel in a_c / b_c
    const int* a_n2;       // i8 (byte images, not twins): n' = |x - 128|^2 per row, exactly (norms hold 2 floor(n' / 2)): what turns a
    const int* b_n2;       //   candidate's accumulator into the exact S (pf_exact_candidates_kernel<4>)
};

__device__ __forceinline__ void glds_copy_bytes(const void* g, void* lds, int bytes, int tid, int nthreads) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const char* gp = reinterpret_cast<const char*>(g);
    char* lp = reinterpret_cast<char*>(lds);
#pragma unroll 1
    for (int piece = wave; piece * 1024 < bytes; piece += nthreads / 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lp + piece * 1024), 16, 0, 0);
}

__device__ __forceinline__ void v2_merge(float& s0, float& s1, float b0, float b1) {
    s1 = fminf(fmaxf(s0, b0), fminf(s1, b1));
    s0 = fminf(s0, b0);
}

#include "msfm_sweep.hip.h"
#include "msfm_sweep_i8.hip.h"

// thresholds: fold the pass-1 partials; T = S~(2) + 2 eps, stored in u- / v-space.
// With `prune` (match lists, not the knnMatch-level API) a row / column that PROVABLY cannot yield a match
// gets T = -inf: pass 1 knows the exact minimum S~min of the row and an upper bound S~2ub of its second
// smallest, so S0_exact >= S~min - eps and S1_exact <= S~2ub + eps; if sqrt(S0lb) >= ratio * sqrt(S1ub) the
// Lowe test d0 < ratio * d1 fails whatever the exact values are, and if sqrt(S0lb) > max_distance the
// distance cut removes it.  pf_finalize_kernel reports such rows as "no neighbour".
// grid = (ceil(max_npad/256), n_pairs)
struct PruneParams {
    int prune;
    float ratio;
    float max_distance;  // rounded up to float
};

__device__ __forceinline__ bool pf_dead(float s0, float s1, float nrm, float eps, float other_max, PruneParams pr) {
    if (!pr.prune) return false;
    const float tiny = 1e-5f * (fabsf(s0) + fabsf(s1) + nrm + other_max);
    const float s0lb = fmaxf(s0 - eps - tiny, 0.f);
    const float s1ub = s1 + eps + tiny;
    const float d0lb = sqrtf(s0lb) * (1.f - 1e-6f);
    const bool ratio_fails = pr.ratio > 0.f && d0lb >= pr.ratio * sqrtf(s1ub) * (1.f + 1e-5f);
    const bool too_far = d0lb > pr.max_distance * (1.f + 1e-5f);
    return ratio_fails || too_far;
}

struct PlanPair {             // where the members of a pair sit in the member arrays
    int fwd_member;           // -1: the pair is not on the compacted path
    int rev_member0;          // members rev_member0 + bit, bit < rev_bits
    int rev_bits;
    int pad;
};


// live counting for the plan of the compacted sweep 2 (msfm_plan.hip.h), done while the thresholds are in registers:
// rows per member (dir 0: live rows of image 1; dir 1: per mask bit the live columns that carry it) and per group
struct PlanCounts {
    const PlanPair* pp_plan;     // null: no plan (dense sweep 2)
    const int* member_group;
    int* cnt;                    // [members], zeroed
    int* gtot;                   // [groups], zeroed
};

__global__ void pf_thresholds_kernel(const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf,
                                     const float* __restrict__ rp_s0, const float* __restrict__ rp_s1,
                                     const float* __restrict__ cp_s0, unsigned* __restrict__ colmask,
                                     float* __restrict__ tu, float* __restrict__ tv, PruneParams pr, PlanCounts plan,
                                     int marked /* route Q: tu / tv arrive holding -inf for what pf_prune_q8_kernel found dead: skipped */) {
    MSFM_TAIL_PRIO();
    const PairDesc pd = pairs[blockIdx.y];
    const PfPair pp = pf[blockIdx.y];
    if (!pd.valid || !pp.use) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    bool row_live = false;       // for the plan counts below
    unsigned col_bits = 0;
    const float eps_norm = 4.8828125e-4f * fmaxf(pp.a_c, pp.b_c);  // 2^-11 c: fp16 subnormal flush of the norm quadruples
    // ---- every load of this thread first, at clamped (always valid) indices: one memory round trip per wave instead of a chain of
    // them (rows, the columns' partials, their norms); see pf_prune_q8_kernel
    const float2* cp2 = reinterpret_cast<const float2*>(cp_s0);
    const int* cpk = reinterpret_cast<const int*>(cp_s0);   // (the integer sweeps write 4-byte packed partials: msfm_cp_pack)
    const int nb = pd.a_blocks256;
    const bool in_r = e < pd.n1pad, in_c = e < pd.n2pad;
    const int er = in_r ? e : 0, ec = in_c ? e : 0;
    const float ld_rs0 = rp_s0[pd.rp_off + er], ld_rs1 = rp_s1[pd.rp_off + er];
    const float ld_anrm = pp.a_nrm[er], ld_bnrm = pp.b_nrm[ec];
    const float ld_tu = marked ? tu[pp.tu_off + er] : 0.f, ld_tv = marked ? tv[pp.tv_off + ec] : 0.f;
    float2 ld_cp[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const long long i = pd.cp_off + (long long)min(p, nb - 1) * pd.n2pad + ec;
        ld_cp[p] = pp.i8 ? make_float2(__int_as_float(cpk[i]), 0.f) : cp2[i];
    }
    auto partial16 = [&](int p) -> float2 { return pp.i8 ? i8_cp_unpack(__float_as_int(ld_cp[p].x)) : ld_cp[p]; };
    if (in_r && !(marked && ld_tu == -f_inf())) {
        float s0 = f_inf(), s1 = f_inf();
        v2_merge(s0, s1, ld_rs0, ld_rs1);
        for (int p = 1; p < pd.ranges; ++p) {   // (a pair split into B ranges: small batches only)
            const long long o = pd.rp_off + (long long)p * pd.n1pad + e;
            v2_merge(s0, s1, rp_s0[o], rp_s1[o]);
        }
        // s1 = S~(2); eps_row = rel*(na + nb_max) + abs*(sqrt(na)+sqrt(nb_max)) + eps_norm; threshold in S-space
        const float na = ld_anrm;
        const float eps = pp.i8 ? kI8Eps : kEpsRel * (na + pp.b_nrm_max) + kEpsAbs * (sqrtf(na) + sqrtf(pp.b_nrm_max)) + eps_norm;
        // + roundings here and in sweep 2's test (none on the integer path: S~, T and the test are exact integers)
        const float slack = pp.i8 ? 2.f * eps : 2.f * eps + 1e-5f * (fabsf(s1) + na + pp.b_nrm_max);
        const bool live = (e < pd.n1) && !pf_dead(s0, s1, na, eps, pp.b_nrm_max, pr);
        tu[pp.tu_off + e] = live ? s1 + slack : -f_inf();
        row_live = live;
    }
    if (in_c && marked && ld_tv == -f_inf()) {
        if (colmask) colmask[pp.tv_off + e] = 0;
    } else if (in_c) {
        // column partials of sweep 1: per 512-row A block the two largest of the accumulator maxima (-S~/2) over four
        // disjoint row classes; the second smallest S~ over all of them is an upper bound of the column's second-smallest.
        // Up to 16 blocks (8192 rows) the blocks' minima stay in registers for the mask below.
        float s0 = f_inf(), s1 = f_inf();
        auto partial = [&](long long i) -> float2 { return pp.i8 ? i8_cp_unpack(cpk[i]) : cp2[i]; };
        float bmin[16];
        if (nb <= 16) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                bmin[p] = f_inf();
                if (p < nb) {
                    const float2 m = partial16(p);
                    bmin[p] = -2.f * m.x;
                    v2_merge(s0, s1, -2.f * m.x, -2.f * m.y);
                }
            }
        } else {
            for (int p = 0; p < nb; ++p) {
                const float2 m = partial(pd.cp_off + (long long)p * pd.n2pad + e);
                v2_merge(s0, s1, -2.f * m.x, -2.f * m.y);
            }
        }
        const float nb_ = ld_bnrm;
        const float eps = pp.i8 ? kI8Eps : kEpsRel * (nb_ + pp.a_nrm_max) + kEpsAbs * (sqrtf(nb_) + sqrtf(pp.a_nrm_max)) + eps_norm;
        const float slack = pp.i8 ? 2.f * eps : 2.f * eps + 1e-5f * (fabsf(s1) + nb_ + pp.a_nrm_max);
        const bool live = (e < pd.n2) && !pf_dead(s0, s1, nb_, eps, pp.a_nrm_max, pr);
        const float T = live ? s1 + slack : -f_inf();
        tv[pp.tv_off + e] = T;
        if (colmask) {
            // which 512-row blocks of image 1 can hold a candidate of this column at all: the block's smallest S~ must
            // not exceed T.  Bit b covers blocks [b g, (b + 1) g), g = ceil(blocks / 32) (1 up to 16384 rows).  The
            // reverse direction of sweep 2 only visits those blocks (pf_plan.hip.h).
            const int g = (nb + 31) / 32;
            unsigned mask = 0;
            if (live) {
                if (nb <= 16) {
#pragma unroll
                    for (int p = 0; p < 16; ++p)
                        if (p < nb && bmin[p] <= T) mask |= 1u << p;   // (g = 1)
                } else {
                    for (int p = 0; p < nb; ++p) {
                        const float smin = -2.f * partial(pd.cp_off + (long long)p * pd.n2pad + e).x;
                        if (smin <= T) mask |= 1u << (p / g);
                    }
                }
            }
            colmask[pp.tv_off + e] = mask;
            col_bits = mask;
        }
    }
    if (plan.pp_plan) {
        const PlanPair pl = plan.pp_plan[blockIdx.y];
        if (pl.fwd_member < 0) return;   // (block-uniform)
        __shared__ int hist[33];
        if (threadIdx.x < 33) hist[threadIdx.x] = 0;
        __syncthreads();
        const unsigned long long rl = __ballot(row_live);
        if ((threadIdx.x & 63) == 0 && rl) atomicAdd(&hist[32], __popcll(rl));
        while (col_bits) {
            const int bit = __builtin_ctz(col_bits);
            col_bits &= col_bits - 1;
            atomicAdd(&hist[bit], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0 && hist[32]) {
            atomicAdd(&plan.cnt[pl.fwd_member], hist[32]);
            atomicAdd(&plan.gtot[plan.member_group[pl.fwd_member]], hist[32]);
        }
        if ((int)threadIdx.x < pl.rev_bits && hist[threadIdx.x]) {
            atomicAdd(&plan.cnt[pl.rev_member0 + threadIdx.x], hist[threadIdx.x]);
            atomicAdd(&plan.gtot[plan.member_group[pl.rev_member0 + threadIdx.x]], hist[threadIdx.x]);
        }
    }
}

// A candidate list: the records one sweep produced.  mode 0: real (q, t) of pair `pair`; mode 1: (k, t) with
// q = live_idx[k] (compacted live rows of image 1 against image 2); mode 2: (k, q) with t = live_idx[k]
// (compacted live rows of image 2 against image 1).  A compacted list serves a GROUP of pairs that share the
// streamed image: row k belongs to pair row_pair[k].
struct CandList {
    int pair;   // mode 0 only
    int mode;
    long long off;
    int cap;  // 0: unused list
    int pad;
    const int* live_idx;
    const int* row_pair;
    // integer-core route (byte pairs): what pf_exact_candidates_kernel<4> needs to turn a candidate's accumulator into the exact S --
    // per compacted row its hit-level source T - 2 h_a and its n', per row of the streamed image its n'
    const float* cmp_tu;
    const int* cmp_n2;
    const int* b_n2;
};

__device__ __forceinline__ unsigned long long pf_key(float s, int idx) {
    return ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)idx;  // s >= 0: uint order == float order
}

// ---------------------------------------------------------------------------------------------------------------------------
// Exact pinned-order S for every candidate, and the best / second-best (S, index) per row and column among them.
//
// What this kernel costs is decided by three things that round 3's version (16-candidate windows of every list dealt round-robin over
// all workgroups, two returning 64-bit atomics behind every candidate) got wrong -- measured, profiles/r04_exact_recheck.txt:
//  (1) ATOMICS, not bytes, were its limit: ~60 M device-scope 64-bit atomics per 25 M candidates at the ~19 G/s the part sustains
//      = the kernel's 3.6 ms, whatever the gather did (fetching 37 % less changed nothing).  Now every candidate first LOOKS at
//      its slots with plain loads (issued together with the metadata, long before they are needed): a key that does not beat the
//      slot's current best can only be a candidate for `second`, and only if it beats the current second -- most candidates of a row
//      are neither (a row has ~5; the running minimum of a random sequence changes ~2 times) and issue NO atomic.  A stale look is
//      harmless: the slots only ever decrease, so "not smaller than what I saw" implies "not smaller than what is there".
//  (2) The dependent chain record -> row tables -> pair table -> rows was walked once per candidate and wave, four round trips in a
//      row with nothing else in flight.  Now a workgroup STAGES a span of 256 consecutive candidates: thread t resolves candidate t
//      (three round trips, 256 candidates at once), leaves the two row addresses in LDS, and the 16-lane groups then stream through
//      the span with one round trip per candidate, two candidates in flight per group.
//  (3) WHERE a candidate is evaluated: a list (= one group of sweep 2) names rows of ONE streamed image (<= 2.5 MB at 5000 rows) and
//      arrives in chunks of <= 256 written by one wave of sweep 2 (64 compacted rows, each ~2-3 times per chunk).  List l is
//      evaluated on XCD l mod 8 only (a workgroup's XCD is blockIdx.x mod 8: hardware round-robin), its spans handed out by a
//      per-XCD cursor (a static deal left 40 % of the time to the slowest workgroup): the streamed image stays in that XCD's 4 MB L2,
//      a compacted row's repeats hit.  Memory fetches 21 GB -> 13 GB per 25 M candidates.
// 16 lanes per candidate (SSE order: lane L owns the lane partial k = L mod 16; AVX2 order: 32 partials -> 2 per lane; AVX-512: 64
// -> 4 per lane): in every order lane L needs the elements 16 j + L, j = 0..7, of both rows -- coalesced 64-byte reads.
// grid = 8 * workgroups per XCD (persistent), block = kExSpan threads.
constexpr int kExSpan = 256;
constexpr int kExLists = 1024;   // lists per XCD and round of the prefix table (4 per thread)

// lane l of a 16-lane row <- lane l + N of the same row (0 beyond the row): a DPP row shift, a VALU modifier -- the reductions below
// used eight LDS-crossbar shuffles (ds_bpermute) per candidate in round 3
template <int N>
__device__ __forceinline__ float pf_row_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}

// The named order's reduction of one candidate's 128 squared differences; av / bv: the lane's elements 16 j + sub, j = 0..7, of the two
// rows.  The result is valid in lane 0 of the 16-lane group (sub == 0) -- the lane that stores it.
template <int ORDER>
__device__ __forceinline__ float pf_exact_combine(const float (&av)[8], const float (&bv)[8]) {
    if (ORDER == 0) {
        // SSE: 16 lane partials p[L] = sum_j t(16 j + L)^2 (rounded product, rounded add, j ascending);
        // s[l] = ((p[l] + p[4+l]) + p[8+l]) + p[12+l], l = 0..3; result (s0 + s2) + (s1 + s3)
        float p = 0.f;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const float t = av[it] - bv[it];
            p = (it == 0) ? t * t : p + t * t;
        }
        const float s = ((p + pf_row_shl<4>(p)) + pf_row_shl<8>(p)) + pf_row_shl<12>(p);   // lanes 0..3
        const float u = s + pf_row_shl<2>(s);                                              // lane 0: s0 + s2, lane 1: s1 + s3
        return u + pf_row_shl<1>(u);
    } else if (ORDER == 3) {
        // AVX-512: 64 partials p[16 v + L]; lane L: four accumulators x two iterations, fused; y_l = (s_l + s_{l+8}) + (s_{l+4} + s_{l+12}),
        // result (y0 + y2) + (y1 + y3)
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float t0 = av[v] - bv[v];
            const float t1 = av[4 + v] - bv[4 + v];
            const float p = __builtin_fmaf(t1, t1, t0 * t0);
            s = (v == 0) ? p : s + p;
        }
        const float x = s + pf_row_shl<8>(s);   // lanes 0..7
        const float y = x + pf_row_shl<4>(x);   // lanes 0..3
        const float z = y + pf_row_shl<2>(y);   // lanes 0, 1
        return z + pf_row_shl<1>(z);
    } else {
        // AVX2: 32 partials; lane L owns L (elements 32 it + L) and L + 16 (elements 32 it + L + 16), fused;
        // s[l] = ((p[l] + p[8+l]) + p[16+l]) + p[24+l], l = 0..7; result ((s0+s1)+(s2+s3)) + ((s4+s5)+(s6+s7))
        float pa = 0.f, pb = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float ta = av[2 * it] - bv[2 * it];
            const float tb = av[2 * it + 1] - bv[2 * it + 1];
            pa = (it == 0) ? ta * ta : __builtin_fmaf(ta, ta, pa);
            pb = (it == 0) ? tb * tb : __builtin_fmaf(tb, tb, pb);
        }
        const float s = ((pa + pf_row_shl<8>(pa)) + pb) + pf_row_shl<8>(pb);   // lanes 0..7
        const float t = s + pf_row_shl<1>(s);                                  // lanes 0, 2, 4, 6
        const float v = t + pf_row_shl<2>(t);                                  // lanes 0, 4
        return v + pf_row_shl<4>(v);
    }
}

// One key into a slot's (best, second): msfm_fold_key (msfm_hostutil.h, where a g++-built test drives the same code through random
// arrival orders and stale looks) with the device's 64-bit atomicMin.
__device__ __forceinline__ void pf_fold(unsigned long long* __restrict__ best, unsigned long long* __restrict__ second, long long slot,
                                        unsigned long long key, unsigned long long sb, unsigned long long ss) {
    msfm_fold_key(best + slot, second + slot, key, sb, ss,
                  [](unsigned long long* w, unsigned long long k) -> unsigned long long { return atomicMin(w, k); });
}

// ORDER 4 (round 5) -- byte pairs on the integer-core route: NO row is read.  sweep_i8_kernel<3> hands over every candidate's accumulator
// acc = a'.b' - h_b + C (C = the compacted row's hit level, the C operand of its tile's first MFMA), and S = n'_a + (n'_b & 1) - 2 (acc - C)
// is the exact integer |a - b|^2 -- which on byte data IS the pinned fp32 order's result under every named order (every partial sum is an
// integer below 2^24; oracle/int_oracle.py pins that).  Phase 2 -- the gather of two 512-byte rows per candidate, 39 % of a step's HBM
// bytes on the float job -- does not exist here; the staging, the per-XCD span cursor and the look-before-atomic fold are shared.
template <int ORDER>
__global__ __launch_bounds__(kExSpan) void pf_exact_candidates_kernel(
    const PairDesc* __restrict__ pairs, const CandList* __restrict__ lists, const unsigned long long* __restrict__ cand_count,
    const int2* __restrict__ cand, unsigned long long* __restrict__ best, unsigned long long* __restrict__ second, int n_lists,
    int* __restrict__ cursors /* [8], zero: spans handed out per XCD */, const int* __restrict__ cand_val /* ORDER 4 */, int b_h0_unused) {
    MSFM_TAIL_PRIO();
    __shared__ int s_base[kExLists + 1];   // s_base[i] = spans of the XCD's lists before list i of the round
    __shared__ int s_part[kExSpan];
    __shared__ const float* s_a[kExSpan];
    __shared__ const float* s_b[kExSpan];
    __shared__ float s_res[kExSpan];
    __shared__ int s_g;
    const int xcd = blockIdx.x & 7, tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    const int my_lists = n_lists > xcd ? (n_lists - xcd + 7) / 8 : 0;
    if (tid == 0) s_g = atomicAdd(&cursors[xcd], 1);
    int round_base = 0;   // spans of the XCD's earlier rounds
  for (int l0 = 0; l0 < my_lists; l0 += kExLists) {
    const int nl = min(kExLists, my_lists - l0);
    // ---- span counts of this round's lists -> exclusive prefix sums in s_base ----
    int cnt4[4], mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 4 * tid + j;
        int ns = 0;
        if (i < nl) {
            const int lid = xcd + 8 * (l0 + i);
            const int cap = lists[lid].cap;
            const unsigned long long cc = cand_count[lid];
            const int n = (int)(cc < (unsigned long long)cap ? cc : (unsigned long long)cap);
            ns = (n + kExSpan - 1) / kExSpan;
        }
        cnt4[j] = ns;
        mine += ns;
    }
    __syncthreads();   // (the previous round's readers of s_base are done; thread 0's first s_g is visible)
    s_part[tid] = mine;
    __syncthreads();
    for (int d = 1; d < kExSpan; d <<= 1) {
        const int v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - mine;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_base[4 * tid + j] = run;
        run += cnt4[j];
    }
    if (tid == kExSpan - 1) s_base[kExLists] = run;
    __syncthreads();
    const int total = s_base[kExLists];
   for (;;) {
    const int g = s_g - round_base;      // this workgroup's span of the round (uniform)
    if (g >= total) break;               // the rest belongs to the next round, if any (s_g is kept)
    // the list of span g: the last i with s_base[i] <= g (lists without spans share their successor's base: "last" skips them)
    int lo = 0, hi = kExLists;           // invariant: s_base[lo] <= g < s_base[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_base[mid] <= g) lo = mid; else hi = mid;
    }
    const int lid = xcd + 8 * (l0 + lo);
    const CandList L = lists[lid];
    const int n = (int)(cand_count[lid] < (unsigned long long)L.cap ? cand_count[lid] : (unsigned long long)L.cap);
    const int c0 = (g - s_base[lo]) * kExSpan, c1 = min(n, c0 + kExSpan);
    // ---- phase 1: thread t resolves candidate t of the span (the tail's threads resolve the span's first one again: valid
    //      addresses for the whole-wave shuffles of phase 2) and looks at its slots
    const bool live = c0 + tid < c1;
    int2 qt = cand[L.off + (live ? c0 + tid : c0)];
    float s_int = 0.f;
    if (ORDER == 4) {
        // (compacted lists only: the integer route has no dense sweep 2)  C exactly as sweep_i8_kernel<3> forms its C operand
        const int acc = cand_val[L.off + (live ? c0 + tid : c0)];
        const int C = (int)floorf(fminf(fmaxf(0.5f * L.cmp_tu[qt.x], -5.0e8f), 5.0e8f));
        s_int = (float)(L.cmp_n2[qt.x] + (L.b_n2[qt.y] & 1) - 2 * (acc - C));
    }
    const int pair = L.mode == 0 ? L.pair : L.row_pair[qt.x];
    if (L.mode == 1) qt.x = L.live_idx[qt.x];
    if (L.mode == 2) qt = make_int2(qt.y, L.live_idx[qt.x]);
    const PairDesc* pd = pairs + pair;
    if (ORDER != 4) {
        s_a[tid] = pd->a_rawp + (size_t)qt.x * kDim;
        s_b[tid] = pd->b_rawp + (size_t)qt.y * kDim;
    }
    // A mode-1 list only serves the row direction, a mode-2 list only the column direction: the live rows of the OTHER
    // direction get their complete candidate sets from their own list.
    const long long slot_r = pd->kf_off + qt.x, slot_c = pd->kr_off + qt.y;
    unsigned long long sb_r = ~0ull, ss_r = ~0ull, sb_c = ~0ull, ss_c = ~0ull;
    if (live && L.mode != 2) sb_r = best[slot_r], ss_r = second[slot_r];
    if (live && L.mode != 1) sb_c = best[slot_c], ss_c = second[slot_c];
    __syncthreads();
    // the next span's number: fetched now (everybody has read this span's: that read is above the barrier), in flight during phase 2
    int g_next = 0;
    if (tid == 0) g_next = atomicAdd(&cursors[xcd], 1);
    // ---- phase 2: the 16-lane groups stream through the span, two candidates in flight per group
    const int n4 = ORDER == 4 ? 0 : (c1 - c0 + 3) & ~3;   // (whole waves take part in the shuffles)
    for (int k0 = grp; k0 < n4; k0 += 32) {
        const int k1 = k0 + 16 < n4 ? k0 + 16 : k0;   // (uniform per wave: a wave's four groups are four consecutive candidates)
        const float* a0 = s_a[k0];
        const float* b0 = s_b[k0];
        const float* a1 = s_a[k1];
        const float* b1 = s_b[k1];
        // (the permuted row copy: the lane's eight elements 16 j + sub are two float4, each load 256 contiguous bytes per group -- four
        // wave-level loads per candidate instead of sixteen: 55 % of the kernel's wave cycles were SQ_WAIT_INST_ANY, vector-memory issue)
        float av0[8], bv0[8], av1[8], bv1[8];
        {
            const float4* pa0 = reinterpret_cast<const float4*>(a0) + sub;
            const float4* pb0 = reinterpret_cast<const float4*>(b0) + sub;
            const float4* pa1 = reinterpret_cast<const float4*>(a1) + sub;
            const float4* pb1 = reinterpret_cast<const float4*>(b1) + sub;
            const float4 x0 = pa0[0], x1 = pa0[16], y0 = pb0[0], y1 = pb0[16];
            const float4 z0 = pa1[0], z1 = pa1[16], w0 = pb1[0], w1 = pb1[16];
            av0[0] = x0.x, av0[1] = x0.y, av0[2] = x0.z, av0[3] = x0.w, av0[4] = x1.x, av0[5] = x1.y, av0[6] = x1.z, av0[7] = x1.w;
            bv0[0] = y0.x, bv0[1] = y0.y, bv0[2] = y0.z, bv0[3] = y0.w, bv0[4] = y1.x, bv0[5] = y1.y, bv0[6] = y1.z, bv0[7] = y1.w;
            av1[0] = z0.x, av1[1] = z0.y, av1[2] = z0.z, av1[3] = z0.w, av1[4] = z1.x, av1[5] = z1.y, av1[6] = z1.z, av1[7] = z1.w;
            bv1[0] = w0.x, bv1[1] = w0.y, bv1[2] = w0.z, bv1[3] = w0.w, bv1[4] = w1.x, bv1[5] = w1.y, bv1[6] = w1.z, bv1[7] = w1.w;
        }
        const float r0 = pf_exact_combine<ORDER>(av0, bv0);
        const float r1 = pf_exact_combine<ORDER>(av1, bv1);
        if (sub == 0) {
            s_res[k0] = r0;
            s_res[k1] = r1;
        }
    }
    if (tid == 0) s_g = g_next;
    __syncthreads();
    // ---- phase 3: thread t folds candidate t into its slots
    if (live) {
        const float res = ORDER == 4 ? s_int : s_res[tid];
        if (res < f_inf()) {   // batchDistance never inserts a distance >= FLT_MAX
            if (L.mode != 2) pf_fold(best, second, slot_r, pf_key(res, qt.y), sb_r, ss_r);
            if (L.mode != 1) pf_fold(best, second, slot_c, pf_key(res, qt.x), sb_c, ss_c);
        }
    }
   }
    round_base += total;
  }
}

#include "msfm_plan.hip.h"
#include "msfm_q8.hip.h"
#include "msfm_store.hip.h"

// finalize: the same outputs as merge_knn_kernel (idx0, d0, d1, tie queue)
__global__ void pf_finalize_kernel(const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf,
                                   const float* __restrict__ tuv,
                                   const unsigned long long* __restrict__ best, const unsigned long long* __restrict__ second,
                                   int* __restrict__ k_i0, float* __restrict__ k_d0, float* __restrict__ k_d1,
                                   int* __restrict__ fix_count, int4* __restrict__ fix_list, int fix_cap) {
    MSFM_TAIL_PRIO();
    const PairDesc pd = pairs[blockIdx.y];
    const PfPair pp = pf[blockIdx.y];
    if (!pd.valid || !pp.use) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    for (int dir = 0; dir < 2; ++dir) {
        const int n = dir == 0 ? pd.n1 : pd.n2;
        if (e >= n) continue;
        const long long ko = (dir == 0 ? pd.kf_off : pd.kr_off) + e;
        // a pruned row / column (threshold -inf) may still own stray candidates of the other direction:
        // they are not its neighbour set, report "no neighbour"
        const bool dead = tuv[ko] == -f_inf();
        const unsigned long long b = dead ? ~0ull : best[ko], s = dead ? ~0ull : second[ko];
        int i0 = -1;
        float d0 = 3.402823466e+38f, d1 = 3.402823466e+38f;
        if (b != ~0ull) { i0 = (int)(unsigned)(b & 0xffffffffu); d0 = sqrtf(__uint_as_float((unsigned)(b >> 32))); }
        if (s != ~0ull) d1 = sqrtf(__uint_as_float((unsigned)(s >> 32)));
        k_i0[ko] = i0;
        k_d0[ko] = d0;
        k_d1[ko] = d1;
        if (i0 >= 0 && d0 == d1) {
            const int slot = atomicAdd(fix_count, 1);
            if (slot < fix_cap) fix_list[slot] = make_int4((int)blockIdx.y, dir, e, 0);
        }
    }
}

}  // namespace msfm
