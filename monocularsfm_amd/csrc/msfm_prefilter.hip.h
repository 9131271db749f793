// msfm_prefilter.hip.h -- MFMA prefilter + exact re-check: the fast way to the SAME bits.
//
// The exact-order kernel (dist_top2_kernel) spends 384 un-fusable VALU ops on every one of the
// n1*n2 descriptor pairs although only the two nearest neighbours per row/column matter.  This path
// finds a small, PROVABLY sufficient candidate set with the matrix cores and evaluates the pinned
// fp32 order only on it:
//
//   sweep 1  approx_kernel<1>  S~ = |a|^2 + |b|^2 - 2 a~.b~ with fp16 operands on
//                              v_mfma_f32_32x32x16_f16 -- the norms ride in a ninth k-step (below), so the
//                              accumulator IS -S~/2; per row / column the two smallest S~ values.
//   thresholds_kernel          T = S~(2) + 2*eps, eps = rigorous bound on |S~ - S_exact| (below); rows /
//                              columns that provably cannot match get T = -inf (match lists only).
//   sweep 2  approx_kernel<3>  live rows of one image (compacted) against the other image, per direction:
//                              the row threshold rides in the ninth k-step too, a hit is accumulator >= 0;
//            approx_kernel<2>  dense variant (nothing pruned): S~ <= T_row[q] or S~ <= T_col[t];
//                              hits are appended to the pair's candidate list (a few per row).
//   exact_candidates_kernel    S_exact in the pinned accumulation order for the candidates only.
//   reduce (3 tiny kernels)    per row / column the best and second best (S_exact, index) among
//                              the candidates -> the same kNN arrays merge_knn_kernel produces.
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

#ifndef MSFM_RB
#define MSFM_RB 2
#endif
constexpr int kPfRB = MSFM_RB;            // 32-row MFMA blocks per wave (A fragments of 32 kPfRB rows stay in registers)
constexpr int kPfWaveRows = 32 * kPfRB;
constexpr int kPfWgRows = 4 * kPfWaveRows; // A rows per workgroup / work item
constexpr int kPfThreads = 256;           // 4 waves
constexpr int kPfBT = 64;                 // B rows per tile
constexpr int kHalfRowBytes = kDim * 2;   // one fp16 descriptor = 256 B = 16 granules of 16 B
constexpr int kPfLdsB = kPfBT * kHalfRowBytes;  // 16 KiB per slot
constexpr int kPfCandBuf = 256;                  // per-wave LDS candidate buffer (pass 2), int2 entries
constexpr int kPfExtB = kPfBT * 16;              // the tile's norm quadruples: 16 B per row
// B ring 48 KiB | per-wave quadruple rings 12 KiB | per-wave column-threshold rings 3 KiB | zero granule |
// per-wave candidate buffers 8 KiB | column partials 6 KiB
constexpr int kPfLdsBytes = 3 * kPfLdsB + 4 * 3 * kPfExtB + 4 * 3 * 64 * 4 + 64 + 4 * kPfCandBuf * 8 + 3 * 4 * 64 * 8;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct PfPair {            // per-pair extras of the prefilter path (parallel to PairDesc)
    const _Float16* a_h;   // fp16 swizzled blocks
    const _Float16* b_h;
    const float* a_nrm;    // |row|^2, +inf on padding rows
    const float* b_nrm;
    const _Float16* a_ext; // norm quadruples [rows][8 halfs] (the A image's are used when it plays B in a compacted sweep)
    const _Float16* b_ext;
    float a_c, b_c;        // the images' scales c (powers of two)
    float a_nrm_max, b_nrm_max;
    long long tu_off;      // row thresholds T (S-space) [n1pad]; compacted sweep: T - |a|^2 per live row
    long long tv_off;      // column thresholds T (S-space) [n2pad]
    long long cand_off;    // candidate list base
    int cand_cap;
    int use;               // 1: prefiltered; 0: not safe -> exact brute force
};

// ---------------------------------------------------------------------------------------------
// upload-time preparation: fp16 swizzled blocks, row norms, maxima
//   block layout: [64 rows][16 granules]; granule g of row r sits at position g ^ (r & 15), so a
//   linear LDS-DMA of the block lands bank-conflict-free for the MFMA operand reads
// ---------------------------------------------------------------------------------------------
__global__ void pf_prepare_kernel(const float* __restrict__ raw, _Float16* __restrict__ h, float* __restrict__ nrm,
                                  unsigned* __restrict__ maxima /* [0]=nrm_max bits, [1]=abs_max bits */,
                                  int n, int npad) {
    const long long total = (long long)npad * 16;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(e >> 4), g = (int)(e & 15);
        h8 v;
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float x = row < n ? raw[(size_t)row * kDim + g * 8 + k] : 0.f;
            v[k] = (_Float16)x;
            amax = fmaxf(amax, fabsf(x));
            if (!(fabsf(x) <= 3.0e38f)) amax = f_inf();  // NaN / inf
        }
        *reinterpret_cast<h8*>(h + ((size_t)row * 16 + (g ^ (row & 15))) * 8) = v;
        if (amax > 0.f) atomicMax(&maxima[1], __float_as_uint(amax));
    }
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < npad; row += gridDim.x * blockDim.x) {
        float s = f_inf();
        if (row < n) {
            s = 0.f;
            for (int k = 0; k < kDim; ++k) {
                const float x = raw[(size_t)row * kDim + k];
                s = fmaf(x, x, s);
            }
            atomicMax(&maxima[0], __float_as_uint(s));  // s >= 0: uint order == float order; NaN bits sort high
        }
        nrm[row] = s;
    }
}

// norm quadruples [h_hi, h_lo, c, c, 0, 0, 0, 0], h = |row|^2 / 2 / c (padding rows: +inf -> never selected)
__global__ void pf_ext_kernel(const float* __restrict__ nrm, _Float16* __restrict__ ext, int npad, float c) {
    const float inv_c = 1.f / c;  // power of two: exact
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < npad; row += gridDim.x * blockDim.x) {
        const float h = 0.5f * nrm[row] * inv_c;
        const _Float16 hi = (_Float16)h;
        const float rest = h - (float)hi;
        const _Float16 lo = (rest == rest && fabsf(rest) < 3.0e38f) ? (_Float16)rest : (_Float16)0.f;
        h8 v;
        v[0] = hi; v[1] = lo; v[2] = (_Float16)c; v[3] = (_Float16)c;
        v[4] = v[5] = v[6] = v[7] = (_Float16)0.f;
        *reinterpret_cast<h8*>(ext + (size_t)row * 8) = v;
    }
}

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

// ---------------------------------------------------------------------------------------------
// approx_kernel<PASS>: one workgroup (4 waves) = one 256-row A block x a range of 64-row B tiles.
//   Wave w owns A rows w*64..w*64+63 for the whole item: its fp16 A fragments (two 32-row MFMA
//   blocks x 9 k-steps, the ninth holding the norm / threshold quadruple) live in registers, loaded
//   once straight from HBM.  Only B tiles stream through LDS: a ring of three 16 KiB slots filled by
//   LDS-DMA two tiles ahead, synchronised with raw s_barrier + COUNTED s_waitcnt vmcnt(N) so the two
//   younger tiles stay in flight across the barrier.  The tile's norm quadruples (and the dense sweep's
//   column thresholds) ride along as per-wave DMAs, so the loop contains no ordinary global load that
//   would make hipcc drain vmcnt(0).
//   Per tile and column block: 2 x (8 MFMA 32x32x16 f16 + 1 MFMA 32x32x8 f16 for the norm quadruple), then the
//   epilogue on the 2 x 16 results.
//   MFMA layout: lane l feeds A[row l&31][k (l>>5)*8..+7] and B[col l&31][same k]; it receives for
//   column l&31 the 16 rows (r&3) + 8 (r>>2) + 4 (l>>5), r = 0..15.
// PASS 1: accumulator = -S~/2.  Row maxima (1 op / element), column maxima (v_max3: 0.5 op / element);
//         the two smallest S~ per row (merged over the 32 lanes at the end) and per column (lane pair
//         merged, the four waves folded in LDS) are written as partials.
// PASS 2: accumulator = -S~/2; append (q, t) where S~ <= T_row[q] or S~ <= T_col[t].
// PASS 3: A = compacted live rows, accumulator = -(S~ - T_row)/2; append (k, t) where it is >= 0.
//         PASS 2 / 3 first reduce the block to "any hit?" with v_max3 and only then build the bit mask.
// VMEM LOADS per wave per tile (the counted wait depends on it): PASS 1 / 3: 5 DMA, PASS 2: 6 DMA.
// Stores and the rare candidate flushes only add ops, which makes the wait more conservative.
// ---------------------------------------------------------------------------------------------
constexpr int kPfRing = 3;

// Timing experiments only (results are WRONG unless 0): 1 no epilogue VALU, 2 no MFMA, 3 B fragments read
// from LDS once per item instead of per block, 4 no barrier / DMA in the loop, 5 = 1 + 3, 6 = 1 + 3 + 4
#ifndef MSFM_ABL
#define MSFM_ABL 0
#endif
#ifndef MSFM_MERGE_LATE
#define MSFM_MERGE_LATE 0
#endif
#ifndef MSFM_DMA_LATE
#define MSFM_DMA_LATE 5
#endif
#ifndef MSFM_PIPE
#define MSFM_PIPE 0    // 1: block-level software pipeline (experiment, see the loop)
#endif
#ifndef MSFM_SCHED
#define MSFM_SCHED 0   // scheduling experiments (tools/variant_bench.sh); 0 = production
#endif

constexpr bool kAblNoEpi = MSFM_ABL == 1 || MSFM_ABL == 5 || MSFM_ABL == 6;
constexpr bool kAblNoMfma = MSFM_ABL == 2;
constexpr bool kAblNoLds = MSFM_ABL == 3 || MSFM_ABL == 5 || MSFM_ABL == 6;
constexpr bool kAblNoSync = MSFM_ABL == 4 || MSFM_ABL == 6;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }  // folds to v_max3_f32

template <int PASS>
__global__ __launch_bounds__(kPfThreads, kPfRB <= 2 ? 2 : 1) void approx_kernel(
    const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf, const WorkItem* __restrict__ items,
    float* __restrict__ rp_s0, float* __restrict__ rp_s1, float* __restrict__ cp_s0, float* __restrict__ cp_s1,
    const float* __restrict__ tu, const float* __restrict__ tv, int2* __restrict__ cand,
    unsigned long long* __restrict__ cand_count /* 64-bit: n1 * n2 hits of a flooded list do not fit 32 bits */) {
    typedef const __attribute__((address_space(1))) float* gfloat_p;  // keep these loads off the FLAT path
    typedef const __attribute__((address_space(1))) h8* gh8_p;
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    char* sB = pf_smem;
    char* sExt = pf_smem + kPfRing * kPfLdsB;                                    // [wave][slot][64 rows x 16 B]
    float* sThr = reinterpret_cast<float*>(sExt + 4 * kPfRing * kPfExtB);        // [wave][slot][64]
    char* sZero = reinterpret_cast<char*>(sThr + 4 * kPfRing * 64);              // 64 B, first 16 used
    char* sCand = sZero + 64;

    const WorkItem item = items[blockIdx.x];
    if (item.pair < 0) return;
    const PfPair pp = pf[item.pair];
    if (!pp.use) return;
    const PairDesc pd = pairs[item.pair];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lcol = lane & 31, lhalf = lane >> 5;

    // 64-row tiles; the last 128-row block of the B image may hold an all-padding second tile: skip it
    const int t_begin = item.bt_begin * 2, t_end = min(item.bt_end * 2, max(item.bt_begin * 2 + 1, (pd.n2 + kPfBT - 1) / kPfBT));
    const char* gB = reinterpret_cast<const char*>(pp.b_h);
    const char* gE = reinterpret_cast<const char*>(pp.b_ext);
    const gfloat_p g_anrm = (gfloat_p)pp.a_nrm;
    const gfloat_p g_tu = (gfloat_p)tu;
    char* ext_w = sExt + wave * (kPfRing * kPfExtB);   // this wave's private copies
    float* thr_w = sThr + wave * (kPfRing * 64);

    // DMA group of tile tt (clamped: the tail re-fetches the last tile so every iteration issues the
    // same number of VMEM ops): this wave's quarter of the 16 KiB tile + its private quadruples / thresholds
    auto dma_tile = [&](int tt) {
        const int tc = tt < t_end ? tt : t_end - 1;
        const int sl = (tt - t_begin) % kPfRing;
        const char* g = gB + (size_t)tc * kPfLdsB + wave * 4096 + lane * 16;
        char* l = sB + sl * kPfLdsB + wave * 4096;
#pragma unroll
        for (int k = 0; k < (MSFM_ABL == 9 ? 2 : 4); ++k)  // ablation 9: half of the tile DMA (timing experiment)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + k * 1024),
                                             (__attribute__((address_space(3))) void*)(l + k * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gE + (size_t)tc * kPfExtB + lane * 16),
                                         (__attribute__((address_space(3))) void*)(ext_w + sl * kPfExtB), 16, 0, 0);
        if (PASS == 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tv + pp.tv_off + tc * kPfBT + lane),
                                             (__attribute__((address_space(3))) void*)(thr_w + sl * 64), 4, 0, 0);
    };
    constexpr int kDmaOps = ((PASS == 2) ? 6 : 5) - (MSFM_ABL == 9 ? 2 : 0);
    // where tile t+2's DMA pieces are issued in iteration t: 0 right after the barrier, 1 at the end of the iteration,
    // 2 / 3 between / before the epilogues (MSFM_DMA_LATE: 4 = end of iteration in sweep 1 only; 5 = the default:
    // before the epilogues in sweep 1 (-2.4 .. -4.7 %: among the epilogue's VALU work a 1-KiB piece costs the wave
    // less than among the ds_reads and MFMAs right after the barrier), right after the barrier in sweep 2)
    constexpr int kDmaLate = (MSFM_DMA_LATE == 4) ? (PASS == 1 ? 1 : 0) : (MSFM_DMA_LATE == 5) ? (PASS == 1 ? 3 : 0) : MSFM_DMA_LATE;

    dma_tile(t_begin);
    dma_tile(t_begin + 1);
    if (tid < 4) reinterpret_cast<float*>(sZero)[tid] = 0.f;

    // A fragments: rows a_blk*256 + wave*64 + rb*32 + lcol, granule 2*ks + lhalf (stored at ^ (row & 15));
    // ninth k-step: [-c, -c, x_hi, x_lo, 0...] in the lhalf == 0 lanes (k = 128..135), zeros in the others
    h8 af[kPfRB][9];
    const float inv_c = 1.f / pp.b_c;
#pragma unroll
    for (int rb = 0; rb < kPfRB; ++rb) {
        const int frow = item.a_blk * kPfWgRows + wave * kPfWaveRows + rb * 32 + lcol;
        const gh8_p ga = (gh8_p)(pp.a_h) + (size_t)frow * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) af[rb][ks] = ga[(2 * ks + lhalf) ^ (frow & 15)];
        // padding rows: X = -inf -> accumulator -inf, never a maximum, never a hit
        float X;
        if (PASS == 3) X = frow < pd.n1 ? 0.5f * g_tu[pp.tu_off + frow] : -f_inf();
        else X = frow < pd.n1 ? -0.5f * g_anrm[frow] : -f_inf();
        const float xs = X * inv_c;
        const _Float16 hi = (_Float16)xs;
        const float rest = xs - (float)hi;
        const _Float16 lo = (rest == rest && fabsf(rest) < 3.0e38f) ? (_Float16)rest : (_Float16)0.f;
        h8 e;
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (_Float16)0.f;
        if (lhalf == 0) {
            e[0] = (_Float16)(-pp.b_c);
            e[1] = (_Float16)(-pp.b_c);
            e[2] = hi;
            e[3] = lo;
        }
        af[rb][8] = e;
    }

    // this lane's 32 result rows: (rb, r) -> row = a_blk*256 + wave*64 + rb*32 + (r&3) + 8*(r>>2) + 4*lhalf
    const int arow_base = item.a_blk * kPfWgRows + wave * kPfWaveRows + 4 * lhalf;
    // rs0: PASS 1 running row maximum of the accumulator (-S~min/2); PASS 2 the row's hit level -T_row/2
    float rs0[kPfRB][16];
#pragma unroll
    for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = arow_base + rb * 32 + (r & 3) + 8 * (r >> 2);
            rs0[rb][r] = (PASS == 2) ? (rr < pd.n1 ? -0.5f * g_tu[pp.tu_off + rr] : f_inf()) : -f_inf();
        }
    // Make hipcc itself wait for the fragment loads here (a register use it can see): otherwise its
    // scoreboard still holds them as pending at the first MFMA and it drains vmcnt(0) INSIDE the loop,
    // which would serialise the DMA ring.
#pragma unroll
    for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) asm volatile("" ::"v"(af[rb][ks]));
    wait_vmcnt<0>();  // prologue loads (and the first two DMA groups) are done: counted waits start clean

    // sweep 2: wave-private candidate buffer in LDS
    int2* cbuf = reinterpret_cast<int2*>(sCand) + wave * kPfCandBuf;
    const unsigned cbuf_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)cbuf;
    // sweep 1: column partials of the four waves meet in LDS ([ring slot][wave][64 columns] x (s0, s1)) and
    // are merged by one wave two tiles later: 4x less partial traffic to HBM than one slot per wave
    const unsigned colbuf_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)(sCand + 4 * kPfCandBuf * 8);
    int n_buf = 0;  // wave-uniform
    auto flush_candidates = [&]() {
        if (n_buf == 0) return;
        unsigned long long base64 = 0;
        if (lane == 0) base64 = atomicAdd(&cand_count[item.pair], (unsigned long long)n_buf);
        // beyond the capacity nothing is stored: the clamped base keeps the test below false for every k
        const int base = __builtin_amdgcn_readfirstlane((int)(base64 < (unsigned long long)pp.cand_cap ? base64 : (unsigned long long)pp.cand_cap));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the asm ds_writes below are not tracked by hipcc
        for (int k = lane; k < n_buf; k += 64)
            if (base + k < pp.cand_cap) cand[pp.cand_off + base + k] = cbuf[k];
        n_buf = 0;
    };

    const int xb = lcol & 15;  // both column blocks: row & 15 == lcol & 15
    const bool wave_active = item.a_blk * kPfWgRows + wave * kPfWaveRows < pd.n1;  // wave-uniform
    if (PASS == 1 && !wave_active) {
        // its column partials are never written: park (+inf, +inf) in all three ring slots once
        const float2 pr = make_float2(f_inf(), f_inf());
#pragma unroll
        for (int sl = 0; sl < kPfRing; ++sl)
            asm volatile("ds_write_b64 %0, %1" ::"v"(colbuf_lds + (unsigned)((sl * 4 + wave) * 64 + lane) * 8u), "v"(pr) : "memory");
    }

    // ---- the two halves of the software pipeline -------------------------------------------------
    // A "block" is (tile, column block): per wave 2 x 16 MFMA results per lane.
    // stage(): issue the 18 MFMAs of block k+1 into `nxt` while the VALU epilogue of block k (in `cur`)
    // runs -- in ONE basic block, so the scheduler can interleave them: co-resident waves run in
    // lockstep (same barriers), only the overlap inside a wave keeps both pipes busy.
    struct BlockMeta { float hc; int col; int cslot; };  // hc: PASS 2 column hit level -T_col/2; cslot: LDS slot of the column partials
    const int zero_off = (int)(sZero - pf_smem);
    int abl_t = t_begin;
    auto load_bf = [&](const char* pb, int pe_off, int cb, h8 (&bf)[9]) {
        if (kAblNoLds && abl_t != t_begin) return;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            bf[ks] = *reinterpret_cast<const h8*>(pb + cb * 32 * kHalfRowBytes + (((2 * ks + lhalf) ^ xb) << 4));
        // ninth k-step: the quadruple of column cb*32 + lcol for k = 128..135, the zero granule for k = 136..143
        // (one base pointer + selected offset: a select between two pointers makes hipcc drain vmcnt(0))
        bf[8] = *reinterpret_cast<const h8*>(pf_smem + (lhalf == 0 ? pe_off + (cb * 32 + lcol) * 16 : zero_off));
    };
    auto mfma_block = [&](const h8 (&bf)[9], f16v (&acc)[kPfRB]) {
        if (kAblNoMfma) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int rb = 0; rb < kPfRB; ++rb) acc[rb][r] += (float)bf[(r + rb) & 7][0] + (float)bf[8][r & 7];
            return;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb) acc[rb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[rb][ks], bf[ks], acc[rb], 0, 0, 0);
        {   // the quadruple only fills k = 0..3: the K = 8 instruction (lane l: k = 4 (l >> 5) .. +3) takes half the passes
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 b4 = __builtin_shufflevector(bf[8], bf[8], 0, 1, 2, 3);
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(af[rb][8], af[rb][8], 0, 1, 2, 3), b4, acc[rb], 0, 0, 0);
        }
    };
    // branch-free part of the epilogue; returns "this lane saw a hit" for the sweep-2 variants
    auto epilogue_valu = [&](const f16v (&acc)[kPfRB], const BlockMeta& bm, bool do_rows = true) -> bool {
        if (kAblNoEpi) {
            rs0[0][0] = fmaxf(rs0[0][0], acc[0][0] + acc[kPfRB - 1][15]);
            return false;
        }
        // column maximum of the accumulator over this lane's 32 rows: 16 v_max3
        float m = -f_inf();
        if (PASS != 2) {
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) m = max3f(m, acc[rb][r], acc[rb][r + 1]);
        }
        if (PASS == 1) {
            // Only MAXIMA are tracked: the second smallest of the minima of S~ over disjoint subsets is an
            // upper bound of the true second-smallest S~, which is all the threshold needs (it is exact
            // unless both neighbours fall into one subset).
            if (do_rows) {
#pragma unroll
                for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rs0[rb][r] = fmaxf(rs0[rb][r], acc[rb][r]);
            }
            // partner lane (l ^ 32): the other 32 rows of the wave
            const float other = __shfl_xor(m, 32);
            // (s0, s1) of this wave's 64 rows for column bm.col -> LDS (inline asm: see append_hits)
            if (lhalf == 0) {
                const float2 pr = make_float2(-2.f * fmaxf(m, other), -2.f * fminf(m, other));
                asm volatile("ds_write_b64 %0, %1" ::"v"(colbuf_lds + (unsigned)(bm.cslot + lcol) * 8u), "v"(pr) : "memory");
            }
            return false;
        } else if (PASS == 3) {
            return m >= 0.f;
        } else {
            // row criterion: max over (acc - level_row) >= 0; column criterion: max over acc >= level_col
            float mr = -f_inf(), mc = -f_inf();
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    mr = max3f(mr, acc[rb][r] - rs0[rb][r], acc[rb][r + 1] - rs0[rb][r + 1]);
                    mc = max3f(mc, acc[rb][r], acc[rb][r + 1]);
                }
            return mr >= 0.f || mc >= bm.hc;
        }
    };
    // sweep 2, rare path: the block holds at least one hit -> bit mask per lane (element k = rb*16 + r at
    // bit 31-k), slotted with ballot/popcount into this wave's LDS buffer -- no atomics in the loop --
    // and flushed to the pair's global list when the buffer fills up
    auto append_hits = [&](bool any, const f16v (&acc)[kPfRB], const BlockMeta& bm) {
        if (__ballot(any) == 0ull) return;
        unsigned long long mask = 0;   // element k = rb*16 + r at bit (16 kPfRB - 1 - k)
        constexpr int kEl = 16 * kPfRB;
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // padding rows / columns hold -inf: with an infinite threshold (fewer than two real elements in
                // a subset) the hit level is -inf as well, and -inf >= -inf must not count
                const bool hit = (PASS == 3) ? (acc[rb][r] >= 0.f)
                                             : (acc[rb][r] > -f_inf() && (acc[rb][r] >= rs0[rb][r] || acc[rb][r] >= bm.hc));
                mask = mask + mask + (hit ? 1ull : 0ull);
            }
        while (__ballot(mask != 0ull) != 0ull) {
            const bool hit = mask != 0ull;
            const int k = __clzll((long long)mask) - (64 - kEl);  // first remaining element of this lane
            const unsigned long long mm = __ballot(hit);
            if (n_buf + 64 > kPfCandBuf) flush_candidates();
            if (hit) {
                mask &= ~(1ull << (kEl - 1 - k));
                const int slt = n_buf + __popcll(mm & ((1ull << lane) - 1ull));
                // inline asm on purpose: hipcc would first drain vmcnt(0) for a compiler-visible LDS store
                const int2 e = make_int2(arow_base + (k >> 4) * 32 + (k & 3) + 8 * ((k & 15) >> 2), bm.col);
                asm volatile("ds_write_b64 %0, %1" ::"v"(cbuf_lds + slt * 8), "v"(e) : "memory");
            }
            n_buf += __popcll(mm);
        }
    };
    // sweep 1: lane = column of tile tt; fold the four waves' (s0, s1) and store one partial per A block
    auto merge_columns = [&](int tt) {
        const unsigned base = colbuf_lds + (unsigned)((((tt - t_begin) % kPfRing) * 4) * 64 + lane) * 8u;
        float2 w0, w1, w2, w3;
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\t"
                     "ds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(base) : "memory");
        v2_merge(w0.x, w0.y, w1.x, w1.y);
        v2_merge(w2.x, w2.y, w3.x, w3.y);
        v2_merge(w0.x, w0.y, w2.x, w2.y);
        const long long o = pd.cp_off + (long long)item.a_blk * pd.n2pad + tt * kPfBT + lane;
        cp_s0[o] = w0.x;
        cp_s1[o] = w0.y;
    };

    if constexpr (MSFM_PIPE == 2) {
        // EXPERIMENT (-DMSFM_PIPE=2): epilogue of block k-1 placed in the MFMA gaps of block k (one fragment buffer,
        // two accumulator sets alternating statically; sched_barrier between the halves keeps hipcc from sinking the
        // row updates across them, which is what cost 32 register copies per tile in the first version of this loop)
        f16v accA[kPfRB], accB[kPfRB];
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[rb][r] = -f_inf();
        BlockMeta metaA = {0.f, 0, 0}, metaB = {f_inf(), t_begin * kPfBT + 32 + lcol, (2 * 4 + wave) * 64 + 32};
        auto hint = [&]() {
#pragma unroll
            for (int k = 0; k < 18; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
        };
#pragma unroll 1
        for (int t = t_begin; t < t_end; ++t) {
            if (t - t_begin >= 2) wait_vmcnt<kDmaOps>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            dma_tile(t + 2);
            if (PASS == 1 && t - t_begin >= 2 && wave == (t & 3)) merge_columns(t - 2);
            const int sl = (t - t_begin) % kPfRing;
            const char* pb = sB + sl * kPfLdsB + lcol * kHalfRowBytes;
            const int pe_off = (int)(ext_w - pf_smem) + sl * kPfExtB;
            const float* thr = thr_w + sl * 64;
            abl_t = t;
            if (wave_active) {
                h8 bf[9];
                load_bf(pb, pe_off, 0, bf);
                metaA.hc = (PASS == 2) ? -0.5f * thr[lcol] : 0.f;
                metaA.col = t * kPfBT + lcol;
                metaA.cslot = (sl * 4 + wave) * 64;
                mfma_block(bf, accA);
                const bool anyB = epilogue_valu(accB, metaB, true);
                hint();
                if (PASS >= 2) append_hits(anyB, accB, metaB);
                __builtin_amdgcn_sched_barrier(0);
                load_bf(pb, pe_off, 1, bf);
                metaB.hc = (PASS == 2) ? -0.5f * thr[32 + lcol] : 0.f;
                metaB.col = t * kPfBT + 32 + lcol;
                metaB.cslot = (sl * 4 + wave) * 64 + 32;
                mfma_block(bf, accB);
                const bool anyA = epilogue_valu(accA, metaA, true);
                hint();
                if (PASS >= 2) append_hits(anyA, accA, metaA);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wave_active) {
            const bool any = epilogue_valu(accB, metaB, true);
            if (PASS >= 2) append_hits(any, accB, metaB);
        }
    } else if constexpr (MSFM_PIPE == 1) {
    // EXPERIMENT (-DMSFM_PIPE=1, measured and not adopted): software pipeline at block granularity.  At two waves
    // per SIMD the second fragment buffer spills in sweep 1 (+35 %); in the compacted sweep it gains 1.6 %; at one
    // wave per SIMD (MSFM_RB=4) the accumulators land in AGPRs and the epilogue pays v_accvgpr_read copies (+35 %).  While block k multiplies, the B fragments
    // of block k+1 are read from LDS (second fragment buffer) and the epilogue of block k-1 runs on the other
    // accumulator set; one barrier per tile at the start of its second half, DMA one tile ahead (vmcnt(0) there:
    // the loads are a full tile old).
    f16v accA[kPfRB], accB[kPfRB];
#pragma unroll
    for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[rb][r] = -f_inf();
    BlockMeta metaA = {0.f, 0, 0}, metaB = {f_inf(), t_begin * kPfBT + 32 + lcol, (2 * 4 + wave) * 64 + 32};
    h8 bfA[9], bfB[9];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile t_begin is in LDS (every wave waited for its own DMA part above)
    asm volatile("" ::: "memory");
    if (wave_active) load_bf(sB + lcol * kHalfRowBytes, (int)(ext_w - pf_smem), 0, bfA);
#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int sl = (t - t_begin) % kPfRing;
        const char* pb = sB + sl * kPfLdsB + lcol * kHalfRowBytes;
        const int pe_off = (int)(ext_w - pf_smem) + sl * kPfExtB;
        const float* thr = thr_w + sl * 64;
        abl_t = t;
        // ---- first half: block (t, 0) multiplies; fragments of (t, 1) arrive; epilogue of (t-1, 1) ----
        if (wave_active) {
            load_bf(pb, pe_off, 1, bfB);
            metaA.hc = (PASS == 2) ? -0.5f * thr[lcol] : 0.f;
            metaA.col = t * kPfBT + lcol;
            metaA.cslot = (sl * 4 + wave) * 64;
            mfma_block(bfA, accA);
            const bool anyB = epilogue_valu(accB, metaB, true);
            if (PASS >= 2) append_hits(anyB, accB, metaB);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- tile t+1 must have landed; tile t-1's slot is free for tile t+2 ----
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma_tile(t + 2);
        if (PASS == 1 && t - t_begin >= 2 && wave == (t & 3)) merge_columns(t - 2);
        // ---- second half: block (t, 1) multiplies; fragments of (t+1, 0) arrive; epilogue of (t, 0) ----
        if (wave_active) {
            const int tn = t + 1 < t_end ? t + 1 : t;   // the last prefetch re-reads the current tile (unused)
            const int sn = (tn - t_begin) % kPfRing;
            load_bf(sB + sn * kPfLdsB + lcol * kHalfRowBytes, (int)(ext_w - pf_smem) + sn * kPfExtB, 0, bfA);
            metaB.hc = (PASS == 2) ? -0.5f * thr[32 + lcol] : 0.f;
            metaB.col = t * kPfBT + 32 + lcol;
            metaB.cslot = (sl * 4 + wave) * 64 + 32;
            mfma_block(bfB, accB);
            const bool anyA = epilogue_valu(accA, metaA, true);
            if (PASS >= 2) append_hits(anyA, accA, metaA);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wave_active) {   // drain: epilogue of the last block
        const bool any = epilogue_valu(accB, metaB, true);
        if (PASS >= 2) append_hits(any, accB, metaB);
    }
    } else {
    // Both column blocks of a tile are multiplied first (four independent accumulator chains), then their
    // epilogues run together: rows take ONE v_max3 per pair of elements (running maximum, block cb 0, block
    // cb 1), and no accumulator lives across the loop back-edge -- carrying the second block's accumulator
    // into the next iteration (to overlap its epilogue with the next MFMAs) cost 32 register copies per tile
    // and bought nothing: MFMA and VALU issue of a SIMD do not overlap on this part (tools/ubench_mfma_valu).
    f16v accA[kPfRB], accB[kPfRB];
    BlockMeta metaA = {0.f, 0, 0}, metaB = {0.f, 0, 0};
#if MSFM_ABL
    h8 bf[9];  // must survive the iteration when the reads are ablated
#endif
#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        // Tile t must have landed.  Its DMA group is followed by exactly one younger group of LOADS
        // (tile t+1, kDmaOps of them).  Loads retire in order among themselves, but on gfx9-class
        // vmcnt stores may retire out of order with respect to loads, so the count must not rely on
        // the (sweep-1) stores: "at most kDmaOps outstanding" implies every load of tile t is done,
        // because a pending load of tile t would keep all kDmaOps loads of tile t+1 pending as well.
        if (!kAblNoSync) {
            if (MSFM_ABL != 8 && t - t_begin >= 2) wait_vmcnt<kDmaOps>();  // ablation 8: do not wait for the DMA (timing experiment)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MSFM_ABL != 7 || ((t - t_begin) & 1) == 0)  // ablation 7: a barrier every other tile only (timing experiment)
            __builtin_amdgcn_s_barrier();  // every wave's part of tile t is in LDS; slot of tile t-1 is free
            asm volatile("" ::: "memory");
            if (!kDmaLate) dma_tile(t + 2);
        }
        // tile t-2's column partials are complete in LDS (its epilogues ran before the previous barrier)
        if (!MSFM_MERGE_LATE && PASS == 1 && t - t_begin >= 2 && wave == (t & 3)) merge_columns(t - 2);
        const int sl = (t - t_begin) % kPfRing;
        const char* pb = sB + sl * kPfLdsB + lcol * kHalfRowBytes;
        const int pe_off = (int)(ext_w - pf_smem) + sl * kPfExtB;
        const float* thr = thr_w + sl * 64;
        abl_t = t;
        // A wave whose 64 rows are all padding (tail of an image, tail of a compacted row set) still takes
        // part in the DMA and the barriers, but leaves the matrix pipe to the co-resident workgroup
        if (wave_active) {
#if !MSFM_ABL
            h8 bf[9];
#endif
#if MSFM_SCHED == 4
            __builtin_amdgcn_iglp_opt(0);
#elif MSFM_SCHED == 5
            __builtin_amdgcn_iglp_opt(1);
#endif
            load_bf(pb, pe_off, 0, bf);
            metaA.hc = (PASS == 2) ? -0.5f * thr[lcol] : 0.f;
            metaA.col = t * kPfBT + lcol;
            metaA.cslot = (sl * 4 + wave) * 64;
#if MSFM_SCHED == 3
            __builtin_amdgcn_s_setprio(2);
#endif
            mfma_block(bf, accA);
#if MSFM_SCHED == 1
            __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 18, 0);
#endif
#if MSFM_SCHED == 2
            h8 bf2[9];
            load_bf(pb, pe_off, 1, bf2);
            mfma_block(bf2, accB);
            __builtin_amdgcn_sched_group_barrier(0x100, 18, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 36, 0);
#else
            load_bf(pb, pe_off, 1, bf);
            mfma_block(bf, accB);
#endif
#if MSFM_SCHED == 1
            __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 18, 0);
#endif
#if MSFM_SCHED == 3
            __builtin_amdgcn_s_setprio(0);
#endif
            metaB.hc = (PASS == 2) ? -0.5f * thr[32 + lcol] : 0.f;
            metaB.col = t * kPfBT + 32 + lcol;
            metaB.cslot = (sl * 4 + wave) * 64 + 32;
            if (kDmaLate == 3 && !kAblNoSync) dma_tile(t + 2);
            const bool anyA = epilogue_valu(accA, metaA, false);
            if (kDmaLate == 2 && !kAblNoSync) dma_tile(t + 2);
            const bool anyB = epilogue_valu(accB, metaB, false);
            if (PASS == 1 && !kAblNoEpi) {
#pragma unroll
                for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rs0[rb][r] = max3f(rs0[rb][r], accA[rb][r], accB[rb][r]);
            }
            if (PASS >= 2) {
                append_hits(anyA, accA, metaA);
                append_hits(anyB, accB, metaB);
            }
        }
        // the slot of tile t-1 has been free since this iteration's barrier; issuing the pieces here, after the
        // epilogue's VALU work, is cheaper than among the ds_reads and MFMAs right after the barrier
        if ((kDmaLate == 1 || (kDmaLate >= 2 && !wave_active)) && !kAblNoSync) dma_tile(t + 2);
        // (the merge of tile t-2's column partials may run anywhere in iteration t; at the end it keeps the head of
        // the iteration -- the first cycles after the barrier release -- free of VALU work)
        if (MSFM_MERGE_LATE && PASS == 1 && t - t_begin >= 2 && wave == (t & 3)) merge_columns(t - 2);
    }
    }
    if (PASS >= 2) flush_candidates();
    if (PASS == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's last column partials are in LDS
        asm volatile("" ::: "memory");
        if (wave == 0 && t_end - 2 >= t_begin) merge_columns(t_end - 2);  // (a range may consist of a single tile)
        if (wave == 1) merge_columns(t_end - 1);
    }

    if (PASS == 1) {
        // rows: S~ = -2 * accumulator; the two smallest of the 32 lanes' minima; one partial slot per B range
        float rs1[kPfRB][16];
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                rs0[rb][r] = -2.f * rs0[rb][r];
                rs1[rb][r] = f_inf();
#pragma unroll
                for (int m = 1; m < 32; m <<= 1)
                    v2_merge(rs0[rb][r], rs1[rb][r], __shfl_xor(rs0[rb][r], m), __shfl_xor(rs1[rb][r], m));
            }
        if (lcol == 0) {
            const long long o = pd.rp_off + (long long)item.range * pd.n1pad + arow_base;
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = rb * 32 + (r & 3) + 8 * (r >> 2);
                    rp_s0[o + off] = rs0[rb][r];
                    rp_s1[o + off] = rs1[rb][r];
                }
        }
    }
}

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

__global__ void pf_thresholds_kernel(const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf,
                                     const float* __restrict__ rp_s0, const float* __restrict__ rp_s1,
                                     const float* __restrict__ cp_s0, const float* __restrict__ cp_s1,
                                     float* __restrict__ tu, float* __restrict__ tv, PruneParams pr) {
    const PairDesc pd = pairs[blockIdx.y];
    const PfPair pp = pf[blockIdx.y];
    if (!pd.valid || !pp.use) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const float eps_norm = 4.8828125e-4f * fmaxf(pp.a_c, pp.b_c);  // 2^-11 c: fp16 subnormal flush of the norm quadruples
    if (e < pd.n1pad) {
        float s0 = f_inf(), s1 = f_inf();
        for (int p = 0; p < pd.ranges; ++p) {
            const long long o = pd.rp_off + (long long)p * pd.n1pad + e;
            v2_merge(s0, s1, rp_s0[o], rp_s1[o]);
        }
        // s1 = S~(2); eps_row = rel*(na + nb_max) + abs*(sqrt(na)+sqrt(nb_max)) + eps_norm; threshold in S-space
        const float na = pp.a_nrm[e];
        const float eps = kEpsRel * (na + pp.b_nrm_max) + kEpsAbs * (sqrtf(na) + sqrtf(pp.b_nrm_max)) + eps_norm;
        const float slack = 2.f * eps + 1e-5f * (fabsf(s1) + na + pp.b_nrm_max);  // + roundings here and in sweep 2's test
        const bool live = (e < pd.n1) && !pf_dead(s0, s1, na, eps, pp.b_nrm_max, pr);
        tu[pp.tu_off + e] = live ? s1 + slack : -f_inf();
    }
    if (e < pd.n2pad) {
        float s0 = f_inf(), s1 = f_inf();
        for (int p = 0; p < pd.a_blocks256; ++p) {
            const long long o = pd.cp_off + (long long)p * pd.n2pad + e;
            v2_merge(s0, s1, cp_s0[o], cp_s1[o]);
        }
        const float nb = pp.b_nrm[e];
        const float eps = kEpsRel * (nb + pp.a_nrm_max) + kEpsAbs * (sqrtf(nb) + sqrtf(pp.a_nrm_max)) + eps_norm;
        const float slack = 2.f * eps + 1e-5f * (fabsf(s1) + nb + pp.a_nrm_max);
        const bool live = (e < pd.n2) && !pf_dead(s0, s1, nb, eps, pp.a_nrm_max, pr);
        tv[pp.tv_off + e] = live ? s1 + slack : -f_inf();
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
};

// exact pinned-order S for every candidate: 16 lanes per candidate (SSE order: lane L owns the
// lane partial k = L mod 16; AVX2 order: 32 partials -> 2 per lane), coalesced 64-B row reads.
// Records are rewritten in place as real (q, t).   grid = (x, n_lists)
template <int ORDER>
__global__ void pf_exact_candidates_kernel(const PairDesc* __restrict__ pairs, const CandList* __restrict__ lists,
                                           const int* __restrict__ active /* ids of the non-empty lists */,
                                           const unsigned long long* __restrict__ cand_count, int2* __restrict__ cand,
                                           float* __restrict__ cand_s, int* __restrict__ cand_pair) {
    const int lid = active[blockIdx.y];
    const CandList L = lists[lid];
    if (L.cap == 0) return;
    const int n = (int)(cand_count[lid] < (unsigned long long)L.cap ? cand_count[lid] : (unsigned long long)L.cap);
    const int sub = threadIdx.x & 15;
    for (int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; c < ((n + 3) & ~3); c += (gridDim.x * blockDim.x) >> 4) {
        const bool live = c < n;
        int2 qt = live ? cand[L.off + c] : make_int2(0, 0);
        int pair = L.mode == 0 ? L.pair : L.row_pair[live ? qt.x : 0];
        if (!live && L.mode != 0) pair = L.row_pair[0], qt = make_int2(0, 0);
        if (live && L.mode == 1) qt.x = L.live_idx[qt.x];
        if (live && L.mode == 2) qt = make_int2(qt.y, L.live_idx[qt.x]);
        const float* a = pairs[pair].a_raw + (size_t)qt.x * kDim;
        const float* b = pairs[pair].b_raw + (size_t)qt.y * kDim;
        float res;
        if (ORDER == 0) {
            float p = 0.f;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const float t = a[16 * it + sub] - b[16 * it + sub];
                p = (it == 0) ? t * t : p + t * t;
            }
            // s[l] = ((p[l] + p[4+l]) + p[8+l]) + p[12+l]; result = (s0+s2)+(s1+s3)
            const float p4 = __shfl(p, (threadIdx.x & ~15) | ((sub & 3) + 4), 64);
            const float p8 = __shfl(p, (threadIdx.x & ~15) | ((sub & 3) + 8), 64);
            const float p12 = __shfl(p, (threadIdx.x & ~15) | ((sub & 3) + 12), 64);
            const float p0 = __shfl(p, (threadIdx.x & ~15) | (sub & 3), 64);
            const float s = ((p0 + p4) + p8) + p12;  // valid in every lane for l = sub & 3
            const float s0 = __shfl(s, (threadIdx.x & ~15) | 0, 64), s1 = __shfl(s, (threadIdx.x & ~15) | 1, 64);
            const float s2 = __shfl(s, (threadIdx.x & ~15) | 2, 64), s3 = __shfl(s, (threadIdx.x & ~15) | 3, 64);
            res = (s0 + s2) + (s1 + s3);
        } else {
            // 32 partials: lane `sub` owns L = sub and L = sub + 16
            float pa = 0.f, pb = 0.f;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float ta = a[32 * it + sub] - b[32 * it + sub];
                const float tb = a[32 * it + sub + 16] - b[32 * it + sub + 16];
                pa = (it == 0) ? ta * ta : __builtin_fmaf(ta, ta, pa);
                pb = (it == 0) ? tb * tb : __builtin_fmaf(tb, tb, pb);
            }
            // s[l] = ((p[l]+p[8+l])+p[16+l])+p[24+l], l = 0..7: p[l]=pa(l), p[8+l]=pa(8+l), p[16+l]=pb(l), p[24+l]=pb(8+l)
            const int l = sub & 7, base = threadIdx.x & ~15;
            const float q0 = __shfl(pa, base | l, 64), q1 = __shfl(pa, base | (8 + l), 64);
            const float q2 = __shfl(pb, base | l, 64), q3 = __shfl(pb, base | (8 + l), 64);
            const float s = ((q0 + q1) + q2) + q3;
            float sv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) sv[k] = __shfl(s, base | k, 64);
            res = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
        }
        if (live && sub == 0) {
            cand[L.off + c] = qt;
            cand_s[L.off + c] = res;
            cand_pair[L.off + c] = pair;
        }
    }
}

__device__ __forceinline__ unsigned long long pf_key(float s, int idx) {
    return ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)idx;  // s >= 0: uint order == float order
}

// reduce phase A: best (S, idx) per row and per column among the candidates (64-bit atomicMin).
// A mode-1 list only serves the row direction, a mode-2 list only the column direction: the live
// rows of the OTHER direction get their complete candidate sets from their own list.
__global__ void pf_reduce_best_kernel(const PairDesc* __restrict__ pairs, const CandList* __restrict__ lists,
                                      const int* __restrict__ active, const unsigned long long* __restrict__ cand_count,
                                      const int2* __restrict__ cand, const float* __restrict__ cand_s,
                                      const int* __restrict__ cand_pair, unsigned long long* __restrict__ best) {
    const int lid = active[blockIdx.y];
    const CandList L = lists[lid];
    if (L.cap == 0) return;
    const int n = (int)(cand_count[lid] < (unsigned long long)L.cap ? cand_count[lid] : (unsigned long long)L.cap);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const int2 qt = cand[L.off + c];
        const float s = cand_s[L.off + c];
        if (!(s < f_inf())) continue;  // batchDistance never inserts a distance >= FLT_MAX
        const int pair = cand_pair[L.off + c];
        if (L.mode != 2) atomicMin(&best[pairs[pair].kf_off + qt.x], pf_key(s, qt.y));
        if (L.mode != 1) atomicMin(&best[pairs[pair].kr_off + qt.y], pf_key(s, qt.x));
    }
}
// reduce phase B: second best = min over the candidates that are not the best one
__global__ void pf_reduce_second_kernel(const PairDesc* __restrict__ pairs, const CandList* __restrict__ lists,
                                        const int* __restrict__ active, const unsigned long long* __restrict__ cand_count,
                                        const int2* __restrict__ cand, const float* __restrict__ cand_s,
                                        const int* __restrict__ cand_pair, const unsigned long long* __restrict__ best,
                                        unsigned long long* __restrict__ second) {
    const int lid = active[blockIdx.y];
    const CandList L = lists[lid];
    if (L.cap == 0) return;
    const int n = (int)(cand_count[lid] < (unsigned long long)L.cap ? cand_count[lid] : (unsigned long long)L.cap);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const int2 qt = cand[L.off + c];
        const float s = cand_s[L.off + c];
        if (!(s < f_inf())) continue;
        const long long kfo = pairs[cand_pair[L.off + c]].kf_off, kro = pairs[cand_pair[L.off + c]].kr_off;
        const unsigned long long kf = pf_key(s, qt.y), kr = pf_key(s, qt.x);
        if (L.mode != 2 && kf != best[kfo + qt.x]) atomicMin(&second[kfo + qt.x], kf);
        if (L.mode != 1 && kr != best[kro + qt.y]) atomicMin(&second[kro + qt.y], kr);
    }
}

// live rows (threshold > -inf) per (pair, direction); grid = 2 * n_pairs blocks of 256
__global__ void pf_count_live_kernel(const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf,
                                     const float* __restrict__ tuv, int* __restrict__ live_cnt) {
    const int p = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const PairDesc pd = pairs[p];
    const PfPair pp = pf[p];
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    if (pd.valid && pp.use) {
        const int n = dir ? pd.n2 : pd.n1;
        const long long off = dir ? pp.tv_off : pp.tu_off;
        int c = 0;
        for (int e = threadIdx.x; e < n; e += blockDim.x) c += (tuv[off + e] != -f_inf()) ? 1 : 0;
        for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
        if ((threadIdx.x & 63) == 0) atomicAdd(&total, c);
    }
    __syncthreads();
    if (threadIdx.x == 0) live_cnt[blockIdx.x] = total;
}

// compaction of the live rows of one (pair, direction): indices, thresholds and the fp16 rows
// (re-swizzled for their new row number).   grid = n_jobs blocks of 256
struct GatherJob {
    const _Float16* src_h;   // image's fp16 blocks
    const float* src_nrm;    // its |row|^2
    long long src_thr_off;   // into tuv
    long long dst_row;       // first row of this job in the compact arrays (its group starts at a multiple of 256)
    long long zero_upto;     // rows [dst_row + live, zero_upto) are zero-filled (the group's tail; else == dst_row + live)
    int n;                   // rows of the image
    int pair;                // batch index of the pair the rows belong to
};
__global__ void pf_gather_live_kernel(const GatherJob* __restrict__ jobs, const float* __restrict__ tuv,
                                      int* __restrict__ live_idx, int* __restrict__ row_pair, float* __restrict__ cmp_tu,
                                      _Float16* __restrict__ cmp_h) {
    const GatherJob J = jobs[blockIdx.x];
    __shared__ int wsum[4];
    __shared__ int running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e0 = 0; e0 < J.n; e0 += 256) {
        const int e = e0 + threadIdx.x;
        const float t = e < J.n ? tuv[J.src_thr_off + e] : -f_inf();
        const bool live = t != -f_inf();
        const unsigned long long bal = __ballot(live);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int k = running + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) k += wsum[w];
        if (live) {
            live_idx[J.dst_row + k] = e;
            row_pair[J.dst_row + k] = J.pair;
            cmp_tu[J.dst_row + k] = t - J.src_nrm[e];  // sweep 2 folds (T - |a|^2)/2 into the MFMA
        }
        __syncthreads();
        if (threadIdx.x == 0) running += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    const int cnt = running;
    // fp16 rows: 16 threads per row, one 16-byte granule each; granule g of row r lives at g ^ (r & 15)
    // (r = the row number inside its 256-aligned group: dst_row + k has the same low bits)
    for (int k = threadIdx.x >> 4; k < cnt; k += 16) {
        const int r = live_idx[J.dst_row + k];
        const int g = threadIdx.x & 15;
        const long long d = J.dst_row + k;
        const h8 v = *reinterpret_cast<const h8*>(J.src_h + ((size_t)r * 16 + (g ^ (r & 15))) * 8);
        *reinterpret_cast<h8*>(cmp_h + ((size_t)d * 16 + (g ^ (int)(d & 15))) * 8) = v;
    }
    // the group's tail up to the next multiple of 256 is swept too: zero it (its rows count as dead, but an
    // fp16 inf / NaN from stale memory must not reach the matrix core)
    h8 z;
    for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
    for (long long d = J.dst_row + cnt + (threadIdx.x >> 4); d < J.zero_upto; d += 16)
        *reinterpret_cast<h8*>(cmp_h + ((size_t)d * 16 + (threadIdx.x & 15)) * 8) = z;
}

// finalize: the same outputs as merge_knn_kernel (idx0, d0, d1, tie queue)
__global__ void pf_finalize_kernel(const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf,
                                   const float* __restrict__ tuv,
                                   const unsigned long long* __restrict__ best, const unsigned long long* __restrict__ second,
                                   int* __restrict__ k_i0, float* __restrict__ k_d0, float* __restrict__ k_d1,
                                   int* __restrict__ fix_count, int4* __restrict__ fix_list, int fix_cap) {
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
