// msfm_kernels.hip.h -- gfx950 (CDNA4) device code of the ComputeMatches hot path.
//
// What the kernels replace (reference = nebula-beta/MonocularSfM):
//   dist_top2_kernel   the two cv::BFMatcher::knnMatch(k=2) sweeps per image pair
//                      (src/Feature/FeatureUtils.cpp:146-149 called twice from :168-169), i.e.
//                      cv::batchDistance + hal::normL2Sqr_; BOTH directions come from one pass
//                      over the distance tile because (a-b)^2 == (b-a)^2 bitwise.
//   merge_knn_kernel   the K=2 insertion pass of batchDistance (lowest index wins ties) + sqrt.
//   tie_fixup_kernel   restores knnMatch's sqrt-space tie rule where the d^2-space selection
//                      above can pick a different index (distinct d^2, equal sqrtf).
//   epilogue_kernel    Lowe ratio (FeatureUtils.cpp:150-156), CrossCheck incl. the operator[]
//                      quirk (:281-310), FilterMatchesByDistance (:208-218), stream compaction.
//
// Numerics contract (bit-exact): S(q,t) is accumulated in fp32 in the order of
// the named OpenCV build (MSFM_ORDER_*): no reassociation, and for the SSE order no FMA
// contraction (this translation unit is compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace msfm {

constexpr int kDim = 128;
constexpr int kBM = 128;      // A rows (query descriptors) per workgroup
constexpr int kBN = 128;      // B rows (train descriptors) per tile
constexpr int kTM = 8;        // per-thread micro tile: 8 A rows x 4 B rows
constexpr int kTN = 4;
constexpr int kThreads = 512; // 8 waves; lane = tx*16 + ty (tx 0..3, ty 0..15)
constexpr int kChunkPos = 32; // storage positions per LDS chunk of a B tile
constexpr int kChunks = kDim / kChunkPos;
constexpr int kPanelFloats = kBM * kDim; // one 128-row block of an image, 64 KiB
constexpr int kSlotFloats = kChunkPos * kBN; // one B chunk of a tile in HBM, 16 KiB
constexpr int kWaveCols = kBN / 8;            // B rows per wave per tile
constexpr int kWaveSlotFloats = kChunkPos * kWaveCols;  // one wave-private chunk slot, 2 KiB
// LDS carve-up (floats): A panel 64 KiB | 8 waves x 2 chunk slots (32 KiB; reused for the final
// row merge: 8 waves x 3 x 128 floats = 12 KiB)
constexpr int kLdsA = 0;
constexpr int kLdsB = kPanelFloats;
constexpr int kLdsFloats = kLdsB + 8 * 2 * kWaveSlotFloats;
constexpr int kLdsBytes = kLdsFloats * 4;

// the sixteen-group order (AVX-512) parks the rows' running argmin indices in LDS between tile epilogues (dist_top2_kernel)
constexpr int kLdsBytesIdxStash = kLdsBytes + kTM * kThreads * 4;
template <int ORDER> struct OrderTraits;
// OpenCV 4.x SSE baseline: lanes l=0..3 x accumulators v=0..3, 8 iterations, no FMA,
// ((d0+d1)+d2)+d3 per lane then (l0+l2)+(l1+l3): groups are processed in lane order 0,2,1,3
// so that the final tree is a balanced in-order reduction over the processing order.
template <> struct OrderTraits<0> {
    static constexpr int kGroups = 4, kIters = 8, kGroupPos = 32;
    static constexpr bool kFused = false;
    __host__ __device__ static int pos_to_k(int pos) {
        const int g = pos >> 5, v = (pos >> 3) & 3, it = pos & 7;
        const int lane = (g == 0) ? 0 : (g == 1) ? 2 : (g == 2) ? 1 : 3;
        return 16 * it + 4 * v + lane;
    }
};
// OpenCV 4.x AVX2+FMA3: lanes l=0..7 x accumulators v=0..3, 4 iterations, fused,
// ((d0+d1)+d2)+d3 per lane then ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)).
template <> struct OrderTraits<1> {
    static constexpr int kGroups = 8, kIters = 4, kGroupPos = 16;
    static constexpr bool kFused = true;
    __host__ __device__ static int pos_to_k(int pos) {
        const int g = pos >> 4, v = (pos >> 2) & 3, it = pos & 3;
        return 32 * it + 8 * v + g;
    }
};

// The small kernels of a sub-batch's tail run on one stream while the persistent sweep 1 of the next sub-batch owns every CU
// on the other.  At equal priority the SIMD arbiter keeps serving the (older) sweep waves: measured 8 x slower for the
// thresholds kernel, 55 x for the single-workgroup plan kernel -- the whole tail then takes as long as the sweep it should
// hide behind.  With the highest wave priority they get their issue slots and are gone in their stand-alone time.
#define MSFM_TAIL_PRIO() __builtin_amdgcn_s_setprio(3)

// AVX-512 + FMA3 (the third named order, MSFM_ORDER_AVX512_FMA = 3): lanes l = 0..15 x accumulators v = 0..3, 2 iterations,
// fused, ((d0+d1)+d2)+d3 per lane, then y_l = (s_l + s_{l+8}) + (s_{l+4} + s_{l+12}) and (y0 + y2) + (y1 + y3): a balanced
// in-order tree over the lanes processed as 0,8,4,12, 2,10,6,14, 1,9,5,13, 3,11,7,15.
template <> struct OrderTraits<3> {
    static constexpr int kGroups = 16, kIters = 2, kGroupPos = 8;
    static constexpr bool kFused = true;
    __host__ __device__ static int pos_to_k(int pos) {
        const int g = pos >> 3, v = (pos >> 1) & 3, it = pos & 1;
        // g = 4 a + b: lane = (a's quarter: 0, 2, 1, 3) + (b's step: 0, 8, 4, 12)
        const int a = g >> 2, b = g & 3;
        const int lane = ((a == 0) ? 0 : (a == 1) ? 2 : (a == 2) ? 1 : 3) + ((b == 0) ? 0 : (b == 1) ? 8 : (b == 2) ? 4 : 12);
        return 64 * it + 16 * v + lane;
    }
};

struct PairDesc {
    const float* a_panel;  // image id1 (query), panel layout
    const float* b_panel;  // image id2 (train)
    const float* a_rawp;   // the fp32 rows, PERMUTED for the exact re-check: position 64 h + 4 L + c of a row holds its element 16 (4 h + c) + L
    const float* b_rawp;   // (lane L of a 16-lane group needs the elements 16 j + L, j = 0..7, in every accumulation order: two 16-byte loads);
                           // the sqrt-space tie fix-up reads them through rawp_pos (msfm_store.hip.h).  Null for byte images on the integer route
    int n1, n2;
    int a_blocks, b_tiles;
    int n1pad, n2pad;
    int ranges;            // B-tile ranges the pair was split into
    int valid;             // 0: a side is empty -> no neighbours, no device work
    int path;              // 0: brute-force exact kernel, 1: MFMA prefilter + exact re-check
    int a_blocks256;       // 256-row A blocks (prefilter work items)
    int exact_int;         // both images were uploaded as bytes: S is an exact integer under ANY accumulation order
    long long rp_off;      // row partials  [ranges][n1pad]
    long long cp_off;      // column partials [a_blocks][n2pad]
    long long kf_off;      // final forward knn arrays [n1pad]
    long long kr_off;      // final reverse knn arrays [n2pad]
    long long out_off;     // staged matches [n1]
};

struct WorkItem {
    int pair;     // -1: padding item (XCD interleave), nothing to do
    int a_blk;
    int bt_begin, bt_end;
    int range;
    int pad[3];
};

struct Top2 {
    float s0; int i0; float s1;
};

__device__ __forceinline__ float f_inf() { return __builtin_huge_valf(); }

// candidate with a HIGHER index than everything folded so far (strict <: earlier wins ties)
__device__ __forceinline__ void top2_push(float& s0, int& i0, float& s1, float s, int idx) {
    const bool lt = s < s0;
    s1 = __builtin_amdgcn_fmed3f(s0, s1, s);  // second smallest of {s0<=s1, s}
    i0 = lt ? idx : i0;
    s0 = fminf(s0, s);
}
// general merge: ties on s0 go to the lower index
__device__ __forceinline__ void top2_merge(float& s0, int& i0, float& s1, float bs0, int bi0, float bs1) {
    const bool take_b = (bs0 < s0) || (bs0 == s0 && (unsigned)bi0 < (unsigned)i0);
    s1 = fminf(fmaxf(s0, bs0), fminf(s1, bs1));
    i0 = take_b ? bi0 : i0;
    s0 = fminf(s0, bs0);
}

// ---------------------------------------------------------------------------------------
// async global -> LDS copy of `bytes` (multiple of 8 KiB) by the whole 512-thread group:
// each wave instruction moves 64 lanes x 16 B = 1 KiB, LDS image = global image (lane-linear).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void glds_copy(const float* __restrict__ g, float* lds, int floats, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll 1
    for (int piece = wave; piece * 256 < floats; piece += kThreads / 64) {
        const float* gp = g + piece * 256 + lane * 4;
        float* lp = lds + piece * 256;  // wave-uniform base; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// One group of 4 accumulator chains (one SIMD lane of the OpenCV loop) for the 8x4 micro tile,
// in packed fp32 (v_pk_add/mul_f32: two A rows per instruction, the B value broadcast):
// s[i][j] = ((p0+p1)+p2)+p3 for row pair i x column j, p_v = sum over `kIters` storage positions
// in order.  Per position the stages (sub, mul, accumulate) are issued as blocks of 8 independent
// packed ops so no instruction waits on its predecessor, and the LDS reads of the next position
// are issued before the current one is consumed.
//   sa: this lane's A rows at position 0 (rows 4ty..4ty+3; +64 floats: rows 64+4ty..), stride kBM
//   sb: this lane's 4 B rows at position 0 of the wave-private chunk, stride kWaveCols
template <int ORDER>
__device__ __forceinline__ void group_chains(const float* __restrict__ sa, const float* __restrict__ sb,
                                             v2f (&s)[4][kTN]) {
    using OT = OrderTraits<ORDER>;
    constexpr int kPos = 4 * OT::kIters;
    v4f a_lo = *reinterpret_cast<const v4f*>(sa);
    v4f a_hi = *reinterpret_cast<const v4f*>(sa + 64);
    v4f bb = *reinterpret_cast<const v4f*>(sb);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        v2f p[4][kTN];
#pragma unroll
        for (int it = 0; it < OT::kIters; ++it) {
            const int pos = v * OT::kIters + it;
            const v2f ap[4] = {a_lo.xy, a_lo.zw, a_hi.xy, a_hi.zw};
            const float bj[kTN] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // (the sixteen-group order keeps five partial-sum sets of the tree alive: its operands of the next position are
                // fetched after this one is consumed instead -- twelve registers less, no spills; it is the cross-check order)
                if (OT::kGroups != 16 && h == 1 && pos + 1 < kPos) {
                    // next position's LDS reads go out between the two half blocks: the wait for THIS
                    // position's operands (hipcc emits lgkmcnt(0)) then never covers a just-issued read
                    __builtin_amdgcn_sched_barrier(0);
                    a_lo = *reinterpret_cast<const v4f*>(sa + (pos + 1) * kBM);
                    a_hi = *reinterpret_cast<const v4f*>(sa + (pos + 1) * kBM + 64);
                    bb = *reinterpret_cast<const v4f*>(sb + (pos + 1) * kWaveCols);
                }
                if (OT::kGroups == 16) {   // (one row pair at a time: four difference registers alive instead of sixteen)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        v2f u[kTN];
#pragma unroll
                        for (int j = 0; j < kTN; ++j) u[j] = ap[2 * h + i] - (v2f)(bj[j]);
#pragma unroll
                        for (int j = 0; j < kTN; ++j)
                            p[2 * h + i][j] = (it == 0) ? u[j] * u[j] : __builtin_elementwise_fma(u[j], u[j], p[2 * h + i][j]);
                    }
                    continue;
                }
                v2f t[2][kTN];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < kTN; ++j) t[i][j] = ap[2 * h + i] - (v2f)(bj[j]);
                if (OT::kFused) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < kTN; ++j)
                            p[2 * h + i][j] = (it == 0) ? t[i][j] * t[i][j]
                                                        : __builtin_elementwise_fma(t[i][j], t[i][j], p[2 * h + i][j]);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < kTN; ++j) t[i][j] = t[i][j] * t[i][j];  // rounded product
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < kTN; ++j)
                            p[2 * h + i][j] = (it == 0) ? t[i][j] : p[2 * h + i][j] + t[i][j];  // 0 + x == x
                }
            }
            // keep the scheduler from hoisting later positions' LDS reads over this block (it
            // otherwise runs out of VGPRs and spills)
            __builtin_amdgcn_sched_barrier(0);
            if (OT::kGroups == 16 && pos + 1 < kPos) {
                a_lo = *reinterpret_cast<const v4f*>(sa + (pos + 1) * kBM);
                a_hi = *reinterpret_cast<const v4f*>(sa + (pos + 1) * kBM + 64);
                bb = *reinterpret_cast<const v4f*>(sb + (pos + 1) * kWaveCols);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < kTN; ++j) s[i][j] = (v == 0) ? p[i][j] : s[i][j] + p[i][j];
    }
}

// ---------------------------------------------------------------------------------------
// dist_top2_kernel: one workgroup (8 waves) = one 128-row block of image A x a range of 128-row
// tiles of image B.  The A panel is shared in LDS (loaded once).  Each WAVE owns 16 of a tile's
// 128 B rows and streams them through its own double-buffered LDS slots with LDS-DMA, so the
// main loop has no workgroup barrier: waves drift freely and hide each other's LDS latency and
// epilogues.  Lane (tx = lane>>4, ty = lane&15) computes an 8x4 micro tile: A rows
// {4ty..4ty+3} U {64+4ty..64+4ty+3} x B rows tile*128 + wave*16 + 4tx..+3, S in exact order.
//   rows:    running top-2 (S0, argmin, S1) per A row in registers across all tiles, merged over
//            tx (shuffles) and the 8 waves (LDS) once at the end;
//   columns: top-2 over the block's 128 A rows = the wave's 16 ty lanes (shuffles), written as a
//            partial per (A block, B row).
// ---------------------------------------------------------------------------------------
template <int ORDER>
__global__ __launch_bounds__(kThreads) void dist_top2_kernel(
    const PairDesc* __restrict__ pairs, const WorkItem* __restrict__ items,
    float* __restrict__ rp_s0, int* __restrict__ rp_i0, float* __restrict__ rp_s1,
    float* __restrict__ cp_s0, int* __restrict__ cp_i0, float* __restrict__ cp_s1) {
    using OT = OrderTraits<ORDER>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem + kLdsA;

    const WorkItem item = items[blockIdx.x];
    if (item.pair < 0) return;
    const PairDesc pd = pairs[item.pair];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = lane & 15;
    const int tx = lane >> 4;
    float* sBw = smem + kLdsB + wave * (2 * kWaveSlotFloats);  // this wave's two chunk slots

    const int n_steps = (item.bt_end - item.bt_begin) * kChunks;
    // wave-private LDS-DMA of chunk `st` of the item: 32 positions x 16 B rows = 2 KiB = two
    // 1-KiB wave instructions; lane l fetches 16 B: position l/4 (+16), B rows 4*(l%4)..+3
    const float* gB = pd.b_panel + (size_t)item.bt_begin * kPanelFloats + wave * kWaveCols + (lane >> 2) * kBN + (lane & 3) * 4;
    auto dma_chunk = [&](int st) {
        const float* g = gB + (size_t)st * kSlotFloats;  // chunks of consecutive tiles are contiguous
        float* l = sBw + (st & 1) * kWaveSlotFloats;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 16 * kBN),
                                         (__attribute__((address_space(3))) void*)(l + 256), 16, 0, 0);
    };

    // prologue: A panel (all waves) + this wave's first B chunk, one barrier
    glds_copy(pd.a_panel + (size_t)item.a_blk * kPanelFloats, sA, kPanelFloats, tid);
    dma_chunk(0);
    __syncthreads();  // includes vmcnt(0): the A panel and chunk 0 have landed

    const int arow0 = item.a_blk * kBM + 4 * ty;  // rows arow0..+3 and arow0+64..+3
    float r_s0[kTM], r_s1[kTM];
    int r_i0[kTM];
#pragma unroll
    for (int i = 0; i < kTM; ++i) { r_s0[i] = f_inf(); r_s1[i] = f_inf(); r_i0[i] = -1; }
    // Sixteen groups: five partial-sum sets of the balanced tree are alive inside the fourth chunk (160 registers) next to the
    // 32 chain accumulators -- the eight running argmin indices wait in LDS ([i][thread]: conflict-free) while the sums are
    // formed and are only in registers during a tile's epilogue (256 VGPRs, no spill; the other orders keep them in registers).
    constexpr bool kStashIdx = OT::kGroups == 16;
    int* sIdx = reinterpret_cast<int*>(smem + kLdsFloats);
    if (kStashIdx) {
#pragma unroll
        for (int i = 0; i < kTM; ++i) sIdx[i * kThreads + tid] = -1;
    }

    v2f lvl0[4][kTN], lvl1[4][kTN], lvl2[4][kTN];  // [row pair][column]
    (void)lvl2;

    int step = 0;
    // start of a chunk step: this wave's DMA for the step has landed; refill the other slot (its
    // reads, issued during the previous step by this same wave, have all been consumed)
    auto begin_chunk = [&](int c, const float*& sa, const float*& sb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (step + 1 < n_steps) dma_chunk(step + 1);
        sa = sA + c * kChunkPos * kBM + 4 * ty;
        sb = sBw + (step & 1) * kWaveSlotFloats + 4 * tx;
        ++step;
    };
#define MSFM_FOREACH(expr)                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) \
        _Pragma("unroll") for (int j = 0; j < kTN; ++j) { expr; }

#pragma unroll 1
    for (int bt = item.bt_begin; bt < item.bt_end; ++bt) {
        // Balanced in-order tree over the lane groups (a binary-counter stack).
        const float *sa, *sb;
        if (OT::kGroups == 4) {  // one group per chunk: ((g0+g1)+(g2+g3)) -> lvl1
#pragma unroll 1
            for (int c = 0; c < kChunks; ++c) {
                begin_chunk(c, sa, sb);
                v2f s[4][kTN];
                group_chains<ORDER>(sa, sb, s);
                if ((c & 1) == 0) { MSFM_FOREACH(lvl0[i][j] = s[i][j]) }
                else if (c == 1) { MSFM_FOREACH(lvl1[i][j] = lvl0[i][j] + s[i][j]) }
                else { MSFM_FOREACH(lvl1[i][j] = lvl1[i][j] + (lvl0[i][j] + s[i][j])) }
            }
        } else if (OT::kGroups == 16) {  // four groups per chunk: q_c = (g0+g1)+(g2+g3), result (q0+q1)+(q2+q3) -> lvl2
#pragma unroll 1
            for (int c = 0; c < kChunks; ++c) {
                begin_chunk(c, sa, sb);
                v2f s[4][kTN], q[4][kTN];
                group_chains<ORDER>(sa, sb, lvl0);
                group_chains<ORDER>(sa + OT::kGroupPos * kBM, sb + OT::kGroupPos * kWaveCols, s);
                MSFM_FOREACH(q[i][j] = lvl0[i][j] + s[i][j])
                group_chains<ORDER>(sa + 2 * OT::kGroupPos * kBM, sb + 2 * OT::kGroupPos * kWaveCols, lvl0);
                group_chains<ORDER>(sa + 3 * OT::kGroupPos * kBM, sb + 3 * OT::kGroupPos * kWaveCols, s);
                MSFM_FOREACH(q[i][j] = q[i][j] + (lvl0[i][j] + s[i][j]))
                if (c == 0 || c == 2) { MSFM_FOREACH(lvl1[i][j] = q[i][j]) }
                else if (c == 1) { MSFM_FOREACH(lvl2[i][j] = lvl1[i][j] + q[i][j]) }
                else { MSFM_FOREACH(lvl2[i][j] = lvl2[i][j] + (lvl1[i][j] + q[i][j])) }
            }
        } else {  // two groups per chunk: (((g0+g1)+(g2+g3)) + ((g4+g5)+(g6+g7))) -> lvl2
#pragma unroll 1
            for (int c = 0; c < kChunks; ++c) {
                begin_chunk(c, sa, sb);
                group_chains<ORDER>(sa, sb, lvl0);
                v2f s[4][kTN];
                group_chains<ORDER>(sa + OT::kGroupPos * kBM, sb + OT::kGroupPos * kWaveCols, s);
                if (c == 0 || c == 2) { MSFM_FOREACH(lvl1[i][j] = lvl0[i][j] + s[i][j]) }
                else if (c == 1) { MSFM_FOREACH(lvl2[i][j] = lvl1[i][j] + (lvl0[i][j] + s[i][j])) }
                else { MSFM_FOREACH(lvl2[i][j] = lvl2[i][j] + (lvl1[i][j] + (lvl0[i][j] + s[i][j]))) }
            }
        }
        // micro-tile row i: i < 4 -> A row arow0 + i (pair i/2), i >= 4 -> A row arow0 + 64 + (i-4)
        float fin[kTM][kTN];
#pragma unroll
        for (int i = 0; i < kTM; ++i)
#pragma unroll
            for (int j = 0; j < kTN; ++j)
                // v_min_f32 returns the non-NaN operand: a NaN distance becomes +inf, which is never inserted
                // (batchDistance only replaces its FLT_MAX-initialised slots by strictly smaller values) -- and
                // v_med3_f32 in top2_push is only well-defined without NaNs
                fin[i][j] = fminf((OT::kGroups == 4) ? lvl1[i >> 1][j][i & 1] : lvl2[i >> 1][j][i & 1], f_inf());

        // ---- tile epilogue -------------------------------------------------------------
        const int col0 = bt * kBN + wave * kWaveCols + tx * kTN;  // first B row (train index) of this lane
        if ((bt + 1) * kBN > pd.n2 || (item.a_blk + 1) * kBM > pd.n1) {
#pragma unroll
            for (int i = 0; i < kTM; ++i)
#pragma unroll
                for (int j = 0; j < kTN; ++j)
                    if (arow0 + (i & 3) + (i >> 2) * 64 >= pd.n1 || col0 + j >= pd.n2) fin[i][j] = f_inf();
        }
        // rows: this lane's 4 train indices, ascending
        if (kStashIdx) {
#pragma unroll
            for (int i = 0; i < kTM; ++i) r_i0[i] = sIdx[i * kThreads + tid];
        }
#pragma unroll
        for (int i = 0; i < kTM; ++i)
#pragma unroll
            for (int j = 0; j < kTN; ++j) top2_push(r_s0[i], r_i0[i], r_s1[i], fin[i][j], col0 + j);
        if (kStashIdx) {
#pragma unroll
            for (int i = 0; i < kTM; ++i) sIdx[i * kThreads + tid] = r_i0[i];
        }
        // columns: this lane's 8 query indices (ascending), then the 16 ty lanes of the wave
#pragma unroll
        for (int j = 0; j < kTN; ++j) {
            float c_s0 = f_inf(), c_s1 = f_inf();
            int c_i0 = -1;
#pragma unroll
            for (int i = 0; i < kTM; ++i)
                top2_push(c_s0, c_i0, c_s1, fin[i][j], arow0 + (i & 3) + (i >> 2) * 64);
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                const float o_s0 = __shfl_xor(c_s0, m);
                const int o_i0 = __shfl_xor(c_i0, m);
                const float o_s1 = __shfl_xor(c_s1, m);
                top2_merge(c_s0, c_i0, c_s1, o_s0, o_i0, o_s1);
            }
            if (ty == 0) {
                const long long o = pd.cp_off + (long long)item.a_blk * pd.n2pad + col0 + j;
                cp_s0[o] = c_s0;
                cp_i0[o] = c_i0;
                cp_s1[o] = c_s1;
            }
        }
    }

    // ---- rows: merge the 4 tx lane groups of the wave, then the 8 waves through LDS ----------
    if (kStashIdx) {
#pragma unroll
        for (int i = 0; i < kTM; ++i) r_i0[i] = sIdx[i * kThreads + tid];
    }
#pragma unroll
    for (int i = 0; i < kTM; ++i) {
#pragma unroll
        for (int m = 16; m < 64; m <<= 1) {
            const float o_s0 = __shfl_xor(r_s0[i], m);
            const int o_i0 = __shfl_xor(r_i0[i], m);
            const float o_s1 = __shfl_xor(r_s1[i], m);
            top2_merge(r_s0[i], r_i0[i], r_s1[i], o_s0, o_i0, o_s1);
        }
    }
    __syncthreads();  // every wave is done with its B slots: reuse the region as merge scratch
    float* sR = smem + kLdsB;  // [wave][3][128 rows]
    if (tx == 0) {
#pragma unroll
        for (int i = 0; i < kTM; ++i) {
            const int r = 4 * ty + (i & 3) + (i >> 2) * 64;
            sR[(wave * 3 + 0) * kBM + r] = r_s0[i];
            sR[(wave * 3 + 1) * kBM + r] = __int_as_float(r_i0[i]);
            sR[(wave * 3 + 2) * kBM + r] = r_s1[i];
        }
    }
    __syncthreads();
    if (tid < kBM) {
        float s0 = sR[0 * kBM + tid];
        int i0 = __float_as_int(sR[1 * kBM + tid]);
        float s1 = sR[2 * kBM + tid];
#pragma unroll
        for (int w = 1; w < 8; ++w)
            top2_merge(s0, i0, s1, sR[(w * 3 + 0) * kBM + tid], __float_as_int(sR[(w * 3 + 1) * kBM + tid]),
                       sR[(w * 3 + 2) * kBM + tid]);
        const long long o = pd.rp_off + (long long)item.range * pd.n1pad + item.a_blk * kBM + tid;
        rp_s0[o] = s0;
        rp_i0[o] = i0;
        rp_s1[o] = s1;
    }
#undef MSFM_FOREACH
}

// ---------------------------------------------------------------------------------------
// merge_knn_kernel: fold the partials of one pair into the final knnMatch(k=2) result per
// query (forward) and per train row (reverse): idx0, d0 = sqrtf(S0), d1 = sqrtf(S1).
// Rows whose two best distances collide in sqrt space are queued for tie_fixup_kernel.
// grid = (ceil(max(n1pad,n2pad)/256), n_pairs)
// ---------------------------------------------------------------------------------------
__global__ void merge_knn_kernel(const PairDesc* __restrict__ pairs,
                                 const float* __restrict__ rp_s0, const int* __restrict__ rp_i0,
                                 const float* __restrict__ rp_s1, const float* __restrict__ cp_s0,
                                 const int* __restrict__ cp_i0, const float* __restrict__ cp_s1,
                                 int* __restrict__ k_i0, float* __restrict__ k_d0, float* __restrict__ k_d1,
                                 int* __restrict__ fix_count, int4* __restrict__ fix_list, int fix_cap) {
    MSFM_TAIL_PRIO();
    const PairDesc pd = pairs[blockIdx.y];
    if (!pd.valid || pd.path != 0) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    for (int dir = 0; dir < 2; ++dir) {
        const int n = dir == 0 ? pd.n1 : pd.n2;
        if (e >= n) continue;
        const int npad = dir == 0 ? pd.n1pad : pd.n2pad;
        const int parts = dir == 0 ? pd.ranges : pd.a_blocks;
        const long long base = (dir == 0 ? pd.rp_off : pd.cp_off) + e;
        const float* ps0 = dir == 0 ? rp_s0 : cp_s0;
        const int* pi0 = dir == 0 ? rp_i0 : cp_i0;
        const float* ps1 = dir == 0 ? rp_s1 : cp_s1;
        float s0 = ps0[base], s1 = ps1[base];
        int i0 = pi0[base];
        for (int p = 1; p < parts; ++p) {
            const long long o = base + (long long)p * npad;
            top2_merge(s0, i0, s1, ps0[o], pi0[o], ps1[o]);
        }
        // FLT_MAX / -1 where batchDistance leaves its initial values (fewer than k neighbours)
        const float d0 = (s0 < f_inf()) ? sqrtf(s0) : 3.402823466e+38f;
        const float d1 = (s1 < f_inf()) ? sqrtf(s1) : 3.402823466e+38f;
        if (!(s0 < f_inf())) i0 = -1;
        const long long ko = (dir == 0 ? pd.kf_off : pd.kr_off) + e;
        k_i0[ko] = i0;
        k_d0[ko] = d0;
        k_d1[ko] = d1;
        // knnMatch orders by (sqrtf(S), index): if the two best collide after sqrt, a lower
        // index with a slightly larger S but the same sqrtf may be the true first neighbour.
        if (i0 >= 0 && d0 == d1) {
            const int slot = atomicAdd(fix_count, 1);
            if (slot < fix_cap) fix_list[slot] = make_int4((int)blockIdx.y, dir, e, 0);
        }
    }
}

// position of element k of a row in its permuted copy (PairDesc::a_rawp; msfm_store.hip.h has the same function for the build kernels)
__device__ __forceinline__ int tie_rawp_pos(int k) { return 64 * (k >> 6) + 4 * (k & 15) + ((k >> 4) & 3); }

// exact-order S for two descriptors given as permuted rows (used only on the rare tie path)
template <int ORDER>
__device__ float l2sqr_rowmajor(const float* __restrict__ a, const float* __restrict__ b) {
    using OT = OrderTraits<ORDER>;
    float lvl[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // binary-counter stack: the balanced in-order tree over the groups
    for (int g = 0; g < OT::kGroups; ++g) {
        float s = 0.f;
        for (int v = 0; v < 4; ++v) {
            float p = 0.f;
            for (int it = 0; it < OT::kIters; ++it) {
                const int k = tie_rawp_pos(OT::pos_to_k(g * OT::kGroupPos + v * OT::kIters + it));
                const float t = a[k] - b[k];
                if (it == 0) p = t * t;
                else if (OT::kFused) p = __builtin_fmaf(t, t, p);
                else p = p + t * t;
            }
            s = (v == 0) ? p : s + p;
        }
        int j = 0;
        for (int gg = g; gg & 1; gg >>= 1, ++j) s = lvl[j] + s;   // (the earlier subtree is the left operand)
        lvl[j] = s;
    }
    return lvl[OT::kGroups == 4 ? 2 : (OT::kGroups == 8 ? 3 : 4)];
}

// one 64-lane workgroup per queued row: lowest index whose sqrtf(S) equals d0
template <int ORDER>
__global__ void tie_fixup_kernel(const PairDesc* __restrict__ pairs, const int* __restrict__ fix_count,
                                 const int4* __restrict__ fix_list, int fix_cap,
                                 int* __restrict__ k_i0, const float* __restrict__ k_d0) {
    MSFM_TAIL_PRIO();
    const int nfix = min(*fix_count, fix_cap);
    for (int f = blockIdx.x; f < nfix; f += gridDim.x) {
        const int4 w = fix_list[f];
        const PairDesc pd = pairs[w.x];
        const bool fwd = (w.y == 0);
        const float* me = (fwd ? pd.a_rawp : pd.b_rawp) + (size_t)w.z * kDim;
        const float* other = fwd ? pd.b_rawp : pd.a_rawp;
        const int n_other = fwd ? pd.n2 : pd.n1;
        const long long ko = (fwd ? pd.kf_off : pd.kr_off) + w.z;
        const float d0 = k_d0[ko];
        int best = 0x7fffffff;
        for (int t = threadIdx.x; t < n_other; t += 64) {
            // forward: S(q=me, t); reverse: S(q=other row, t=me) -- same bits, (a-b)^2 == (b-a)^2
            const float s = fwd ? l2sqr_rowmajor<ORDER>(me, other + (size_t)t * kDim)
                                : l2sqr_rowmajor<ORDER>(other + (size_t)t * kDim, me);
            if (sqrtf(s) == d0) { best = t; break; }  // ascending t per lane: first hit is the lane's lowest
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) best = min(best, __shfl_xor(best, m));
        if (threadIdx.x == 0 && best != 0x7fffffff) k_i0[ko] = best;
    }
}

// ---------------------------------------------------------------------------------------
// epilogue_kernel: one workgroup per pair.  Lowe ratio both directions, CrossCheck with the
// reference's operator[] quirk (missing key reads 0), max-distance cut, ordered compaction.
// ---------------------------------------------------------------------------------------
struct EpiParams {
    float ratio;
    int cross_check;
    double max_distance;
};

// ---- order-invariance certificate ------------------------------------------------------------------------------
// The database stores (queryIdx, trainIdx) only.  Which rows could come out differently under ANOTHER conforming fp32
// evaluation order of hal::normL2Sqr_ (a different OpenCV build: other SIMD width, other reduction tree, FMA or not)?
//
// Bound.  S = sum of 128 non-negative terms (a_c - b_c)^2.  In fp32 (u = 2^-24) each term carries the rounding of the
// difference (twice, it is squared) and of the product (none if fused), and passes through at most 127 additions in any
// association: |S^ - S| <= gamma_130 S, gamma_k = k u / (1 - k u) = 7.75e-6 (Higham, Accuracy and Stability, 4.2 / 3.1;
// valid for every order because all terms are >= 0).  d = sqrtf(S^) adds one rounding: d in sqrt(S) (1 +- (gamma/2 + u)).
// Two conforming builds therefore differ by at most a relative kOrderEps = 1e-5 > gamma_130 + 2 u + the rounding of
// fl(ratio * d1) in any distance they report.
//
// What this path knows per row: (i0, d0, d1) in the pinned order, and that every other element j has S^_j >= S^_1
// (brute force: by selection; prefilter: non-candidates are provably worse than the true second neighbour).  Hence under
// any other order B:  d0_B >= d0 (1 - e),  d1_B <= d1 (1 + e)  always, and if d0 (1 + e) < d1 (1 - e) the first
// neighbour is the same element with d0_B <= d0 (1 + e), d1_B >= d1 (1 - e) -- and no tie in sqrt space can hand the
// index to a lower-numbered element.  A row is CERTIFIED when the reference's decisions are the same on the whole
// interval:
//     passes the ratio test:  d0 (1+e) < ratio d1 (1-e)  and  d0 (1+e) < d1 (1-e)  and the distance cut does not
//                             straddle max_distance (forward direction only: FilterMatchesByDistance sees forward d0);
//     fails it:               d0 (1-e) >= ratio d1 (1+e)   (whichever element is first: it yields no match);
//     beyond the cut:         d0 (1-e) > max_distance      (forward: no match whatever the ratio test says).
// Rows pruned by the prefilter are dead with a margin of eps >= 1.5e-3 (na + nb) >> gamma S (msfm_prefilter.hip.h).
// Pairs of byte images are exact integers under every order.  Rows that are not certified are "order-sensitive";
// zero of them in a call <=> the stored rows are identical under any conforming normL2Sqr_ build.
constexpr double kOrderEps = 1.0e-5;

__device__ __forceinline__ bool order_sensitive(int i0, float d0, float d1, float ratio, double max_distance, bool forward) {
    if (i0 < 0 || !(d1 < 3.402823466e+38f)) return false;   // no match under any order (pruned with margin / < 2 neighbours)
    const double lo0 = (double)d0 * (1.0 - kOrderEps), hi0 = (double)d0 * (1.0 + kOrderEps);
    const double lo1 = (double)d1 * (1.0 - kOrderEps), hi1 = (double)d1 * (1.0 + kOrderEps);
    if (forward && lo0 > max_distance) return false;        // cut by FilterMatchesByDistance whatever the ratio test says
    const bool pass = d0 < ratio * d1;
    bool certified;
    if (pass)
        certified = hi0 < (double)ratio * lo1 && hi0 < lo1 && (!forward || hi0 <= max_distance || lo0 > max_distance);
    else
        certified = lo0 >= (double)ratio * hi1;
    return !certified;
}

// Where the epilogue takes a row's (idx0, d0, d1) from: the final kNN arrays (brute-force route, knnMatch-level API, tie fix-up), or --
// matrix-core route, match lists -- straight from the reduce slots of the exact re-check: a row whose threshold is -inf was pruned
// (it may own stray candidates of the other direction: "no neighbour"), a live one carries its best / second (S, index) keys.
// The second form saves writing and re-reading 12 bytes for every one of a batch's ~83 M row / column slots, 94 % of them dead.
struct KnnFromArrays {
    const int* i0;
    const float* d0;
    const float* d1;
    __device__ __forceinline__ void get(long long slot, int& i, float& a, float& b) const {
        i = i0[slot];
        a = d0[slot];
        b = d1[slot];
    }
};
struct KnnFromKeys {
    const float* tuv;
    const unsigned long long* best;
    const unsigned long long* second;
    __device__ __forceinline__ void get(long long slot, int& i, float& a, float& b) const {
        i = -1;
        a = b = 3.402823466e+38f;
        if (tuv[slot] == -f_inf()) return;
        const unsigned long long kb = best[slot], ks = second[slot];
        if (kb != ~0ull) { i = (int)(unsigned)(kb & 0xffffffffu); a = sqrtf(__uint_as_float((unsigned)(kb >> 32))); }
        if (ks != ~0ull) b = sqrtf(__uint_as_float((unsigned)(ks >> 32)));
    }
};

template <class KNN>
__global__ __launch_bounds__(256) void epilogue_kernel(const PairDesc* __restrict__ pairs, EpiParams prm, KNN knn,
                                                       int2* __restrict__ st_qt, float* __restrict__ st_d,
                                                       int* __restrict__ counts, int* __restrict__ sens_counts) {
    MSFM_TAIL_PRIO();
    const PairDesc pd = pairs[blockIdx.x];
    __shared__ int wsum[4];
    __shared__ int running, n_sens;
    if (threadIdx.x == 0) { running = 0; n_sens = 0; }
    __syncthreads();
    // The reference indexes m[1] unconditionally (FeatureUtils.cpp:152): a direction whose train
    // set has < 2 rows is undefined there.  Build-defined: that direction yields no matches, and a
    // cross-checked pair with such a direction yields none at all.
    if (!pd.valid || pd.n2 < 2 || (prm.cross_check && pd.n1 < 2)) {
        if (threadIdx.x == 0) { counts[blockIdx.x] = 0; sens_counts[blockIdx.x] = 0; }
        return;
    }
    const bool certify = !pd.exact_int;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q0 = 0; q0 < pd.n1; q0 += 256) {
        const int q = q0 + threadIdx.x;
        bool keep = false, sens = false;
        int t = -1;
        float d0 = 0.f;
        if (q < pd.n1) {
            float d1;
            knn.get(pd.kf_off + q, t, d0, d1);
            // m[0].distance < distance_ratio * m[1].distance, fp32 product, strict
            keep = (t >= 0) && (d1 < 3.402823466e+38f) && (d0 < prm.ratio * d1);
            sens = certify && order_sensitive(t, d0, d1, prm.ratio, prm.max_distance, true);
            if (keep && prm.cross_check) {
                int rq;
                float rd0, rd1;
                knn.get(pd.kr_off + t, rq, rd0, rd1);
                const bool rkeep = (rq >= 0) && (rd1 < 3.402823466e+38f) && (rd0 < prm.ratio * rd1);
                const int vis = rkeep ? rq : 0;  // unordered_map::operator[] default-inserts 0
                keep = (vis == q);
            }
            if (keep && ((double)d0 > prm.max_distance)) keep = false;
        }
        const unsigned long long sb = __ballot(sens);
        if (lane == 0 && sb) atomicAdd(&n_sens, __popcll(sb));
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (keep) {
            st_qt[pd.out_off + off + before] = make_int2(q, t);
            st_d[pd.out_off + off + before] = d0;
        }
        __syncthreads();
        if (threadIdx.x == 0) running += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    // the reverse rows decide the cross-check
    if (certify && prm.cross_check)
        for (int t0 = 0; t0 < pd.n2; t0 += 256) {
            const int t = t0 + threadIdx.x;
            int ri = -1;
            float rd0 = 0.f, rd1 = 0.f;
            if (t < pd.n2) knn.get(pd.kr_off + t, ri, rd0, rd1);
            const bool s = t < pd.n2 && order_sensitive(ri, rd0, rd1, prm.ratio, prm.max_distance, false);
            const unsigned long long sb = __ballot(s);
            if (lane == 0 && sb) atomicAdd(&n_sens, __popcll(sb));
        }
    __syncthreads();
    if (threadIdx.x == 0) { counts[blockIdx.x] = running; sens_counts[blockIdx.x] = n_sens; }
}

// exclusive scan of per-pair counts (single workgroup; P is at most a few thousand per batch)
__global__ void scan_counts_kernel(const int* __restrict__ counts, long long* __restrict__ offsets, int n) {
    MSFM_TAIL_PRIO();
    __shared__ long long carry;
    __shared__ long long wtot[4];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        long long v = (i < n) ? counts[i] : 0;
        long long incl = v;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const long long o = __shfl_up(incl, m);
            if (lane >= m) incl += o;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        long long woff = carry;
        for (int w = 0; w < wave; ++w) woff += wtot[w];
        if (i < n) offsets[i] = woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = carry;
}

// copy each pair's staged matches to its CSR position
__global__ void gather_kernel(const PairDesc* __restrict__ pairs, const int* __restrict__ counts,
                              const long long* __restrict__ offsets, const int2* __restrict__ st_qt,
                              const float* __restrict__ st_d, int2* __restrict__ out_qt,
                              float* __restrict__ out_d) {
    MSFM_TAIL_PRIO();
    const PairDesc pd = pairs[blockIdx.x];
    const int c = counts[blockIdx.x];
    const long long o = offsets[blockIdx.x];
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
        out_qt[o + i] = st_qt[pd.out_off + i];
        out_d[o + i] = st_d[pd.out_off + i];
    }
}

}  // namespace msfm
