// msfm_sweep.hip.h -- the MFMA sweep kernels of the prefilter path (included by msfm_prefilter.hip.h).
//
// sweep_kernel<PASS>: one workgroup = 512 A rows x a range of 64-row B tiles, EIGHT waves = two per SIMD.
//
//   Wave w owns A rows w*64 .. w*64+63 for the whole work item: its fp16 A fragments (two 32-row MFMA blocks x
//   9 k-steps, the ninth holding the norm / threshold quadruple) live in registers, loaded once from HBM.  Only B
//   tiles stream through LDS: a ring of four 16 KiB slots filled by LDS-DMA three tiles ahead (every wave moves 2 KiB
//   of a tile plus its private copy of the tile's norm quadruples).
//
//   PING-PONG.  Waves w and w + 4 sit on the same SIMD (a workgroup's waves are dealt to the SIMDs cyclically).  The
//   two halves of the workgroup run the same per-tile program half a tile apart:
//
//        phase      2t                2t+1               2t+2               2t+3
//        waves 0-3  MFMA(t)           EPI(t)+DMA(t+3)    MFMA(t+1)          EPI(t+1)+DMA(t+4)
//        waves 4-7  EPI(t-1)+DMA(t+2) MFMA(t)            EPI(t)+DMA(t+3)    MFMA(t+1)
//
//   with one s_barrier between phases.  MFMA(t): the 36 matrix instructions of a tile (2 column blocks x 2 row blocks x
//   9 k-steps) and the LDS reads of the second column block's fragments -- nothing else.  EPI(t): everything that is
//   not matrix work -- the v_max3 epilogues on the 64 results per lane, the column partials, the DMA pieces of tile
//   t+3, the fold of the previous tile's column partials, and the LDS reads of the NEXT tile's first fragments, so that
//   the following MFMA phase starts on registers.  While one wave of a SIMD feeds the matrix pipe, the other one does
//   its VALU / LDS / VMEM work: the pipe sees MFMAs back to back (MI355X_MICROARCH.md, "Two waves per SIMD"), where
//   four unsynchronised waves of the round-1 kernel (each reads -> MFMAs -> epilogue) kept it 56 % busy.
//
//   Synchronisation: raw s_barrier + COUNTED s_waitcnt vmcnt(N) (loads only: stores may retire out of order), so
//   the DMA groups of the two younger tiles stay in flight across barriers.  Tile u is complete in LDS one phase
//   before its first MFMA (so it can be pre-read): waves 0-3 wait for their share at the end of MFMA(u-1), waves 4-7 at
//   the end of EPI(u-2); both then have exactly one younger DMA group outstanding.  The loop contains no ordinary
//   global load (hipcc would drain vmcnt(0) for it); LDS stores inside the loop are inline-asm ds_write_b64 for the
//   same reason, and the A-fragment loads are pinned by a register-use asm before the loop.
//
//   MFMA layout (v_mfma_f32_32x32x16_f16): lane l feeds A[row l&31][k (l>>5)*8..+7] and B[col l&31][same k]; it
//   receives for column l&31 the 16 rows (r&3) + 8 (r>>2) + 4 (l>>5), r = 0..15.  The fp16 blocks are stored
//   [row][granule ^ (row & 15)], so the lane-linear DMA image is bank-conflict-free for the ds_read_b128 operand reads.
//
// PASS 1: accumulator = -S~/2.  Row maxima (v_max3 over running / block 0 / block 1: 0.5 op per element), column
//         maxima (v_max3 chains: 0.5 op per element); the two smallest S~ per row (merged over the 32 lanes at the
//         end) and per column (lane pair merged, the eight waves folded in LDS) are written as partials.
// PASS 2: accumulator = -S~/2; append (q, t) where S~ <= T_row[q] or S~ <= T_col[t]   (dense sweep 2).
// PASS 3: A = compacted live rows, accumulator = -(S~ - T_row)/2; append (k, t) where it is >= 0.
//         PASS 2 / 3 first reduce a block to "any hit?" with v_max3 and only then build the bit mask.
// VMEM LOADS per wave per tile (the counted wait depends on it): PASS 1 / 3: 3 DMA, PASS 2: 4 DMA.
#pragma once
// (included inside namespace msfm)

// Diagnostic build only (tools/gpu_probe.sh, -DMSFM_SWEEP_PROBE): per-wave cycle sums of the four segments of a tile
// (MFMA phase, its wait + barrier, EPI phase, its wait + barrier), printed by the library after every sweep 1.
#ifdef MSFM_SWEEP_PROBE
__device__ unsigned long long g_sweep_probe[kPfWaves][8];
#define MSFM_PROBE_BEGIN unsigned long long pb_t = __builtin_amdgcn_s_memtime(), pb_acc[4] = {0, 0, 0, 0};
#define MSFM_PROBE(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pb_acc[k] += n_ - pb_t; pb_t = n_; }
#define MSFM_PROBE_END                                                                                     \
    if (PASS == 1 && lane == 0) {                                                                          \
        for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&g_sweep_probe[wave][k_], pb_acc[k_]);                    \
        atomicAdd(&g_sweep_probe[wave][4], (unsigned long long)(t_end - t_begin));                         \
    }
#else
#define MSFM_PROBE_BEGIN
#define MSFM_PROBE(k)
#define MSFM_PROBE_END
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }  // folds to v_max3_f32

template <int PASS>
__global__ __launch_bounds__(kPfThreads, 2) void sweep_kernel(
    const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf, const WorkItem* __restrict__ items,
    float* __restrict__ rp_s0, float* __restrict__ rp_s1, float* __restrict__ cp_s0, float* __restrict__ cp_s1,
    const float* __restrict__ tu, const float* __restrict__ tv, int2* __restrict__ cand,
    unsigned long long* __restrict__ cand_count /* 64-bit: n1 * n2 hits of a flooded list do not fit 32 bits */) {
    typedef const __attribute__((address_space(1))) float* gfloat_p;  // keep these loads off the FLAT path
    typedef const __attribute__((address_space(1))) h8* gh8_p;
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    char* sB = pf_smem;
    char* sExt = pf_smem + kPfRing * kPfLdsB;                                          // [wave][slot][64 rows x 16 B]
    float* sThr = reinterpret_cast<float*>(sExt + kPfWaves * kPfRing * kPfExtB);       // [wave][slot][64]
    char* sZero = reinterpret_cast<char*>(sThr + kPfWaves * kPfRing * 64);             // 64 B, first 16 used
    char* sCand = sZero + 64;                                                           // [wave][kPfCandBuf] int2
    char* sCol = sCand + kPfWaves * kPfCandBuf * 8;                                     // [col slot][wave][64] float2

    const WorkItem item = items[blockIdx.x];
    if (item.pair < 0) return;
    const PfPair pp = pf[item.pair];
    if (!pp.use) return;
    const PairDesc pd = pairs[item.pair];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;          // 0: MFMA in the even phases, 1: in the odd ones
    const int lcol = lane & 31, lhalf = lane >> 5;

    // 64-row tiles; the last 128-row block of the B image may hold an all-padding second tile: skip it
    const int t_begin = item.bt_begin * 2, t_end = min(item.bt_end * 2, max(item.bt_begin * 2 + 1, (pd.n2 + kPfBT - 1) / kPfBT));
    const char* gB = reinterpret_cast<const char*>(pp.b_h);
    const char* gE = reinterpret_cast<const char*>(pp.b_ext);
    const gfloat_p g_anrm = (gfloat_p)pp.a_nrm;
    const gfloat_p g_tu = (gfloat_p)tu;
    char* ext_w = sExt + wave * (kPfRing * kPfExtB);   // this wave's private copies
    float* thr_w = sThr + wave * (kPfRing * 64);

    // DMA group of tile tt (clamped: the tail re-fetches the last tile so every EPI phase issues the same number of
    // VMEM loads): this wave's eighth of the 16 KiB tile + its private quadruples / thresholds
    auto dma_tile = [&](int tt) {
        const int tc = tt < t_end ? tt : t_end - 1;
        const int sl = (tt - t_begin) & (kPfRing - 1);
        const char* g = gB + (size_t)tc * kPfLdsB + wave * 2048 + lane * 16;
        char* l = sB + sl * kPfLdsB + wave * 2048;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + k * 1024),
                                             (__attribute__((address_space(3))) void*)(l + k * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gE + (size_t)tc * kPfExtB + lane * 16),
                                         (__attribute__((address_space(3))) void*)(ext_w + sl * kPfExtB), 16, 0, 0);
        if (PASS == 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tv + pp.tv_off + tc * kPfBT + lane),
                                             (__attribute__((address_space(3))) void*)(thr_w + sl * 64), 4, 0, 0);
    };
    constexpr int kDmaOps = (PASS == 2) ? 4 : 3;

    dma_tile(t_begin);
    dma_tile(t_begin + 1);
    dma_tile(t_begin + 2);
    if (tid < 4) reinterpret_cast<float*>(sZero)[tid] = 0.f;

    // A fragments: rows a_blk*512 + wave*64 + rb*32 + lcol, granule 2*ks + lhalf (stored at ^ (row & 15));
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

    // this lane's 32 result rows: (rb, r) -> row = a_blk*512 + wave*64 + rb*32 + (r&3) + 8*(r>>2) + 4*lhalf
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
    // Make hipcc itself wait for the fragment loads here (a register use it can see): otherwise its scoreboard
    // still holds them as pending at the first MFMA and it drains vmcnt(0) INSIDE the loop, which would serialise
    // the DMA ring.
#pragma unroll
    for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) asm volatile("" ::"v"(af[rb][ks]));
    wait_vmcnt<0>();  // prologue loads and the first three DMA groups are done: counted waits start clean

    // sweep 2: wave-private candidate buffer in LDS
    int2* cbuf = reinterpret_cast<int2*>(sCand) + wave * kPfCandBuf;
    const unsigned cbuf_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)cbuf;
    // sweep 1: the column partials of the eight waves meet in LDS ([col slot][wave][64 columns] x (s0, s1)) and are
    // folded by one wave a tile later: one partial per 512-row A block and column reaches HBM
    const unsigned colbuf_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sCol;
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
        // its column partials are never written: park (+inf, +inf) in every slot once
        const float2 pr = make_float2(f_inf(), f_inf());
#pragma unroll
        for (int sl = 0; sl < kPfColRing; ++sl)
            asm volatile("ds_write_b64 %0, %1" ::"v"(colbuf_lds + (unsigned)((sl * kPfWaves + wave) * 64 + lane) * 8u), "v"(pr) : "memory");
    }

    struct BlockMeta { float hc; int col; int cslot; };  // hc: PASS 2 column hit level -T_col/2; cslot: LDS slot of the column partials
    const int zero_off = (int)(sZero - pf_smem);
    // B fragments of the tile in ring slot sl: the 8 data k-steps of column block 0, and the ninth k-step of BOTH
    // column blocks -- the quadruple of column cb*32 + lcol for k = 128..131 in the lhalf == 0 lanes, zeros (k = 132..135)
    // in the others (one base pointer + selected offset: a select between two pointers makes hipcc drain vmcnt(0)).
    // The quadruple only fills k = 0..3 of its k-step: the K = 8 instruction (lane l: k = 4 (l >> 5) .. +3) takes half
    // the passes of a K = 16 one.
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    auto load_bf = [&](int sl, h8 (&bf)[8], h4 (&be)[2]) {
        const char* pb = sB + sl * kPfLdsB + lcol * kHalfRowBytes;
        const int pe_off = (int)(ext_w - pf_smem) + sl * kPfExtB;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bf[ks] = *reinterpret_cast<const h8*>(pb + (((2 * ks + lhalf) ^ xb) << 4));
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            be[cb] = *reinterpret_cast<const h4*>(pf_smem + (lhalf == 0 ? pe_off + (cb * 32 + lcol) * 16 : zero_off));
    };
    auto mfma_block = [&](const h8 (&bf)[8], h4 be, f16v (&acc)[kPfRB]) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb) acc[rb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[rb][ks], bf[ks], acc[rb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb)
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(af[rb][8], af[rb][8], 0, 1, 2, 3), be, acc[rb], 0, 0, 0);
    };
    // Column block 0 of the tile in slot sl on the fragments already in `bf`; as soon as the two MFMAs of a k-step have
    // been issued its registers take the fragment of column block 1 (the read lands 16 MFMAs before it is needed).
    // The sched_group_barriers pin that interleave: left alone, hipcc issues all 18 MFMAs, then the reads, then waits.
    auto mfma_block_reload = [&](int sl, h8 (&bf)[8], h4 be, f16v (&acc)[kPfRB]) {
        const char* pb = sB + sl * kPfLdsB + (32 + lcol) * kHalfRowBytes;
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb) acc[rb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[rb][ks], bf[ks], acc[rb], 0, 0, 0);
            bf[ks] = *reinterpret_cast<const h8*>(pb + (((2 * ks + lhalf) ^ xb) << 4));
        }
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb)
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(af[rb][8], af[rb][8], 0, 1, 2, 3), be, acc[rb], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    };
    // branch-free part of the epilogue of one column block; returns "this lane saw a hit" for the sweep-2 variants
    auto epilogue_valu = [&](const f16v (&acc)[kPfRB], const BlockMeta& bm) -> bool {
        // column maximum of the accumulator over this lane's 32 rows: 16 v_max3
        float m = -f_inf();
        if (PASS != 2) {
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) m = max3f(m, acc[rb][r], acc[rb][r + 1]);
        }
        if (PASS == 1) {
            // Only MAXIMA are tracked: the second smallest of the minima of S~ over disjoint subsets is an upper
            // bound of the true second-smallest S~, which is all the threshold needs (it is exact unless both
            // neighbours fall into one subset).  Partner lane (l ^ 32): the other 32 rows of the wave.
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
    // sweep 2, rare path: the block holds at least one hit -> bit mask per lane (element k = rb*16 + r at bit 31-k),
    // slotted with ballot/popcount into this wave's LDS buffer -- no atomics in the loop -- and flushed to the
    // pair's global list when the buffer fills up
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
    // sweep 1: lane = column of tile tt; fold the eight waves' (s0, s1) and store one partial per A block
    auto merge_columns = [&](int tt) {
        const unsigned base = colbuf_lds + (unsigned)((((tt - t_begin) & (kPfColRing - 1)) * kPfWaves) * 64 + lane) * 8u;
        float2 w[8];
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\t"
                     "ds_read_b64 %3, %8 offset:1536\n\tds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\t"
                     "ds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                     : "v"(base) : "memory");
        v2_merge(w[0].x, w[0].y, w[1].x, w[1].y);
        v2_merge(w[2].x, w[2].y, w[3].x, w[3].y);
        v2_merge(w[4].x, w[4].y, w[5].x, w[5].y);
        v2_merge(w[6].x, w[6].y, w[7].x, w[7].y);
        v2_merge(w[0].x, w[0].y, w[2].x, w[2].y);
        v2_merge(w[4].x, w[4].y, w[6].x, w[6].y);
        v2_merge(w[0].x, w[0].y, w[4].x, w[4].y);
        const long long o = pd.cp_off + (long long)item.a_blk * pd.n2pad + tt * kPfBT + lane;
        cp_s0[o] = w[0].x;
        cp_s1[o] = w[0].y;
    };

    lds_barrier();   // every wave's share of tiles 0..2 (and the zero granule, the parked partials) is in LDS
    // pre-read of the next tile's first fragments at the end of the EPI phase (not in the dense sweep 2: its
    // epilogue keeps the row levels and the thresholds live as well, the 36 registers would spill)
    constexpr bool kPreRead = PASS != 2;
    h8 bf[8];
    h4 be[2];
    if (kPreRead && wave_active) load_bf(0, bf, be);   // first fragments of the first tile
    if (grp == 1) lds_barrier();          // the odd half starts half a tile later

    f16v accA[kPfRB], accB[kPfRB];
    BlockMeta metaA = {0.f, 0, 0}, metaB = {0.f, 0, 0};
    MSFM_PROBE_BEGIN
#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int sl = (t - t_begin) & (kPfRing - 1);
        // ---- MFMA phase: the matrix pipe is this wave's; its SIMD partner is in its EPI phase ----------------
        // A wave whose 64 rows are all padding (tail of an image / of a compacted row set) still takes part in the DMA
        // and the barriers, but leaves the matrix pipe alone
        if (wave_active) {
#ifndef MSFM_SWEEP_NOPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            if (!kPreRead) load_bf(sl, bf, be);
            mfma_block_reload(sl, bf, be[0], accA);
            mfma_block(bf, be[1], accB);
#ifndef MSFM_SWEEP_NOPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        // tile t+1 must be complete one phase before its first MFMA: this half waits here for its share (its DMA
        // group of tile t+2 may stay in flight), the other half at the end of its EPI phase
        MSFM_PROBE(0)
        if (grp == 0) wait_vmcnt<kDmaOps>();
        lds_barrier();
        MSFM_PROBE(1)
        // ---- EPI phase: everything that is not matrix work ---------------------------------------------------
        dma_tile(t + 3);   // into the slot of tile t-1, dead since the barrier before last
        if (wave_active) {
            const float* thr = thr_w + sl * 64;
            metaA.hc = (PASS == 2) ? -0.5f * thr[lcol] : 0.f;
            metaA.col = t * kPfBT + lcol;
            metaA.cslot = (((t - t_begin) & (kPfColRing - 1)) * kPfWaves + wave) * 64;
            metaB.hc = (PASS == 2) ? -0.5f * thr[32 + lcol] : 0.f;
            metaB.col = t * kPfBT + 32 + lcol;
            metaB.cslot = metaA.cslot + 32;
            const bool anyA = epilogue_valu(accA, metaA);
            const bool anyB = epilogue_valu(accB, metaB);
            if (PASS == 1) {
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
        // tile t-1's column partials are complete in LDS (both halves wrote them before the last barrier): one wave of
        // the even half folds them
        if (PASS == 1 && grp == 0 && t > t_begin && wave == ((t - t_begin) & 3)) merge_columns(t - 1);
        if (kPreRead && wave_active && t + 1 < t_end) load_bf((sl + 1) & (kPfRing - 1), bf, be);   // pre-read: the next MFMA phase starts on registers
        MSFM_PROBE(2)
        if (grp == 1) wait_vmcnt<kDmaOps>();
        lds_barrier();
        MSFM_PROBE(3)
    }
    MSFM_PROBE_END
    if (grp == 0) lds_barrier();   // the odd half's last EPI phase
    if (PASS >= 2) flush_candidates();
    if (PASS == 1 && wave == 0) merge_columns(t_end - 1);

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

