// msfm_sweep.hip.h -- the MFMA sweep kernels of the prefilter path (included by msfm_prefilter.hip.h).
//
// sweep_kernel<PASS>: one workgroup = 512 A rows x a range of 64-row B tiles, EIGHT waves = two per SIMD.
//
//   Wave w owns A rows w*64 .. w*64+63 for the whole work item: its fp16 A fragments (two 32-row MFMA blocks x
//   9 k-steps, the ninth holding the norm / threshold quadruple) live in registers, loaded once from HBM.  Only B
//   tiles stream through LDS: a ring of four 17 KiB slots (64 operand rows of 272 B: 128 halfs + the row's norm
//   quadruple) filled by LDS-DMA three tiles ahead, 17 pieces of 1 KiB per tile dealt over the eight waves.
//
//   PING-PONG.  Waves w and w + 4 sit on the same SIMD (a workgroup's waves are dealt to the SIMDs cyclically).  The
//   two halves of the workgroup run the same per-tile program half a tile apart:
//
//        phase      2t                2t+1               2t+2               2t+3
//        waves 0-3  MFMA(t)           EPI(t)+DMA(t+3)    MFMA(t+1)          EPI(t+1)+DMA(t+4)
//        waves 4-7  EPI(t-1)+DMA(t+2) MFMA(t)            EPI(t)+DMA(t+3)    MFMA(t+1)
//
//   with one s_barrier between phases.  MFMA(t): the 36 matrix instructions of a tile (2 column blocks x 2 row blocks x
//   9 k-steps) and the LDS reads of the second column block's fragments (base + immediate offset, no address
//   arithmetic) -- nothing else.  EPI(t): everything that is not matrix work -- the v_max3 epilogues on the 64 results per
//   lane, the column maxima (one LDS atomic per wave and tile), the DMA pieces of tile t+3 and the LDS reads of the NEXT
//   tile's first fragments, so that the following MFMA phase starts on registers.  While one wave of a SIMD feeds the
//   matrix pipe, the other one does its VALU / LDS / VMEM work (MI355X_MICROARCH.md, "Two waves per SIMD").
//
//   Synchronisation: raw s_barrier + COUNTED s_waitcnt vmcnt(N), so the DMA group of the youngest tile stays in flight
//   across barriers.  Tile u is complete in LDS one phase before its first MFMA (so it can be pre-read): waves 0-3 wait
//   for their share at the end of MFMA(u-1), waves 4-7 at the end of EPI(u-2); both then have exactly one younger DMA
//   group outstanding.  (Loads retire in order among themselves; a store may retire out of order, so the count relies
//   on loads only: with N = pieces per group, "at most N outstanding" implies the older group has landed, because one
//   of its loads pending would keep all N younger loads pending as well.)  The loop contains no ordinary global load
//   (hipcc would drain vmcnt(0) for it); the A-fragment loads are pinned by a register-use asm before the loop.
//
//   MFMA layout (v_mfma_f32_32x32x16_f16): lane l feeds A[row l&31][k (l>>5)*8..+7] and B[col l&31][same k]; it
//   receives for column l&31 the 16 rows (r&3) + 8 (r>>2) + 4 (l>>5), r = 0..15.
//
// PASS 1: accumulator = -S~/2.  Rows: running maxima (v_max3 over running / block 0 / block 1: 0.5 op per element),
//         merged over the 32 lanes at the end of the item -> the two smallest S~ per row as one partial per B range.
//         Columns: v_max3 chains over the lane's 32 rows (0.5 op per element), then one ds_max_f32 per column block into
//         the tile's class array [4 classes][64 columns] (class = 2 (wave & 1) + lane half: the maxima over four disjoint
//         quarters of the 512 rows); a rotating wave copies the array of the previous tile to HBM (one float4 per column
//         and A block).  pf_thresholds_kernel takes the minimum and
//         the second smallest of the class minima: an upper bound of the column's second-smallest S~, which is all the
//         threshold needs.
// PASS 2: accumulator = -S~/2; append (q, t) where S~ <= T_row[q] or S~ <= T_col[t]   (dense sweep 2).
// PASS 3: A = compacted live rows, accumulator = -(S~ - T_row)/2; append (k, t) where it is >= 0.
//         PASS 2 / 3 first reduce a block to "any hit?" with v_max3 and only then build the bit mask.
// PASS 4: A = compacted live rows (as PASS 3), accumulator = -S~/2, ROW results only (as PASS 1): the two smallest S~ of
//         every compacted row over the streamed tile range.  The fp16-accurate sweep 1 of the rows an int8-quantised
//         first sweep left alive (msfm_match.hip, route Q).
#pragma once
// (included inside namespace msfm)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Phase boundary.  sched_barrier(0): NOTHING may be scheduled across it, register-only instructions included.  The
// "memory" clobbers only order memory operations; without the scheduling barriers hipcc hoisted the address arithmetic of
// the next phase's DMA (a v_mad_i64_i32 whose carry-out lands in an SGPR pair, followed by an s_add into the same SGPR
// that feeds M0) above the s_barrier, and with that placement single-active-wave workgroups issued DMA pieces to a wrong
// LDS slot (observed on gfx950 / ROCm 7.2: thresholds right, sweep 2 flooded with spurious hits; any scheduling barrier
// at the phase boundaries removes it).  The DMA addressing below also stays in 32-bit offsets from a uniform base so
// that no VALU instruction with an SGPR carry-out is involved.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }  // folds to v_max3_f32

template <int PASS>
__global__ __launch_bounds__(kPfThreads, 2) void sweep_kernel(
    const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf, const WorkItem* __restrict__ items,
    float* __restrict__ rp_s0, float* __restrict__ rp_s1, float* __restrict__ cp_s0, float* __restrict__ cp_s1,
    const float* __restrict__ tu, const float* __restrict__ tv, int2* __restrict__ cand,
    unsigned long long* __restrict__ cand_count /* 64-bit: n1 * n2 hits of a flooded list do not fit 32 bits */,
    const int* __restrict__ n_items_dev /* item count in device memory (a plan built on the device), or null */, int n_items_host,
    int* __restrict__ dyn_next /* 8 per-XCD cursors: items are fetched dynamically (sweep 2: unequal items), or null */) {
    typedef const __attribute__((address_space(1))) float* gfloat_p;  // keep these loads off the FLAT path
    typedef const __attribute__((address_space(1))) h8* gh8_p;
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    char* sB = pf_smem;                                                                // [ring slot][64 rows x 272 B]
    float* sThr = reinterpret_cast<float*>(pf_smem + kPfRing * kPfTileBytes);          // [wave][slot][64] (PASS 2)
    char* sCand = reinterpret_cast<char*>(sThr + kPfWaves * kPfRing * 64);             // [wave][kPfCandBuf] int2 (PASS 2 / 3)
    float* sCol = reinterpret_cast<float*>(sCand + kPfWaves * kPfCandBuf * 8);         // [2 tiles][64 columns][4 classes] (PASS 1)

    // Persistent workgroups: one per CU (the LDS ring allows no more), striding over the item list.  The list is
    // XCD-interleaved (item i belongs to XCD i % 8) and the grid is a multiple of 8, so a workgroup keeps streaming the B
    // panels its XCD's L2 already holds.
    // Uniform items (sweep 1) are taken in a fixed stride; the unequal items of the compacted sweep 2 (79-tile forward
    // sweeps next to 8-tile reverse ones) are fetched from a per-XCD cursor, longest first.
    const int n_items = n_items_dev ? *n_items_dev : n_items_host;
    __shared__ int s_next_item;
    bool first_item = true;
#pragma unroll 1
    for (int it = blockIdx.x;; it += gridDim.x) {
    if (!first_item || dyn_next) lds_barrier();   // the previous item's last LDS accesses (class arrays, the cursor) are done
    first_item = false;
    if (dyn_next) {
        if (threadIdx.x == 0) s_next_item = atomicAdd(&dyn_next[blockIdx.x & 7], 1);
        lds_barrier();
        it = s_next_item * 8 + (int)(blockIdx.x & 7);
    }
    if (it >= n_items) break;
    const WorkItem item = items[it];
    if (item.pair < 0) continue;
    const PfPair pp = pf[item.pair];
    if (!pp.use) continue;
    const PairDesc pd = pairs[item.pair];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;          // 0: MFMA in the even phases, 1: in the odd ones
    const int lcol = lane & 31, lhalf = lane >> 5;

    // 64-row tiles; the last 128-row block of the B image may hold an all-padding second tile: skip it
    const int t_begin = item.bt_begin * 2, t_end = min(item.bt_end * 2, max(item.bt_begin * 2 + 1, (pd.n2 + kPfBT - 1) / kPfBT));
    const char* gB = reinterpret_cast<const char*>(pp.b_h);
    const gfloat_p g_anrm = (gfloat_p)pp.a_nrm;
    const gfloat_p g_tu = (gfloat_p)tu;
    float* thr_w = sThr + wave * (kPfRing * 64);

    // DMA group of tile tt (clamped: the tail re-fetches the last tile so every EPI phase issues the same number of
    // VMEM loads): pieces w and w + 8 of the tile's 17 (wave 0: piece 16 as well) + in PASS 2 the wave's private copy of
    // the tile's column thresholds
    const unsigned lane_off = (unsigned)(wave * 1024 + lane * 16);   // 32-bit offsets from the uniform image base
    auto dma_tile = [&](int tt) {
        const int tc = tt < t_end ? tt : t_end - 1;
        const int sl = (tt - t_begin) & (kPfRing - 1);
        const unsigned off = (unsigned)tc * (unsigned)kPfTileBytes + lane_off;   // < 2^31: an image has < 2^18 rows of 272 B
        char* l = sB + sl * kPfTileBytes + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + off), (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + (off + 8192u)),
                                         (__attribute__((address_space(3))) void*)(l + 8192), 16, 0, 0);
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + (off + 16384u)),
                                             (__attribute__((address_space(3))) void*)(l + 16384), 16, 0, 0);
        if (PASS == 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tv + pp.tv_off + tc * kPfBT + lane),
                                             (__attribute__((address_space(3))) void*)(thr_w + sl * 64), 4, 0, 0);
    };
    // loads per group: the counted waits below must name the issuing wave's own number
    constexpr int kDmaOps = (PASS == 2) ? 3 : 2;   // waves 1..7; wave 0: one more
    auto wait_older_group = [&]() {
        if (wave == 0) wait_vmcnt<kDmaOps + 1>();
        else wait_vmcnt<kDmaOps>();
    };

    dma_tile(t_begin);
    dma_tile(t_begin + 1);
    dma_tile(t_begin + 2);
    if (PASS == 1) sCol[tid] = -f_inf();   // 2 tiles x 4 classes x 64 columns = 512 floats

    // A fragments: rows a_blk*512 + wave*64 + rb*32 + lcol, granule 2*ks + lhalf;
    // ninth k-step: [-c, -c, x_hi, x_lo, 0...] in the lhalf == 0 lanes (k = 128..135), zeros in the others
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h8 af[kPfRB][8];
    h4 ae[kPfRB];     // the ninth k-step's A side: k = 128..131 (lhalf == 0 lanes), two registers -- an h8 would cost four more
                      // VGPRs, and at 226 the kernel leaves 48 of a SIMD's 512 to the tail kernels of the other stream, at <= 224 64
    const float inv_c = 1.f / pp.b_c;
#pragma unroll
    for (int rb = 0; rb < kPfRB; ++rb) {
        const int frow = item.a_blk * kPfWgRows + wave * kPfWaveRows + rb * 32 + lcol;
        // 272-byte rows: 17 aligned granules.  Compacted sweep: the A rows are live rows of many images, named by a
        // pointer table (a row without a source -- the tail of a group -- reads the image-independent zero row)
        const _Float16* arow = pp.a_h + (size_t)frow * kPfRowHalfs;
        if (PASS >= 3) {
            const _Float16* r = pp.a_rows[frow];
            arow = r ? r : pp.a_h;
        }
        const gh8_p ga = (gh8_p)arow;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) af[rb][ks] = ga[2 * ks + lhalf];
        // padding rows: X = -inf -> accumulator -inf, never a maximum, never a hit
        float X;
        if (PASS >= 3) X = frow < pd.n1 ? 0.5f * g_tu[pp.tu_off + frow] : -f_inf();   // (PASS 4: the table holds -|a|^2)
        else X = frow < pd.n1 ? -0.5f * g_anrm[frow] : -f_inf();
        const float xs = X * inv_c;
        const _Float16 hi = (_Float16)xs;
        const float rest = xs - (float)hi;
        const _Float16 lo = (rest == rest && fabsf(rest) < 3.0e38f) ? (_Float16)rest : (_Float16)0.f;
        h4 e;
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (_Float16)0.f;
        if (lhalf == 0) {
            e[0] = (_Float16)(-pp.b_c);
            e[1] = (_Float16)(-pp.b_c);
            e[2] = hi;
            e[3] = lo;
        }
        ae[rb] = e;
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
            if (PASS == 2) {
                const float lvl = -0.5f * g_tu[pp.tu_off + rr];   // unconditional: the threshold array covers the padded rows
                rs0[rb][r] = rr < pd.n1 ? lvl : f_inf();
            } else {
                rs0[rb][r] = -f_inf();
            }
        }
    // Make hipcc itself wait for the fragment loads here (a register use it can see): otherwise its scoreboard
    // still holds them as pending at the first MFMA and it drains vmcnt(0) INSIDE the loop, which would serialise
    // the DMA ring.
#pragma unroll
    for (int rb = 0; rb < kPfRB; ++rb) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" ::"v"(af[rb][ks]));
        asm volatile("" ::"v"(ae[rb]));
    }
    wait_vmcnt<0>();  // prologue loads and the first three DMA groups are done: counted waits start clean

    // sweep 2: wave-private candidate buffer in LDS
    int2* cbuf = reinterpret_cast<int2*>(sCand) + wave * kPfCandBuf;
    const unsigned cbuf_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)cbuf;
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

    // (wave-uniform; readfirstlane keeps it in a scalar register -- as a lane mask it costs two VALU instructions per tile)
    const bool wave_active = __builtin_amdgcn_readfirstlane((int)(item.a_blk * kPfWgRows + wave * kPfWaveRows < pd.n1)) != 0;

    // B fragments of the tile in ring slot sl: the 8 data k-steps of column block 0 (row lcol of the slot, granule
    // 2 ks + lhalf: a constant offset from the lane's row address), and the ninth k-step of BOTH column blocks -- the first
    // 8 bytes of the row's 17th granule ([h_hi, h_lo, c, c] = k 128..131) in the lhalf == 0 lanes, its zero half
    // (k 132..135) in the others.  The quadruple only fills k = 0..3 of its k-step, so the K = 8 instruction (lane l:
    // k = 4 (l >> 5) .. +3) carries it with half the operand registers; it occupies the pipe as long as a K = 16 one
    // (32 cycles, tools/ubench_clock.hip), i.e. the norm k-step is 4 of a tile's 36 MFMA slots.
    const int lane_row_off = lcol * kPfRowBytes + lhalf * 16;
    const int lane_ext_off = lcol * kPfRowBytes + 2 * kDim + lhalf * 8;
    auto load_bf = [&](int sl, h8 (&bf)[8], h4 (&be)[2]) {
        const char* pb = sB + sl * kPfTileBytes + lane_row_off;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bf[ks] = *reinterpret_cast<const h8*>(pb + ks * 32);
        const char* pe = sB + sl * kPfTileBytes + lane_ext_off;
        be[0] = *reinterpret_cast<const h4*>(pe);
        be[1] = *reinterpret_cast<const h4*>(pe + 32 * kPfRowBytes);
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
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x8f16(ae[rb], be, acc[rb], 0, 0, 0);
    };
    // Column block 0 of the tile in slot sl on the fragments already in `bf`; as soon as the two MFMAs of a k-step have
    // been issued its registers take the fragment of column block 1 (the read lands 16 MFMAs before it is needed).
    // The sched_group_barriers pin that interleave: left alone, hipcc issues all 18 MFMAs, then the reads, then waits.
    auto mfma_block_reload = [&](int sl, h8 (&bf)[8], h4 be, f16v (&acc)[kPfRB]) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb) acc[rb][r] = 0.f;
        const char* pb = sB + sl * kPfTileBytes + 32 * kPfRowBytes + lane_row_off;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[rb][ks], bf[ks], acc[rb], 0, 0, 0);
            bf[ks] = *reinterpret_cast<const h8*>(pb + ks * 32);
        }
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb)
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x8f16(ae[rb], be, acc[rb], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    };
    // maximum of the accumulator over this lane's 32 rows of one column block: two independent v_max3 chains.  The odd
    // element out goes through a raw v_max_f32: fmaxf() would first canonicalise both operands (v_max_f32 x, x, x), three
    // extra VALU instructions per column block -- and every VALU instruction is SIMD time here.
    auto column_max = [&](const f16v (&acc)[kPfRB]) -> float {
        static_assert(kPfRB == 2, "two chains");
        float m[kPfRB];
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb) {
            m[rb] = max3f(acc[rb][0], acc[rb][1], acc[rb][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) m[rb] = max3f(m[rb], acc[rb][r], acc[rb][r + 1]);
        }
        float last;
        asm("v_max_f32 %0, %1, %2" : "=v"(last) : "v"(acc[0][15]), "v"(acc[1][15]));
        return max3f(m[0], m[1], last);
    };
    // ---- sweep 2: recording hits ---------------------------------------------------------------------------------
    // A hit goes, slotted with ballot / popcount, into this wave's LDS buffer -- no atomics in the loop -- and the buffer
    // is flushed to the list in HBM when it fills up.  (Inline asm store: hipcc would drain vmcnt(0) for an LDS store
    // it can see.)
    auto record_hits = [&](bool hit, int row, int col) {
        const unsigned long long mm = __ballot(hit);
        if (mm == 0ull) return;
        if (n_buf + 64 > kPfCandBuf) flush_candidates();
        if (hit) {
            const int slt = n_buf + __popcll(mm & ((1ull << lane) - 1ull));
            const int2 e = make_int2(row, col);
            asm volatile("ds_write_b64 %0, %1" ::"v"(cbuf_lds + slt * 8), "v"(e) : "memory");
        }
        n_buf += __popcll(mm);
    };
    // Compacted sweep (PASS 3): a hit is accumulator >= 0, i.e. sign bit clear (the accumulator starts at +0 and a sum
    // that cancels exactly rounds to +0: -0 cannot occur; NaN cannot either on fp16-safe operands).  With every A row
    // alive a 64 x 32 block holds one or two hits, so there is no "nothing here" shortcut worth a branch: every lane
    // shifts the 32 sign bits of a column block into one mask, ONE v_alignbit per element (element k = rb*16 + r ends up
    // at bit 31 - k), and the hits are slotted with ballot / popcount into the wave's LDS buffer, one hit per lane and
    // round (usually one or two rounds).  Measured alternatives, cycles of the epilogue phase per tile against the 1300
    // of the partner wave's MFMA phase it should hide behind: compare + select mask 2150, a ballot + branch per register
    // quad 2600, scalar bookkeeping with v_readlane 2200 (every VALU -> SGPR -> branch round trip costs ~50 cycles).
    auto scan_hits3 = [&](const f16v (&a0)[kPfRB], const f16v (&a1)[kPfRB], int col0) {
        unsigned nm[2] = {0u, 0u};   // bit set: sign set, no hit
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) nm[blk] = __builtin_amdgcn_alignbit(nm[blk], __float_as_uint((blk ? a1 : a0)[rb][r]), 31);
        static_assert(kPfRB == 2, "32 elements per column block and mask");
#pragma unroll 1
        for (int blk = 0; blk < 2; ++blk) {   // a runtime loop: the flush code below exists once
            unsigned hm = ~(blk ? nm[1] : nm[0]);
            const int col = col0 + blk * 32;
            for (unsigned long long mm = __ballot(hm != 0u); mm != 0ull; mm = __ballot(hm != 0u)) {   // one ballot per round
                const bool hit = hm != 0u;
                const int k = __clz((int)hm);   // first remaining element of this lane
                if (n_buf + 64 > kPfCandBuf) flush_candidates();
                if (hit) {
                    hm &= ~(0x80000000u >> k);
                    const int slt = n_buf + __popcll(mm & ((1ull << lane) - 1ull));
                    const int2 e = make_int2(arow_base + (k >> 4) * 32 + (k & 3) + 8 * ((k & 15) >> 2), col);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(cbuf_lds + slt * 8), "v"(e) : "memory");
                }
                n_buf += __popcll(mm);
            }
        }
    };
    // Dense sweep (PASS 2, the kNN-level API): row criterion acc >= level_row, column criterion acc >= level_col; a block
    // is first reduced to "any hit?" with v_max3, then every element is tested.  Padding rows / columns hold -inf: with
    // an infinite threshold (fewer than two real elements in a subset) the hit level is -inf as well, and -inf >= -inf
    // must not count.
    auto scan_hits2 = [&](const f16v (&acc)[kPfRB], float hc, int col) {
        float mr = -f_inf(), mc = -f_inf();
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                mr = max3f(mr, acc[rb][r] - rs0[rb][r], acc[rb][r + 1] - rs0[rb][r + 1]);
                mc = max3f(mc, acc[rb][r], acc[rb][r + 1]);
            }
        if (__ballot(mr >= 0.f || mc >= hc) == 0ull) return;
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                record_hits(acc[rb][r] > -f_inf() && (acc[rb][r] >= rs0[rb][r] || acc[rb][r] >= hc),
                            arow_base + rb * 32 + (r & 3) + 8 * (r >> 2), col);
    };
    // sweep 1, columns.  Lane l holds the maximum of column l & 31 over the 32 rows of its lane half, per column block.
    // It goes, with one LDS atomic per column block, into the tile's class array [4 classes][64 columns], class =
    // 2 (wave & 1) + lane half: the maxima over four disjoint quarters of the block's 512 rows (even / odd waves x the two
    // row interleaves of the MFMA layout -- an image of a single wave still fills two classes).  Inline asm keeps hipcc
    // from draining vmcnt(0) for an LDS access it would see.
    const unsigned col_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sCol +
                             (unsigned)((2 * (wave & 1) + lhalf) * kPfBT + lcol) * 4u;
    auto fold_columns = [&](float mA, float mB, int cs) {
        const unsigned a = col_lds + (unsigned)cs * (kPfBT * kPfColClasses * 4);
        asm volatile("ds_max_f32 %0, %1\n\tds_max_f32 %0, %2 offset:128" ::"v"(a), "v"(mA), "v"(mB) : "memory");
    };
    // sweep 1: lane = column of tile tt: copy its four class maxima to HBM (one float4 per column and A block) and
    // reset them for tile tt + 2
    const unsigned colrow_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sCol + (unsigned)lane * 4u;
    auto store_columns = [&](int tt) {
        const unsigned a = colrow_lds + (unsigned)((tt - t_begin) & 1) * (kPfBT * kPfColClasses * 4);
        v4f v;
        const float ninf = -f_inf();
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "ds_write_b32 %4, %5\n\tds_write_b32 %4, %5 offset:256\n\tds_write_b32 %4, %5 offset:512\n\tds_write_b32 %4, %5 offset:768"
                     : "=&v"(v.x), "=&v"(v.y), "=&v"(v.z), "=&v"(v.w) : "v"(a), "v"(ninf) : "memory");
        // only the two largest of the four class maxima travel (the thresholds need the block's smallest S~ and an upper
        // bound of its second smallest): 8 B per column and A block
        const float m01 = fmaxf(v.x, v.y), n01 = fminf(v.x, v.y), m23 = fmaxf(v.z, v.w), n23 = fminf(v.z, v.w);
        v2f o;
        o.x = fmaxf(m01, m23);
        o.y = fmaxf(fminf(m01, m23), fmaxf(n01, n23));
        reinterpret_cast<v2f*>(cp_s0)[pd.cp_off + (long long)item.a_blk * pd.n2pad + tt * kPfBT + lane] = o;
    };

    lds_barrier();   // every wave's share of tiles 0..2 (and the column class arrays) is in LDS
    // pre-read of the next tile's first fragments at the end of the EPI phase (not in the dense sweep 2: its
    // epilogue keeps the row levels and the thresholds live as well, the 36 registers would spill)
    constexpr bool kPreRead = PASS != 2;
    h8 bf[8];
    h4 be[2];
    if (kPreRead && wave_active) load_bf(0, bf, be);   // first fragments of the first tile
    if (grp == 1) lds_barrier();                         // the odd half starts half a tile later

    f16v accA[kPfRB], accB[kPfRB];
#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int sl = (t - t_begin) & (kPfRing - 1);
        // ---- MFMA phase: the matrix pipe is this wave's; its SIMD partner is in its EPI phase ----------------
        // A wave whose 64 rows are all padding (tail of an image / of a compacted row set) still takes part in the DMA
        // and the barriers, but leaves the matrix pipe alone
        if (wave_active) {
            __builtin_amdgcn_s_setprio(1);
            if (!kPreRead) load_bf(sl, bf, be);
            mfma_block_reload(sl, bf, be[0], accA);
            mfma_block(bf, be[1], accB);
            __builtin_amdgcn_s_setprio(0);
        }
        // tile t+1 must be complete one phase before its first MFMA: this half waits here for its share (its DMA
        // group of tile t+2 may stay in flight), the other half at the end of its EPI phase
        if (grp == 0) wait_older_group();
        lds_barrier();
        // ---- EPI phase: everything that is not matrix work ---------------------------------------------------
        // (the store comes before the DMA group: the counted wait then covers it together with the older group)
        if (PASS == 1 && grp == 0 && t > t_begin && wave == ((t - t_begin) & 3)) store_columns(t - 1);
        dma_tile(t + 3);   // into the slot of tile t-1, dead since the barrier before last
        if (wave_active) {
            if (PASS == 1 || PASS == 4) {
                if (PASS == 1) fold_columns(column_max(accA), column_max(accB), (t - t_begin) & 1);   // block 1: columns 32..63 = +128 B
#pragma unroll
                for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rs0[rb][r] = max3f(rs0[rb][r], accA[rb][r], accB[rb][r]);
            } else if (PASS == 3) {
                scan_hits3(accA, accB, t * kPfBT + lcol);
            } else {
                const float* thr = thr_w + sl * 64;   // column hit levels -T_col / 2
                scan_hits2(accA, -0.5f * thr[lcol], t * kPfBT + lcol);
                scan_hits2(accB, -0.5f * thr[32 + lcol], t * kPfBT + 32 + lcol);
            }
        }
        if (kPreRead && wave_active && t + 1 < t_end) load_bf((sl + 1) & (kPfRing - 1), bf, be);   // the next MFMA phase starts on registers
        if (grp == 1) wait_older_group();
        lds_barrier();
    }
    if (grp == 0) lds_barrier();   // the odd half's last EPI phase
    if (PASS == 2 || PASS == 3) flush_candidates();
    if (PASS == 1 && wave == 0) store_columns(t_end - 1);

    if (PASS == 1 || PASS == 4) {
        wait_vmcnt<0>();   // the tail's DMA groups (re-fetches of the last tile) still write into the ring ...
        lds_barrier();     // ... everybody's have landed: the ring is scratch now
        // rows: lane (lcol, lhalf) holds, for each of its 32 rows, the accumulator maximum over ITS columns; a row's result is
        // the two largest of its 32 lane values (S~ = -2 acc: the smallest S~ and an upper bound of the second smallest).
        // Transposed through the wave's share of the (now idle) B ring -- [64 rows][32 lanes + 1] floats -- so that lane l
        // reduces row l of the wave serially: 32 writes, 32 reads and ~100 VALU per lane instead of 32 x 5 butterfly steps
        // (~1000 shuffles + VALU).  LDS operations of one wave execute in order: no barrier.
        float* tr = reinterpret_cast<float*>(sB) + wave * (kPfWaveRows * 33);
        static_assert(kPfWaves * kPfWaveRows * 33 * 4 <= kPfRing * kPfTileBytes, "the transposition fits the ring");
#pragma unroll
        for (int rb = 0; rb < kPfRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 33 + lcol] = rs0[rb][r];
        float m0 = -f_inf(), m1 = -f_inf();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float v = tr[lane * 33 + k];
            m1 = fmaxf(m1, fminf(m0, v));
            m0 = fmaxf(m0, v);
        }
        const long long o = pd.rp_off + (long long)item.range * pd.n1pad + item.a_blk * kPfWgRows + wave * kPfWaveRows + lane;
        rp_s0[o] = -2.f * m0;      // padding rows: -inf -> +inf
        rp_s1[o] = -2.f * m1;
    }
    }   // item loop
}
