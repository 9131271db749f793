// msfm_sweep_i8.hip.h -- the sweeps of the prefilter path on the INTEGER matrix cores, for stores whose descriptors are
// bytes (`descriptors_u8` side table / MSFM_DTYPE_U8 uploads / float uploads holding only integers 0..255) and, sweep 1 only,
// for the byte twins of float stores (msfm_q8.hip.h).  Included by msfm_prefilter.hip.h after msfm_sweep.hip.h.
//
// A byte descriptor x (0..255) is stored as the signed byte x' = x - 128 (= x ^ 0x80): the distance is shift invariant,
// |a - b|^2 = |a' - b'|^2 = n'_a + n'_b - 2 a'.b' with n' = |x'|^2 <= 2^21, and v_mfma_i32_32x32x32_i8 computes a'.b'
// EXACTLY -- twice the k-depth of the fp16 instruction in (nearly) the same pipe time.  With h = floor(n'/2) the accumulator
//
//        acc = a'.b' - h_a - h_b        ->   S~ = -2 acc = S - (n'_a & 1) - (n'_b & 1),   |S~ - S| <= 2 =: eps
//
// is built WITHOUT a VALU instruction (a VALU instruction beside busy matrix pipes costs ~8 ns of SIMD time, DESIGN.md 5.1.3:
// profiles/r03_ubench_coissue.txt): both norms ride in a
// FIFTH k-step of 32 slots.  Every row carries, behind its 128 operand bytes, 16 signed DIGITS d of V = H0 - h (H0 = its
// image's centre of h) and the 16 CONSTANTS c = [1, -128 x 15]:  sum c_k d_k = d_0 - 128 (d_1 + ... + d_15) represents
// every integer in [-243 968, 245 759] -- an image whose h values spread further than that around their centre (they
// cannot all be SIFT descriptors) is not a byte image for this path and takes the fp16 kernels.
//   * slots 0..15:  A side the constants, B side the streamed row's digits            ->  + (H0_b - h_b);
//   * slots 16..31: sweep 1: A side the A ROW'S OWN digits, B side the constants      ->  + (H0_a - h_a)
//     so the accumulator is  a'.b' - h_a - h_b + K,  K = H0_a + H0_b  uniform over the work item: the row / column maxima
//     are taken on it as they are and K is subtracted where they leave the kernel.  No register holds a row constant:
//     112 VGPRs instead of 127 -- four waves per SIMD then leave 64 of the 512 registers to the tail kernels of the
//     OTHER stream's sub-batch, which run beside this sweep instead of behind it;
//   * compacted sweep 2: slots 16..31 of the A side are zero and the row's hit level floor((T_row - 2 h_a) / 2) - H0_b (a
//     hit is acc >= 0  <=>  S~ <= T_row; too wide for 16 digits) is the C OPERAND of the tile's first MFMA, as before.
// Rows are 176 B = 11 granules (128 operand bytes, 16 digits, 16 constants, 16 B of padding: the same odd-granule bank
// swizzle as the fp16 rows); the float "norm" array of the path holds 2h.  Padding COLUMNS carry zero digits, i.e. they look like a
// real column: the image's last tile masks them in its epilogue; padding ROWS (zero digits as well) exist in one wave of
// an image's last 512-row block only, which masks their accumulators the same way.
// Everything downstream (thresholds, plan, exact re-check in the pinned fp32 order, reduce) is the float pipeline
// unchanged: the results leave this kernel as floats (exact: |acc| < 2^23).  On byte data the pinned fp32 order is itself
// exact integer arithmetic (every partial sum is an integer below 2^24; oracle/int_oracle.py pins that), so eps = 2
// instead of ~1.5e-3 (n_a + n_b) ~ 1000: fewer candidates, and half the matrix time.
//
// Same ingredients as sweep_kernel (msfm_sweep.hip.h): matrix halves and epilogue halves of different waves of a SIMD side by
// side, LDS ring filled by LDS-DMA, counted vmcnt waits, persistent workgroups -- but ONE workgroup barrier per tile (round 3;
// the tile loop below says how the two groups of waves are skewed around it).  PASS 1 and PASS 3 only: the dense sweep 2
// (kNN-level API, ratio > 0.95) stays on the fp16 kernel.
#pragma once
// (included inside namespace msfm)

constexpr int kI8RowBytes = 176;                        // 128 operand bytes + 16 digits of H0 - h + the 16 constants + 16 B padding
constexpr int kI8TileBytes = kPfBT * kI8RowBytes;       // 11264 B = 11 DMA pieces
constexpr int kI8DigitLo = -243968, kI8DigitHi = 245759;   // representable H0 - h (see st_digits, msfm_store.hip.h)
// SIXTEEN waves of 32 rows (four per SIMD): on the integer cores a tile's matrix phase is 16 x 34 cycles per SIMD, and one
// wave issues a VALU instruction every 8 cycles at best -- with two 64-row waves per SIMD the ~130 VALU instructions per
// tile and wave set the pace (profiles/r02_i8_sweep_ablation.txt); four 32-row waves give the epilogue twice the issue slots
// per matrix slot.  Waves w, w+4, w+8, w+12 share a SIMD; the two groups of the tile loop are (w >> 2) & 1.
constexpr int kI8Waves = 16;
constexpr int kI8Threads = 64 * kI8Waves;
constexpr int kI8WaveRows = kPfWgRows / kI8Waves;       // 32: one MFMA row block per wave
// Ring of EIGHT 11-KiB tiles, DMA seven tiles ahead: a tile takes ~1 us here, so the three-tile lead of the fp16
// kernel (1.6 us per tile there) would leave little more than an L2 miss.
constexpr int kI8Ring = 8;
// ring | per-wave candidate buffers (q, t) (PASS 3) | column class maxima (PASS 1) | per-wave accumulator values of the candidates (PASS 3)
constexpr int kI8LdsBytes = kI8Ring * kI8TileBytes + kI8Waves * kPfCandBuf * 8 + 4 * kPfBT * kPfColClasses * 4;
constexpr int kI8LdsBytes3 = kI8LdsBytes + kI8Waves * kPfCandBuf * 4;
static_assert(kI8WaveRows == 32, "one 32-row block per wave");
constexpr int kI8Pad = -(1 << 29);                      // "-inf" of a padding row / column (two of them still fit an int32)
constexpr int kI8PadTest = -(1 << 27);                  // anything below is padding
constexpr float kI8Eps = 2.f;
static_assert(kI8TileBytes % 1024 == 0 && kI8TileBytes / 1024 == 11, "tile = 11 DMA pieces");

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }  // folds to v_max3_i32

// Column partials of the integer sweeps: one packed 32-bit word (msfm_hostutil.h: msfm_cp_pack) instead of the float pipeline's float2
__device__ __forceinline__ float2 i8_cp_unpack(int code) {   // -> the float pipeline's (largest, second largest), -inf for none
    const int hi = msfm_cp_hi(code);
    if (hi == kCpNone) return make_float2(-f_inf(), -f_inf());
    return make_float2((float)hi, msfm_cp_has_second(code) ? (float)(hi - msfm_cp_gap(code)) : -f_inf());
}

template <int PASS>
__global__ __launch_bounds__(kI8Threads) void sweep_i8_kernel(
    const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf, const WorkItem* __restrict__ items,
    float* __restrict__ rp_s0, float* __restrict__ rp_s1, float* __restrict__ cp_s0, const float* __restrict__ tu,
    int2* __restrict__ cand, unsigned long long* __restrict__ cand_count, const int* __restrict__ n_items_dev, int n_items_host,
    int* __restrict__ dyn_next, int* __restrict__ cand_val /* PASS 3: the accumulator of every candidate, parallel to `cand` */) {
    static_assert(PASS == 1 || PASS == 3, "sweep 1 and the compacted sweep 2");
    typedef const __attribute__((address_space(1))) float* gfloat_p;
    typedef const __attribute__((address_space(1))) i4v* gi4_p;
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    char* sB = pf_smem;                                                   // [ring slot][64 rows x 176 B]
    char* sCand = pf_smem + kI8Ring * kI8TileBytes;                       // [wave][kPfCandBuf] int2 (PASS 3)
    int* sCol = reinterpret_cast<int*>(sCand + kI8Waves * kPfCandBuf * 8);  // [4 tiles][4 classes][64 columns] (PASS 1)
    int* sVal = sCol + 4 * kPfBT * kPfColClasses;                           // [wave][kPfCandBuf] (PASS 3; only kI8LdsBytes3 launches have it)

    const int n_items = n_items_dev ? *n_items_dev : n_items_host;
    __shared__ int s_next_item;
    bool first_item = true;
#pragma unroll 1
    for (int it = blockIdx.x;; it += gridDim.x) {
    if (!first_item || dyn_next) lds_barrier();
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
    const int grp = (wave >> 2) & 1;    // 0: MFMA in the even phases, 1: in the odd ones; two waves of each on every SIMD
    const int lcol = lane & 31, lhalf = lane >> 5;

    const int t_begin = item.bt_begin * 2, t_end = min(item.bt_end * 2, max(item.bt_begin * 2 + 1, (pd.n2 + kPfBT - 1) / kPfBT));
    const char* gB = reinterpret_cast<const char*>(pp.b_h);
    const gfloat_p g_tu = (gfloat_p)tu;

    // DMA group of tile tt: the tile's 11 pieces go to waves 0..10, one each
    const bool dma_wave = wave < kI8TileBytes / 1024;   // wave-uniform
    const unsigned lane_off = (unsigned)(wave * 1024 + lane * 16);
    auto dma_tile = [&](int tt) {
        if (!dma_wave) return;
        const int tc = tt < t_end ? tt : t_end - 1;
        const int sl = (tt - t_begin) & (kI8Ring - 1);
        // the tile's offset goes into the SCALAR base, the lane's 32-bit offset is the same register for the whole item: no VALU
        // instruction per tile (hipcc's own lowering of the builtin adds the two in a VGPR pair).  M0 = the wave's LDS destination.
        const unsigned long long tile = (unsigned long long)gB + (unsigned long long)(unsigned)tc * (unsigned long long)kI8TileBytes;
        const unsigned dst = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)(sB + sl * kI8TileBytes + wave * 1024);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // (m0 is a reserved register: naming it as a clobber is the point)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(lane_off), "s"(tile), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
    };
    // One load per group and DMA wave.  At either wait the wave has issued the groups up to tile u + kI8Ring - 3 and needs
    // tile u: "at most kI8Ring - 3 outstanding" = tile u has landed (loads retire in order; see sweep_kernel).
    auto wait_older_group = [&]() {
        if (dma_wave) wait_vmcnt<kI8Ring - 3>();
    };

#pragma unroll
    for (int k = 0; k < kI8Ring - 1; ++k) dma_tile(t_begin + k);
    if (PASS == 1 && tid < 4 * kPfBT * kPfColClasses) sCol[tid] = (int)0x80000000;

    // A fragments: row a_blk*512 + wave*32 + lcol, k-step ks = bytes 32 ks + 16 lhalf .. + 15
    i4v af[4], a_digit;
    {
        const int frow = item.a_blk * kPfWgRows + wave * kI8WaveRows + lcol;
        const char* arow = reinterpret_cast<const char*>(pp.a_h) + (size_t)frow * kI8RowBytes;
        if (PASS == 3) {
            const char* r = reinterpret_cast<const char*>(pp.a_rows[frow]);
            arow = r ? r : reinterpret_cast<const char*>(pp.a_h);   // no source: the zero row
        }
        const gi4_p ga = (gi4_p)arow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[ks] = ga[2 * ks + lhalf];
        // the A side of the digit k-step (this lane: slots 16 lhalf .. + 15): the constants in the lower half; in the upper
        // half the row's own digits (sweep 1) or zeros (sweep 2: its rows keep their hit level in the C operand)
        const i4v cdig = {(int)0x80808001u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
        if (PASS == 1) {
            a_digit = ga[8];
            if (lhalf == 0) a_digit = cdig;
        } else {
            a_digit = lhalf == 0 ? cdig : i4v(0);
        }
    }
    // this lane's 16 result rows: r -> row = a_blk*512 + wave*32 + (r&3) + 8*(r>>2) + 4*lhalf; their constants
    const int arow_base = item.a_blk * kPfWgRows + wave * kI8WaveRows + 4 * lhalf;
    // PASS 3: the lane's 16 row hit levels floor((T - 2 h_a) / 2) - H0_b (T = +inf -> everything hits) = the C operand of the
    // tile's first MFMA.  The loads are unconditional (the array covers the padded rows) and pinned by a register use: hipcc
    // otherwise sinks each of them into its `rr < n1` branch and waits for it there -- 16 round trips instead of 16 loads in
    // flight.  PASS 1 has no row constants (header): its C operand is zero.
    i16v rowc = i16v(0);
    if (PASS == 3) {
        float xs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = arow_base + (r & 3) + 8 * (r >> 2);
            xs[r] = g_tu[pp.tu_off + rr];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(xs[r]));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = arow_base + (r & 3) + 8 * (r >> 2);
            rowc[r] = rr < pd.n1 ? (int)floorf(fminf(fmaxf(0.5f * xs[r], -5.0e8f), 5.0e8f)) - pp.b_h0 : kI8Pad;
        }
    }
    const int item_k = PASS == 1 ? pp.a_h0 + pp.b_h0 : 0;   // sweep 1: every accumulator of the item is K = H0_a + H0_b too large
    // sweep 1: the one wave of an image's last block that holds real AND padding rows masks the padding rows' accumulators
    // (their operand bytes and digits are zero: they would look like a row with h = H0_a)
    const bool wave_partial = PASS == 1 && item.a_blk * kPfWgRows + wave * kI8WaveRows < pd.n1 &&
                              item.a_blk * kPfWgRows + (wave + 1) * kI8WaveRows > pd.n1;   // wave-uniform
    int rs0[16];   // PASS 1: running row maximum of the accumulator
#pragma unroll
    for (int r = 0; r < 16; ++r) rs0[r] = (int)0x80000000;
    // pin the prologue loads before the loop (see sweep_kernel)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(af[ks]));
    asm volatile("" ::"v"(a_digit));
    if (PASS == 3) asm volatile("" ::"v"(rowc));
    wait_vmcnt<0>();

    int2* cbuf = reinterpret_cast<int2*>(sCand) + wave * kPfCandBuf;
    const unsigned cbuf_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)cbuf;
    int* vbuf = sVal + wave * kPfCandBuf;
    const unsigned vbuf_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)vbuf;
    int n_buf = 0;
    auto flush_candidates = [&]() {
        if (n_buf == 0) return;
        unsigned long long base64 = 0;
        if (lane == 0) base64 = atomicAdd(&cand_count[item.pair], (unsigned long long)n_buf);
        const int base = __builtin_amdgcn_readfirstlane((int)(base64 < (unsigned long long)pp.cand_cap ? base64 : (unsigned long long)pp.cand_cap));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int k = lane; k < n_buf; k += 64)
            if (base + k < pp.cand_cap) {
                cand[pp.cand_off + base + k] = cbuf[k];
                cand_val[pp.cand_off + base + k] = vbuf[k];
            }
        n_buf = 0;
    };

    // (readfirstlane: the compiler otherwise carries this wave-uniform flag as a lane mask and spends two VALU instructions per
    // tile on rebuilding it -- under a busy matrix pipe a VALU instruction costs ~8 ns of SIMD time)
    const bool wave_active = __builtin_amdgcn_readfirstlane((int)(item.a_blk * kPfWgRows + wave * kI8WaveRows < pd.n1)) != 0;

    // B fragments.  The two column blocks of a tile are two independent accumulator chains (a single dependent MFMA
    // chain runs at 3/4 of the rate, profiles/r01_ubench_mfma_chains.txt), so a k-step needs BOTH blocks' fragments:
    // k-steps 0 and 1 are pre-read in the EPI phase of the previous tile; k-step 2, k-step 3 and the digits are read
    // behind the first, second and third MFMA pair into the registers those free (the SIMD's other matrix-phase wave
    // fills the waits; more fragments in registers would not fit 128 VGPRs).
    const int lane_row_off = lcol * kI8RowBytes + lhalf * 16;
    auto preread = [&](int sl, i4v (&bf)[2][2]) {
        const char* pb = sB + sl * kI8TileBytes + lane_row_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf[ks][0] = *reinterpret_cast<const i4v*>(pb + ks * 32);
            bf[ks][1] = *reinterpret_cast<const i4v*>(pb + 32 * kI8RowBytes + ks * 32);
        }
    };
    auto column_max = [&](const i16v& acc) -> int {
        int m0 = max3i(acc[0], acc[1], acc[2]), m1 = max3i(acc[3], acc[4], acc[5]);
#pragma unroll
        for (int r = 6; r < 14; r += 4) {
            m0 = max3i(m0, acc[r], acc[r + 1]);
            m1 = max3i(m1, acc[r + 2], acc[r + 3]);
        }
        return max3i(max(m0, acc[14]), m1, acc[15]);
    };
    // compacted sweep: a hit is accumulator >= 0, sign bit clear (see scan_hits3 of sweep_kernel); 16 elements per lane
    // and column block: element r ends up at bit 15 - r of the block's mask
    // With every hit travels its ACCUMULATOR (round 5): acc = a'.b' - h_b + C with C the row's hit level floor((T - 2 h_a) / 2) (the C
    // operand; clamped, see rowc) -- the consumer (pf_exact_candidates_kernel<4>) turns it into the EXACT S = n'_a + (n'_b & 1) - 2 (acc - C):
    // on byte data the pinned fp32 order is exact integer arithmetic (header), so the candidates of a byte pair need no second look at
    // their rows.  acc[k] with a per-lane k: a select chain over the 16 accumulators (15 v_cmp + v_cndmask), paid once per hit -- about one
    // per wave and tile on compacted rows.
    auto pick16 = [&](const i16v& a, int k) -> int {   // (a chain, not a tree: one temporary -- the kernel sits at its register limit)
        int v = a[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) v = k == r ? a[r] : v;
        return v;
    };
    auto scan_block = [&](const i16v& a, unsigned nmask, int col) {
        unsigned hm = ~nmask & 0xffffu;
        for (unsigned long long mm = __ballot(hm != 0u); mm != 0ull; mm = __ballot(hm != 0u)) {
            const bool hit = hm != 0u;
            const int k = (__clz((int)hm) - 16) & 15;   // first remaining element of this lane (lanes without a hit: any valid index)
            if (n_buf + 64 > kPfCandBuf) flush_candidates();
            const int v = pick16(a, k);
            if (hit) {
                hm &= ~(0x8000u >> k);
                const int slt = n_buf + __popcll(mm & ((1ull << lane) - 1ull));
                const int2 e = make_int2(arow_base + (k & 3) + 8 * (k >> 2), col);
                asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3" ::"v"(cbuf_lds + slt * 8), "v"(e), "v"(vbuf_lds + slt * 4), "v"(v) : "memory");
            }
            n_buf += __popcll(mm);
        }
    };
    auto scan_hits3 = [&](const i16v& a0, const i16v& a1, int col0) {
        unsigned nm[2] = {0u, 0u};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            nm[0] = __builtin_amdgcn_alignbit(nm[0], (unsigned)a0[r], 31);
            nm[1] = __builtin_amdgcn_alignbit(nm[1], (unsigned)a1[r], 31);
        }
        // (the common case -- no hit in the wave's 64 x 32 elements -- leaves after one ballot)
        if (__ballot(((~nm[0] | ~nm[1]) & 0xffffu) != 0u) == 0ull) return;
        scan_block(a0, nm[0], col0);
        scan_block(a1, nm[1], col0 + 32);
    };
    // sweep 1, columns: one LDS atomic per column block into the tile's class array (see sweep_kernel)
    const unsigned col_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sCol +
                             (unsigned)((2 * (wave & 1) + lhalf) * kPfBT + lcol) * 4u;
    auto fold_columns = [&](int mA, int mB, int cs) {
        const unsigned a = col_lds + (unsigned)cs * (kPfBT * kPfColClasses * 4);
        asm volatile("ds_max_i32 %0, %1\n\tds_max_i32 %0, %2 offset:128" ::"v"(a), "v"(mA), "v"(mB) : "memory");
    };
    auto store_columns = [&](int tt) {
        // (address and constants are rebuilt per call -- every fourth tile per wave -- instead of living in registers for the
        // whole kernel: at <= 112 VGPRs four waves per SIMD leave 64 registers to the other stream's tail kernels)
        unsigned lane_op = (unsigned)lane;
        asm volatile("" : "+v"(lane_op));
        const unsigned a = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sCol + lane_op * 4u +
                           (unsigned)((tt - t_begin) & 3) * (kPfBT * kPfColClasses * 4);
        i4v v;
        int reset = (int)0x80000000;
        asm volatile("" : "+v"(reset));
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "ds_write_b32 %4, %5\n\tds_write_b32 %4, %5 offset:256\n\tds_write_b32 %4, %5 offset:512\n\tds_write_b32 %4, %5 offset:768"
                     : "=&v"(v.x), "=&v"(v.y), "=&v"(v.z), "=&v"(v.w) : "v"(a), "v"(reset) : "memory");
        // the two largest of the four class maxima of the accumulator (-S~/2), packed (msfm_cp_pack); none where a class saw no real row
        const int m01 = max(v.x, v.y), n01 = min(v.x, v.y), m23 = max(v.z, v.w), n23 = min(v.z, v.w);
        const int hi = max(m01, m23), lo = max(min(m01, m23), max(n01, n23));
        const int o = msfm_cp_pack(hi - item_k, hi > kI8PadTest, lo - item_k, lo > kI8PadTest);
        int* colbase = reinterpret_cast<int*>(cp_s0) + (pd.cp_off + (long long)item.a_blk * pd.n2pad);   // (uniform base + 32-bit lane offset)
        colbase[(unsigned)(tt * kPfBT) + lane_op] = o;
    };

    lds_barrier();
    dma_tile(t_begin + kI8Ring - 1);   // the first interval's DMA group
    i4v bf[2][2];
    if (wave_active) preread(0, bf);

    i16v accA, accB;
    // the matrix half of a tile: 10 MFMA, k-steps 2, 3 and the digits read from LDS behind the MFMA pairs that free their registers
    auto matrix_half = [&](int t) {
        const int sl = (t - t_begin) & (kI8Ring - 1);
        // (no s_setprio: with one barrier per tile the waves of a SIMD interleave well on their own -- priority for the matrix
        // halves measured 0.7 % slower, for the epilogue halves 2 % slower, profiles/r03_i8_sync_experiments.txt)
        const char* pb2 = sB + sl * kI8TileBytes + lane_row_off + 2 * 32;
        if (PASS == 1) {   // C = 0 (an inline constant: no register, no init)
            accA = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[0][0], i16v(0), 0, 0, 0);
            accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[0][1], i16v(0), 0, 0, 0);
        } else {           // C = the rows' hit levels
            accA = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[0][0], rowc, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[0][1], rowc, 0, 0, 0);
        }
        bf[0][0] = *reinterpret_cast<const i4v*>(pb2);                          // k-step 2 into the registers of k-step 0
        bf[0][1] = *reinterpret_cast<const i4v*>(pb2 + 32 * kI8RowBytes);
        accA = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[1], bf[1][0], accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[1], bf[1][1], accB, 0, 0, 0);
        bf[1][0] = *reinterpret_cast<const i4v*>(pb2 + 32);                     // k-step 3 into those of k-step 1
        bf[1][1] = *reinterpret_cast<const i4v*>(pb2 + 32 * kI8RowBytes + 32);
        accA = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[2], bf[0][0], accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[2], bf[0][1], accB, 0, 0, 0);
        bf[0][0] = *reinterpret_cast<const i4v*>(pb2 + 64);                     // the digits (bytes 128 + 16 lhalf ..)
        bf[0][1] = *reinterpret_cast<const i4v*>(pb2 + 32 * kI8RowBytes + 64);
        accA = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[3], bf[1][0], accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[3], bf[1][1], accB, 0, 0, 0);
        accA = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_digit, bf[0][0], accA, 0, 0, 0);   // + (H0_b - h_b) [+ (H0_a - h_a)]
        accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_digit, bf[0][1], accB, 0, 0, 0);
    };
    // the epilogue half of tile t (its accumulators are in accA / accB)
    auto epilogue_half = [&](int t) {
        // the image's last tile: columns >= n2 carry zero digits (they look like a column with h = H0): mask them
        const bool last_tile = (t + 1) * kPfBT > pd.n2;   // wave-uniform
        if (PASS == 1 && last_tile) {
            const bool v0 = t * kPfBT + lcol < pd.n2, v1 = t * kPfBT + 32 + lcol < pd.n2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accA[r] = v0 ? accA[r] : kI8Pad;
                accB[r] = v1 ? accB[r] : kI8Pad;
            }
        }
        if (wave_partial) {
            // (the limit is made opaque per tile: hipcc otherwise hoists the sixteen loop-invariant compares out of the
            // tile loop and keeps their masks alive for every wave of every work item -- 16 registers for a branch that
            // one wave per image takes)
            int row_lim = pd.n1 - arow_base;
            asm volatile("" : "+v"(row_lim));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool real_row = (r & 3) + 8 * (r >> 2) < row_lim;
                accA[r] = real_row ? accA[r] : kI8Pad;
                accB[r] = real_row ? accB[r] : kI8Pad;
            }
        }
        if (PASS == 3 && last_tile) {       // (sign bit set = no hit)
            if (!(t * kPfBT + lcol < pd.n2)) accA = i16v(-1);
            if (!(t * kPfBT + 32 + lcol < pd.n2)) accB = i16v(-1);
        }
        if (PASS == 1) {
            fold_columns(column_max(accA), column_max(accB), (t - t_begin) & 3);
#pragma unroll
            for (int r = 0; r < 16; ++r) rs0[r] = max3i(rs0[r], accA[r], accB[r]);
        } else {
            scan_hits3(accA, accB, t * kPfBT + lcol);
        }
    };

    // ONE barrier per tile.  Every wave runs matrix(u), epilogue(u), matrix(u + 1), ...; the even group passes its barrier behind
    // the epilogues, the odd group behind the matrix halves: in interval v (between barriers B_{v-1} and B_v) the even group does
    // matrix(v), epilogue(v) and the odd group epilogue(v - 1), matrix(v).  On every SIMD two waves issue MFMA while the other two
    // issue the VALU work of an epilogue -- they co-issue, profiles/r03_ubench_coissue.txt -- and a wave that finishes its first
    // half early goes on with its second instead of waiting for the slowest wave of the workgroup at a mid-tile barrier (round 2:
    // two barriers per tile, the halves in lock-step).
    //   ring: B_v guarantees that tiles <= v + 2 have landed (the even group pre-reads tile v + 2's first fragments at the end of
    //   interval v + 1); the DMA waves have issued the groups up to v + kI8Ring - 1 by then: "at most kI8Ring - 3 outstanding" (loads
    //   retire in order).  The group issued at the start of interval v overwrites the slot of tile v - 1, whose last readers (both
    //   groups' matrix(v - 1)) finished before B_{v-1}.
    //   column classes: tile t is folded by the even group in interval t and by the odd one in interval t + 1, stored (and reset)
    //   at the start of interval t + 2 by one wave: four slots.
    auto next_interval = [&](int v) {
        wait_older_group();
        lds_barrier();
        dma_tile(v + kI8Ring - 1);
        if (PASS == 1 && v - 2 >= t_begin && wave == ((v - t_begin) & 3)) store_columns(v - 2);
    };
#pragma unroll 1
    for (int u = t_begin; u < t_end; ++u) {
        if (wave_active) matrix_half(u);
        if (grp == 1) next_interval(u + 1);
        if (wave_active) {
            epilogue_half(u);
            if (u + 1 < t_end) preread((u + 1 - t_begin) & (kI8Ring - 1), bf);
        }
        if (grp == 0) next_interval(u + 1);
    }
    if (PASS == 3) flush_candidates();

    if (PASS == 1) {
        // rows: the two largest of a row's 32 lane maxima, transposed through the wave's share of the idle ring
        // ([32 rows][32 lanes + 1] ints; see sweep_kernel): lane l < 32 reduces row l of the wave
        wait_vmcnt<0>();   // the tail's DMA groups still write into the ring ...
        lds_barrier();     // ... everybody's have landed, and the odd group's last epilogue has folded its columns
        if (wave == 0) store_columns(t_end - 1);
        int* tr = reinterpret_cast<int*>(sB) + wave * (kI8WaveRows * 33);
        static_assert(kI8Waves * kI8WaveRows * 33 * 4 <= kI8Ring * kI8TileBytes, "the transposition fits the ring");
#pragma unroll
        for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * 33 + lcol] = rs0[r];
        if (lane < kI8WaveRows) {
            int m0 = (int)0x80000000, m1 = (int)0x80000000;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const int v = tr[lane * 33 + k];
                m1 = max(m1, min(m0, v));
                m0 = max(m0, v);
            }
            // S~ = -2 * accumulator as a float (exact); padding -> +inf
            const long long o = pd.rp_off + (long long)item.range * pd.n1pad + item.a_blk * kPfWgRows + wave * kI8WaveRows + lane;
            rp_s0[o] = m0 > kI8PadTest ? (float)(-2 * (m0 - item_k)) : f_inf();
            rp_s1[o] = m1 > kI8PadTest ? (float)(-2 * (m1 - item_k)) : f_inf();
        }
    }
    }   // item loop
}
