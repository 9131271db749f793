// msfm_q8.hip.h -- route Q: the first sweep of a FLOAT store on the INTEGER matrix cores (included by msfm_prefilter.hip.h).
//
// v_mfma_i32_32x32x32_i8 sustains ~2.0 POP/s on this part against ~1.25 PFLOP/s for the fp16 instruction: the integer
// sweep of msfm_sweep_i8.hip.h does the same 128-term products in 0.62 of the time.  RootSIFT rows are bounded ([0, 1],
// unit L2 norm: FeatureExtraction's L1-root normalisation, src/Feature/FeatureExtraction.cpp:260-270 of the reference),
// so every image with all values in [0, 1] gets a BYTE TWIN  q = rint(s x),  s = 255 / m (m: the largest value of the context's
// twinned images, rounded up to 1/16 -- below), next to its fp16 operand rows, stored exactly like a byte image (signed rows of
// 176 B with the norm digits).  The integer sweep on the twins yields, per row, the exact integer S^ = |q_a - q_b|^2 of the two
// nearest twins (up to the parity bits: eps = 2, msfm_sweep_i8).
//
// What the twins prove.  With a^ = q_a inv (inv ~ 1 / s: the float itself), e_a = |a - a^|_2 (computed per row at upload, rounded
// up) and d^ = |a^ - b^|:
//        | |a - b| - d^ |  <=  e_a + e_b                         (triangle inequality, real arithmetic)
// so per row q of image 1, with E_2 = max_t e_t:
//        d0  >=  L0 = sqrt(S^min) inv - (e_q + E_2)              (S^min = smallest S^: S~ <= S^ <= S~ + 2)
//        d1  <=  U1 = sqrt(S^(2) + 2) inv + (e_q + E_2)          (S^(2): an upper bound of the second smallest)
// If L0 >= ratio * U1 the Lowe test fails whatever the exact bits are, if L0 > max_distance the distance cut removes
// the row: it is DEAD, exactly as for the fp16 bounds of pf_thresholds_kernel (the pinned fp32 order is within 4e-6
// relative of the real distance, msfm_kernels.hip.h: a factor 1 +- 1e-5 covers it).  On the bench data the same 6 % of the
// rows stay alive as under the fp16 bound (tools/int8_prefilter_study.py).
//
// FINE twins (m <= 0.625, i.e. RootSIFT: s ~ 580, e ~ 0.0056) -- the thresholds of sweep 2 come straight from the twins:
//   sweep 1     sweep_i8_kernel<1> on the twins, every descriptor pair of the batch               (the dominant kernel)
//   prune       pf_prune_q8_kernel(direct): live / dead, T = U1^2 + eps_fp16 per live row / column, the block masks of the
//               reverse direction from the twins' per-block column minima, the live counts of the plan
//   plan, sweep 2 (fp16), exact re-check, ...   as on the fp16 route.  4.9 candidates per live row instead of 2.5, mask density
//               0.50 instead of 0.24 -- and no sweep 1' (12 candidates per live row at s = 255: hence the adaptive scale).
// COARSE twins (values beyond 0.625; MSFM_Q8_DIRECT=0 forces it) -- the live rows get an fp16 sweep 1 of their own first:
//   sweep 1, prune (live / dead only)
//   plan A      the compacted-sweep plan (msfm_plan.hip.h) with every live column in EVERY 512-row block group
//   sweep 1'    sweep_kernel<4>: fp16 S~ top-2 of every compacted live row over its group's tiles  (~13 % of sweep 1's work)
//   scatter     q8_scatter_kernel: those top-2 into the partial arrays pf_thresholds_kernel reads (rows: one range;
//               columns: one entry per block group = the per-block minima the block mask needs)
//   thresholds  pf_thresholds_kernel, unchanged but for skipping what the prune kernel marked dead
//   plan B, sweep 2, exact re-check, ...   as on the fp16 route.
// The candidate sets contain those of the fp16 route restricted to rows that are provably dead anyway: every bit of the result
// is the same.
#pragma once
// (included inside namespace msfm)

// The twins' scale.  A context quantises with ONE scale s = 255 / m, m = the largest value of its twinned images rounded up to
// a multiple of 1/16 (RootSIFT values rarely exceed 0.45: s ~ 580, error norms ~0.0056 instead of 0.0128 at s = 255); a
// twin built under a smaller m is rebuilt at the start of the next matching call (msfm_store_host.hip.h: rebuild_stale_twins -> build_twins, from the resident fp32 rows).  The
// real number the bounds are stated with is the float `inv` ~ 1 / s itself: the twin is a^ = q * inv EXACTLY.
constexpr float kQ8LevelStep = 1.f / 16.f;
// with twins at least this fine the twins' sweep yields thresholds tight enough to collect candidates with directly
// (~4.5 per live row on RootSIFT-like data against 2.5 after an fp16 sweep 1'; 12 at m = 1): no sweep 1'
constexpr float kQ8DirectMaxLevel = 0.625f;

// live / dead from the integer sweep on the twins.  Rows: rp_s0 / rp_s1 hold S~min and an upper bound of the second
// smallest S~ (floats holding integers); columns: per 512-row block the two largest accumulator maxima (-S~/2).
// Live rows / columns get the marker +inf in tu / tv (dead: -inf), every live column carries ALL block bits.
// grid = (ceil(max_npad/256), n_pairs), like pf_thresholds_kernel, whose member counting this kernel repeats.
__global__ void pf_prune_q8_kernel(const PairDesc* __restrict__ pairs, const PfPair* __restrict__ pf, const PfPair* __restrict__ pfq,
                                   const float* __restrict__ rp_s0, const float* __restrict__ rp_s1, const float* __restrict__ cp_s0,
                                   unsigned* __restrict__ colmask, float* __restrict__ tuv, PruneParams pr, PlanCounts plan,
                                   float inv, int direct) {
    MSFM_TAIL_PRIO();
    const PairDesc pd = pairs[blockIdx.y];
    const PfPair pp = pf[blockIdx.y];
    if (!pd.valid || !pp.use) return;
    const PfPair pq = pfq[blockIdx.y];
    if (!pq.use) return;   // (a mixed sub-batch: this pair went through the fp16 sweep 1, pf_thresholds_kernel looks after it)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    bool row_live = false;
    unsigned col_bits = 0;
    // S~ <= S^ <= S~ + 2;  sqrtf and the products below are rounded: the factors (1 -+ 1e-6) keep the bounds one-sided
    auto lower = [&](float s0, float err) -> float { return fmaxf(sqrtf(fmaxf(s0, 0.f)) * inv * (1.f - 1e-6f) - err, 0.f); };
    auto upper = [&](float s1, float err) -> float { return sqrtf(s1 + 2.f) * inv * (1.f + 1e-6f) + err; };
    auto dead = [&](float s0, float s1, float err) -> bool {
        const float l0 = lower(s0, err), u1 = upper(s1, err);
        const bool ratio_fails = pr.ratio > 0.f && l0 * (1.f - 1e-5f) >= pr.ratio * u1 * (1.f + 1e-5f);
        const bool too_far = l0 * (1.f - 1e-5f) > pr.max_distance * (1.f + 1e-5f);
        return ratio_fails || too_far;
    };
    // direct mode: the threshold sweep 2 (fp16) collects candidates with.  Every column that can be the first or second
    // neighbour of the row under the pinned fp32 order has a real S <= U1^2 (1 + 2.1e-5) (U1 bounds the real second distance,
    // the pinned order is within 1e-5 relative of real arithmetic), hence an fp16 S~ <= that + eps; + the roundings here
    // and in sweep 2's test, as in pf_thresholds_kernel
    const float eps_norm = 4.8828125e-4f * fmaxf(pp.a_c, pp.b_c);
    auto threshold = [&](float u1, float nrm, float other_max) -> float {
        const float eps = kEpsRel * (nrm + other_max) + kEpsAbs * (sqrtf(nrm) + sqrtf(other_max)) + eps_norm;
        const float u2 = u1 * u1;
        return u2 * (1.f + 3e-5f) + eps + 1e-5f * (u2 + nrm + other_max);
    };
    // ---- every load of this thread first, at clamped (always valid) indices: the kernel is bookkeeping over ~83 M row / column
    // slots per job and was five dependent memory round trips per wave (rows, their errors, the columns' sixteen partials, their
    // errors, the norms of the live ones: 76 % of its wave cycles in SQ_WAIT_ANY at 2.4 TB/s); now it is one.
    const int* cpk = reinterpret_cast<const int*>(cp_s0);   // packed partials of the integer sweep (msfm_cp_pack)
    const int nb = pd.a_blocks256;
    const bool in_r = e < pd.n1pad, in_c = e < pd.n2pad;
    const int er = in_r ? e : 0, ec = in_c ? e : 0;
    const float ld_rs0 = rp_s0[pd.rp_off + er], ld_rs1 = rp_s1[pd.rp_off + er];
    const float ld_aerr = pq.a_err[min(er, pd.n1 - 1)], ld_anrm = pp.a_nrm[er];
    const float ld_berr = pq.b_err[min(ec, pd.n2 - 1)], ld_bnrm = pp.b_nrm[ec];
    int ld_cp[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) ld_cp[p] = cpk[pd.cp_off + (long long)min(p, nb - 1) * pd.n2pad + ec];
    if (in_r) {
        float s0 = f_inf(), s1 = f_inf();
        v2_merge(s0, s1, ld_rs0, ld_rs1);
        for (int p = 1; p < pd.ranges; ++p) {   // (a pair split into B ranges: small batches only)
            const long long o = pd.rp_off + (long long)p * pd.n1pad + e;
            v2_merge(s0, s1, rp_s0[o], rp_s1[o]);
        }
        bool live = e < pd.n1;
        const float err = live ? (ld_aerr + pq.b_c) * (1.f + 1e-6f) : 0.f;   // (b_c of the twin pair: E of image 2)
        if (live) live = !dead(s0, s1, err);
        float T = live ? f_inf() : -f_inf();
        if (live && direct) T = threshold(upper(s1, err), ld_anrm, pp.b_nrm_max);
        tuv[pp.tu_off + e] = T;
        row_live = live;
    }
    if (in_c) {
        float s0 = f_inf(), s1 = f_inf();
        // up to 16 blocks (8192 rows) the blocks' minima stay in registers for the mask below
        float bmin[16];
        if (nb <= 16) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                float2 m = make_float2(-f_inf(), -f_inf());
                if (p < nb) m = i8_cp_unpack(ld_cp[p]);
                bmin[p] = -2.f * m.x;
                v2_merge(s0, s1, -2.f * m.x, -2.f * m.y);
            }
        } else {
            for (int p = 0; p < nb; ++p) {
                const float2 m = i8_cp_unpack(cpk[pd.cp_off + (long long)p * pd.n2pad + e]);
                v2_merge(s0, s1, -2.f * m.x, -2.f * m.y);
            }
        }
        bool live = e < pd.n2;
        const float err = live ? (ld_berr + pq.a_c) * (1.f + 1e-6f) : 0.f;
        if (live) live = !dead(s0, s1, err);
        const int g = (nb + 31) / 32, bits = (nb + g - 1) / g;
        if (!direct) {
            tuv[pp.tv_off + e] = live ? f_inf() : -f_inf();
            // sweep 1' visits every 512-row block group of image 1 for a live column: one bit per group (see pf_thresholds_kernel)
            col_bits = live ? (bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u)) : 0u;
        } else {
            // the blocks of image 1 that can hold a candidate of this column: the block's nearest row may be as close as
            // lower(block minimum), a candidate is at most U1 away
            const float u1 = live ? upper(s1, err) : 0.f;
            tuv[pp.tv_off + e] = live ? threshold(u1, ld_bnrm, pp.a_nrm_max) : -f_inf();
            if (live) {
                if (nb <= 16) {
#pragma unroll
                    for (int p = 0; p < 16; ++p)
                        if (p < nb && lower(bmin[p], err) * (1.f - 1e-5f) <= u1 * (1.f + 1e-5f)) col_bits |= 1u << p;   // (g = 1)
                } else {
                    for (int p = 0; p < nb; ++p) {
                        const float smin = -2.f * i8_cp_unpack(cpk[pd.cp_off + (long long)p * pd.n2pad + e]).x;
                        if (lower(smin, err) * (1.f - 1e-5f) <= u1 * (1.f + 1e-5f)) col_bits |= 1u << (p / g);
                    }
                }
            }
        }
        colmask[pp.tv_off + e] = col_bits;
    }
    if (plan.pp_plan) {
        const PlanPair pl = plan.pp_plan[blockIdx.y];
        if (pl.fwd_member < 0) return;   // (block-uniform)
        __shared__ int hist[33];
        if (threadIdx.x < 33) hist[threadIdx.x] = 0;
        __syncthreads();
        const unsigned long long rl = __ballot(row_live);
        if ((threadIdx.x & 63) == 0 && rl) atomicAdd(&hist[32], __popcll(rl));
        if (!direct) {
            const unsigned long long cl = __ballot(col_bits != 0u);
            if ((threadIdx.x & 63) == 0 && cl) atomicAdd(&hist[0], __popcll(cl));   // every live column carries every bit
        } else {
            while (col_bits) {
                const int bit = __builtin_ctz(col_bits);
                col_bits &= col_bits - 1;
                atomicAdd(&hist[bit], 1);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && hist[32]) {
            atomicAdd(&plan.cnt[pl.fwd_member], hist[32]);
            atomicAdd(&plan.gtot[plan.member_group[pl.fwd_member]], hist[32]);
        }
        const int mine = direct ? hist[threadIdx.x < 33 ? threadIdx.x : 0] : hist[0];
        if ((int)threadIdx.x < pl.rev_bits && mine) {
            atomicAdd(&plan.cnt[pl.rev_member0 + threadIdx.x], mine);
            atomicAdd(&plan.gtot[plan.member_group[pl.rev_member0 + threadIdx.x]], mine);
        }
    }
}

// sweep 1' results -> the partial arrays pf_thresholds_kernel reads.  One thread per compacted row k of plan A:
// forward group: rp_s0 / rp_s1 of (pair, row) -- range 0 (the other ranges of a split pair get +inf);
// reverse group (image, block bit b): the float2 column partial of the bit's FIRST 512-row block, -inf in the bit's others.
__global__ void q8_scatter_kernel(const PairDesc* __restrict__ pairs, const PairDesc* __restrict__ vpairs, const CandList* __restrict__ lists,
                                  const PlanGroup* __restrict__ groups, int n_groups, const long long* __restrict__ grow0,
                                  const float* __restrict__ cmp_s0, const float* __restrict__ cmp_s1,
                                  float* __restrict__ rp_s0, float* __restrict__ rp_s1, float* __restrict__ cp_s0) {
    MSFM_TAIL_PRIO();
    const int g = blockIdx.y;
    if (g >= n_groups) return;
    const long long row0 = grow0[g];
    if (row0 < 0) return;
    const int rows = vpairs[g].n1;
    const CandList L = lists[g];
    const PlanGroup G = groups[g];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < rows; k += gridDim.x * blockDim.x) {
        const int e = L.live_idx[k];
        const PairDesc pd = pairs[L.row_pair[k]];
        const float s0 = cmp_s0[row0 + k], s1 = cmp_s1[row0 + k];
        if (G.dir == 0) {
            for (int p = 0; p < pd.ranges; ++p) {
                rp_s0[pd.rp_off + (long long)p * pd.n1pad + e] = p == 0 ? s0 : f_inf();
                rp_s1[pd.rp_off + (long long)p * pd.n1pad + e] = p == 0 ? s1 : f_inf();
            }
        } else {
            // the group's tile range [bt_begin, bt_end) in 128-row blocks = the 512-row blocks [bt_begin / 4, ceil(bt_end / 4))
            const int p0 = G.bt_begin / (kPfWgRows / kBM), p1 = (G.bt_end + kPfWgRows / kBM - 1) / (kPfWgRows / kBM);
            float2* cp2 = reinterpret_cast<float2*>(cp_s0);
            for (int p = p0; p < p1 && p < pd.a_blocks256; ++p)
                cp2[pd.cp_off + (long long)p * pd.n2pad + e] = p == p0 ? make_float2(-0.5f * s0, -0.5f * s1) : make_float2(-f_inf(), -f_inf());
        }
    }
}
