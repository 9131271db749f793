// msfm_verify.hip.h -- batched geometric verification on the device: FeatureUtils::FilterMatches
// (reference src/Feature/FeatureUtils.cpp:176-206 = GetAlignedPointsFromMatches + cv::findFundamentalMat(
// FM_RANSAC, 3.0, 0.99) + keep the inliers) for every pair of a batch, right after the match epilogue and
// before the CSR gather, so unverified matches never leave the GPU.
//
// RANSAC as a data-parallel job: all `max_iters` hypotheses of a pair are evaluated at once (one thread =
// one hypothesis: sample 8 matches with a counter-based RNG, normalised 8-point solve, count inliers over
// the pair's matches staged in LDS), then one thread per pair REPLAYS the sequential algorithm's adaptive
// stopping rule over the per-hypothesis counts, so the winner is exactly the hypothesis the sequential loop
// would have ended with.  Refit on the consensus set + final mask + ordered compaction per pair.
// The arithmetic is msfm_fmat.h, shared with the host twin (host/GeometricVerification.cpp): same bits.
// Outside the bit-parity claim with respect to OpenCV; cases mirror findFundamentalMat:
// no matches -> nothing; < 7 -> no model, nothing kept; exactly 7 -> all kept; otherwise RANSAC, and fewer
// than 8 inliers -> nothing kept.
#pragma once
#include "msfm_fmat.h"
#include "msfm_kernels.hip.h"

namespace msfm {

struct VerifyPair {        // per pair: keypoint coordinates (x, y) of the two images
    const float2* k1;
    const float2* k2;
};

struct VerifyParams {
    double thr2;           // squared pixel threshold
    double confidence;
    int max_iters;
    int pad;
    unsigned long long seed;
};

constexpr int kVfChunk = 2048;  // matches staged in LDS at a time (4 float arrays = 32 KiB)

// aligned coordinates of every staged match: GetAlignedPointsFromMatches (FeatureUtils.cpp:262-279)
__global__ void vf_points_kernel(const PairDesc* __restrict__ pairs, const VerifyPair* __restrict__ vp,
                                 const int* __restrict__ counts, const int2* __restrict__ st_qt,
                                 float* __restrict__ x1, float* __restrict__ y1, float* __restrict__ x2, float* __restrict__ y2) {
    MSFM_TAIL_PRIO();
    const PairDesc pd = pairs[blockIdx.x];
    const VerifyPair v = vp[blockIdx.x];
    const int n = counts[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int2 m = st_qt[pd.out_off + i];
        const float2 a = v.k1[m.x], b = v.k2[m.y];
        x1[pd.out_off + i] = a.x;
        y1[pd.out_off + i] = a.y;
        x2[pd.out_off + i] = b.x;
        y2[pd.out_off + i] = b.y;
    }
}

// inlier count of every hypothesis: grid = (ceil(max_iters / 256), n_pairs), thread = hypothesis
__global__ __launch_bounds__(256) void vf_hypotheses_kernel(const PairDesc* __restrict__ pairs, const int* __restrict__ counts,
                                                            const float* __restrict__ x1, const float* __restrict__ y1,
                                                            const float* __restrict__ x2, const float* __restrict__ y2,
                                                            int* __restrict__ hyp_counts, VerifyParams prm) {
    MSFM_TAIL_PRIO();
    __shared__ float sx1[kVfChunk], sy1[kVfChunk], sx2[kVfChunk], sy2[kVfChunk];
    const int p = blockIdx.y;
    const int n = counts[p];
    if (n < 8) return;
    const long long base = pairs[p].out_off;
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    double F[9];
    bool ok = false;
    if (it < prm.max_iters) ok = msfm_fmat::hypothesis(x1 + base, y1 + base, x2 + base, y2 + base, n, prm.seed, it, F);
    int count = 0;
    for (int c0 = 0; c0 < n; c0 += kVfChunk) {
        const int m = min(kVfChunk, n - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            sx1[i] = x1[base + c0 + i];
            sy1[i] = y1[base + c0 + i];
            sx2[i] = x2[base + c0 + i];
            sy2[i] = y2[base + c0 + i];
        }
        __syncthreads();
        if (ok)
            for (int i = 0; i < m; ++i)  // every lane reads the same address: LDS broadcast
                count += (msfm_fmat::epipolar_error(F, sx1[i], sy1[i], sx2[i], sy2[i]) <= prm.thr2) ? 1 : 0;
    }
    if (it < prm.max_iters) hyp_counts[(long long)p * prm.max_iters + it] = ok ? count : 0;
}

// the sequential loop's adaptive stopping rule, replayed: one thread per pair
__global__ void vf_select_kernel(const int* __restrict__ counts, const int* __restrict__ hyp_counts, int n_pairs,
                                 VerifyParams prm, int* __restrict__ best_it, int* __restrict__ best_count) {
    MSFM_TAIL_PRIO();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const int n = counts[p];
    int bc = 0, bi = -1;
    if (n >= 8) {
        const int* hc = hyp_counts + (long long)p * prm.max_iters;
        bi = msfm_fmat::replay_adaptive(n, prm.max_iters, prm.confidence, [&](int it) { return hc[it]; },
                                        &bc);
    }
    best_it[p] = bi;
    best_count[p] = bc;
}

// final mask (best hypothesis, refit on its consensus set if that does not lose inliers) and ordered
// compaction of the pair's staged matches into the second staging buffer.  One workgroup per pair.
__global__ __launch_bounds__(256) void vf_mask_compact_kernel(
    const PairDesc* __restrict__ pairs, const int* __restrict__ counts, const int2* __restrict__ st_qt,
    const float* __restrict__ st_d, const float* __restrict__ x1, const float* __restrict__ y1,
    const float* __restrict__ x2, const float* __restrict__ y2, const int* __restrict__ best_it,
    const int* __restrict__ best_count, unsigned char* __restrict__ flags /* scratch, staged layout */,
    VerifyParams prm, int2* __restrict__ out_qt, float* __restrict__ out_d, int* __restrict__ out_counts) {
    MSFM_TAIL_PRIO();
    __shared__ double sF[9];
    __shared__ int s_ok, s_count, s_base, wsum[4];
    const int p = blockIdx.x;
    const PairDesc pd = pairs[p];
    const int n = counts[p];
    const long long base = pd.out_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // which matches survive
    int mode;  // 0: none, 1: all, 2: by flags
    if (n < 7) mode = 0;
    else if (n == 7) mode = 1;
    else mode = best_it[p] >= 0 ? 2 : 0;
    if (mode == 2) {
        if (tid == 0) {
            double F[9];
            s_ok = msfm_fmat::hypothesis(x1 + base, y1 + base, x2 + base, y2 + base, n, prm.seed, best_it[p], F) ? 1 : 0;
            for (int k = 0; k < 9; ++k) sF[k] = F[k];
            s_count = 0;
        }
        __syncthreads();
        double F[9];
        for (int k = 0; k < 9; ++k) F[k] = sF[k];
        for (int i = tid; i < n; i += blockDim.x)
            flags[base + i] = (s_ok && msfm_fmat::epipolar_error(F, x1[base + i], y1[base + i], x2[base + i], y2[base + i]) <= prm.thr2) ? 1 : 0;
        __syncthreads();
        // refit on the consensus set: sequential accumulation (the host twin's summation order)
        if (tid == 0) {
            using namespace msfm_fmat;
            int m = 0;
            for (int i = 0; i < n; ++i) m += flags[base + i];
            // normaliser over the inliers, in index order
            Norm2D t1{0.0, 0.0, 1.0}, t2{0.0, 0.0, 1.0};
            for (int i = 0; i < n; ++i)
                if (flags[base + i]) {
                    t1.cx += (double)x1[base + i];
                    t1.cy += (double)y1[base + i];
                    t2.cx += (double)x2[base + i];
                    t2.cy += (double)y2[base + i];
                }
            t1.cx /= m; t1.cy /= m; t2.cx /= m; t2.cy /= m;
            double d1 = 0.0, d2 = 0.0;
            for (int i = 0; i < n; ++i)
                if (flags[base + i]) {
                    const double ax = (double)x1[base + i] - t1.cx, ay = (double)y1[base + i] - t1.cy;
                    const double bx = (double)x2[base + i] - t2.cx, by = (double)y2[base + i] - t2.cy;
                    d1 += sqrt(ax * ax + ay * ay);
                    d2 += sqrt(bx * bx + by * by);
                }
            d1 /= m; d2 /= m;
            t1.s = d1 > 1e-12 ? 1.4142135623730951 / d1 : 1.0;
            t2.s = d2 > 1e-12 ? 1.4142135623730951 / d2 : 1.0;
            double M[45];
            for (int k = 0; k < 45; ++k) M[k] = 0.0;
            for (int i = 0; i < n; ++i)
                if (flags[base + i]) moment_add(M, t1, t2, x1[base + i], y1[base + i], x2[base + i], y2[base + i]);
            double F2[9];
            s_ok = solve(M, t1, t2, F2, kFmatRefitSteps) ? 1 : 0;
            for (int k = 0; k < 9; ++k) sF[k] = F2[k];
        }
        __syncthreads();
        if (s_ok) {
            for (int k = 0; k < 9; ++k) F[k] = sF[k];
            int c = 0;
            // second mask in bit 1 of the flag byte
            for (int i = tid; i < n; i += blockDim.x) {
                const int in2 = (msfm_fmat::epipolar_error(F, x1[base + i], y1[base + i], x2[base + i], y2[base + i]) <= prm.thr2) ? 1 : 0;
                flags[base + i] = (unsigned char)(flags[base + i] | (in2 << 1));
                c += in2;
            }
            atomicAdd(&s_count, c);
        }
        __syncthreads();
    }
    const int use_bit = (mode == 2 && s_ok && s_count >= best_count[p]) ? 1 : 0;
    // ordered compaction
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += blockDim.x) {
        const int i = i0 + tid;
        bool keep = false;
        if (i < n) keep = mode == 1 || (mode == 2 && ((flags[base + i] >> use_bit) & 1));
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int pos = s_base + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) pos += wsum[w];
        if (keep) {
            out_qt[base + pos] = st_qt[base + i];
            out_d[base + pos] = st_d[base + i];
        }
        __syncthreads();
        if (tid == 0) s_base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) out_counts[p] = s_base;
}

}  // namespace msfm
