// msfm_fmat.h -- fundamental-matrix arithmetic shared by the device kernels (msfm_verify.hip.h, hipcc)
// and the host twin (host/GeometricVerification.cpp, g++): the geometric-verification hand-off of
// FeatureUtils::FilterMatches (reference src/Feature/FeatureUtils.cpp:176-206, which calls
// cv::findFundamentalMat(pts1, pts2, FM_RANSAC, 3.0, 0.99, mask)).
//
// Everything here is fp64 with +, -, *, /, sqrt only and static loop structure, compiled with
// -ffp-contract=off on both sides, so host and device produce the SAME bits for the same inputs:
// the host implementation is the test oracle of the batched GPU RANSAC.  OpenCV's own RANSAC (its
// RNG stream, its 7-point solver) cannot be reproduced without OpenCV; this row is outside the
// bit-parity claim.
//
// Solver: normalised 8-point.  The null vector of the 9 x 9 moment matrix M = A^T A comes from two
// steps of inverse iteration on M + eps I (Cholesky, no pivoting, no data-dependent indexing -- the
// smallest eigenvalue of M is ~0, the shift makes the factorisation exist, the eigen-gap makes one
// step converge to ~1e-10); rank 2 is enforced by removing the smallest right singular direction
// (3 x 3 cyclic Jacobi, fixed sweep count).
#pragma once

#if defined(__HIPCC__)
#define MSFM_FHD __host__ __device__ inline
#define MSFM_UNROLL _Pragma("unroll")
#else
#define MSFM_FHD inline
#define MSFM_UNROLL
#endif

namespace msfm_fmat {

constexpr int kFmatRefitSteps = 12;  // inverse-iteration steps of the least-squares refit

struct Norm2D {
    double cx, cy, s;
};

// centroid + mean-distance sqrt(2) scaling (Hartley) of the points x[idx[i]], y[idx[i]]
template <typename IndexFn>
MSFM_FHD Norm2D normalizer(const float* x, const float* y, int n, IndexFn at) {
    Norm2D t{0.0, 0.0, 1.0};
    for (int i = 0; i < n; ++i) {
        t.cx += (double)x[at(i)];
        t.cy += (double)y[at(i)];
    }
    t.cx /= n;
    t.cy /= n;
    double d = 0.0;
    for (int i = 0; i < n; ++i) {
        const double dx = (double)x[at(i)] - t.cx, dy = (double)y[at(i)] - t.cy;
        d += sqrt(dx * dx + dy * dy);
    }
    d /= n;
    t.s = d > 1e-12 ? 1.4142135623730951 / d : 1.0;
    return t;
}

// packed upper triangle of a symmetric 9 x 9 matrix: element (a, b), a <= b
MSFM_FHD constexpr int tri(int a, int b) { return a * 9 - a * (a - 1) / 2 + (b - a); }

MSFM_FHD void moment_add(double M[45], const Norm2D& t1, const Norm2D& t2, float px1, float py1, float px2, float py2) {
    const double x1 = ((double)px1 - t1.cx) * t1.s, y1 = ((double)py1 - t1.cy) * t1.s;
    const double x2 = ((double)px2 - t2.cx) * t2.s, y2 = ((double)py2 - t2.cy) * t2.s;
    const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
MSFM_UNROLL
    for (int a = 0; a < 9; ++a)
MSFM_UNROLL
        for (int b = a; b < 9; ++b) M[tri(a, b)] += r[a] * r[b];
}

// one Jacobi rotation of the symmetric 3 x 3 matrix (a00..a22 by reference) in the (p, q) plane
MSFM_FHD void jacobi_rot(double& app, double& aqq, double& apq, double& apr, double& aqr, double& vp0, double& vp1,
                        double& vp2, double& vq0, double& vq1, double& vq2) {
    if (!(apq > 1e-300 || apq < -1e-300)) return;
    const double theta = (aqq - app) / (2.0 * apq);
    const double at = theta >= 0 ? theta : -theta;
    const double t = (theta >= 0 ? 1.0 : -1.0) / (at + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    const double app_n = app - t * apq, aqq_n = aqq + t * apq;
    const double apr_n = c * apr - s * aqr, aqr_n = s * apr + c * aqr;
    app = app_n;
    aqq = aqq_n;
    apq = 0.0;
    apr = apr_n;
    aqr = aqr_n;
    const double p0 = c * vp0 - s * vq0, q0 = s * vp0 + c * vq0;
    const double p1 = c * vp1 - s * vq1, q1 = s * vp1 + c * vq1;
    const double p2 = c * vp2 - s * vq2, q2 = s * vp2 + c * vq2;
    vp0 = p0; vq0 = q0;
    vp1 = p1; vq1 = q1;
    vp2 = p2; vq2 = q2;
}

// F (row-major 3 x 3, unit Frobenius norm, x2^T F x1 = 0) from the moment matrix of normalised points.
// M is destroyed.  `steps` inverse-iteration steps: 2 for an 8-point sample (exact null space), more for a
// least-squares refit whose smallest eigenvalue is the residual.  Returns false for degenerate input.
MSFM_FHD bool solve(double M[45], const Norm2D& t1, const Norm2D& t2, double F[9], int steps) {
    double trace = 0.0;
MSFM_UNROLL
    for (int a = 0; a < 9; ++a) trace += M[tri(a, a)];
    if (!(trace > 0.0) || !(trace < 1e300)) return false;
    const double eps = trace * 1e-13;
MSFM_UNROLL
    for (int a = 0; a < 9; ++a) M[tri(a, a)] += eps;
    // Cholesky M = U^T U in place (U upper triangular, packed)
MSFM_UNROLL
    for (int k = 0; k < 9; ++k) {
        double d = M[tri(k, k)];
MSFM_UNROLL
        for (int j = 0; j < k; ++j) d -= M[tri(j, k)] * M[tri(j, k)];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        M[tri(k, k)] = d;
MSFM_UNROLL
        for (int b = k + 1; b < 9; ++b) {
            double v = M[tri(k, b)];
MSFM_UNROLL
            for (int j = 0; j < k; ++j) v -= M[tri(j, k)] * M[tri(j, b)];
            M[tri(k, b)] = v / d;
        }
    }
    // inverse iteration from a fixed generic start vector
    double x[9] = {0.31, -0.17, 0.43, 0.29, -0.37, 0.23, -0.41, 0.19, 0.47};
    for (int step = 0; step < steps; ++step) {
        // U^T y = x
MSFM_UNROLL
        for (int k = 0; k < 9; ++k) {
            double v = x[k];
MSFM_UNROLL
            for (int j = 0; j < k; ++j) v -= M[tri(j, k)] * x[j];
            x[k] = v / M[tri(k, k)];
        }
        // U z = y
MSFM_UNROLL
        for (int k = 8; k >= 0; --k) {
            double v = x[k];
MSFM_UNROLL
            for (int j = k + 1; j < 9; ++j) v -= M[tri(k, j)] * x[j];
            x[k] = v / M[tri(k, k)];
        }
        double nn = 0.0;
MSFM_UNROLL
        for (int k = 0; k < 9; ++k) nn += x[k] * x[k];
        if (!(nn > 0.0) || !(nn < 1e300)) return false;
        nn = 1.0 / sqrt(nn);
MSFM_UNROLL
        for (int k = 0; k < 9; ++k) x[k] *= nn;
    }
    // rank 2: G = Fn^T Fn, remove the eigen-direction of its smallest eigenvalue
    double g00 = 0, g01 = 0, g02 = 0, g11 = 0, g12 = 0, g22 = 0;
MSFM_UNROLL
    for (int r = 0; r < 3; ++r) {
        g00 += x[3 * r] * x[3 * r];
        g01 += x[3 * r] * x[3 * r + 1];
        g02 += x[3 * r] * x[3 * r + 2];
        g11 += x[3 * r + 1] * x[3 * r + 1];
        g12 += x[3 * r + 1] * x[3 * r + 2];
        g22 += x[3 * r + 2] * x[3 * r + 2];
    }
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;  // v[k][col]
    for (int sweep = 0; sweep < 12; ++sweep) {
        jacobi_rot(g00, g11, g01, g02, g12, v00, v10, v20, v01, v11, v21);  // (0,1), r = 2
        jacobi_rot(g00, g22, g02, g01, g12, v00, v10, v20, v02, v12, v22);  // (0,2), r = 1
        jacobi_rot(g11, g22, g12, g01, g02, v01, v11, v21, v02, v12, v22);  // (1,2), r = 0
    }
    double w0 = v00, w1 = v10, w2 = v20, emin = g00;
    if (g11 < emin) { emin = g11; w0 = v01; w1 = v11; w2 = v21; }
    if (g22 < emin) { emin = g22; w0 = v02; w1 = v12; w2 = v22; }
    double F2[9];
MSFM_UNROLL
    for (int r = 0; r < 3; ++r) {
        const double fv = x[3 * r] * w0 + x[3 * r + 1] * w1 + x[3 * r + 2] * w2;
        F2[3 * r] = x[3 * r] - fv * w0;
        F2[3 * r + 1] = x[3 * r + 1] - fv * w1;
        F2[3 * r + 2] = x[3 * r + 2] - fv * w2;
    }
    // denormalise: F = T2^T F2 T1, T = [s 0 -s cx; 0 s -s cy; 0 0 1]
    double Mx[9];
MSFM_UNROLL
    for (int r = 0; r < 3; ++r) {
        Mx[3 * r] = F2[3 * r] * t1.s;
        Mx[3 * r + 1] = F2[3 * r + 1] * t1.s;
        Mx[3 * r + 2] = F2[3 * r + 2] - F2[3 * r] * (t1.s * t1.cx) - F2[3 * r + 1] * (t1.s * t1.cy);
    }
    double nrm = 0.0;
MSFM_UNROLL
    for (int c = 0; c < 3; ++c) {
        F[c] = t2.s * Mx[c];
        F[3 + c] = t2.s * Mx[3 + c];
        F[6 + c] = Mx[6 + c] - (t2.s * t2.cx) * Mx[c] - (t2.s * t2.cy) * Mx[3 + c];
    }
MSFM_UNROLL
    for (int k = 0; k < 9; ++k) nrm += F[k] * F[k];
    if (!(nrm > 0.0) || !(nrm < 1e300)) return false;
    nrm = 1.0 / sqrt(nrm);
MSFM_UNROLL
    for (int k = 0; k < 9; ++k) F[k] *= nrm;
    return true;
}

// max of the squared distances of x2 to the line F x1 and of x1 to the line F^T x2 (findFundamentalMat's error)
MSFM_FHD double epipolar_error(const double F[9], float ax, float ay, float bx, float by) {
    const double x1 = ax, y1 = ay, x2 = bx, y2 = by;
    double l0 = F[0] * x1 + F[1] * y1 + F[2], l1 = F[3] * x1 + F[4] * y1 + F[5], l2 = F[6] * x1 + F[7] * y1 + F[8];
    const double d2 = x2 * l0 + y2 * l1 + l2;
    const double s2 = 1.0 / (l0 * l0 + l1 * l1);
    l0 = F[0] * x2 + F[3] * y2 + F[6];
    l1 = F[1] * x2 + F[4] * y2 + F[7];
    l2 = F[2] * x2 + F[5] * y2 + F[8];
    const double d1 = x1 * l0 + y1 * l1 + l2;
    const double s1 = 1.0 / (l0 * l0 + l1 * l1);
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return e1 > e2 ? e1 : e2;  // a NaN on either side ends up failing the <= threshold test
}

// counter-based sampling: the 8 distinct match indices of hypothesis `it` (n >= 8)
MSFM_FHD unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
MSFM_FHD void sample8(unsigned long long seed, int it, int n, int idx[8]) {
MSFM_UNROLL
    for (int k = 0; k < 8; ++k) idx[k] = -1;
MSFM_UNROLL
    for (int k = 0; k < 8; ++k) {
        int c = 0;
        for (int attempt = 0;; ++attempt) {
            c = attempt < 32 ? (int)(mix64(seed ^ mix64(((unsigned long long)it << 20) ^ ((unsigned long long)k << 8) ^ (unsigned long long)attempt)) % (unsigned long long)n)
                             : (c + 1) % n;  // linear probe: terminates because n >= 8
            bool dup = false;
MSFM_UNROLL
            for (int j = 0; j < 8; ++j) dup |= (idx[j] == c);
            if (!dup) break;
        }
        idx[k] = c;
    }
}

// hypothesis `it`: F from its 8 sampled matches; x1/y1/x2/y2 are the pair's aligned match coordinates
MSFM_FHD bool hypothesis(const float* x1, const float* y1, const float* x2, const float* y2, int n,
                        unsigned long long seed, int it, double F[9]) {
    int idx[8];
    sample8(seed, it, n, idx);
    float ax[8], ay[8], bx[8], by[8];
MSFM_UNROLL
    for (int k = 0; k < 8; ++k) {
        ax[k] = x1[idx[k]];
        ay[k] = y1[idx[k]];
        bx[k] = x2[idx[k]];
        by[k] = y2[idx[k]];
    }
    const Norm2D t1 = normalizer(ax, ay, 8, [](int i) { return i; });
    const Norm2D t2 = normalizer(bx, by, 8, [](int i) { return i; });
    double M[45];
MSFM_UNROLL
    for (int k = 0; k < 45; ++k) M[k] = 0.0;
MSFM_UNROLL
    for (int k = 0; k < 8; ++k) moment_add(M, t1, t2, ax[k], ay[k], bx[k], by[k]);
    return solve(M, t1, t2, F, 2);
}

// The adaptive iteration count of the sequential RANSAC, replayed over per-hypothesis inlier counts:
// returns the index of the winning hypothesis (-1: none reached 8 inliers) -- what a loop
// "for it < iters: if count[it] > best: best = count[it]; iters = min(iters, need(best))" ends with.
// Natural logarithm from + - x / and bit operations only: the host twin and the device must agree on ceil(need) below,
// and libm's log and the device library's log are different functions.  x > 0, finite, normal.
// x = m 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(y), y = (m - 1) / (m + 1), |y| < 0.172: 20 odd terms reach
// 1e-32 relative -- far below the double rounding of the sum.
MSFM_FHD double det_log(double x) {
    unsigned long long bits;
    static_assert(sizeof(bits) == sizeof(x), "64-bit double");
    __builtin_memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    __builtin_memcpy(&m, &bits, 8);        // [1, 2)
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    const double y = (m - 1.0) / (m + 1.0), y2 = y * y;
    double term = y, sum = 0.0;
    for (int k = 0; k < 20; ++k) {
        sum = sum + term / (double)(2 * k + 1);
        term = term * y2;
    }
    return 2.0 * sum + (double)e * 0.6931471805599453;
}

template <typename CountFn>
MSFM_FHD int replay_adaptive(int n, int max_iters, double confidence, CountFn count_at, int* best_count_out) {
    int best = 0, best_it = -1, iters = max_iters;
    for (int it = 0; it < iters; ++it) {
        const int c = count_at(it);
        if (c > best) {
            best = c;
            best_it = it;
            const double w = (double)c / n;
            const double w2 = w * w, w4 = w2 * w2, w8 = w4 * w4;
            double q = 1.0 - w8;
            if (q < 1e-300) q = 1e-300;
            const double need = det_log(1.0 - confidence) / det_log(q);
            if (need > 0.0 && need < (double)iters) {  // q == 1 (tiny consensus) gives -inf / NaN: no bound
                int ni = (int)need;
                if ((double)ni < need) ni += 1;  // ceil
                iters = ni > it + 1 ? ni : it + 1;
            }
        }
    }
    *best_count_out = best;
    return best >= 8 ? best_it : -1;
}

}  // namespace msfm_fmat
