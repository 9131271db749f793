#include "GeometricVerification.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace MonocularSfM {

namespace {

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9); eigenvectors in columns of V
void JacobiEigen(double* A, int n, double* V, double* evals) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-30) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) evals[i] = A[i * n + i];
}

struct Norm {
    double cx, cy, s;
};

Norm Normalizer(const std::vector<Point2f>& p, const int* idx, int n) {
    Norm t{0, 0, 1};
    for (int i = 0; i < n; ++i) {
        t.cx += p[idx[i]].x;
        t.cy += p[idx[i]].y;
    }
    t.cx /= n;
    t.cy /= n;
    double d = 0;
    for (int i = 0; i < n; ++i) d += std::hypot(p[idx[i]].x - t.cx, p[idx[i]].y - t.cy);
    d /= n;
    t.s = d > 1e-12 ? std::sqrt(2.0) / d : 1.0;
    return t;
}

// normalised 8-point algorithm on the points idx[0..n) ; F maps pts1 -> epipolar lines in image 2
bool EightPoint(const std::vector<Point2f>& p1, const std::vector<Point2f>& p2, const int* idx, int n, double F[9]) {
    const Norm t1 = Normalizer(p1, idx, n), t2 = Normalizer(p2, idx, n);
    double AtA[81] = {0};
    for (int i = 0; i < n; ++i) {
        const double x1 = (p1[idx[i]].x - t1.cx) * t1.s, y1 = (p1[idx[i]].y - t1.cy) * t1.s;
        const double x2 = (p2[idx[i]].x - t2.cx) * t2.s, y2 = (p2[idx[i]].y - t2.cy) * t2.s;
        const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        for (int a = 0; a < 9; ++a)
            for (int b = 0; b < 9; ++b) AtA[a * 9 + b] += r[a] * r[b];
    }
    double V[81], ev[9];
    JacobiEigen(AtA, 9, V, ev);
    int kmin = 0;
    for (int k = 1; k < 9; ++k)
        if (ev[k] < ev[kmin]) kmin = k;
    double Fn[9];
    for (int k = 0; k < 9; ++k) Fn[k] = V[k * 9 + kmin];
    // rank 2: remove the smallest right singular direction, F (I - v v^T)
    double FtF[9] = {0};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            for (int k = 0; k < 3; ++k) FtF[a * 3 + b] += Fn[k * 3 + a] * Fn[k * 3 + b];
    double V3[9], e3[3];
    JacobiEigen(FtF, 3, V3, e3);
    int m = 0;
    for (int k = 1; k < 3; ++k)
        if (e3[k] < e3[m]) m = k;
    const double v[3] = {V3[0 * 3 + m], V3[1 * 3 + m], V3[2 * 3 + m]};
    double F2[9];
    for (int r = 0; r < 3; ++r) {
        const double fv = Fn[r * 3] * v[0] + Fn[r * 3 + 1] * v[1] + Fn[r * 3 + 2] * v[2];
        for (int c = 0; c < 3; ++c) F2[r * 3 + c] = Fn[r * 3 + c] - fv * v[c];
    }
    // denormalise: F = T2^T F2 T1, T = [s 0 -s*cx; 0 s -s*cy; 0 0 1]
    const double T1[9] = {t1.s, 0, -t1.s * t1.cx, 0, t1.s, -t1.s * t1.cy, 0, 0, 1};
    const double T2[9] = {t2.s, 0, -t2.s * t2.cx, 0, t2.s, -t2.s * t2.cy, 0, 0, 1};
    double M[9] = {0};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            for (int k = 0; k < 3; ++k) M[a * 3 + b] += F2[a * 3 + k] * T1[k * 3 + b];
    double nrm = 0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += T2[k * 3 + a] * M[k * 3 + b];
            F[a * 3 + b] = acc;
            nrm += acc * acc;
        }
    if (!(nrm > 0) || !std::isfinite(nrm)) return false;
    nrm = std::sqrt(nrm);
    for (int k = 0; k < 9; ++k) F[k] /= nrm;
    return true;
}

// max of the squared distances of x2 to F x1 and of x1 to F^T x2 (findFundamentalMat's error)
inline double EpipolarError(const double F[9], const Point2f& a, const Point2f& b) {
    const double x1 = a.x, y1 = a.y, x2 = b.x, y2 = b.y;
    double l0 = F[0] * x1 + F[1] * y1 + F[2], l1 = F[3] * x1 + F[4] * y1 + F[5], l2 = F[6] * x1 + F[7] * y1 + F[8];
    const double d2 = x2 * l0 + y2 * l1 + l2;
    const double s2 = 1.0 / (l0 * l0 + l1 * l1);
    l0 = F[0] * x2 + F[3] * y2 + F[6];
    l1 = F[1] * x2 + F[4] * y2 + F[7];
    l2 = F[2] * x2 + F[5] * y2 + F[8];
    const double d1 = x1 * l0 + y1 * l1 + l2;
    const double s1 = 1.0 / (l0 * l0 + l1 * l1);
    return std::max(d1 * d1 * s1, d2 * d2 * s2);
}

struct Rng {  // splitmix64
    unsigned long long s;
    unsigned long long next() {
        unsigned long long z = (s += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    int below(int n) { return (int)(next() % (unsigned long long)n); }
};

}  // namespace

std::vector<unsigned char> FundamentalRansacMask(const std::vector<Point2f>& pts1, const std::vector<Point2f>& pts2,
                                                 double threshold, double confidence, int max_iters,
                                                 unsigned long long seed) {
    const int n = (int)pts1.size();
    if (n < 7) return {};
    if (n == 7) return std::vector<unsigned char>(7, 1);
    const double thr2 = threshold * threshold;
    Rng rng{seed};
    std::vector<unsigned char> best(n, 0), cur(n);
    int best_count = 0;
    int iters = max_iters;
    for (int it = 0; it < iters; ++it) {
        int idx[8];
        for (int k = 0; k < 8;) {
            const int c = rng.below(n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= (idx[j] == c);
            if (!dup) idx[k++] = c;
        }
        double F[9];
        if (!EightPoint(pts1, pts2, idx, 8, F)) continue;
        int count = 0;
        for (int i = 0; i < n; ++i) {
            const double e = EpipolarError(F, pts1[i], pts2[i]);
            cur[i] = (e <= thr2) ? 1 : 0;  // NaN compares false -> outlier
            count += cur[i];
        }
        if (count > best_count) {
            best_count = count;
            best.swap(cur);
            cur.resize(n);
            // adaptive iteration count: log(1-p) / log(1 - w^8)
            const double w = (double)count / n;
            const double denom = std::log(std::max(1.0 - std::pow(w, 8), 1e-300));
            const double need = std::log(1.0 - confidence) / denom;
            if (std::isfinite(need) && need < iters) iters = std::max(it + 1, (int)std::ceil(need));
        }
    }
    if (best_count < 8) return std::vector<unsigned char>(n, 0);
    // one refit on the consensus set, keep it if it does not lose inliers
    std::vector<int> in;
    for (int i = 0; i < n; ++i)
        if (best[i]) in.push_back(i);
    double F[9];
    if (EightPoint(pts1, pts2, in.data(), (int)in.size(), F)) {
        int count = 0;
        for (int i = 0; i < n; ++i) {
            cur[i] = (EpipolarError(F, pts1[i], pts2[i]) <= thr2) ? 1 : 0;
            count += cur[i];
        }
        if (count >= best_count) best.swap(cur);
    }
    return best;
}

void FilterMatches(const std::vector<KeyPoint>& kpts1, const std::vector<KeyPoint>& kpts2,
                   const std::vector<DMatch>& matches, std::vector<DMatch>* prune_matches) {
    if (kpts1.empty() || matches.empty()) return;  // FeatureUtils.cpp:181-184
    std::vector<Point2f> a, b;
    a.reserve(matches.size());
    b.reserve(matches.size());
    for (const DMatch& m : matches) {
        a.push_back(Point2f{kpts1[(size_t)m.queryIdx].x, kpts1[(size_t)m.queryIdx].y});
        b.push_back(Point2f{kpts2[(size_t)m.trainIdx].x, kpts2[(size_t)m.trainIdx].y});
    }
    const std::vector<unsigned char> mask = FundamentalRansacMask(a, b, 3.0, 0.99);
    for (size_t i = 0; i < mask.size(); ++i)
        if (mask[i]) prune_matches->push_back(matches[i]);
}

}  // namespace MonocularSfM
