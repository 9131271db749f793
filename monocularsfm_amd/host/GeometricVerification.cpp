#include "GeometricVerification.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "../csrc/msfm_fmat.h"

namespace MonocularSfM {

// Host twin of the batched device RANSAC (csrc/msfm_verify.hip.h): same sampling, same solver, same error
// measure, same adaptive stopping rule, all through the shared fp64 arithmetic of msfm_fmat.h -- the two give
// identical masks, which is what tests/test_gpu_verify.py checks.  The product path (ComputeMatches) runs the
// device version; this one is the test oracle and serves callers without a device context.
std::vector<unsigned char> FundamentalRansacMask(const std::vector<Point2f>& pts1, const std::vector<Point2f>& pts2,
                                                 double threshold, double confidence, int max_iters,
                                                 unsigned long long seed) {
    using namespace msfm_fmat;
    const int n = (int)pts1.size();
    if (n < 7) return {};
    if (n == 7) return std::vector<unsigned char>(7, 1);
    std::vector<float> x1((size_t)n), y1((size_t)n), x2((size_t)n), y2((size_t)n);
    for (int i = 0; i < n; ++i) {
        x1[(size_t)i] = pts1[(size_t)i].x;
        y1[(size_t)i] = pts1[(size_t)i].y;
        x2[(size_t)i] = pts2[(size_t)i].x;
        y2[(size_t)i] = pts2[(size_t)i].y;
    }
    const double thr2 = threshold * threshold;
    auto count_inliers = [&](const double F[9], unsigned char* mask) {
        int count = 0;
        for (int i = 0; i < n; ++i) {
            const bool in = epipolar_error(F, x1[(size_t)i], y1[(size_t)i], x2[(size_t)i], y2[(size_t)i]) <= thr2;
            if (mask) mask[i] = in ? 1 : 0;
            count += in ? 1 : 0;
        }
        return count;
    };
    // lazily evaluated per-hypothesis counts (the replay only asks for it < current iteration bound)
    auto count_at = [&](int it) {
        double F[9];
        if (!hypothesis(x1.data(), y1.data(), x2.data(), y2.data(), n, seed, it, F)) return 0;
        return count_inliers(F, nullptr);
    };
    int best_count = 0;
    const int best_it = replay_adaptive(n, max_iters, confidence, count_at, &best_count);
    std::vector<unsigned char> best((size_t)n, 0);
    if (best_it < 0) return best;
    double F[9];
    hypothesis(x1.data(), y1.data(), x2.data(), y2.data(), n, seed, best_it, F);
    count_inliers(F, best.data());
    // one refit on the consensus set, kept if it does not lose inliers
    std::vector<int> in;
    for (int i = 0; i < n; ++i)
        if (best[(size_t)i]) in.push_back(i);
    const int m = (int)in.size();
    const Norm2D t1 = normalizer(x1.data(), y1.data(), m, [&](int i) { return in[(size_t)i]; });
    const Norm2D t2 = normalizer(x2.data(), y2.data(), m, [&](int i) { return in[(size_t)i]; });
    double M[45] = {0};
    for (int i = 0; i < m; ++i) {
        const size_t k = (size_t)in[(size_t)i];
        moment_add(M, t1, t2, x1[k], y1[k], x2[k], y2[k]);
    }
    if (solve(M, t1, t2, F, kFmatRefitSteps)) {
        std::vector<unsigned char> cur((size_t)n);
        const int count = count_inliers(F, cur.data());
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
