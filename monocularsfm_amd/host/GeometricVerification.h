// GeometricVerification.h -- host-side twin of FeatureUtils::FilterMatches
// (src/Feature/FeatureUtils.cpp:176-206): cv::findFundamentalMat(pts1, pts2, FM_RANSAC, 3.0, 0.99,
// mask) on raw pixel coordinates, keep the inliers.  OpenCV's RANSAC (its RNG, its 7-point solver)
// cannot be reproduced bit-for-bit without OpenCV, so this row is OUTSIDE the bit-parity claim
// (SURVEY.md 8a row a13 / 8f-1): same model (fundamental matrix), same error measure (max of the
// two squared point-to-epipolar-line distances), same threshold / confidence / iteration cap,
// deterministic seed; acceptance is inlier-set agreement on data with a true epipolar geometry.
#pragma once
#include <vector>

#include "Types.h"

namespace MonocularSfM {

struct Point2f {
    float x, y;
};

// Returns the inlier mask (1 = keep), one entry per match.  Mirrors findFundamentalMat's cases:
// < 7 points -> no model, empty mask (the caller keeps nothing); exactly 7 -> all ones;
// otherwise RANSAC.
std::vector<unsigned char> FundamentalRansacMask(const std::vector<Point2f>& pts1, const std::vector<Point2f>& pts2,
                                                 double threshold = 3.0, double confidence = 0.99,
                                                 int max_iters = 1000, unsigned long long seed = 0x5eed5eedULL);

// FeatureUtils::GetAlignedPointsFromMatches + FilterMatches
void FilterMatches(const std::vector<KeyPoint>& kpts1, const std::vector<KeyPoint>& kpts2,
                   const std::vector<DMatch>& matches, std::vector<DMatch>* prune_matches);

}  // namespace MonocularSfM
