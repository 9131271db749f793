// Types.h -- the reference's id typedefs (include/Common/Types.h:11-14) and the POD rows that
// replace its OpenCV types on this path (cv::KeyPoint -> x,y,size,angle; cv::DMatch -> q,t,dist).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace MonocularSfM {

typedef int image_t;
typedef int image_pair_t;
typedef int point2D_t;

struct KeyPoint {  // the four floats Database stores per keypoint (Database.cpp:114-126)
    float x, y, size, angle;
};

struct DMatch {  // the fields of cv::DMatch the path uses (imgIdx is always 0)
    int queryIdx = -1;
    int trainIdx = -1;
    float distance = 0.f;
};

struct Descriptors {  // cv::Mat CV_32F rows x cols, row-major
    int rows = 0, cols = 0;
    std::vector<float> data;
};

}  // namespace MonocularSfM
