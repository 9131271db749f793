// MatchEmission.h -- how a pair's match list is laid out in its `matches` row.
//
// The consumer is SceneGraph::Load -> AddCorrespondences (/root/reference/src/Reconstruction/SceneGraph.cpp:11-85,
// 170-251): rows in pair_id order, per match a bounds check against NumKeyPoints, a linear std::find_if over
// image1.corrs[idx1] for a duplicate (idx1 = the match's index in the image with the SMALLER id, i.e. column 0 of the
// stored blob, Database.cpp:633-640), then two emplace_backs.  What the producer can do for it without touching the
// consumer or the schema:
//   * never emit a duplicate (idx1, idx2) inside a row and never an index >= the image's keypoint count -- guaranteed by
//     construction (query indices of a list are distinct; with the cross-check train indices are distinct too) and
//     checked by CheckRowContract below, so the find_if never hits and the WARNING paths are never taken;
//   * MSFM_SCENEGRAPH_ORDER=1: order every row by COLUMN 0 instead of by queryIdx.  Both matchers call
//     MatchImagePairs with id1 > id2 (FeatureMatching.cpp:82-97, 110-139), so the reference's rows are sorted by column 1;
//     sorted by column 0 the consumer walks image1.corrs front to back instead of jumping around in it.
//   * MSFM_SCENEGRAPH_MIN_MATCHES=n: a list with fewer than n matches is stored with rows = 0: Load ignores such pairs
//     anyway (min_num_matches), and ReadAllMatches (`WHERE rows > 0`) then does not even fetch them.
// Default (neither variable set): the row is byte-identical to the reference's.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "Types.h"

namespace MonocularSfM {

struct EmissionOptions {
    bool scene_graph_order = false;
    int min_num_matches = 0;
    static EmissionOptions FromEnvironment() {
        EmissionOptions o;
        if (const char* e = std::getenv("MSFM_SCENEGRAPH_ORDER")) o.scene_graph_order = e[0] != '0';
        if (const char* e = std::getenv("MSFM_SCENEGRAPH_MIN_MATCHES")) o.min_num_matches = std::atoi(e);
        return o;
    }
};

// matches: the list of pair (image_id1, image_id2) in DMatch terms (queryIdx -> image_id1, trainIdx -> image_id2)
inline void ApplyEmissionOptions(const EmissionOptions& o, image_t image_id1, image_t image_id2, std::vector<DMatch>* matches) {
    if (o.min_num_matches > 0 && (int)matches->size() < o.min_num_matches) {
        matches->clear();
        return;
    }
    if (!o.scene_graph_order) return;
    const bool swap = image_id1 > image_id2;   // Database::SwapImagePair: column 0 = index in the smaller image id
    std::stable_sort(matches->begin(), matches->end(), [swap](const DMatch& a, const DMatch& b) {
        return swap ? a.trainIdx < b.trainIdx : a.queryIdx < b.queryIdx;
    });
}

// The contract AddCorrespondences relies on: indices in range, no (queryIdx, trainIdx) twice.  0 = holds.
inline int CheckRowContract(const std::vector<DMatch>& matches, size_t num_keypoints1, size_t num_keypoints2) {
    std::vector<std::pair<int, int>> seen;
    seen.reserve(matches.size());
    for (const DMatch& m : matches) {
        if (m.queryIdx < 0 || (size_t)m.queryIdx >= num_keypoints1 || m.trainIdx < 0 || (size_t)m.trainIdx >= num_keypoints2) return 1;
        seen.emplace_back(m.queryIdx, m.trainIdx);
    }
    std::sort(seen.begin(), seen.end());
    return std::adjacent_find(seen.begin(), seen.end()) != seen.end() ? 2 : 0;
}

}  // namespace MonocularSfM
