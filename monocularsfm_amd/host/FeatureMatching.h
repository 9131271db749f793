// FeatureMatching.h -- the reference's matcher classes (include/Feature/FeatureMatching.h:18-130)
// with the same constructor arguments and defaults, running on the gfx950 C ABI
// (include/msfm_match.h).  Differences from the reference, all behind the same interface:
//   * descriptors are uploaded to the GPU once per image instead of being re-read from SQLite
//     for every pair (the "TODO: cache" at src/Feature/FeatureMatching.cpp:31);
//   * the pairs of a whole run are matched as ONE streaming series per GPU (msfm_match_pairs_begin / _next) on device threads,
//     while the calling thread -- the only one that touches SQLite -- writes the rows of the pairs already finished, in the
//     reference's order, groups and transactions (FeatureMatching.cpp:13, 63-72: the reference emits inside its pair loop too).
#pragma once
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "Database.h"
#include "Types.h"
#include "msfm_match.h"

namespace MonocularSfM {

class FeatureMatcher {
public:
    FeatureMatcher(const std::string& database_path, const int& max_num_matches = 10240,
                   const double& max_distance = 0.7, const double& distance_ratio = 0.8,
                   const bool& cross_check = true);
    virtual ~FeatureMatcher();

    // FeatureMatching.cpp:10-73: skip pairs that already have a row, match, distance filter,
    // geometric verification, one matches row per pair (rows may be 0), all in one transaction.
    void MatchImagePairs(const std::vector<std::pair<image_t, image_t>>& image_pairs);
    // Any number of such calls at once: what consecutive MatchImagePairs calls would print and write, group by group (one
    // transaction each), while the GPUs are already working on the pairs of the later groups (see FeatureMatching.cpp).
    void MatchImagePairGroups(const std::vector<std::vector<std::pair<image_t, image_t>>>& groups);
    virtual void RunMatching() = 0;

    // FeatureUtils::FilterMatches (F-matrix RANSAC) is applied unless disabled (MSFM_GEOMETRIC_VERIFICATION=0):
    // on the device by default (msfm_match_pairs_verified), by the host twin with MSFM_GEOMETRIC_VERIFICATION=host
    // (GeometricVerification.h; the two give identical lists).
    void SetGeometricVerification(bool on) { geometric_verification_ = on; }

protected:
    void OpenDatabaseAndDevice();
    void CloseDatabaseAndDevice();
    void EnsureResident(image_t image_id);
    // every image's descriptors (and keypoints) in ONE table sweep, uploaded straight from SQLite's
    // buffers; uses the optional `descriptors_u8` side table when the database has one.  MSFM_BULK_LOAD=0 falls back
    // to the per-image reads.
    void PreloadAllImages();
    const std::vector<KeyPoint>& KeyPointsOf(image_t image_id);  // read once per image

    std::string database_path_;
    int max_num_matches_;  // stored, never read -- as in the reference
    double max_distance_;
    double distance_ratio_;
    bool cross_check_;
    bool geometric_verification_ = true;
    bool verification_on_host_ = false;  // MSFM_GEOMETRIC_VERIFICATION=host
    Database* database_ = nullptr;
    // One context per GPU: MSFM_DEVICE (default 0), MSFM_DEVICES="0,1,..." or "all".  The whole descriptor store is replicated on
    // each; the pairs of a run are dealt to the devices in small cost-balanced blocks, round-robin, so that every device's results
    // arrive at the pace the emitter consumes them (SQLite stays on the calling thread).
    struct Device {
        msfm_ctx* ctx = nullptr;
        std::set<image_t> resident;
        std::set<image_t> top_scale;   // images whose top-scale subset lives in the auxiliary slot MSFM_MAX_IMAGES + id
    };
    std::vector<Device> devices_;
    msfm_ctx* ctx_ = nullptr;             // = devices_[0].ctx
    bool bulk_loaded_ = false;
    std::map<image_t, Descriptors> descriptor_cache_;   // host copies, kept only with several devices and without the bulk load
    void EnsureResidentOn(size_t device_index, image_t image_id);
    std::map<image_t, std::vector<KeyPoint>> keypoints_cache_;  // read once per image (verification, pre-emptive filter)
};

class SequentialFeatureMatcher : public FeatureMatcher {
public:
    SequentialFeatureMatcher(const std::string& database_path, const int& overlap = 3,
                             const int& max_num_matches = 10240, const double& max_distance = 0.7,
                             const double& distance_ratio = 0.8, const bool& cross_check = true)
        : FeatureMatcher(database_path, max_num_matches, max_distance, distance_ratio, cross_check),
          overlap_(overlap) {}
    void RunMatching() override;

private:
    int overlap_;
};

class BruteFeatureMatcher : public FeatureMatcher {
public:
    BruteFeatureMatcher(const std::string& database_path, const int& max_pairs_size = 100,
                        const bool& is_preemtive = true, const int& preemtive_num_features = 100,
                        const int& preemtive_min_num_matches = 4, const int& max_num_matches = 10240,
                        const double& max_distance = 0.7, const double& distance_ratio = 0.8,
                        const bool& cross_check = true)
        : FeatureMatcher(database_path, max_num_matches, max_distance, distance_ratio, cross_check),
          max_pairs_size_(max_pairs_size),
          is_preemtive_(is_preemtive),
          preemtive_num_features_(preemtive_num_features),
          preemtive_min_num_matches_(preemtive_min_num_matches) {}
    void RunMatching() override;

private:
    // Wu, "Towards Linear-Time Incremental Structure from Motion", 3DV 2013 (pre-emptive matching)
    std::vector<std::pair<image_t, image_t>> PreemptivelyFilterImagePairs(
        std::vector<std::pair<image_t, image_t>> image_pairs);
    void PreemptivelyFilterGroups(std::vector<std::vector<std::pair<image_t, image_t>>>* groups);
    std::vector<char> PreemptiveKeepFlags(const std::vector<std::pair<image_t, image_t>>& image_pairs);
    int GetTopScaleDescriptors(const image_t& image_id);  // returns the auxiliary store slot (on the first device)
    bool HasTopScaleDescriptorsCache(const image_t& image_id);

    int max_pairs_size_;
    bool is_preemtive_;
    int preemtive_num_features_;
    int preemtive_min_num_matches_;
};

}  // namespace MonocularSfM
