// HostTestApi.cpp -- C-linkage shims over the host-only pieces (YAML reader, Database, F-RANSAC)
// so that the CPU test-suite can exercise them through ctypes without a GPU.
#include <cstring>
#include <string>
#include <vector>

#include "Database.h"
#include "GeometricVerification.h"
#include "YamlConfig.h"

using namespace MonocularSfM;

extern "C" {

int host_yaml_is_opened(const char* path) {
    msfm_host::YamlConfig y;
    y.Open(path);
    return y.isOpened() ? 1 : 0;
}

int host_yaml_get_string(const char* path, const char* key, char* out, int cap) {
    msfm_host::YamlConfig y;
    if (!y.Open(path)) return -1;
    std::string v;
    y.Get(key, &v);
    std::strncpy(out, v.c_str(), (size_t)cap - 1);
    out[cap - 1] = 0;
    return y.Has(key) ? 1 : 0;
}

double host_yaml_get_double(const char* path, const char* key, double dflt) {
    msfm_host::YamlConfig y;
    y.Open(path);
    double v = dflt;
    y.Get(key, &v);
    return v;
}

int host_yaml_get_int(const char* path, const char* key, int dflt) {
    msfm_host::YamlConfig y;
    y.Open(path);
    int v = dflt;
    y.Get(key, &v);
    return v;
}

int host_yaml_get_bool(const char* path, const char* key, int dflt) {
    msfm_host::YamlConfig y;
    y.Open(path);
    bool v = dflt != 0;
    y.Get(key, &v);
    return v ? 1 : 0;
}

int host_db_num_images(const char* path) {
    Database db;
    db.Open(path);
    const int n = (int)db.ReadAllImages().size();
    db.Close();
    return n;
}

// returns rows; copies min(rows*cols, cap) floats
int host_db_read_descriptors(const char* path, int image_id, float* out, long long cap, int* cols) {
    Database db;
    db.Open(path);
    const Descriptors d = db.ReadDescriptors(image_id);
    db.Close();
    *cols = d.cols;
    const long long n = std::min<long long>((long long)d.data.size(), cap);
    if (n > 0) std::memcpy(out, d.data.data(), (size_t)n * 4);
    return d.rows;
}

int host_db_read_keypoints(const char* path, int image_id, float* out, long long cap) {
    Database db;
    db.Open(path);
    const std::vector<KeyPoint> k = db.ReadKeyPoints(image_id);
    db.Close();
    const long long n = std::min<long long>((long long)k.size() * 4, cap);
    if (n > 0) std::memcpy(out, k.data(), (size_t)n * 4);
    return (int)k.size();
}

int host_db_write_matches(const char* path, int id1, int id2, const int* qt, int m) {
    Database db;
    db.Open(path);
    std::vector<DMatch> ms((size_t)m);
    for (int i = 0; i < m; ++i) {
        ms[(size_t)i].queryIdx = qt[2 * i];
        ms[(size_t)i].trainIdx = qt[2 * i + 1];
    }
    db.BeginTransaction();
    db.WriteMatches(id1, id2, ms);
    db.EndTransaction();
    db.Close();
    return 0;
}

int host_db_exist_matches(const char* path, int id1, int id2) {
    Database db;
    db.Open(path);
    const int e = db.ExistMatches(id1, id2) ? 1 : 0;
    db.Close();
    return e;
}

// returns m; writes (queryIdx, trainIdx) pairs as seen from (id1, id2)
int host_db_read_matches(const char* path, int id1, int id2, int* qt, int cap_pairs) {
    Database db;
    db.Open(path);
    const std::vector<DMatch> ms = db.ReadMatches(id1, id2);
    db.Close();
    for (size_t i = 0; i < ms.size() && (int)i < cap_pairs; ++i) {
        qt[2 * i] = ms[i].queryIdx;
        qt[2 * i + 1] = ms[i].trainIdx;
    }
    return (int)ms.size();
}

int host_pair_id(int id1, int id2) { return Database::ImagePairToPairId(id1, id2); }

// mask length n (0/1); returns number of mask entries written (0 if no model)
int host_fundamental_ransac(const float* p1, const float* p2, int n, unsigned char* mask) {
    std::vector<Point2f> a((size_t)n), b((size_t)n);
    for (int i = 0; i < n; ++i) {
        a[(size_t)i] = Point2f{p1[2 * i], p1[2 * i + 1]};
        b[(size_t)i] = Point2f{p2[2 * i], p2[2 * i + 1]};
    }
    const std::vector<unsigned char> m = FundamentalRansacMask(a, b);
    for (size_t i = 0; i < m.size(); ++i) mask[i] = m[i];
    return (int)m.size();
}

}  // extern "C"
