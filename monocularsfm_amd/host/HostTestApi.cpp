// HostTestApi.cpp -- C-linkage shims over the host-only pieces (YAML reader, Database, F-RANSAC)
// so that the CPU test-suite can exercise them through ctypes without a GPU.
#include <cstring>
#include <string>
#include <vector>

#include "Database.h"
#include "GeometricVerification.h"
#include "MatchEmission.h"
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

// ---- bulk loader + u8 side table --------------------------------------------------------------------
int host_db_write_descriptors_u8(const char* path, int image_id, const unsigned char* data, int rows, int cols) {
    Database db;
    db.Open(path);
    db.CreateDescriptorsU8Table();
    db.BeginTransaction();
    db.WriteDescriptorsU8(image_id, data, (size_t)rows, (size_t)cols);
    db.EndTransaction();
    db.Close();
    return 0;
}

// One sweep over a table (which: 0 descriptors, 1 keypoints, 2 descriptors_u8): per row image id, rows, cols, a checksum
// of the blob bytes; returns the number of rows visited (-1: the u8 side table does not exist).
int host_db_visit_all(const char* path, int which, int* ids, int* rows, int* cols, unsigned long long* checksums, int cap) {
    struct Acc {
        int *ids, *rows, *cols;
        unsigned long long* sums;
        int cap, n;
    } acc{ids, rows, cols, checksums, cap, 0};
    auto visit = [](void* user, image_t id, const void* data, size_t r, size_t c, size_t elem) {
        Acc* a = static_cast<Acc*>(user);
        if (a->n < a->cap) {
            unsigned long long h = 1469598103934665603ull;   // FNV-1a over the blob
            const unsigned char* b = static_cast<const unsigned char*>(data);
            for (size_t i = 0; i < r * c * elem; ++i) h = (h ^ b[i]) * 1099511628211ull;
            a->ids[a->n] = (int)id;
            a->rows[a->n] = (int)r;
            a->cols[a->n] = (int)c;
            a->sums[a->n] = h;
        }
        a->n += 1;
    };
    Database db;
    db.Open(path);
    size_t n = 0;
    if (which == 2 && !db.HasDescriptorsU8()) {
        db.Close();
        return -1;
    }
    if (which == 0) n = db.VisitAllDescriptors(visit, &acc);
    else if (which == 1) n = db.VisitAllKeyPoints(visit, &acc);
    else n = db.VisitAllDescriptorsU8(visit, &acc);
    db.Close();
    return (int)n;
}

// ---- row emission -------------------------------------------------------------------------------------
// in / out: m (queryIdx, trainIdx) pairs of pair (id1, id2); returns the new count
int host_apply_emission(int id1, int id2, int scene_graph_order, int min_num_matches, int* qt, int m) {
    std::vector<DMatch> ms((size_t)m);
    for (int i = 0; i < m; ++i) {
        ms[(size_t)i].queryIdx = qt[2 * i];
        ms[(size_t)i].trainIdx = qt[2 * i + 1];
    }
    EmissionOptions o;
    o.scene_graph_order = scene_graph_order != 0;
    o.min_num_matches = min_num_matches;
    ApplyEmissionOptions(o, id1, id2, &ms);
    for (size_t i = 0; i < ms.size(); ++i) {
        qt[2 * i] = ms[i].queryIdx;
        qt[2 * i + 1] = ms[i].trainIdx;
    }
    return (int)ms.size();
}

int host_check_row_contract(const int* qt, int m, int num_keypoints1, int num_keypoints2) {
    std::vector<DMatch> ms((size_t)m);
    for (int i = 0; i < m; ++i) {
        ms[(size_t)i].queryIdx = qt[2 * i];
        ms[(size_t)i].trainIdx = qt[2 * i + 1];
    }
    return CheckRowContract(ms, (size_t)num_keypoints1, (size_t)num_keypoints2);
}

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
