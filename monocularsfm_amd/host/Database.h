// Database.h -- the reference's SQLite surface on the ComputeMatches path
// (include/Database/Database.h:17-134, src/Database/Database.cpp), same method names, same
// tables/pragmas/blob layout, OpenCV types replaced by the PODs in Types.h.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "SqliteDyn.h"
#include "Types.h"

namespace MonocularSfM {

class Database {
public:
    struct Image {
        image_t id;
        std::string name;
    };
    const static int kSchemaVersion = 1;

    Database();
    ~Database();
    void Open(const std::string& path);
    void Close();

    void BeginTransaction() const;
    void EndTransaction() const;

    bool ExistImageById(const image_t image_id) const;
    bool ExistKeyPoints(const image_t image_id) const;
    bool ExistDescriptors(const image_t image_id) const;
    bool ExistMatches(const image_pair_t pair_id) const;
    bool ExistMatches(const image_t image_id1, const image_t image_id2) const;

    size_t NumImages() const;
    size_t NumMatches(const image_t image_id1, const image_t image_id2) const;

    std::vector<Image> ReadAllImages() const;
    std::vector<KeyPoint> ReadKeyPoints(const image_t image_id) const;
    Descriptors ReadDescriptors(const image_t image_id) const;
    std::vector<DMatch> ReadMatches(const image_t image_id1, const image_t image_id2) const;
    std::vector<std::pair<image_pair_t, std::vector<DMatch>>> ReadAllMatches() const;

    image_t WriteImage(const Image& image, const bool use_image_id = false) const;
    void WriteKeyPoints(const image_t image_id, const std::vector<KeyPoint>& keypoints) const;
    void WriteDescriptors(const image_t image_id, const Descriptors& descriptors) const;
    void WriteMatches(const image_t image_id1, const image_t image_id2, const std::vector<DMatch>& matches) const;

    static image_pair_t ImagePairToPairId(const image_t image_id1, const image_t image_id2);
    static void PairIdToImagePair(const image_pair_t pair_id, image_t* image_id1, image_t* image_id2);
    static bool SwapImagePair(const image_t image_id1, const image_t image_id2);

private:
    void CreateTables() const;
    void UpdateSchema() const;
    void PrepareSQLStatements();
    void FinalizeSQLStatements();
    bool ExistRowId(sqlite3_stmt* sql_stmt, const size_t row_id) const;
    size_t CountRows(const std::string& table) const;

    sqlite3* database_;
    std::vector<sqlite3_stmt*> sql_stmts_;
    sqlite3_stmt *sql_stmt_exists_image_id_, *sql_stmt_exists_keypoints_, *sql_stmt_exists_descriptors_,
        *sql_stmt_exists_matches_;
    sqlite3_stmt *sql_stmt_read_images_, *sql_stmt_read_keypoints_, *sql_stmt_read_descriptors_,
        *sql_stmt_read_matches_, *sql_stmt_read_matches_num_, *sql_stmt_read_matches_all_;
    sqlite3_stmt *sql_stmt_add_image_, *sql_stmt_add_keypoints_, *sql_stmt_add_descriptors_, *sql_stmt_add_matches_;
};

}  // namespace MonocularSfM
