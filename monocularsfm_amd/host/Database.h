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

    // ---- bulk loader --------------------------------------------------------------------------------
    // One `SELECT image_id, rows, cols, data FROM descriptors ORDER BY image_id` sweep instead of one prepared-statement
    // read per image (the reference: two per PAIR, FeatureMatching.cpp:32-33; Database.cpp:482-523).  The callback gets
    // the BLOB in SQLite's own buffer (valid during the call only): the matcher uploads it to the GPU from there, the
    // rows never pass through a host-side container.  elem_size is 4 (float32, the reference's table) or 1 (uint8, the
    // optional side table below).  Returns the number of rows visited.
    typedef void (*BlobVisitor)(void* user, image_t image_id, const void* data, size_t rows, size_t cols, size_t elem_size);
    size_t VisitAllDescriptors(BlobVisitor visit, void* user) const;
    size_t VisitAllKeyPoints(BlobVisitor visit, void* user) const;
    // Optional side table `descriptors_u8(image_id, rows, cols, data BLOB)`: raw SIFT descriptors are integers 0..255; stored
    // as bytes they are a quarter of the float table and feed the library's uint8 upload (integer distances are exact
    // on the matrix cores).  Not part of the reference's schema: only read when present, only written on request.
    bool HasDescriptorsU8() const;
    void CreateDescriptorsU8Table() const;
    void WriteDescriptorsU8(const image_t image_id, const unsigned char* data, size_t rows, size_t cols) const;
    size_t VisitAllDescriptorsU8(BlobVisitor visit, void* user) const;

    image_t WriteImage(const Image& image, const bool use_image_id = false) const;
    void WriteKeyPoints(const image_t image_id, const std::vector<KeyPoint>& keypoints) const;
    void WriteDescriptors(const image_t image_id, const Descriptors& descriptors) const;
    void WriteMatches(const image_t image_id1, const image_t image_id2, const std::vector<DMatch>& matches) const;
    // The same row from a list that already has the STORED layout (count x 2, column 0 = the index in the image with the smaller id,
    // Database.cpp:633-640): what WriteMatches builds from the DMatch vector, without the vector -- the matcher's device threads lay the
    // rows out while the SQLite thread writes the previous ones (host/FeatureMatching.cpp).
    void WriteMatchesStored(const image_t image_id1, const image_t image_id2, const point2D_t* stored_rows, size_t count) const;
    // Every pair_id that has a `matches` row, ascending: one index sweep instead of one ExistMatches SELECT per pair of a large job
    // (FeatureMatcher::MatchImagePairs asks per pair, src/Feature/FeatureMatching.cpp:21-25).
    std::vector<image_pair_t> ReadAllMatchPairIds() const;

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
