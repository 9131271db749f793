// Database.cpp -- see Database.h.  Written against the behavioural contract of the reference's
// src/Database/Database.cpp (tables :710-764, pragmas :299-302, statements :779-892, blob codec
// :230-278, pair ids :656-694); errors print and exit like the reference's SQLITE3_CALL (:8-22).
#include "Database.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace MonocularSfM {

using msfm_host::Sqlite;

namespace {

const size_t kMaxNumImages = 10000;

int Check(int rc, const char* what, int line) {
    if (rc == msfm_host::SQLITE_OK_ || rc == msfm_host::SQLITE_ROW_ || rc == msfm_host::SQLITE_DONE_) return rc;
    std::fprintf(stderr, "SQLite error [%s, line %i]: %s\n", what, line, Sqlite().errstr(rc));
    std::exit(EXIT_FAILURE);
}
#define SQL_CALL(expr) Check((expr), __FILE__, __LINE__)

void Exec(sqlite3* db, const char* sql) {
    char* err = nullptr;
    const int rc = Sqlite().exec(db, sql, nullptr, nullptr, &err);
    if (rc != msfm_host::SQLITE_OK_) {  // the reference only reports exec failures (SQLITE3_EXEC :26-36)
        std::fprintf(stderr, "SQLite error [%s]: %s\n", sql, err ? err : "?");
        if (err) Sqlite().free(err);
    }
}

// rows, cols, data columns starting at `col` -> raw bytes (checked against rows*cols*elem)
template <typename T>
std::vector<T> ReadBlob(sqlite3_stmt* stmt, int rc, int col, size_t* rows, size_t* cols) {
    assert(rc == msfm_host::SQLITE_ROW_);
    (void)rc;
    *rows = (size_t)Sqlite().column_int64(stmt, col + 0);
    *cols = (size_t)Sqlite().column_int64(stmt, col + 1);
    const size_t num_bytes = (size_t)Sqlite().column_bytes(stmt, col + 2);
    if ((*rows) * (*cols) * sizeof(T) != num_bytes) {
        std::fprintf(stderr, "Database: blob of %zu bytes does not match rows=%zu cols=%zu\n", num_bytes, *rows, *cols);
        std::exit(EXIT_FAILURE);
    }
    std::vector<T> out((*rows) * (*cols));
    if (num_bytes) std::memcpy(out.data(), Sqlite().column_blob(stmt, col + 2), num_bytes);
    return out;
}

template <typename T>
void WriteBlob(sqlite3_stmt* stmt, const T* data, size_t rows, size_t cols, int col) {
    SQL_CALL(Sqlite().bind_int64(stmt, col + 0, (int64_t)rows));
    SQL_CALL(Sqlite().bind_int64(stmt, col + 1, (int64_t)cols));
    // The reference binds malloc(rows * cols * sizeof(T)) (Database.cpp:57-61, 270-277): for an empty list that is
    // a non-null pointer with length 0, i.e. a zero-length BLOB -- not NULL, which a null data pointer would give.
    static const char kEmpty = 0;
    const void* ptr = (rows * cols == 0 || data == nullptr) ? static_cast<const void*>(&kEmpty) : static_cast<const void*>(data);
    SQL_CALL(Sqlite().bind_blob(stmt, col + 2, ptr, (int)(rows * cols * sizeof(T)), nullptr /* SQLITE_STATIC */));
}

}  // namespace

Database::Database() : database_(nullptr) {}
Database::~Database() { Close(); }

void Database::Open(const std::string& path) {
    SQL_CALL(Sqlite().open_v2(path.c_str(), &database_,
                              msfm_host::SQLITE_OPEN_CREATE_ | msfm_host::SQLITE_OPEN_READWRITE_ |
                                  msfm_host::SQLITE_OPEN_NOMUTEX_,
                              nullptr));
    Exec(database_, "PRAGMA synchronous=OFF");
    Exec(database_, "PRAGMA journal_mode=WAL");
    Exec(database_, "PRAGMA temp_store=MEMORY");
    Exec(database_, "PRAGMA foreign_keys=ON");
    // Connection-local read path, not part of the on-disk contract: SQLite maps the file instead of copying every page through its
    // cache with pread (the bulk load reads the whole descriptors table once: 341 MB for the South-Building-shaped database).
    // MSFM_SQLITE_MMAP=0: off.
    {
        const char* e = std::getenv("MSFM_SQLITE_MMAP");
        if (!(e && e[0] == '0')) Exec(database_, "PRAGMA mmap_size=4294967296");
    }
    // MSFM_SQLITE_PRAGMAS="name=value;name=value": further connection-local pragmas (experiments with the write path: cache_size,
    // wal_autocheckpoint, locking_mode -- tools/emit_bench.cpp); applied after the reference's four, before the tables are touched
    if (const char* extra = std::getenv("MSFM_SQLITE_PRAGMAS")) {
        std::string all(extra);
        size_t at = 0;
        while (at < all.size()) {
            size_t end = all.find(';', at);
            if (end == std::string::npos) end = all.size();
            if (end > at) Exec(database_, ("PRAGMA " + all.substr(at, end - at)).c_str());
            at = end + 1;
        }
    }
    CreateTables();
    UpdateSchema();
    PrepareSQLStatements();
}

void Database::Close() {
    if (database_ != nullptr) {
        FinalizeSQLStatements();
        Sqlite().close_v2(database_);
        database_ = nullptr;
    }
}

void Database::BeginTransaction() const { Exec(database_, "BEGIN TRANSACTION"); }
void Database::EndTransaction() const { Exec(database_, "END TRANSACTION"); }

bool Database::ExistImageById(const image_t image_id) const { return ExistRowId(sql_stmt_exists_image_id_, image_id); }
bool Database::ExistKeyPoints(const image_t image_id) const { return ExistRowId(sql_stmt_exists_keypoints_, image_id); }
bool Database::ExistDescriptors(const image_t image_id) const { return ExistRowId(sql_stmt_exists_descriptors_, image_id); }
bool Database::ExistMatches(const image_pair_t pair_id) const { return ExistRowId(sql_stmt_exists_matches_, pair_id); }
bool Database::ExistMatches(const image_t image_id1, const image_t image_id2) const {
    return ExistMatches(ImagePairToPairId(image_id1, image_id2));
}

size_t Database::NumImages() const { return CountRows("images"); }

size_t Database::NumMatches(const image_t image_id1, const image_t image_id2) const {
    SQL_CALL(Sqlite().bind_int64(sql_stmt_read_matches_num_, 1, ImagePairToPairId(image_id1, image_id2)));
    const int rc = SQL_CALL(Sqlite().step(sql_stmt_read_matches_num_));
    size_t num = 0;
    if (rc == msfm_host::SQLITE_ROW_) num = (size_t)Sqlite().column_int64(sql_stmt_read_matches_num_, 0);
    SQL_CALL(Sqlite().reset(sql_stmt_read_matches_num_));
    return num;
}

std::vector<Database::Image> Database::ReadAllImages() const {
    std::vector<Image> images;
    images.reserve(NumImages());
    while (SQL_CALL(Sqlite().step(sql_stmt_read_images_)) == msfm_host::SQLITE_ROW_) {
        Image im;
        im.id = (image_t)Sqlite().column_int64(sql_stmt_read_images_, 0);
        im.name = reinterpret_cast<const char*>(Sqlite().column_text(sql_stmt_read_images_, 1));
        images.push_back(im);
    }
    SQL_CALL(Sqlite().reset(sql_stmt_read_images_));
    return images;
}

std::vector<KeyPoint> Database::ReadKeyPoints(const image_t image_id) const {
    SQL_CALL(Sqlite().bind_int64(sql_stmt_read_keypoints_, 1, image_id));
    const int rc = SQL_CALL(Sqlite().step(sql_stmt_read_keypoints_));
    size_t rows = 0, cols = 0;
    std::vector<float> blob = ReadBlob<float>(sql_stmt_read_keypoints_, rc, 0, &rows, &cols);
    SQL_CALL(Sqlite().reset(sql_stmt_read_keypoints_));
    assert(cols == 4 || rows == 0);
    std::vector<KeyPoint> kps(rows);
    for (size_t i = 0; i < rows; ++i) kps[i] = KeyPoint{blob[4 * i], blob[4 * i + 1], blob[4 * i + 2], blob[4 * i + 3]};
    return kps;
}

Descriptors Database::ReadDescriptors(const image_t image_id) const {
    SQL_CALL(Sqlite().bind_int64(sql_stmt_read_descriptors_, 1, image_id));
    const int rc = SQL_CALL(Sqlite().step(sql_stmt_read_descriptors_));
    size_t rows = 0, cols = 0;
    Descriptors d;
    d.data = ReadBlob<float>(sql_stmt_read_descriptors_, rc, 0, &rows, &cols);
    SQL_CALL(Sqlite().reset(sql_stmt_read_descriptors_));
    d.rows = (int)rows;
    d.cols = (int)cols;
    return d;
}

namespace {
size_t VisitTable(sqlite3* db, const char* sql, size_t elem_size, Database::BlobVisitor visit, void* user) {
    sqlite3_stmt* stmt = nullptr;
    SQL_CALL(Sqlite().prepare_v2(db, sql, -1, &stmt, nullptr));
    size_t n = 0;
    int rc;
    while ((rc = SQL_CALL(Sqlite().step(stmt))) == msfm_host::SQLITE_ROW_) {
        const image_t id = (image_t)Sqlite().column_int64(stmt, 0);
        const size_t rows = (size_t)Sqlite().column_int64(stmt, 1), cols = (size_t)Sqlite().column_int64(stmt, 2);
        const size_t num_bytes = (size_t)Sqlite().column_bytes(stmt, 3);
        if (rows * cols * elem_size != num_bytes) {
            std::fprintf(stderr, "Database: blob of %zu bytes does not match rows=%zu cols=%zu (image %d)\n", num_bytes, rows, cols, (int)id);
            std::exit(EXIT_FAILURE);
        }
        visit(user, id, num_bytes ? Sqlite().column_blob(stmt, 3) : nullptr, rows, cols, elem_size);
        ++n;
    }
    SQL_CALL(Sqlite().finalize(stmt));
    return n;
}
}  // namespace

size_t Database::VisitAllDescriptors(BlobVisitor visit, void* user) const {
    return VisitTable(database_, "SELECT image_id, rows, cols, data FROM descriptors ORDER BY image_id;", sizeof(float), visit, user);
}

size_t Database::VisitAllKeyPoints(BlobVisitor visit, void* user) const {
    return VisitTable(database_, "SELECT image_id, rows, cols, data FROM keypoints ORDER BY image_id;", sizeof(float), visit, user);
}

bool Database::HasDescriptorsU8() const {
    sqlite3_stmt* stmt = nullptr;
    SQL_CALL(Sqlite().prepare_v2(database_, "SELECT 1 FROM sqlite_master WHERE type = 'table' AND name = 'descriptors_u8';", -1, &stmt, nullptr));
    const bool has = SQL_CALL(Sqlite().step(stmt)) == msfm_host::SQLITE_ROW_;
    SQL_CALL(Sqlite().finalize(stmt));
    return has;
}

void Database::CreateDescriptorsU8Table() const {
    Exec(database_,
         "CREATE TABLE IF NOT EXISTS descriptors_u8"
         "  (image_id    INTEGER    PRIMARY KEY    NOT NULL,"
         "   rows        INTEGER                   NOT NULL,"
         "   cols        INTEGER                   NOT NULL,"
         "   data        BLOB,"
         "FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE)");
}

void Database::WriteDescriptorsU8(const image_t image_id, const unsigned char* data, size_t rows, size_t cols) const {
    sqlite3_stmt* stmt = nullptr;
    SQL_CALL(Sqlite().prepare_v2(database_, "INSERT OR REPLACE INTO descriptors_u8(image_id, rows, cols, data) VALUES(?, ?, ?, ?);", -1, &stmt, nullptr));
    SQL_CALL(Sqlite().bind_int64(stmt, 1, image_id));
    WriteBlob<unsigned char>(stmt, data, rows, cols, 2);
    SQL_CALL(Sqlite().step(stmt));
    SQL_CALL(Sqlite().finalize(stmt));
}

size_t Database::VisitAllDescriptorsU8(BlobVisitor visit, void* user) const {
    return VisitTable(database_, "SELECT image_id, rows, cols, data FROM descriptors_u8 ORDER BY image_id;", 1, visit, user);
}

std::vector<DMatch> Database::ReadMatches(const image_t image_id1, const image_t image_id2) const {
    SQL_CALL(Sqlite().bind_int64(sql_stmt_read_matches_, 1, ImagePairToPairId(image_id1, image_id2)));
    const int rc = SQL_CALL(Sqlite().step(sql_stmt_read_matches_));
    size_t rows = 0, cols = 0;
    std::vector<point2D_t> blob = ReadBlob<point2D_t>(sql_stmt_read_matches_, rc, 0, &rows, &cols);
    SQL_CALL(Sqlite().reset(sql_stmt_read_matches_));
    const bool swap = SwapImagePair(image_id1, image_id2);
    std::vector<DMatch> matches(rows);
    for (size_t i = 0; i < rows; ++i) {
        matches[i].queryIdx = blob[2 * i + (swap ? 1 : 0)];
        matches[i].trainIdx = blob[2 * i + (swap ? 0 : 1)];
    }
    return matches;
}

std::vector<std::pair<image_pair_t, std::vector<DMatch>>> Database::ReadAllMatches() const {
    std::vector<std::pair<image_pair_t, std::vector<DMatch>>> results;
    int rc;
    while ((rc = SQL_CALL(Sqlite().step(sql_stmt_read_matches_all_))) == msfm_host::SQLITE_ROW_) {
        const image_pair_t pair_id = (image_pair_t)Sqlite().column_int64(sql_stmt_read_matches_all_, 0);
        size_t rows = 0, cols = 0;
        std::vector<point2D_t> blob = ReadBlob<point2D_t>(sql_stmt_read_matches_all_, rc, 1, &rows, &cols);
        std::vector<DMatch> m(rows);
        for (size_t i = 0; i < rows; ++i) {
            m[i].queryIdx = blob[2 * i];
            m[i].trainIdx = blob[2 * i + 1];
        }
        results.emplace_back(pair_id, std::move(m));
    }
    SQL_CALL(Sqlite().reset(sql_stmt_read_matches_all_));
    return results;
}

image_t Database::WriteImage(const Image& image, const bool use_image_id) const {
    if (use_image_id) {
        assert(!ExistImageById(image.id));
        SQL_CALL(Sqlite().bind_int64(sql_stmt_add_image_, 1, image.id));
    } else {
        SQL_CALL(Sqlite().bind_null(sql_stmt_add_image_, 1));
    }
    SQL_CALL(Sqlite().bind_text(sql_stmt_add_image_, 2, image.name.c_str(), (int)image.name.size(), nullptr));
    SQL_CALL(Sqlite().step(sql_stmt_add_image_));
    SQL_CALL(Sqlite().reset(sql_stmt_add_image_));
    return (image_t)Sqlite().last_insert_rowid(database_);
}

void Database::WriteKeyPoints(const image_t image_id, const std::vector<KeyPoint>& keypoints) const {
    static_assert(sizeof(KeyPoint) == 16, "KeyPoint must be 4 packed floats");
    SQL_CALL(Sqlite().bind_int64(sql_stmt_add_keypoints_, 1, image_id));
    WriteBlob<float>(sql_stmt_add_keypoints_, reinterpret_cast<const float*>(keypoints.data()), keypoints.size(), 4, 2);
    SQL_CALL(Sqlite().step(sql_stmt_add_keypoints_));
    SQL_CALL(Sqlite().reset(sql_stmt_add_keypoints_));
}

void Database::WriteDescriptors(const image_t image_id, const Descriptors& d) const {
    SQL_CALL(Sqlite().bind_int64(sql_stmt_add_descriptors_, 1, image_id));
    WriteBlob<float>(sql_stmt_add_descriptors_, d.data.data(), (size_t)d.rows, (size_t)d.cols, 2);
    SQL_CALL(Sqlite().step(sql_stmt_add_descriptors_));
    SQL_CALL(Sqlite().reset(sql_stmt_add_descriptors_));
}

void Database::WriteMatches(const image_t image_id1, const image_t image_id2, const std::vector<DMatch>& matches) const {
    SQL_CALL(Sqlite().bind_int64(sql_stmt_add_matches_, 1, ImagePairToPairId(image_id1, image_id2)));
    // column 0 always indexes the image with the smaller id (SwapMatchesBlob, Database.cpp:633-640)
    const bool swap = SwapImagePair(image_id1, image_id2);
    std::vector<point2D_t> blob(matches.size() * 2);
    for (size_t i = 0; i < matches.size(); ++i) {
        blob[2 * i + (swap ? 1 : 0)] = matches[i].queryIdx;
        blob[2 * i + (swap ? 0 : 1)] = matches[i].trainIdx;
    }
    WriteBlob<point2D_t>(sql_stmt_add_matches_, blob.data(), matches.size(), 2, 2);
    SQL_CALL(Sqlite().step(sql_stmt_add_matches_));
    SQL_CALL(Sqlite().reset(sql_stmt_add_matches_));
}

void Database::WriteMatchesStored(const image_t image_id1, const image_t image_id2, const point2D_t* stored_rows, size_t count) const {
    SQL_CALL(Sqlite().bind_int64(sql_stmt_add_matches_, 1, ImagePairToPairId(image_id1, image_id2)));
    WriteBlob<point2D_t>(sql_stmt_add_matches_, stored_rows, count, 2, 2);   // bound SQLITE_STATIC: the caller's rows must outlive the step below
    SQL_CALL(Sqlite().step(sql_stmt_add_matches_));
    SQL_CALL(Sqlite().reset(sql_stmt_add_matches_));
}

std::vector<image_pair_t> Database::ReadAllMatchPairIds() const {
    std::vector<image_pair_t> ids;
    sqlite3_stmt* stmt = nullptr;
    SQL_CALL(Sqlite().prepare_v2(database_, "SELECT pair_id FROM matches ORDER BY pair_id;", -1, &stmt, nullptr));
    while (SQL_CALL(Sqlite().step(stmt)) == msfm_host::SQLITE_ROW_) ids.push_back((image_pair_t)Sqlite().column_int64(stmt, 0));
    SQL_CALL(Sqlite().finalize(stmt));
    return ids;
}

image_pair_t Database::ImagePairToPairId(const image_t image_id1, const image_t image_id2) {
    assert(image_id1 >= 0 && image_id2 >= 0);
    assert((size_t)image_id1 < kMaxNumImages && (size_t)image_id2 < kMaxNumImages);
    if (SwapImagePair(image_id1, image_id2)) return (image_pair_t)(kMaxNumImages * image_id2 + image_id1);
    return (image_pair_t)(kMaxNumImages * image_id1 + image_id2);
}

void Database::PairIdToImagePair(const image_pair_t pair_id, image_t* image_id1, image_t* image_id2) {
    *image_id2 = (image_t)(pair_id % kMaxNumImages);
    *image_id1 = (image_t)((pair_id - *image_id2) / kMaxNumImages);
}

bool Database::SwapImagePair(const image_t image_id1, const image_t image_id2) { return image_id1 > image_id2; }

void Database::CreateTables() const {
    Exec(database_,
         "CREATE TABLE IF NOT EXISTS images"
         "(  image_id  INTEGER PRIMARY KEY AUTOINCREMENT   NOT NULL,"
         "   name      TEXT                                NOT NULL UNIQUE)");
    const char* per_image[] = {"keypoints", "colors", "descriptors"};
    for (const char* t : per_image) {
        const std::string sql = std::string("CREATE TABLE IF NOT EXISTS ") + t +
                                "  (image_id    INTEGER    PRIMARY KEY    NOT NULL,"
                                "   rows        INTEGER                   NOT NULL,"
                                "   cols        INTEGER                   NOT NULL,"
                                "   data        BLOB,"
                                "FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE)";
        Exec(database_, sql.c_str());
    }
    Exec(database_,
         "CREATE TABLE IF NOT EXISTS matches"
         "   (pair_id    INTEGER    PRIMARY KEY    NOT NULL,"
         "    rows       INTEGER                   NOT NULL,"
         "    cols       INTEGER                   NOT NULL,"
         "    data       BLOB);");
}

void Database::UpdateSchema() const {
    const std::string sql = "PRAGMA user_version = " + std::to_string(kSchemaVersion) + ";";
    Exec(database_, sql.c_str());
}

void Database::PrepareSQLStatements() {
    sql_stmts_.clear();
    auto prep = [this](const char* sql, sqlite3_stmt** stmt) {
        SQL_CALL(Sqlite().prepare_v2(database_, sql, -1, stmt, nullptr));
        sql_stmts_.push_back(*stmt);
    };
    prep("SELECT 1 FROM images WHERE image_id = ?;", &sql_stmt_exists_image_id_);
    prep("SELECT 1 FROM keypoints WHERE image_id = ?;", &sql_stmt_exists_keypoints_);
    prep("SELECT 1 FROM descriptors WHERE image_id = ?;", &sql_stmt_exists_descriptors_);
    prep("SELECT 1 FROM matches WHERE pair_id = ?;", &sql_stmt_exists_matches_);
    prep("SELECT * FROM images;", &sql_stmt_read_images_);
    prep("SELECT rows, cols, data FROM keypoints WHERE image_id = ?;", &sql_stmt_read_keypoints_);
    prep("SELECT rows, cols, data FROM descriptors WHERE image_id = ?;", &sql_stmt_read_descriptors_);
    prep("SELECT rows, cols, data FROM matches WHERE pair_id = ?;", &sql_stmt_read_matches_);
    prep("SELECT rows FROM matches WHERE pair_id = ?;", &sql_stmt_read_matches_num_);
    prep("SELECT * FROM matches WHERE rows > 0;", &sql_stmt_read_matches_all_);
    prep("INSERT INTO images(image_id, name) VALUES(?, ?);", &sql_stmt_add_image_);
    prep("INSERT INTO keypoints(image_id, rows, cols, data) VALUES(?, ?, ?, ?);", &sql_stmt_add_keypoints_);
    prep("INSERT INTO descriptors(image_id, rows, cols, data) VALUES(?, ?, ?, ?);", &sql_stmt_add_descriptors_);
    prep("INSERT INTO matches(pair_id, rows, cols, data) VALUES(?, ?, ?, ?);", &sql_stmt_add_matches_);
}

void Database::FinalizeSQLStatements() {
    for (sqlite3_stmt* s : sql_stmts_) SQL_CALL(Sqlite().finalize(s));
    sql_stmts_.clear();
}

bool Database::ExistRowId(sqlite3_stmt* sql_stmt, const size_t row_id) const {
    SQL_CALL(Sqlite().bind_int64(sql_stmt, 1, (int64_t)row_id));
    const int rc = SQL_CALL(Sqlite().step(sql_stmt));
    SQL_CALL(Sqlite().reset(sql_stmt));
    return rc == msfm_host::SQLITE_ROW_;
}

size_t Database::CountRows(const std::string& table) const {
    const std::string sql = "SELECT COUNT(*) FROM " + table + ";";
    sqlite3_stmt* stmt;
    SQL_CALL(Sqlite().prepare_v2(database_, sql.c_str(), -1, &stmt, nullptr));
    size_t count = 0;
    if (SQL_CALL(Sqlite().step(stmt)) == msfm_host::SQLITE_ROW_) count = (size_t)Sqlite().column_int64(stmt, 0);
    SQL_CALL(Sqlite().finalize(stmt));
    return count;
}

}  // namespace MonocularSfM
