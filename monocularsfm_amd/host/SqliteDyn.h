// SqliteDyn.h -- the ~20 SQLite C API entry points the reference's Database.cpp uses, bound at run
// time from the system libsqlite3.so.0 with self-declared prototypes (the image ships the shared
// object but no headers; the reference vendors a SQLite amalgamation whose sqlite3.c is absent).
#pragma once
#include <cstdint>

struct sqlite3;
struct sqlite3_stmt;

namespace msfm_host {

constexpr int SQLITE_OK_ = 0, SQLITE_ROW_ = 100, SQLITE_DONE_ = 101;
constexpr int SQLITE_OPEN_READWRITE_ = 0x2, SQLITE_OPEN_CREATE_ = 0x4, SQLITE_OPEN_NOMUTEX_ = 0x8000;

struct SqliteApi {
    int (*open_v2)(const char*, sqlite3**, int, const char*);
    int (*close_v2)(sqlite3*);
    int (*exec)(sqlite3*, const char*, int (*)(void*, int, char**, char**), void*, char**);
    void (*free)(void*);
    int (*prepare_v2)(sqlite3*, const char*, int, sqlite3_stmt**, const char**);
    int (*bind_int64)(sqlite3_stmt*, int, int64_t);
    int (*bind_blob)(sqlite3_stmt*, int, const void*, int, void (*)(void*));
    int (*bind_text)(sqlite3_stmt*, int, const char*, int, void (*)(void*));
    int (*bind_null)(sqlite3_stmt*, int);
    int (*step)(sqlite3_stmt*);
    int (*reset)(sqlite3_stmt*);
    int (*finalize)(sqlite3_stmt*);
    int64_t (*column_int64)(sqlite3_stmt*, int);
    int (*column_int)(sqlite3_stmt*, int);
    int (*column_bytes)(sqlite3_stmt*, int);
    const void* (*column_blob)(sqlite3_stmt*, int);
    const unsigned char* (*column_text)(sqlite3_stmt*, int);
    const char* (*errstr)(int);
    const char* (*errmsg)(sqlite3*);
    int64_t (*last_insert_rowid)(sqlite3*);
    const char* (*libversion)();
};

// Loads libsqlite3 on first use; prints to stderr and exits if it cannot be found (the reference
// exits on every SQLite error, Database.cpp:8-22).
const SqliteApi& Sqlite();

}  // namespace msfm_host
