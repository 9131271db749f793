#include "SqliteDyn.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>

namespace msfm_host {

const SqliteApi& Sqlite() {
    static SqliteApi api;
    static bool loaded = false;
    if (loaded) return api;
    const char* names[] = {"libsqlite3.so.0", "libsqlite3.so", "/opt/conda/lib/libsqlite3.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        std::fprintf(stderr, "cannot load libsqlite3: %s\n", dlerror());
        std::exit(EXIT_FAILURE);
    }
#define BIND(field, sym)                                                        \
    do {                                                                        \
        *reinterpret_cast<void**>(&api.field) = dlsym(h, sym);                  \
        if (!api.field) {                                                       \
            std::fprintf(stderr, "libsqlite3 lacks %s\n", sym);                 \
            std::exit(EXIT_FAILURE);                                            \
        }                                                                       \
    } while (0)
    BIND(open_v2, "sqlite3_open_v2");
    BIND(close_v2, "sqlite3_close_v2");
    BIND(exec, "sqlite3_exec");
    BIND(free, "sqlite3_free");
    BIND(prepare_v2, "sqlite3_prepare_v2");
    BIND(bind_int64, "sqlite3_bind_int64");
    BIND(bind_blob, "sqlite3_bind_blob");
    BIND(bind_text, "sqlite3_bind_text");
    BIND(bind_null, "sqlite3_bind_null");
    BIND(step, "sqlite3_step");
    BIND(reset, "sqlite3_reset");
    BIND(finalize, "sqlite3_finalize");
    BIND(column_int64, "sqlite3_column_int64");
    BIND(column_int, "sqlite3_column_int");
    BIND(column_bytes, "sqlite3_column_bytes");
    BIND(column_blob, "sqlite3_column_blob");
    BIND(column_text, "sqlite3_column_text");
    BIND(errstr, "sqlite3_errstr");
    BIND(errmsg, "sqlite3_errmsg");
    BIND(last_insert_rowid, "sqlite3_last_insert_rowid");
    BIND(libversion, "sqlite3_libversion");
#undef BIND
    loaded = true;
    return api;
}

}  // namespace msfm_host
