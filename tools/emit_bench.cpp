// emit_bench.cpp -- micro-benchmark of the executable's EMISSION path (no GPU): all pairs (i, j < i) of n_img images x m matches per pair
// through Database::WriteMatchesStored with the reference's pragmas, 100 rows per transaction, in the reference's order (i outermost:
// the row key kMaxNumImages * j + i jumps between ~n_img key bands) or -- EMIT_SORTED=1 -- in ascending key order (SQLite appends).
//   g++ -O2 -std=c++17 -Iinclude -Imonocularsfm_amd/host -o /tmp/emit_bench tools/emit_bench.cpp monocularsfm_amd/host/Database.cpp monocularsfm_amd/host/SqliteDyn.cpp -ldl
//   /tmp/emit_bench [n_img 600] [m 409] [stdout 1] [db 1] [path] > /dev/null      MSFM_SQLITE_PRAGMAS="a=b;c=d" adds connection pragmas
// Numbers: profiles/r06_emission_study.txt.  Diagnostic tool, not part of the product.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <iostream>
#include "Database.h"
#include "Timer.h"
using namespace MonocularSfM;
int main(int argc, char** argv) {
    const int n_img = argc > 1 ? atoi(argv[1]) : 600;
    const int m = argc > 2 ? atoi(argv[2]) : 409;
    const int do_stdout = argc > 3 ? atoi(argv[3]) : 1;
    const int do_db = argc > 4 ? atoi(argv[4]) : 1;
    const char* path = argc > 5 ? argv[5] : "/tmp/emit/bench.db";
    remove(path); remove((std::string(path) + "-wal").c_str()); remove((std::string(path) + "-shm").c_str());
    Database db; db.Open(path);
    if (const char* p = getenv("EMIT_PRAGMA")) { /* ';'-separated */ }
    std::vector<int> rows(2 * m);
    for (int i = 0; i < 2 * m; ++i) rows[i] = i * 7 % 8192;
    auto t0 = std::chrono::steady_clock::now();
    long long pairs = 0; std::string out; char buf[160];
    if (getenv("EMIT_SORTED")) {   // the same rows in ascending key order: j outermost
        for (int j = 0; j < n_img; ++j)
            for (int i0 = j + 1; i0 < n_img; i0 += 100) {
                if (do_db) db.BeginTransaction();
                for (int i = i0; i < n_img && i < i0 + 100; ++i) {
                    if (do_db) db.WriteMatchesStored(i, j, rows.data(), m);
                    ++pairs;
                }
                if (do_db) db.EndTransaction();
            }
    } else
    for (int i = 0; i < n_img; ++i) {
        for (int j0 = 0; j0 < i; j0 += 100) {
            if (do_db) db.BeginTransaction();
            out.clear();
            for (int j = j0; j < i && j < j0 + 100; ++j) {
                if (do_stdout) {
                    std::snprintf(buf, sizeof(buf), "Compute Matches %d - %d ... \n\t matches num : %zu\n\t ", i, j, (size_t)m);
                    out += buf; out += Timer::Format(0.0000123, "seconds"); out += "\n";
                }
                if (do_db) db.WriteMatchesStored(i, j, rows.data(), m);
                ++pairs;
            }
            if (do_stdout) std::cout << out << std::flush;
            if (do_db) db.EndTransaction();
        }
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto t1 = std::chrono::steady_clock::now();
    db.Close();
    double dc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    std::fprintf(stderr, "%lld pairs x %d matches: %.3f s (%.2f us per pair, %.1f ns per match), close %.3f s\n", pairs, m, dt, dt / pairs * 1e6, dt / pairs / m * 1e9, dc);
}
