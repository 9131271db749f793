"""The concurrency pieces of the executable's run (monocularsfm_amd/host/Pipeline.h), built here with g++ -- once plain, once under
ThreadSanitizer -- and driven without a GPU or a database:

  * ChunkQueue: several device threads produce chunks of different sizes into bounded queues while ONE consumer takes them in the
    order the emitter does (round-robin over the devices' blocks): every chunk arrives once, in order, with its bytes intact; a queue
    over its cap stalls its producer (the bytes waiting never exceed cap + one chunk); Close() ends a drained queue, Abort() frees a
    blocked producer;
  * DeviceCrew: fn(g) runs exactly once per device and call, for thousands of calls, g = 0 on the calling thread;
  * DealBlockEnds: every pair in exactly one block, blocks ascending and within a pair of their cost share, a small job still gives
    every device a block.
The reference has no counterpart (one thread, one pair after the other: /root/reference/src/Feature/FeatureMatching.cpp:13-72)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(HERE, "..", "monocularsfm_amd", "host")

DRIVER = r"""
#include "Pipeline.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <random>
using namespace MonocularSfM;

static int fail(const char* what) { std::printf("FAILED: %s\n", what); return 1; }

int main() {
    // ---- ChunkQueue: G producers, one in-order consumer, a small cap
    const size_t G = 5, chunks_per_dev = 400, cap = 64 * 1024;
    std::vector<std::unique_ptr<ChunkQueue>> q;
    for (size_t g = 0; g < G; ++g) q.emplace_back(new ChunkQueue(cap));
    std::vector<std::atomic<long long>> waiting(G);
    for (auto& w : waiting) w = 0;
    std::atomic<long long> worst(0);
    std::vector<std::thread> producers;
    for (size_t g = 0; g < G; ++g)
        producers.emplace_back([&, g] {
            std::mt19937 rng((unsigned)g + 7);
            size_t first = 0;
            for (size_t c = 0; c < chunks_per_dev; ++c) {
                std::unique_ptr<ResultChunk> ch(new ResultChunk());
                ch->first = first;
                ch->n = 1 + rng() % 40;
                ch->offsets.resize(ch->n + 1);
                long long at = 0;
                for (size_t p = 0; p < ch->n; ++p) {
                    ch->offsets[p] = at;
                    at += rng() % 300;
                }
                ch->offsets[ch->n] = at;
                ch->rows.resize((size_t)at * 2);
                for (size_t i = 0; i < ch->rows.size(); ++i) ch->rows[i] = (point2D_t)(g * 1000003 + first * 31 + i);
                first += ch->n;
                const long long bytes = (long long)ch->Bytes();
                if (!q[g]->Push(std::move(ch))) return;
                const long long now = (waiting[g] += bytes);
                long long w = worst.load();
                while (now > w && !worst.compare_exchange_weak(w, now)) {}
            }
            q[g]->Close();
        });
    long long got_chunks = 0;
    std::vector<size_t> next_first(G, 0);
    std::vector<bool> open(G, true);
    size_t live = G;
    for (size_t turn = 0; live > 0; ++turn) {
        const size_t g = turn % G;
        if (!open[g]) continue;
        if (turn % 97 == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));   // a slow emitter now and then
        std::unique_ptr<ResultChunk> ch = q[g]->Pop();
        if (!ch) {
            open[g] = false;
            --live;
            continue;
        }
        waiting[g] -= (long long)ch->Bytes();
        if (ch->first != next_first[g]) return fail("chunks of a device out of order");
        next_first[g] += ch->n;
        if (ch->offsets.size() != ch->n + 1 || ch->rows.size() != (size_t)ch->offsets[ch->n] * 2) return fail("chunk shape");
        for (size_t i = 0; i < ch->rows.size(); i += 17)
            if (ch->rows[i] != (point2D_t)(g * 1000003 + ch->first * 31 + i)) return fail("chunk bytes");
        ++got_chunks;
    }
    for (auto& t : producers) t.join();
    if (got_chunks != (long long)(G * chunks_per_dev)) return fail("chunk count");
    // bounded: what waits in a queue never exceeds the cap by more than the chunks in the producer's hands (one being pushed, one counted late)
    if (worst.load() > (long long)cap + 3 * (40 * 300 * 8 + 1024)) { std::printf("worst %lld\n", worst.load()); return fail("queue not bounded"); }

    // ---- Abort() frees a producer blocked on a full queue; Pop() on a closed empty queue returns nullptr
    {
        ChunkQueue small(16);
        std::atomic<int> pushed(0), refused(0);
        std::thread p([&] {
            for (int k = 0; k < 3; ++k) {
                std::unique_ptr<ResultChunk> ch(new ResultChunk());
                ch->rows.resize(1000);
                if (small.Push(std::move(ch))) ++pushed; else { ++refused; return; }
            }
        });
        while (pushed.load() < 1) std::this_thread::yield();
        std::this_thread::sleep_for(std::chrono::milliseconds(20));   // the producer now blocks on its second chunk
        small.Abort();
        p.join();
        if (refused.load() != 1) return fail("Abort did not free the blocked producer");
        ChunkQueue empty(16);
        empty.Close();
        if (empty.Pop()) return fail("Pop on a closed empty queue");
    }

    // ---- DeviceCrew: fn(g) once per device and call
    for (size_t n : {(size_t)1, (size_t)2, (size_t)8}) {
        DeviceCrew crew(n);
        std::vector<long long> count(n, 0);
        const std::thread::id me = std::this_thread::get_id();
        bool zero_on_caller = true;
        for (int call = 0; call < 3000; ++call)
            crew.Run([&](size_t g) {
                count[g] += 1;                                  // (each g is touched by one thread at a time: no atomics needed -- TSan checks)
                if (g == 0 && std::this_thread::get_id() != me) zero_on_caller = false;
            });
        for (size_t g = 0; g < n; ++g)
            if (count[g] != 3000) return fail("DeviceCrew: a device missed a call");
        if (!zero_on_caller) return fail("DeviceCrew: g = 0 must run on the calling thread");
    }

    // ---- DealBlockEnds
    std::mt19937_64 rng(99);
    for (int trial = 0; trial < 300; ++trial) {
        const size_t P = trial < 5 ? (size_t)trial : 1 + rng() % 5000, G2 = 1 + rng() % 8, bp = 1 + rng() % 700;
        std::vector<double> cum(P + 1, 0.0);
        double biggest = 0;
        for (size_t w = 0; w < P; ++w) {
            const double c = 1.0 + (double)(rng() % 1000) * (double)(rng() % 1000);
            biggest = std::max(biggest, c);
            cum[w + 1] = cum[w] + c;
        }
        const std::vector<size_t> ends = DealBlockEnds(cum, G2, bp);
        if (P == 0) { if (!ends.empty()) return fail("deal of nothing"); continue; }
        if (ends.empty() || ends.back() != P) return fail("deal does not cover the pairs");
        size_t begin = 0;
        const double share = cum[P] / (double)ends.size();
        for (size_t b = 0; b < ends.size(); ++b) {
            if (ends[b] < begin) return fail("block ends not ascending");
            const double cost = cum[ends[b]] - cum[begin];
            if (cost > share + 2 * biggest + 1e-6) return fail("a block far above its cost share");
            begin = ends[b];
        }
        if (P >= 4 * G2 && ends.size() < 4 * G2 && bp * ends.size() < P) return fail("too few blocks for the devices");
        if (ends.size() > P) return fail("more blocks than pairs");
    }
    std::printf("pipeline ok: %lld chunks through %zu bounded queues (worst %lld bytes waiting)\n", got_chunks, G, worst.load());
    return 0;
}
"""


@pytest.mark.parametrize("sanitizer", ["", "thread"])
def test_pipeline_pieces(tmp_path, sanitizer):
    src = tmp_path / "pipeline_driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / ("pipeline_driver_" + (sanitizer or "plain"))
    flags = ["-O1", "-g", "-std=c++17", "-pthread", "-I", HOST]
    if sanitizer:
        flags += ["-fsanitize=" + sanitizer]
    r = subprocess.run(["g++"] + flags + ["-o", str(exe), str(src)], capture_output=True, text=True)
    if r.returncode != 0 and sanitizer:
        pytest.skip("g++ cannot link -fsanitize=%s here: %s" % (sanitizer, r.stderr[-200:]))
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "pipeline ok" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
