// Pipeline.h -- the concurrency pieces of the executable's run (host/FeatureMatching.cpp): what travels from a device thread to the
// SQLite thread, the bounded queue between them, the helper threads that drive several devices at once, and the deal of the pairs to
// the devices.  No GPU, no SQLite in here: tests/test_host_pipeline.py builds it with g++ (-fsanitize=thread) and runs producers,
// consumers and crews against each other.
// The reference has no counterpart: it computes and writes one pair after the other on one thread
// (/root/reference/src/Feature/FeatureMatching.cpp:13-72).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "Types.h"

namespace MonocularSfM {

// What a device thread hands to the emitter: the lists of `n` consecutive pairs of ITS pair list, already in the stored layout
// (count x 2 int32, column 0 = the index in the image with the smaller id, Database.cpp:633-640).
struct ResultChunk {
    size_t first = 0, n = 0;             // local pair indices [first, first + n)
    std::vector<int64_t> offsets;        // n + 1
    std::vector<point2D_t> rows;         // 2 * offsets[n]
    double seconds_per_pair = 0;         // wall clock of the chunk on the device thread / n (the "Elapsed time" line of a pair)
    size_t Bytes() const { return rows.size() * sizeof(point2D_t) + offsets.size() * 8 + sizeof(*this); }
};

// Bounded single-producer / single-consumer queue: the producer blocks while more than `cap` bytes wait (emission is the slower side
// then, and the device idles rather than the host buffering the whole job); one chunk always fits.
class ChunkQueue {
public:
    explicit ChunkQueue(size_t cap_bytes) : cap_(cap_bytes) {}
    // false: the consumer has given up (error elsewhere): stop producing
    bool Push(std::unique_ptr<ResultChunk> c) {
        std::unique_lock<std::mutex> l(mu_);
        not_full_.wait(l, [&] { return aborted_ || q_.empty() || bytes_ <= cap_; });
        if (aborted_) return false;
        bytes_ += c->Bytes();
        q_.push_back(std::move(c));
        not_empty_.notify_one();
        return true;
    }
    // nullptr: the producer has finished (or failed) and nothing is left
    std::unique_ptr<ResultChunk> Pop() {
        std::unique_lock<std::mutex> l(mu_);
        not_empty_.wait(l, [&] { return closed_ || !q_.empty(); });
        if (q_.empty()) return nullptr;
        std::unique_ptr<ResultChunk> c = std::move(q_.front());
        q_.pop_front();
        bytes_ -= c->Bytes();
        not_full_.notify_one();
        return c;
    }
    void Close() {
        std::lock_guard<std::mutex> l(mu_);
        closed_ = true;
        not_empty_.notify_all();
    }
    void Abort() {
        std::lock_guard<std::mutex> l(mu_);
        aborted_ = true;
        not_full_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable not_empty_, not_full_;
    std::deque<std::unique_ptr<ResultChunk>> q_;
    size_t bytes_ = 0, cap_;
    bool closed_ = false, aborted_ = false;
};

// fn(g) for every device g at once: g = 0 on the calling thread, the others on persistent helper threads (a thread spawn per image and
// device of the bulk load would cost more than the uploads).  A context is only ever driven by the thread that runs ITS g.
class DeviceCrew {
public:
    explicit DeviceCrew(size_t n) : n_(n) {
        for (size_t g = 1; g < n_; ++g) helpers_.emplace_back([this, g] { Loop(g); });
    }
    ~DeviceCrew() {
        {
            std::lock_guard<std::mutex> l(mu_);
            quit_ = true;
            ++generation_;
        }
        start_.notify_all();
        for (auto& t : helpers_) t.join();
    }
    template <class F>
    void Run(F&& fn) {
        if (n_ <= 1) {
            fn((size_t)0);
            return;
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            fn_ = [&fn](size_t g) { fn(g); };
            pending_ = n_ - 1;
            ++generation_;
        }
        start_.notify_all();
        fn((size_t)0);
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [&] { return pending_ == 0; });
    }

private:
    void Loop(size_t g) {
        unsigned long long seen = 0;
        while (true) {
            std::function<void(size_t)> fn;
            {
                std::unique_lock<std::mutex> l(mu_);
                start_.wait(l, [&] { return generation_ != seen; });
                seen = generation_;
                if (quit_) return;
                fn = fn_;
            }
            fn(g);
            {
                std::lock_guard<std::mutex> l(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    size_t n_;
    std::vector<std::thread> helpers_;
    std::mutex mu_;
    std::condition_variable start_, done_;
    std::function<void(size_t)> fn_;
    size_t pending_ = 0;
    unsigned long long generation_ = 0;
    bool quit_ = false;
};


// The deal of a run's pairs (in the order they are computed; cum[w + 1] - cum[w] = cost of pair w, cum.size() = P + 1) to G devices:
// blocks of about `block_pairs` pairs -- at least four blocks per device where the pairs allow it, so that a small job still gives
// every device something -- and equal COST: block b ends at the pair nearest to its share of the total, block b goes to device b mod G.
// -> the end (exclusive) of every block, ascending, the last one P.
inline std::vector<size_t> DealBlockEnds(const std::vector<double>& cum, size_t G, size_t block_pairs) {
    const size_t P = cum.empty() ? 0 : cum.size() - 1;
    std::vector<size_t> ends;
    if (P == 0) return ends;
    G = std::max<size_t>(1, G);
    const size_t per_block = std::max<size_t>(1, std::min(std::max<size_t>(1, block_pairs), (P + 4 * G - 1) / (4 * G)));
    const size_t n_blocks = (P + per_block - 1) / per_block;
    size_t begin = 0;
    for (size_t b = 0; b < n_blocks; ++b) {
        size_t end = b + 1 == n_blocks ? P
                                       : (size_t)(std::lower_bound(cum.begin(), cum.end(), cum[P] * (double)(b + 1) / (double)n_blocks) - cum.begin());
        end = std::min(P, std::max(end, begin));
        ends.push_back(end);
        begin = end;
    }
    return ends;
}

}  // namespace MonocularSfM
