"""Pure host-testable pieces of the library (monocularsfm_amd/csrc/msfm_hostutil.h), compiled here with g++:

  * the 4-byte packing of the integer sweeps' column partials: the largest accumulator must survive exactly, the decoded second
    largest may only be SMALLER than the true one (it is used as an upper bound of the column's second-smallest distance: every
    pruning / threshold bound of the prefilter stays valid) and by at most 1/16 of the gap;
  * the cost marks of a large call's sub-batches (msfm_set_pipeline): increasing, ending at the total, parts shrinking towards the
    end, the last one `taper` of the average;
  * the device scratch one image pair adds to a sub-batch (what a call is cut by), and its Python twin in tools/config4_full.py;
  * the fold of a candidate's key into a slot's (best, second) in the exact re-check -- the device kernel calls this very function
    with atomicMin: 200 000 random arrival sequences with repeated keys, every arrival acting on a STALE look at the slot;
  * the room a sub-batch's sweep-2 plan gets before the plan exists (msfm_plan_room): the executable's order of calls -- the pre-emptive
    filter's subsets, then the full images -- with the numbers measured on the South-Building job; the page-locked pieces of the lists.
The reference's counterpart of the second is the fixed 100-pair flush of BruteFeatureMatcher::RunMatching
(/root/reference/src/Feature/FeatureMatching.cpp:118-139)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "monocularsfm_amd", "csrc")

DRIVER = r"""
#include "msfm_hostutil.h"
#include <cstdio>
#include <cstdlib>
#include <random>
int main() {
    std::mt19937_64 rng(12345);
    long long checked = 0;
    auto check = [&](int hi, int lo) -> bool {
        const int code = msfm_cp_pack(hi, true, lo, true);
        if (msfm_cp_hi(code) != hi) { std::printf("hi lost: %d -> %d\n", hi, msfm_cp_hi(code)); return false; }
        const int gap = hi - lo;
        if (!msfm_cp_has_second(code)) return gap > 1000000 - 64 || gap >= ((31 << 15) - 16 - (1 << 15));   // only huge gaps may drop the second
        const int dec = msfm_cp_gap(code);
        if (dec < gap) { std::printf("gap rounded DOWN: %d -> %d\n", gap, dec); return false; }
        if (gap < 16 && dec != gap) { std::printf("small gap not exact: %d -> %d\n", gap, dec); return false; }
        if ((long long)dec * 16 > (long long)(gap + 16) * 17) { std::printf("gap too coarse: %d -> %d\n", gap, dec); return false; }
        ++checked;
        return true;
    };
    for (int gap = 0; gap < 70000; ++gap)
        if (!check(-12345, -12345 - gap)) return 1;
    for (int k = 0; k < 2000000; ++k) {
        const int hi = 1 - (int)(rng() % 4200000ull);
        const int gap = (int)(rng() % (k & 1 ? 3000ull : 5000000ull));
        if (!check(hi, hi - gap)) return 1;
    }
    for (int e = 0; e < 23; ++e)
        for (int d = -2; d <= 2; ++d) {
            const int gap = (1 << e) + d;
            if (gap >= 0 && !check(0, -gap)) return 1;
        }
    // markers
    if (msfm_cp_hi(msfm_cp_pack(-5, false, -7, true)) != kCpNone) { std::printf("no-row marker\n"); return 1; }
    if (msfm_cp_has_second(msfm_cp_pack(-5, true, -7, false))) { std::printf("no-second marker\n"); return 1; }
    if (msfm_cp_hi(msfm_cp_pack(-4200000, true, -4200001, true)) != -4200000 || msfm_cp_hi(msfm_cp_pack(1, true, 0, true)) != 1) { std::printf("range\n"); return 1; }
    // marks
    const long long totals[] = {30000000001LL, 206310000000LL, 59220619689984LL};
    const double tapers[] = {0.05, 0.3, 1.0};
    for (long long total : totals)
        for (long long n = 2; n <= 9; ++n)
            for (double taper : tapers) {
                const std::vector<long long> m = msfm_pipeline_marks(total, n, taper);
                if ((long long)m.size() != n + 1 || m.front() != 0 || m.back() != total) { std::printf("marks ends\n"); return 1; }
                long long prev_part = -1;
                for (long long k = 0; k < n; ++k) {
                    const long long part = m[k + 1] - m[k];
                    if (part <= 0) { std::printf("marks not increasing\n"); return 1; }
                    if (prev_part >= 0 && part > prev_part + 2) { std::printf("parts grow: %lld after %lld\n", part, prev_part); return 1; }
                    prev_part = part;
                }
                const double avg = (double)total / (double)n, last = (double)(m[n] - m[n - 1]);
                if (last < taper * avg * 0.999 - 2 || last > taper * avg * 1.001 + 2) { std::printf("last part %.0f vs %.0f\n", last, taper * avg); return 1; }
            }
    if (!msfm_pipeline_marks(1000, 1, 0.3).empty() || !msfm_pipeline_marks(1000, 0, 0.3).empty()) { std::printf("marks for one part\n"); return 1; }
    // scratch per pair: monotone in the sizes, the three routes differ, printed for the Python twin in tools/config4_full.py
    const int sizes[][2] = {{8192, 8192}, {16384, 16384}, {5038, 4711}, {100, 100}, {1, 700}};
    for (auto& z : sizes) {
        const int n1 = z[0], n2 = z[1], n1pad = (n1 + 511) / 512 * 512, n2pad = (n2 + 511) / 512 * 512;
        const long long r1 = msfm_pair_scratch_bytes(n1, n2, n1pad, n2pad, (n1 + 127) / 128, n1pad / 512, 1);
        const long long r0 = msfm_pair_scratch_bytes(n1, n2, n1pad, n2pad, (n1 + 127) / 128, n1pad / 512, 0);
        const long long r2 = msfm_pair_scratch_bytes(n1, n2, n1pad, n2pad, (n1 + 127) / 128, n1pad / 512, 2);
        if (r0 <= 0 || r1 <= 0 || r2 <= r1 - 84LL * (((long long)n1 + (long long)n2 * 32) / 16 + 1024)) { std::printf("scratch routes\n"); return 1; }
        if (msfm_pair_scratch_bytes(n1 + 512, n2, n1pad + 512, n2pad, (n1 + 639) / 128, n1pad / 512 + 1, 1) <= r1) { std::printf("scratch not monotone\n"); return 1; }
        std::printf("scratch %d %d %lld\n", n1, n2, r1);
    }
    if (msfm_pair_scratch_bytes(0, 700, 0, 1024, 0, 0, 1) != 0) { std::printf("empty side\n"); return 1; }
    // the fold of the exact re-check: random keys (with repeats) in random order, every arrival looking at an OLDER state of the slot
    {
        auto amin = [](unsigned long long* w, unsigned long long k) { const unsigned long long o = *w; if (k < o) *w = k; return o; };
        for (int trial = 0; trial < 200000; ++trial) {
            const int n = 1 + (int)(rng() % 9);
            unsigned long long keys[16];
            for (int i = 0; i < n; ++i) keys[i] = (i > 0 && rng() % 4 == 0) ? keys[rng() % i] : ((rng() % 50) << 32 | (rng() % 7));
            unsigned long long best = ~0ull, second = ~0ull, hb[17], hs[17];
            for (int i = 0; i < n; ++i) {
                hb[i] = best; hs[i] = second;                       // the slot's history: state before arrival i
                const int look = (int)(rng() % (i + 1));            // a look taken at any earlier (or the current) state
                msfm_fold_key(&best, &second, keys[i], hb[look], hs[look], amin);
            }
            unsigned long long m0 = ~0ull, m1 = ~0ull;
            for (int i = 0; i < n; ++i) if (keys[i] < m0) m0 = keys[i];
            for (int i = 0; i < n; ++i) if (keys[i] != m0 && keys[i] < m1) m1 = keys[i];
            if (best != m0 || second != m1) { std::printf("fold: best %llx second %llx, want %llx %llx (n = %d)\n", best, second, m0, m1, n); return 1; }
        }
    }
    // room for the sweep-2 plan: the ComputeMatches executable's order of calls (all pairs on 100-row subsets, then on the full images)
    {
        const int W = 512;
        // the small call, nothing known: every row fits (its sub-batch is below 2 M rows)
        const long long ub_s = 1625600, g_s = 255;
        MsfmPlanRoom a = msfm_plan_room(ub_s, ub_s * 5 / 16, g_s, 1, W, 0, 0, 0, 0, 0, 0, 0);
        if (a.hinted || a.rows < ub_s || a.cand < 8 * a.rows || a.items < 2 * (a.rows / W + g_s)) { std::printf("plan room: small call\n"); return 1; }
        // the large call's first sub-batch, the small call's needs as the hint and its buffers in place: the hint is void (factor 25),
        // the buffers are too small for 5/16 of the rows, fresh ones hold what the job really needs (measured: 10 457 600 rows)
        const long long ub_l = 40875057, g_l = 8128;
        MsfmPlanRoom b = msfm_plan_room(ub_l, ub_l * 5 / 16, g_l, 2, W, ub_s, 932352, 7335936, 1920, a.rows, a.cand, a.items);
        if (b.hinted || b.rows < 10457600 + 10457600 / 8 || b.rows <= a.rows || b.cand < 83105792 || b.items < 20960) { std::printf("plan room: large call after small\n"); return 1; }
        // (round 5's first half kept a.rows here: the absolute hint + 1/8 + slack was below them)
        // its second sub-batch: alike -> hinted, the buffers are kept
        MsfmPlanRoom c = msfm_plan_room(41025592, 41025592LL * 5 / 16, g_l, 2, W, ub_l, 10457600, 83105792, 20960, b.rows, b.cand, b.items);
        if (!c.hinted || c.rows != b.rows || c.cand != b.cand || c.items != b.items) { std::printf("plan room: alike sub-batch re-sized\n"); return 1; }
        // a sub-batch of the same kind, 1.8 x the size: the prediction scales with it
        MsfmPlanRoom d = msfm_plan_room(ub_l * 9 / 5, ub_l * 9 / 5 * 5 / 16, g_l * 2, 2, W, ub_l, 10457600, 83105792, 20960, b.rows, b.cand, b.items);
        if (!d.hinted || d.rows < 10457600LL * 9 / 5 * 9 / 8 || d.cand < 83105792LL * 9 / 5 || d.items < 20960LL * 9 / 5) { std::printf("plan room: scaled hint\n"); return 1; }
        // back to the small call: the large call's buffers hold it, nothing is replaced
        MsfmPlanRoom e = msfm_plan_room(ub_s, ub_s * 5 / 16, g_s, 1, W, 41025592, 10250240, 81693696, 42448, d.rows, d.cand, d.items);
        if (e.hinted || e.rows != d.rows || e.cand != d.cand || e.items != d.items) { std::printf("plan room: small call after large\n"); return 1; }
        // monotone: more rows never get less room (fresh buffers)
        long long prev = 0;
        for (long long ub = 1000; ub < 400000000LL; ub = ub * 3 / 2) {
            const MsfmPlanRoom r = msfm_plan_room(ub, ub * 5 / 16, 64, 1, W, 0, 0, 0, 0, 0, 0, 0);
            if (r.rows < prev || r.cand < 8 * r.rows) { std::printf("plan room: not monotone at %lld\n", ub); return 1; }
            prev = r.rows;
        }
    }
    // page-locked pieces of the result lists: they tile the range, 1 MiB first, 32 MiB from 32 MiB on, every end 4-KiB aligned
    {
        unsigned long long off = 0, n = 0;
        const unsigned long long MiB = 1ull << 20, want[] = {1, 2, 4, 8, 16, 32, 64, 96};
        while (off < 96 * MiB) {
            const unsigned long long end = msfm_pinned_piece_end(off);
            if (end != want[n] * MiB) { std::printf("pieces: %llu -> %llu\n", off, end); return 1; }
            for (unsigned long long inside : {off, off + 1, (off + end) / 2, end - 1})
                if (msfm_pinned_piece_end(inside) != end) { std::printf("pieces: inside %llu\n", inside); return 1; }
            off = end;
            ++n;
        }
    }
    std::printf("ok %lld\n", checked);
    return 0;
}
"""


def test_packed_column_partials_and_pipeline_marks(tmp_path):
    src = tmp_path / "driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-I", CSRC, "-o", str(exe), str(src)], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = run.stdout.splitlines()
    assert lines[-1].startswith("ok ") and int(lines[-1].split()[1]) > 1000000   # (gaps beyond ~1e6 drop the second: not counted)
    # the sub-batch cut predicted by tools/config4_full.py uses a Python twin of msfm_pair_scratch_bytes: same numbers
    import importlib.util
    spec = importlib.util.spec_from_file_location("config4_full", os.path.join(HERE, "..", "tools", "config4_full.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    twins = [l.split() for l in lines if l.startswith("scratch ")]
    assert len(twins) == 5
    for _, n1, n2, want in twins:
        assert mod.pair_scratch_bytes(int(n1), int(n2)) == int(want), (n1, n2)
    # config 4: ~2 MB per pair -> the default 64 GiB / 3 sets holds ~8 000 pairs per sub-batch (round 3: 4 400 in "48 GiB")
    assert 1.5e6 < mod.pair_scratch_bytes(8192, 8192) < 3.0e6
