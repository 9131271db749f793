// msfm_job.hip.h -- one msfm_match_pairs / _verified call (or one msfm_match_pairs_begin .. _next series) as a state machine.
// Included by msfm_match.hip only.
//
// The call is cut into device sub-batches (memory, pair count, and -- for a large call -- at least ctx->pipeline of them); sub-batch
// k + 1 is LAUNCHED on the other stream / scratch set before the host waits for sub-batch k, so that the bandwidth-bound tail of k runs
// under sweep 1 of k + 1:
//
//      issue(0)  issue(1) complete(0)  issue(2) complete(1)  issue(3) complete(2)  ...  complete(last)
//
// issue(k)    = every launch of the sub-batch (sweeps, plan, exact re-check, epilogue, [verification], CSR gather into the
//               scratch set's own list buffer) + the copies of the words the host needs into page-locked memory;
// complete(k) = wait for k's stream; a queue / plan buffer was too small -> drain everything, re-run k alone (grown by
//               then), carry on behind it; else append k's lists to the call's lists (device-to-device for
//               msfm_fetch_matches_device, device-to-host into the page-locked result buffers: asynchronous, on k's stream) -- or,
//               in the STREAMING form, leave them in the scratch set's own buffers for the caller to read before it asks for the next
//               chunk: the reference streams by construction, one transaction per <= 100 pairs
//               (src/Feature/FeatureMatching.cpp:13, 70-72, 118-139), and a job whose lists do not fit memory needs the same.
// Results do not depend on the cut (tests force it every which way).
#pragma once

struct MatchJob {
    msfm_ctx* ctx = nullptr;
    std::vector<int32_t> pairs_own;   // streaming form: the caller's pair list is copied (it need not outlive _begin)
    const int32_t* pairs = nullptr;
    int n_pairs = 0;
    msfm_match_params prm = {0.8f, 1, 0.7};
    PruneParams prune = {1, 0.8f, 0.7f};
    msfm_verify_params verify_store = {};
    const msfm_verify_params* verify = nullptr;
    bool streaming = false;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    int kSets = kInFlight;
    long long kScratchBytes = 1;
    int scratch_route = 1;
    int kMaxPairsPerBatch = kDefaultMaxPairsPerBatch;
    bool coarse_twins = false;   // float stores whose twins need the fp16 sweep 1' (plan A: more compacted rows per pair)
    std::vector<long long> marks;
    long long cost_done = 0;   // cost of the sub-batches built so far (a re-built sub-batch starts from its own begin: see build)
    bool need_fix = false;
    SubBatch sb[kInFlight];
    int next_begin = 0;
    long long issued = 0, completed = 0;   // sub-batch k lives in slot k % kSets
    bool open = false;
    bool memory_wait = false;   // a sub-batch did not fit the device's memory: nothing new is issued until what is in flight has been completed
    int memory_shrinks = 0;

    // everything before the first launch
    int start(msfm_ctx* c, const int32_t* pairs_in, int n, const msfm_match_params* params, const msfm_verify_params* vp, bool stream_mode) {
        ctx = c;
        pairs = pairs_in;
        n_pairs = n;
        streaming = stream_mode;
        verify = nullptr;
        if (vp) {
            verify_store = *vp;
            verify = &verify_store;
        }
        for (SubBatch& w : sb) w = SubBatch{};
        next_begin = 0;
        issued = completed = 0;
        memory_wait = false;
        memory_shrinks = 0;
        ctx->last_hip_error = 0;
    prm = msfm_match_params{0.8f, 1, 0.7};
    if (params) prm = *params;
    // rows / columns that provably fail the ratio test or the distance cut need no exact neighbours
    prune = PruneParams{1, prm.ratio, (float)prm.max_distance};
    if ((double)prune.max_distance < prm.max_distance) prune.max_distance = nextafterf(prune.max_distance, __builtin_huge_valf());
    if (!(prm.ratio > 0.f) || !(prm.ratio <= 1.f)) prune.ratio = 0.f;  // outside (0, 1]: no ratio-based pruning
    if (!(prm.max_distance >= 0.0)) prune.max_distance = __builtin_huge_valf();  // NaN / negative: no distance-based pruning
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->cur = &ctx->sc[0];
    ctx->have_results = false;
    for (Scratch& s : ctx->sc) {
        s.pf_pending = PfPending{};
        s.sweep1_recorded = false;
        s.sweep2_recorded = false;
    }
    ctx->last_sweep1 = nullptr;
    if (!ctx->deferred.empty()) {   // (what the call before replaced: nothing is in flight now, and nobody waits for a result)
        HIPCHK(ctx, hipDeviceSynchronize());
        ctx->deferred.flush();
    }
    // uploads since the last use are built now (msfm_store_host.hip.h); twins built before a later upload raised the context's level
    // (msfm_q8.hip.h) are rebuilt: nothing is in flight
    {
        int rc = settle_store(ctx);
        if (rc != MSFM_OK) return rc;
        rc = rebuild_stale_twins(ctx, pairs, 2 * n_pairs);
        if (rc != MSFM_OK) return rc;
    }
    ctx->res_offsets.assign((size_t)n_pairs + 1, 0);
    ctx->res_sens.assign((size_t)n_pairs, 0);
    ctx->res_count = 0;
    ctx->prof = msfm_profile{};
    ctx->out_used = 0;
    for (OutSeg& s : ctx->out_segs) s.count = 0;
    if (!streaming) {
        // address space for the host lists: a pair yields at most min(n1, n2) matches (each query row one, each train row once under the
        // cross-check; without it n1)
        long long bound = 1;
        for (int k = 0; k < n_pairs; ++k) {
            const int i = pairs[2 * k], j = pairs[2 * k + 1];
            if (i < 0 || i >= kSlots || j < 0 || j >= kSlots) return fail(ctx, MSFM_E_INVALID, "image id out of range");
            const long long n1 = ctx->images[(size_t)i].n, n2 = ctx->images[(size_t)j].n;
            if (n1 > 0 && n2 > 0) bound += prm.cross_check ? std::min(n1, n2) : n1;
        }
        if (ctx->res_qt.reserve((size_t)bound * 8) != hipSuccess || ctx->res_dist.reserve((size_t)bound * 4) != hipSuccess)
            return fail(ctx, MSFM_E_DEVICE, "cannot reserve address space for the result lists");
    }

    ev_begin = get_event(ctx, 0), ev_end = get_event(ctx, 1);
    if (!ev_begin || !ev_end) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
    HIPCHK(ctx, hipEventRecord(ev_begin, ctx->sc[0].stream));

    // ---- the cut: scratch memory per set, pair count, and a cost limit that gives a large call >= `pipeline` sub-batches
    kSets = ctx->in_flight;
    // the scratch sets in flight SHARE the budget (three sets at half of it each were 1.5 x the documented limit)
    // (the automatic budget is derived from the device's free memory ONCE per state of the context -- the store's size and the limit
    // set by the caller -- not per call: a hipMemGetInfo per call made the cut depend on what else was momentarily allocated on the GPU,
    // and the pre-emptive filter's many small calls paid the driver query each time)
    long long budget = ctx->scratch_bytes > 0 ? ctx->scratch_bytes : kDefaultScratchBytes;
    if (ctx->budget_cached > 0 && ctx->budget_for_store == ctx->store.bytes() && ctx->budget_for_limit == ctx->scratch_bytes) {
        budget = ctx->budget_cached;
    } else {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            long long held = 0;   // what the scratch sets hold already is theirs to use
            for (Scratch& sc : ctx->sc) held += sc.device_bytes();
            const long long avail = (long long)free_b + held;
            // (an explicit limit is honoured up to half of what is free: the estimate per pair is the cut, not a cap -- buffers that turn
            // out too small are re-grown)
            budget = std::min(budget, ctx->scratch_bytes > 0 ? avail / 2 : avail / 4);
        }
        ctx->budget_cached = budget;
        ctx->budget_for_store = ctx->store.bytes();
        ctx->budget_for_limit = ctx->scratch_bytes;
    }
    kScratchBytes = std::max<long long>(1, budget / kSets);
    // which buffers a pair needs (msfm_pair_scratch_bytes): 0 brute force, 1 matrix cores + compacted sweep 2 (3: on the integer cores), 2 + dense sweep 2
    scratch_route = !ctx->prefilter ? 0 : ((prune.ratio > 0.f && prune.ratio <= 0.95f) ? 1 : 2);
    kMaxPairsPerBatch = ctx->max_pairs_per_batch;
    coarse_twins = ctx->prefilter == 1 && ctx->q8_route && ctx->q8_level > 0.f &&
                   !(ctx->q8_direct == 2 || (ctx->q8_direct == 1 && ctx->q8_level <= kQ8DirectMaxLevel));
    // cumulative-cost marks of the parts (empty: no cost cut): msfm_pipeline_marks (msfm_hostutil.h) -- shrinking parts
    marks.clear();
    if (ctx->pipeline > 1 && n_pairs > 1) {
        long long total = 0;
        for (int k = 0; k < n_pairs; ++k) {
            const int i = pairs[2 * k], j = pairs[2 * k + 1];
            if (i < 0 || i >= kSlots || j < 0 || j >= kSlots) return fail(ctx, MSFM_E_INVALID, "image id out of range");
            const long long n1 = ctx->images[i].n, n2 = ctx->images[j].n;
            if (n1 > 0 && n2 > 0) total += n1 * n2;
        }
        marks = msfm_pipeline_marks(total, std::min<long long>(ctx->pipeline, total / kMinPipelineCost), ctx->pipeline_taper);
    }
    cost_done = 0;
    // a tie in sqrt space can only surface in a match list when a row with d0 == d1 can pass the ratio test
    need_fix = !(prm.ratio <= 1.f);

        open = true;
        ctx->series_open = streaming;
        return MSFM_OK;
    }

    // pairs [begin, ...) -> sb.b (host tables); force_exact: pairs of this sub-batch whose candidate list overflowed in an
    // earlier attempt take the brute-force path
    int build(SubBatch& w, int begin, const std::vector<char>& force_exact) {
        w.b = Batch{};
        w.begin = begin;
        if (begin == 0) cost_done = 0;
        const long long cost_begin = (begin == w.begin_of_cost) ? w.cost_begin : cost_done;
        w.begin_of_cost = begin;
        w.cost_begin = cost_begin;
        // the first cumulative-cost mark behind this sub-batch's start
        // (a part ends at the pair nearest to its mark, so its successor may start a little before or behind one)
        long long mark = 0;
        if (!marks.empty()) {
            size_t i = 0;
            while (i + 2 < marks.size() && cost_begin >= (marks[i] + marks[i + 1]) / 2) ++i;
            mark = marks[i + 1];
        }
        long long est = 0, cost = 0;
        int end = begin;
        while (end < n_pairs && (end - begin) < kMaxPairsPerBatch) {
            PairDesc pd;
            PfPair pp;
            int rc = fill_pair(ctx, pairs[2 * end], pairs[2 * end + 1], pd, pp);
            if (rc != MSFM_OK) return rc;
            if (verify) {
                const Image& ia = ctx->images[pairs[2 * end]];
                const Image& ib = ctx->images[pairs[2 * end + 1]];
                if (ia.nk < ia.n || ib.nk < ib.n)
                    return fail(ctx, MSFM_E_STATE, "geometric verification needs msfm_upload_keypoints for image " +
                                                       std::to_string(ia.nk < ia.n ? pairs[2 * end] : pairs[2 * end + 1]));
            }
            const bool bytes_pair = ctx->prefilter == 1 && ctx->images[(size_t)pairs[2 * end]].is_u8 && ctx->images[(size_t)pairs[2 * end + 1]].is_u8;
            const long long need = pd.valid ? msfm_pair_scratch_bytes(pd.n1, pd.n2, pd.n1pad, pd.n2pad, pd.a_blocks, pd.a_blocks256,
                                                                     !pp.use ? 0 : (scratch_route == 1 && bytes_pair ? 3 : (scratch_route == 1 && coarse_twins ? 4 : scratch_route))) : 0;
            const long long c = pd.valid ? (long long)pd.n1 * pd.n2 : 0;
            if (end > begin && est + need > kScratchBytes) break;
            if (end > begin && !marks.empty() && cost_begin + cost + c / 2 > mark) break;
            est += need;
            cost += c;
            const size_t k = w.b.pairs.size();
            if (k < force_exact.size() && force_exact[k]) {
                pp.use = 0;
                pd.path = 0;
            }
            w.b.pairs.push_back(pd);
            w.b.pf.push_back(pp);
            w.b.id1.push_back(pairs[2 * end]);
            w.b.id2.push_back(pairs[2 * end + 1]);
            ++end;
        }
        w.end = end;
        cost_done = cost_begin + cost;
        // the forms of the images this sub-batch's routes read (derived on demand for byte images), then the pairs' pointers
        return prepare_batch_images(ctx, w.b, prune);
    }

    // every launch of the sub-batch on the CURRENT scratch set (ctx->cur), nothing waits
    int issue(SubBatch& w, size_t ev_base) {
        Batch& b = w.b;
        const size_t P = b.pairs.size();
        const int begin = w.begin;
        w.ev_base = ev_base;
        w.exact_launched = false;
        SC.prof = msfm_profile{};
        SC.seq = ctx->issue_seq++;
        SC.sweep2_recorded = false;
        int rc = run_knn(ctx, b, ev_base, &w.exact_launched, prune, need_fix, true);
        if (rc != MSFM_OK) return rc;
        const long long oe = std::max<long long>(1, b.out_elems);
        HIPCHK(ctx, SC.d_st_qt.ensure(oe * sizeof(int2)));
        HIPCHK(ctx, SC.d_st_d.ensure(oe * 4));
        HIPCHK(ctx, SC.d_sub_qt.ensure(oe * sizeof(int2)));
        HIPCHK(ctx, SC.d_sub_d.ensure(oe * 4));
        HIPCHK(ctx, SC.d_counts.ensure(P * 4));
        HIPCHK(ctx, SC.d_sens.ensure(P * 4));
        HIPCHK(ctx, SC.d_offsets.ensure((P + 1) * 8));
        EpiParams ep = {prm.ratio, prm.cross_check, prm.max_distance};
        if (SC.keys_epilogue)
            hipLaunchKernelGGL(epilogue_kernel<KnnFromKeys>, dim3((unsigned)P), dim3(256), 0, SC.stream, SC.d_pairs.as<PairDesc>(), ep,
                               KnnFromKeys{SC.d_tu.as<float>(), SC.d_best.as<unsigned long long>(), SC.d_second.as<unsigned long long>()},
                               SC.d_st_qt.as<int2>(), SC.d_st_d.as<float>(), SC.d_counts.as<int>(), SC.d_sens.as<int>());
        else
            hipLaunchKernelGGL(epilogue_kernel<KnnFromArrays>, dim3((unsigned)P), dim3(256), 0, SC.stream, SC.d_pairs.as<PairDesc>(), ep,
                               KnnFromArrays{SC.d_k_i0.as<int>(), SC.d_k_d0.as<float>(), SC.d_k_d1.as<float>()},
                               SC.d_st_qt.as<int2>(), SC.d_st_d.as<float>(), SC.d_counts.as<int>(), SC.d_sens.as<int>());
        HIPCHK(ctx, hipGetLastError());
        const int* d_counts = SC.d_counts.as<int>();
        const int2* d_st_qt = SC.d_st_qt.as<int2>();
        const float* d_st_d = SC.d_st_d.as<float>();
        if (verify) {
            // FeatureUtils::FilterMatches on the staged lists: all hypotheses of all pairs at once
            VerifyParams vprm = {verify->threshold * verify->threshold, verify->confidence, verify->max_iters, 0, verify->seed};
            std::vector<VerifyPair>& vpairs = b.verify_pairs;   // (lives as long as the sub-batch: the copy below may still be in flight)
            vpairs.resize(P);
            for (size_t p = 0; p < P; ++p)
                vpairs[p] = VerifyPair{ctx->images[pairs[2 * (begin + (int)p)]].kxy, ctx->images[pairs[2 * (begin + (int)p) + 1]].kxy};
            HIPCHK(ctx, SC.d_vf_pairs.ensure(P * sizeof(VerifyPair)));
            HIPCHK(ctx, SC.d_vf_x1.ensure(oe * 4));
            HIPCHK(ctx, SC.d_vf_y1.ensure(oe * 4));
            HIPCHK(ctx, SC.d_vf_x2.ensure(oe * 4));
            HIPCHK(ctx, SC.d_vf_y2.ensure(oe * 4));
            HIPCHK(ctx, SC.d_vf_flags.ensure(oe));
            HIPCHK(ctx, SC.d_vf_hyp.ensure(P * (size_t)vprm.max_iters * 4));
            HIPCHK(ctx, SC.d_vf_best_it.ensure(P * 4));
            HIPCHK(ctx, SC.d_vf_best_count.ensure(P * 4));
            HIPCHK(ctx, SC.d_st2_qt.ensure(oe * sizeof(int2)));
            HIPCHK(ctx, SC.d_st2_d.ensure(oe * 4));
            HIPCHK(ctx, SC.d_counts2.ensure(P * 4));
            HIPCHK(ctx, hipMemcpyAsync(SC.d_vf_pairs.p, vpairs.data(), P * sizeof(VerifyPair), hipMemcpyHostToDevice, SC.stream));
            hipEvent_t v0 = get_event(ctx, ev_base + 6), v1 = get_event(ctx, ev_base + 7);
            if (!v0 || !v1) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
            HIPCHK(ctx, hipEventRecord(v0, SC.stream));
            const PairDesc* dp = SC.d_pairs.as<PairDesc>();
            float *x1 = SC.d_vf_x1.as<float>(), *y1 = SC.d_vf_y1.as<float>(), *x2 = SC.d_vf_x2.as<float>(), *y2 = SC.d_vf_y2.as<float>();
            hipLaunchKernelGGL(vf_points_kernel, dim3((unsigned)P), dim3(256), 0, SC.stream, dp, SC.d_vf_pairs.as<VerifyPair>(),
                               d_counts, d_st_qt, x1, y1, x2, y2);
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(vf_hypotheses_kernel, dim3((unsigned)((vprm.max_iters + 255) / 256), (unsigned)P), dim3(256), 0, SC.stream,
                               dp, d_counts, (const float*)x1, (const float*)y1, (const float*)x2, (const float*)y2,
                               SC.d_vf_hyp.as<int>(), vprm);
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(vf_select_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, SC.stream, d_counts,
                               (const int*)SC.d_vf_hyp.as<int>(), (int)P, vprm, SC.d_vf_best_it.as<int>(), SC.d_vf_best_count.as<int>());
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(vf_mask_compact_kernel, dim3((unsigned)P), dim3(256), 0, SC.stream, dp, d_counts, d_st_qt, d_st_d,
                               (const float*)x1, (const float*)y1, (const float*)x2, (const float*)y2,
                               (const int*)SC.d_vf_best_it.as<int>(), (const int*)SC.d_vf_best_count.as<int>(),
                               SC.d_vf_flags.as<unsigned char>(), vprm, SC.d_st2_qt.as<int2>(), SC.d_st2_d.as<float>(),
                               SC.d_counts2.as<int>());
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipEventRecord(v1, SC.stream));
            d_counts = SC.d_counts2.as<int>();
            d_st_qt = SC.d_st2_qt.as<int2>();
            d_st_d = SC.d_st2_d.as<float>();
        }
        hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(256), 0, SC.stream, d_counts,
                           SC.d_offsets.as<long long>(), (int)P);
        HIPCHK(ctx, hipGetLastError());
        // CSR order, into this scratch set's own list buffer: where the lists go in the call's buffers is only known when
        // the sub-batches before this one have been completed
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)P), dim3(256), 0, SC.stream, SC.d_pairs.as<PairDesc>(),
                           d_counts, SC.d_offsets.as<long long>(), d_st_qt,
                           d_st_d, SC.d_sub_qt.as<int2>(), SC.d_sub_d.as<float>());
        HIPCHK(ctx, hipGetLastError());
        rc = queue_tail_copies(ctx, P);
        if (rc != MSFM_OK) return rc;
        w.active = true;
        return MSFM_OK;
    }

    // wait for the sub-batch of the CURRENT scratch set; *retry: a queue / plan buffer was too small (grown by now)
    int complete(SubBatch& w, std::vector<char>& force_exact, bool* retry_out) {
        Batch& b = w.b;
        const size_t P = b.pairs.size();
        w.active = false;
        *retry_out = false;
        HIPCHK(ctx, hipStreamSynchronize(SC.stream));
        bool retry = false, retry_pf = false;
        force_exact.resize(P, 0);
        int rc = check_fix_overflow(ctx, &retry);
        if (rc != MSFM_OK) return rc;
        rc = finish_prefilter(ctx, b, force_exact, &retry_pf);
        if (rc != MSFM_OK) return rc;
        if (retry || retry_pf) {   // only the re-run counters of a dropped attempt count
            ctx->prof.tie_queue_regrows += SC.prof.tie_queue_regrows;
            ctx->prof.plan_regrows += SC.prof.plan_regrows;
            ctx->prof.fallback_pairs += SC.prof.fallback_pairs;
            *retry_out = true;
            return MSFM_OK;
        }
        SC.prof.fallback_pairs = 0;   // (counted when the attempt that found them was dropped)
        const char* h = SC.h_tail.as<char>();
        const long long* offs = reinterpret_cast<const long long*>(h + 8);
        const int32_t* sens = reinterpret_cast<const int32_t*>(h + 8 + (P + 1) * 8);
        const long long total = offs[P];
        const size_t base = streaming ? 0 : ctx->res_count;
        if (streaming) {
            // the streaming form (msfm_match_pairs_begin / _next): the sub-batch's lists go to page-locked buffers of ITS scratch set
            // and stay, with the device copy in d_sub_qt / d_sub_d, until the caller asks for the next chunk -- nothing accumulates
            HIPCHK(ctx, SC.h_sub_qt.ensure(((size_t)total + 1) * 8, 0));
            HIPCHK(ctx, SC.h_sub_d.ensure(((size_t)total + 1) * 4, 0));
            if (total > 0) {
                HIPCHK(ctx, hipMemcpyAsync(SC.h_sub_qt.p, SC.d_sub_qt.p, (size_t)total * 8, hipMemcpyDeviceToHost, SC.stream));
                HIPCHK(ctx, hipMemcpyAsync(SC.h_sub_d.p, SC.d_sub_d.p, (size_t)total * 4, hipMemcpyDeviceToHost, SC.stream));
                HIPCHK(ctx, hipStreamSynchronize(SC.stream));
            }
            w.offsets.assign(offs, offs + P + 1);
            w.sens.assign(sens, sens + P);
            for (size_t p = 0; p < P; ++p) SC.prof.order_sensitive_rows += sens[p];
            ctx->res_count += (size_t)total;   // (matches of the job so far)
        } else {
            // the call's lists: the host range page-locks the pieces this sub-batch reaches into (GrowPinned), the device side takes a
            // segment with room for the whole sub-batch -- nothing moves, nothing is freed, no stream is drained
            const size_t need = base + (size_t)total + 1;
            if (ctx->res_qt.ensure_pinned(need * 8) != hipSuccess || ctx->res_dist.ensure_pinned(need * 4) != hipSuccess)
                return fail(ctx, MSFM_E_DEVICE, "cannot page-lock the result lists (beyond the reserved range or the host's memory): use msfm_match_pairs_begin / _next");
            if (total > 0) {
                OutSeg* seg = ctx->out_used ? &ctx->out_segs[ctx->out_used - 1] : nullptr;
                if (!seg || seg->cap - seg->count < (size_t)total) {
                    // the next kept segment that is large enough, or a new one sized for ~8 sub-batches like this
                    while (true) {
                        if (ctx->out_used == ctx->out_segs.size()) {
                            ctx->out_segs.push_back(OutSeg{});
                            OutSeg& s = ctx->out_segs.back();
                            const size_t cap = std::max<size_t>((size_t)total * 8, (size_t)4 << 20);
                            HIPCHK(ctx, s.qt.ensure(cap * 8));
                            HIPCHK(ctx, s.d.ensure(cap * 4));
                            s.cap = cap;
                        }
                        seg = &ctx->out_segs[ctx->out_used++];
                        seg->first = base;
                        seg->count = 0;
                        if (seg->cap >= (size_t)total) break;
                    }
                }
                HIPCHK(ctx, ctx->res_qt.copy_in(base * 8, SC.d_sub_qt.p, (size_t)total * 8, SC.stream));
                HIPCHK(ctx, ctx->res_dist.copy_in(base * 4, SC.d_sub_d.p, (size_t)total * 4, SC.stream));
                HIPCHK(ctx, hipMemcpyAsync(seg->qt.as<int2>() + seg->count, SC.d_sub_qt.p, (size_t)total * 8, hipMemcpyDeviceToDevice, SC.stream));
                HIPCHK(ctx, hipMemcpyAsync(seg->d.as<float>() + seg->count, SC.d_sub_d.p, (size_t)total * 4, hipMemcpyDeviceToDevice, SC.stream));
                seg->count += (size_t)total;
            }
            ctx->res_count = base + (size_t)total;
            for (size_t p = 0; p < P; ++p) {
                ctx->res_offsets[(size_t)w.begin + p + 1] = (int64_t)base + offs[p + 1];
                ctx->res_sens[(size_t)w.begin + p] = sens[p];
                SC.prof.order_sensitive_rows += sens[p];
            }
        }
        SC.prof.descriptor_pairs += b.desc_pairs;
        SC.prof.dist_algo_bytes += b.algo_bytes;
        rc = accumulate_kernel_time(ctx, w.ev_base, w.exact_launched);
        if (rc != MSFM_OK) return rc;
        if (verify) {
            float vms = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&vms, ctx->ev_pool[w.ev_base + 6], ctx->ev_pool[w.ev_base + 7]));
            SC.prof.verify_ms += vms;
        }
        SC.prof.sub_batches += 1;
        add_profile(ctx->prof, SC.prof);
        return MSFM_OK;
    }

    bool more() const { return completed < issued || next_begin < n_pairs; }   // (a sub-batch waiting for memory has not advanced next_begin)

    // launch ahead as many sub-batches as there are free scratch sets, wait for the oldest one (re-running it alone if a buffer was too
    // small): *slot_out = the scratch set whose sub-batch has just been completed
    // The device ran out of memory while a sub-batch was being sized or launched (another tenant of the GPU; several contexts on one
    // device -- the budget is derived from the free memory ONCE per state of the store, msfm_set_limits): not an error yet.  What the
    // failed attempt launched is drained and dropped; nothing new is issued until the sub-batches in flight have been completed (their
    // buffers are theirs); then every scratch buffer goes back to the device, the share of a sub-batch is halved -- and never more
    // than a sixth of what is free now -- and the call cuts again from the same pair.  Results do not depend on the cut.
    bool out_of_memory(int rc) const { return rc == MSFM_E_DEVICE && ctx->last_hip_error == (int)hipErrorOutOfMemory && memory_shrinks < 4; }
    void drop_attempt(int slot) {
        Scratch& sc = ctx->sc[slot];
        if (sc.stream) (void)hipStreamSynchronize(sc.stream);
        (void)hipGetLastError();   // (the runtime keeps the allocation's error for the next hipGetLastError: a launch check would see it)
        sb[slot].active = false;
        sc.pf_pending = PfPending{};
        sc.sweep1_recorded = false;
        sc.sweep2_recorded = false;
        if (ctx->last_sweep1 == &sc) ctx->last_sweep1 = nullptr;
        ctx->last_hip_error = 0;
        memory_wait = true;
    }
    int shrink_memory() {
        int rc = drain_streams(ctx);
        if (rc != MSFM_OK) return rc;
        ctx->deferred.flush();
        for (Scratch& sc : ctx->sc) sc.release_device();   // (streams, events and the page-locked host side stay)
        size_t free_b = 0, total_b = 0;
        long long share = kScratchBytes / 2;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) share = std::min<long long>(share, (long long)(free_b / (2 * (size_t)kSets)));   // all sets together: half of what is free
        kScratchBytes = std::max<long long>(1, share);
        ctx->budget_cached = kScratchBytes * kSets;   // (later calls on this state of the store start from what worked)
        ctx->hint_rows_ub = 0;                        // (the plan's prediction was relative to buffers that are gone)
        memory_wait = false;
        memory_shrinks += 1;
        ctx->prof.memory_shrinks += 1;
        if (std::getenv("MSFM_DEBUG_TIMING"))
            std::fprintf(stderr, "[msfm alloc] out of device memory: scratch given back, %.2f GiB per sub-batch from pair %d on (%zu MiB free)\n",
                         kScratchBytes / 1073741824.0, next_begin, free_b >> 20);
        return MSFM_OK;
    }

    int step(int* slot_out) {
        const DeferFreesScope defer(&ctx->deferred);   // buffers replaced below are freed once the streams have been drained
        const std::vector<char> no_force;
        while (true) {
            while (!memory_wait && next_begin < n_pairs && issued - completed < kSets) {
                const int slot = (int)(issued % kSets);
                int rc = ensure_scratch_set(ctx, slot);
                if (rc != MSFM_OK) return rc;
                ctx->cur = &ctx->sc[slot];
                rc = build(sb[slot], next_begin, no_force);
                if (rc == MSFM_OK) rc = issue(sb[slot], 2 + 12 * (size_t)slot);
                if (out_of_memory(rc)) {
                    drop_attempt(slot);
                    break;
                }
                if (rc != MSFM_OK) return rc;
                next_begin = sb[slot].end;
                ++issued;
            }
            if (issued > completed) break;          // something is in flight: complete the oldest below
            if (!memory_wait) break;                // (nothing left to do: the caller checks more())
            const int rc = shrink_memory();          // nothing in flight and the next sub-batch did not fit: smaller ones
            if (rc != MSFM_OK) return rc;
        }
        if (issued == completed) return fail(ctx, MSFM_E_STATE, "step() without work");
        const int slot = (int)(completed % kSets);
        ctx->cur = &ctx->sc[slot];
        std::vector<char> force_exact;
        bool retry = false;
        int rc = complete(sb[slot], force_exact, &retry);
        if (rc != MSFM_OK) return rc;
        if (retry) {
            // drop what is in flight behind it, re-run this sub-batch alone until it fits, carry on from its end
            rc = drain_streams(ctx);
            if (rc != MSFM_OK) return rc;
            ctx->deferred.flush();
            for (int k = 0; k < kSets; ++k)
                if (k != slot) {
                    sb[k].active = false;
                    ctx->sc[k].pf_pending = PfPending{};
                    ctx->sc[k].sweep1_recorded = false;
                    ctx->sc[k].sweep2_recorded = false;
                }
            ctx->last_sweep1 = nullptr;
            for (int attempt = 1;; ++attempt) {
                rc = build(sb[slot], sb[slot].begin, force_exact);
                if (rc != MSFM_OK) return rc;
                rc = issue(sb[slot], 2 + 12 * (size_t)slot);
                if (rc != MSFM_OK) return rc;
                rc = complete(sb[slot], force_exact, &retry);
                if (rc != MSFM_OK) return rc;
                if (!retry) break;
                if (attempt >= 6) return fail(ctx, MSFM_E_DEVICE, "batch kept overflowing its queues");
            }
            next_begin = sb[slot].end;
            issued = completed + 1;
        }
        ++completed;
        *slot_out = slot;
        return MSFM_OK;
    }

    // behind the last sub-batch
    int finish() {
        int rc = drain_streams(ctx);
        if (rc != MSFM_OK) return rc;
        // (ctx->deferred is flushed by the next call's start or msfm_destroy: a hipFree is ~0.1 ms, and the caller is waiting for its lists)
        ctx->cur = &ctx->sc[0];
        HIPCHK(ctx, hipEventRecord(ev_end, SC.stream));
        HIPCHK(ctx, hipEventSynchronize(ev_end));
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ev_begin, ev_end));
        ctx->prof.total_device_ms = ms;
        open = false;
        ctx->series_open = false;
        if (std::getenv("MSFM_DEBUG_TIMING")) {
            AllocClock& c = alloc_clock();
            std::fprintf(stderr, "[msfm alloc] since the last report: %d hipMalloc %.1f ms (%.2f GiB), %d hipFree %.1f ms, %d hipHostMalloc %.1f ms (%.2f GiB)\n",
                         c.dev_n, c.dev_ms, c.dev_bytes / 1073741824.0, c.free_n, c.free_ms, c.pin_n, c.pin_ms, c.pin_bytes / 1073741824.0);
            c = AllocClock{};
        }
        return MSFM_OK;
    }
};
