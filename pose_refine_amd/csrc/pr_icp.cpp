// pr_icp.cpp -- the batched ICP driver (cuda_icp/icp.cu:156-217 for many clouds): device-solve loop, host-solve pipeline, graph cache
#include "pr_runtime.h"

namespace prr {

hipError_t launch_pass(const prk::IcpBatch &b, const SceneSel &sc, uint32_t P, hipStream_t st, hipEvent_t *nn_marks)
{
    if (!st) st = g->stream;
    if (sc.kind == PR_SCENE_NN) {
        if (sc.nn_split && b.nn_prev) {                            // search (applies the pending update, leaves the winners in nn_prev) ...
            hipError_t e = prk::launch_nn_search(b, sc.nn, P, sc.nn_max_points, (uint32_t)std::max(1, opt.nn_run), st, nn_marks);
            if (e != hipSuccess) return e;
            prk::IcpBatch bb = b;                                   // ... then the canonical-order pass over the winners
            bb.pre_transformed = 1;
            const prk::SceneNNWinners w{ sc.nn.max_dist_diff, sc.nn.pts, sc.nn.normal, b.nn_prev };
            return prk::launch_icp_pass_nn_winners(bb, w, P, st);
        }
        return prk::launch_icp_pass_nn(b, sc.nn, P, st);
    }
    if (sc.packed) return prk::launch_icp_pass_proj_packed(b, sc.pk, P, st);
    return prk::launch_icp_pass_proj_aos(b, sc.aos, P, st);
}

// Streams are created only when they are first needed: the runtime multiplexes all streams of a process onto a handful of
// hardware queues (4 by default) in creation order, and two streams that share a queue serialise -- with every possible
// stream created up front, a slot's render stream ended up behind the other slot's 21 queued passes.
int ensure_stream(hipStream_t &st, hipEvent_t *ev)
{
    if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (ev && !*ev) HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return PR_OK;
}

// ---- the batched ICP driver -----------------------------------------------------------------------
// clouds: cloud i = cloud_base[start_h[i] .. start_h[i]+count_h[i]).  start/count must already be in
// g->start / g->counts on the device when dev_meta_ready, otherwise they are uploaded here.
int icp_drive(pr_vec3 *cloud_base, const uint32_t *start_h, const uint32_t *count_h, uint32_t P, const SceneSel &sc_in,
              pr_criteria crit, pr_result *results_host, pr_result *results_dev)
{
    if (P == 0) return PR_OK;
    SceneSel sc = sc_in;
    if (crit.max_iteration < 0) { set_error("max_iteration must be >= 0"); return PR_ERR_INVALID; }
    const uint32_t steps = (uint32_t)std::max(1, opt.steps);
    const uint32_t ppb = steps * prk::kPointsPerStep;
    uint32_t max_n = 0; uint64_t sum_n = 0;
    for (uint32_t i = 0; i < P; ++i) { max_n = std::max(max_n, count_h[i]); sum_n += count_h[i]; }
    const uint32_t nblk = (max_n + ppb - 1) / ppb;

    PR_TRY(g->meta.ensure(sizeof(prk::PoseMeta) * P));
    PR_TRY(g->partial.ensure(sizeof(float) * prk::kAccStride * (size_t)std::max(1u, nblk) * P));
    PR_TRY(g->sums.ensure(sizeof(float) * prk::kAccStride * P));
    PR_TRY(g->h_meta.ensure(sizeof(prk::PoseMeta) * P));
    PR_TRY(g->h_sums.ensure(sizeof(float) * prk::kAccStride * P + 64));        // (+ the pose groups' flags, PR_SOLVE_HOST with host_poll)
    PR_TRY(g->h_results.ensure(sizeof(pr_result) * P));

    prk::IcpBatch b{};
    b.cloud = cloud_base; b.meta = g->meta.as<prk::PoseMeta>(); b.partial = g->partial.as<float>();
    b.nblk = nblk; b.steps = steps;
    sc.nn_split = (sc.kind == PR_SCENE_NN && sc.nn.rec32 && opt.nn_split) ? 1u : 0u;
    sc.nn_max_points = max_n;
    if (sc.nn_split && opt.nn_count) {                            // instrumented run: kCounterPasses x 8 counters, accumulated until read
        const bool fresh = g->nn_counters.p == nullptr;
        PR_TRY(g->nn_counters.ensure(sizeof(unsigned long long) * 8 * kCounterPasses));
        if (fresh) HIP_TRY(hipMemsetAsync(g->nn_counters.p, 0, sizeof(unsigned long long) * 8 * kCounterPasses, g->stream));
        if ((uint32_t)crit.max_iteration + 1 <= kCounterPasses) sc.nn.counters = g->nn_counters.as<unsigned long long>();
    }
    if (sc.kind == PR_SCENE_NN && sc.nn.rec32 && (opt.nn_seed || sc.nn_split)) {     // winners, indexed like the cloud points
        size_t span = 1;
        for (uint32_t i = 0; i < P; ++i) span = std::max(span, (size_t)start_h[i] + count_h[i]);
        // winners | slack of the keep-the-winner test | queue 1 and queue 2 of unsettled queries (8 B per entry) | queue counters per hypothesis
        PR_TRY(g->nn_prev.ensure(sizeof(uint32_t) * (span * prk::kNNWordsPerPoint + prk::kQCountStride * (size_t)P) + 64));
        b.nn_prev = g->nn_prev.as<uint32_t>();
        b.nn_slack = reinterpret_cast<float *>(b.nn_prev + span);
        b.nn_queue = reinterpret_cast<uint2 *>(b.nn_prev + 2 * span);
        b.nn_queue2 = reinterpret_cast<uint2 *>(b.nn_prev + 4 * span);
        b.nn_qcount = b.nn_prev + 6 * span;
        HIP_TRY(prk::launch_fill_i32(reinterpret_cast<int32_t *>(b.nn_qcount), (size_t)prk::kQCountStride * P, 0, g->stream));     // (a kernel, not a memset command: no runtime copy / fill path on a per-call path, see below)
    }

    prk::PoseMeta *h_meta = g->h_meta.as<prk::PoseMeta>();
    float *h_sums = g->h_sums.as<float>();
    pr_result *res = g->h_results.as<pr_result>();
    for (uint32_t i = 0; i < P; ++i) {
        identity16(res[i].T); res[i].inlier_rmse = 0.0f; res[i].fitness = 0.0f;   // icp.h:29-31
        std::memset(&h_meta[i], 0, sizeof(prk::PoseMeta));
        h_meta[i].start = start_h[i]; h_meta[i].count = count_h[i];
        h_meta[i].state = (count_h[i] > 0) ? prk::kRun : prk::kSkip;    // empty cloud: count==0 -> identity result (icp.cu:183)
    }

    if (opt.solve_mode == PR_SOLVE_DEVICE) {
        PR_TRY(g->dstate.ensure(sizeof(prk::DevIcpState) * P));
        PR_TRY(g->h_dstate.ensure(sizeof(prk::DevIcpState) * P));
        prk::DevIcpState *init = g->h_dstate.as<prk::DevIcpState>();
        for (uint32_t i = 0; i < P; ++i) { identity16(init[i].T); init[i].fitness = 0; init[i].rmse = 0; init[i].done = (h_meta[i].state == prk::kSkip); init[i].passes = 0; }
        pr_result *dres = results_dev;
        if (!dres) { PR_TRY(g->dresults.ensure(sizeof(pr_result) * P)); dres = g->dresults.as<pr_result>(); }
        const bool may_exit_early = (crit.relative_fitness > 0.0f && crit.relative_rmse > 0.0f);
        const bool fused = opt.fused_solve != 0;
        if (fused) PR_TRY(g->arrive.ensure(sizeof(uint32_t) * P));

        // profile==2: every sample_period-th call is a timed call -- it runs synchronously, as one pose group, with the other slot
        // drained, and times every correspondence launch of its loop
        const uint64_t tick = g->sample_clock++;
        const bool sample_call = (opt.profile == 2) && (tick % (uint64_t)std::max(1, opt.sample_period) == 0);
        // Pose groups: the batch is split over up to four streams so that one group's serial solve tail (and the ragged end
        // of its pass) overlaps another group's pass.  Timed launches (profile 1, the sampled call of profile 2) run as a
        // single group so that the measured kernel has the chip to itself.
        const bool timed_call = (opt.profile == 1 || opt.profile == 3) || sample_call;   // (profile 3 only keeps SUBMITTED batches asynchronous: whatever runs here is a synchronous timed call)
        const uint32_t n_groups = timed_call ? 1u : std::max(1u, std::min({ pose_groups_for(sc.kind), 4u, P / 32u }));
        // the fixed sequence: state upload, (max_iteration+1) x [correspondence pass, finalize+solve], result pack
        auto enqueue_all = [&](std::vector<hipEvent_t> *, bool host_checks) -> int {
            // start state and per-hypothesis records: PULLED from the pinned arrays by a kernel, results pushed into the pinned array by one -- no copy
            // command on a per-call path (round 6: the runtime's copy-engine path is where one-off multi-millisecond stalls came from, see below)
            void *init_dev = nullptr, *meta_host_dev = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&init_dev, init, 0));
            HIP_TRY(hipHostGetDevicePointer(&meta_host_dev, h_meta, 0));
            HIP_TRY(prk::launch_stage_words(init_dev, g->dstate.p, sizeof(prk::DevIcpState) * P, g->stream));
            HIP_TRY(prk::launch_stage_words(meta_host_dev, g->meta.p, sizeof(prk::PoseMeta) * P, g->stream));
            if (fused) HIP_TRY(prk::launch_fill_i32(g->arrive.as<int32_t>(), P, 0, g->stream));
            // Two pose groups on two streams: while the (latency-bound, one wavefront per pose) finalize+solve
            // of one group runs, the correspondence pass of the other group keeps the chip busy.
            auto group_begin = [&](uint32_t grp) { return (uint32_t)(((uint64_t)P * grp) / n_groups); };
            if (n_groups > 1) {
                for (uint32_t k = 1; k < n_groups; ++k) PR_TRY(ensure_stream(g->side[k - 1], &g->ev_join[k - 1]));
                HIP_TRY(hipEventRecord(g->ev_fork, g->stream));
                for (uint32_t k = 1; k < n_groups; ++k) HIP_TRY(hipStreamWaitEvent(g->side[k - 1], g->ev_fork, 0));
            }
            for (uint32_t it = 0; it <= (uint32_t)crit.max_iteration; ++it) {
                for (uint32_t grp = 0; grp < n_groups; ++grp) {
                    const uint32_t p0 = group_begin(grp), np = group_begin(grp + 1) - p0;
                    hipStream_t st = grp ? g->side[grp - 1] : g->stream;
                    prk::IcpBatch bb = b;
                    bb.meta += p0; bb.partial += (size_t)p0 * nblk * prk::kAccStride; if (bb.nn_qcount) bb.nn_qcount += prk::kQCountStride * (size_t)p0;
                    bb.iter = it;
                    if (fused) { bb.fused = 1; bb.crit = crit; bb.st = g->dstate.as<prk::DevIcpState>() + p0; bb.arrive = g->arrive.as<uint32_t>() + p0; }
                    bb.score_only = (it == (uint32_t)crit.max_iteration) ? 1u : 0u;
                    if (grp == 0 && timed_call) {           // a timed call times every launch of its loop
                        SpanGuard sp(kSpanIcp); HIP_TRY(launch_pass(bb, sc, np, st));
                        uint64_t pts = 0; for (uint32_t i = p0; i < p0 + np; ++i) pts += count_h[i];
                        // algorithmic bytes per point: 12 read + 24 gathered, + 12 written back once a transform is
                        // pending; the score-only last pass needs the scene point but not its normal (12 + 12 + 12)
                        const bool first = (it == 0), last = (it == (uint32_t)crit.max_iteration);
                        g->icp_points += pts; g->icp_bytes += pts * ((first || last) ? 36u : 48u);
                    } else HIP_TRY(launch_pass(bb, sc, np, st));
                    if (!fused) HIP_TRY(prk::launch_icp_finalize_solve(bb.partial, g->meta.as<prk::PoseMeta>() + p0, nblk, steps,
                                                                       g->dstate.as<prk::DevIcpState>() + p0, crit, it, np, st));
                }
                if (host_checks && may_exit_early && (it & 3) == 3 && it < (uint32_t)crit.max_iteration) {
                    for (uint32_t k = 1; k < n_groups; ++k) HIP_TRY(hipStreamSynchronize(g->side[k - 1]));
                    HIP_TRY(hipMemcpyAsync(h_meta, g->meta.p, sizeof(prk::PoseMeta) * P, hipMemcpyDeviceToHost, g->stream));
                    HIP_TRY(hipStreamSynchronize(g->stream));
                    bool any = false;
                    for (uint32_t i = 0; i < P; ++i) any |= (h_meta[i].state != prk::kSkip);
                    if (!any) break;
                }
            }
            for (uint32_t k = 1; k < n_groups; ++k) { HIP_TRY(hipEventRecord(g->ev_join[k - 1], g->side[k - 1])); HIP_TRY(hipStreamWaitEvent(g->stream, g->ev_join[k - 1], 0)); }
            HIP_TRY(prk::launch_pack_results(g->dstate.as<prk::DevIcpState>(), dres, P, g->stream));
            // results go to the pinned staging buffer (a pageable destination is not capturable)
            if (results_host) {
                void *res_dev = nullptr;
                HIP_TRY(hipHostGetDevicePointer(&res_dev, res, 0));
                HIP_TRY(prk::launch_stage_words64(dres, res_dev, sizeof(pr_result) * P, g->stream));
            }
            return PR_OK;
        };

        if (opt.use_graph && n_groups == 1 && (opt.profile == 0 || (opt.profile == 2 && !sample_call))) {   // HIP events recorded inside a captured graph cannot be timed
            GraphKey key;
            key.add(P); key.add(nblk); key.add(steps); key.add(crit); key.add(sc); key.add(cloud_base); key.add(g->meta.p); key.add(g->partial.p);
            key.add(g->dstate.p); key.add(dres); key.add(results_host != nullptr); key.add(res); key.add(h_meta); key.add(init); key.add(opt.profile); key.add(pose_groups_for(sc.kind)); key.add(opt.fused_solve); key.add(g->arrive.p); key.add(b.nn_prev);
            key.add(b.nn_slack); key.add(b.nn_queue); key.add(b.nn_queue2); key.add(b.nn_qcount);    // laid out behind nn_prev at multiples of `span` (max start+count): same P / max_n, other offsets => other addresses
            CachedGraph *hit = nullptr;
            for (auto &c : g->graphs) if (c.exec && c.key == key) { hit = &c; break; }
            if (!hit) {
                if (g->graphs.size() >= 8) {                         // evict the least recently used entry
                    size_t lru = 0;
                    for (size_t i = 1; i < g->graphs.size(); ++i) if (g->graphs[i].stamp < g->graphs[lru].stamp) lru = i;
                    destroy_graph(g->graphs[lru]);
                    g->graphs.erase(g->graphs.begin() + lru);
                }
                CachedGraph c; c.key = key;
                HIP_TRY(hipStreamSynchronize(g->stream));
                HIP_TRY(hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal));
                int rc = enqueue_all(nullptr, /*host_checks=*/false);
                hipError_t ce = hipStreamEndCapture(g->stream, &c.graph);
                if (rc != PR_OK || ce != hipSuccess) { destroy_graph(c); if (rc == PR_OK) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce)); rc = PR_ERR_HIP; } return rc; }
                HIP_TRY(hipGraphInstantiate(&c.exec, c.graph, nullptr, nullptr, 0));
                g->graphs.push_back(std::move(c));
                hit = &g->graphs.back();
            }
            hit->stamp = ++g->graph_clock;
            HIP_TRY(hipGraphLaunch(hit->exec, g->stream));
            HIP_TRY(hipStreamSynchronize(g->stream));
            if (results_host) std::memcpy(results_host, res, sizeof(pr_result) * P);
            drain_spans();
            return PR_OK;
        }

        PR_TRY(enqueue_all(nullptr, /*host_checks=*/true));
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (results_host) std::memcpy(results_host, res, sizeof(pr_result) * P);
        drain_spans();
        return PR_OK;
    }

    // PR_SOLVE_HOST: per iteration one launch + one small D2H, then the per-pose logic of icp.cu:178-212 on the host.  The batch runs
    // as up to four pose groups, each on a stream of its own, software-pipelined: while the host waits for, solves and re-arms one
    // group, the correspondence pass of the next group is on the chip -- the host's share of an iteration (copy latency, wake-up,
    // 160 ns per 6x6 solve) is as long as the pass itself, and a single group leaves the GPU idle for all of it.  Results per
    // hypothesis do not depend on the grouping (sums, solve and state are per hypothesis).  A timed call (profile 1) is one group.
    // profile 2: one call in sample_period times ONE of its passes (a different iteration from call to call); that call runs as one pose
    // group like the calls of profile 1 / 3, so that the timed launch has the chip to itself -- the other calls keep their pipeline
    const uint64_t host_tick = g->sample_clock++;
    const bool host_sample_call = (opt.profile == 2) && (host_tick % (uint64_t)std::max(1, opt.sample_period) == 0);
    const uint32_t host_sample_it = (uint32_t)(((host_tick / (uint64_t)std::max(1, opt.sample_period)) * 5 + 1) % (uint64_t)(crit.max_iteration + 1));
    const uint32_t n_groups = (opt.profile == 1 || opt.profile == 3 || host_sample_call) ? 1u : std::max(1u, std::min({ pose_groups_for(sc.kind), 4u, P / 32u }));
    auto group_begin = [&](uint32_t grp) { return (uint32_t)(((uint64_t)P * grp) / n_groups); };
    if (opt.fused_solve) { PR_TRY(g->arrive.ensure(sizeof(uint32_t) * (P + 4))); HIP_TRY(prk::launch_fill_i32(g->arrive.as<int32_t>(), (size_t)P + 4, 0, g->stream)); }   // (+ one counter per pose group)
    if (n_groups > 1) {
        for (uint32_t k = 1; k < n_groups; ++k) PR_TRY(ensure_stream(g->side[k - 1], &g->ev_join[k - 1]));
        HIP_TRY(hipEventRecord(g->ev_fork, g->stream));               // the groups start behind the render / cloud work of this call
        for (uint32_t k = 1; k < n_groups; ++k) HIP_TRY(hipStreamWaitEvent(g->side[k - 1], g->ev_fork, 0));
    }
    auto group_stream = [&](uint32_t grp) { return grp ? g->side[grp - 1] : g->stream; };
    // With option fused_solve the pass also adds up the workgroup sums of a hypothesis (the workgroup that arrives last does, as in the
    // device-solve loop) and stores the 29 totals straight into the pinned host array: no finalize launch, no copy command -- two
    // API calls and one copy-engine round trip less per group and iteration on a path that is bound by exactly those.
    const bool host_fused = opt.fused_solve != 0;
    float *sums_dev = nullptr;
    if (host_fused) {
        PR_TRY(g->arrive.ensure(sizeof(uint32_t) * (P + 4)));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&sums_dev), h_sums, 0));
    }
    // Projective scenes also read the per-hypothesis state (64 bytes: cloud span, pending update) from the pinned host array instead
    // of receiving it through a copy command: one uniform load per workgroup over the host link, 1.62 -> 1.54 ms per 256-hypothesis
    // batch.  The four kernels of a kd-tree pass have ten times the workgroups; there the copy is cheaper (8.3 against 8.8 ms).
    const prk::PoseMeta *meta_dev = nullptr, *h_meta_mapped = nullptr;          // h_meta as the device sees it
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(const_cast<prk::PoseMeta **>(&h_meta_mapped)), h_meta, 0));
    if (host_fused && sc.kind != PR_SCENE_NN) HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(const_cast<prk::PoseMeta **>(&meta_dev)), h_meta, 0));
    // Round 6: the host waits for a pose group's FLAG in pinned memory (stored by the workgroup that delivers the group's last
    // hypothesis) instead of for the stream -- it solves while the launch winds down.  Option blocking_wait (several ranks sharing few CPUs) and
    // host_poll = 0 keep the stream wait.
    const bool poll_flag = host_fused && !opt.blocking_wait && opt.host_poll;
    volatile uint32_t *h_flags = reinterpret_cast<volatile uint32_t *>(h_sums + (size_t)P * prk::kAccStride);
    if (poll_flag) { for (int k = 0; k < 4; ++k) h_flags[k] = 0u; for (uint32_t i = 0; i < P; ++i) reinterpret_cast<volatile uint32_t *>(h_sums + (size_t)i * prk::kAccStride)[31] = 0u; }
    // one iteration of one group goes onto its stream: state upload, pass, block sums -> pose sums, download
    auto enqueue_group = [&](uint32_t grp, uint32_t it) -> int {
        const uint32_t p0 = group_begin(grp), np = group_begin(grp + 1) - p0;
        hipStream_t st = group_stream(grp);
        prk::IcpBatch bb = b;
        if (meta_dev) bb.meta = meta_dev;                            // the pass reads the 64-byte state of its hypothesis from the pinned host array
        else HIP_TRY(prk::launch_stage_words(h_meta_mapped + p0, g->meta.as<prk::PoseMeta>() + p0, sizeof(prk::PoseMeta) * np, st));   // (pulled by a kernel: no copy command per iteration)
        bb.meta += p0; bb.partial += (size_t)p0 * nblk * prk::kAccStride; if (bb.nn_qcount) bb.nn_qcount += prk::kQCountStride * (size_t)p0;
        bb.iter = it;
        if (host_fused) { bb.fused = 2; bb.arrive = g->arrive.as<uint32_t>() + p0; bb.sums_out = sums_dev + (size_t)p0 * prk::kAccStride; }
        if (poll_flag) {                                             // the group's count and flag (see pass_deliver): how many hypotheses will deliver
            uint32_t expected = 0;
            for (uint32_t i = p0; i < p0 + np; ++i) expected += (h_meta[i].state != prk::kSkip) ? 1u : 0u;
            bb.grp_count = g->arrive.as<uint32_t>() + P + grp; bb.grp_expected = expected;
            bb.grp_flag = reinterpret_cast<uint32_t *>(sums_dev + (size_t)P * prk::kAccStride) + grp;
        }
        if (opt.profile == 1 || opt.profile == 3 || (host_sample_call && it == host_sample_it)) {
            SpanGuard sp(kSpanIcp); HIP_TRY(launch_pass(bb, sc, np, st));
            for (uint32_t i = p0; i < p0 + np; ++i) if (h_meta[i].state != prk::kSkip) { g->icp_points += count_h[i]; g->icp_bytes += (uint64_t)count_h[i] * (it == 0 ? 36u : 48u); }
        } else HIP_TRY(launch_pass(bb, sc, np, st));
        if (!host_fused) {
            HIP_TRY(prk::launch_icp_finalize(bb.partial, g->meta.as<prk::PoseMeta>() + p0, nblk, steps, g->sums.as<float>() + (size_t)p0 * prk::kAccStride, np, st));
            HIP_TRY(hipMemcpyAsync(h_sums + (size_t)p0 * prk::kAccStride, g->sums.as<float>() + (size_t)p0 * prk::kAccStride, sizeof(float) * prk::kAccStride * np,
                                   hipMemcpyDeviceToHost, st));
        }
        return PR_OK;
    };
    // the host's part of an iteration for one hypothesis whose 29 sums have arrived (icp.cu:178-212); returns whether the hypothesis goes on
    auto solve_one = [&](uint32_t i, uint32_t it) -> bool {
        const float *Ab = h_sums + (size_t)i * prk::kAccStride;
        pr_result &r = res[i];
        const float prev_fit = r.fitness, prev_rmse = r.inlier_rmse;
        const float cnt = Ab[28], err = Ab[27];
        if (cnt == 0) { h_meta[i].state = prk::kSkip; return false; }                         // icp.cu:183
        r.fitness = cnt / (float)count_h[i];                                              // icp.cu:185
        r.inlier_rmse = std::sqrt(err / cnt);                                             // icp.cu:186
        if (it == (uint32_t)crit.max_iteration) { h_meta[i].state = prk::kSkip; return false; }    // icp.cu:189
        if (std::fabs(r.fitness - prev_fit) < crit.relative_fitness &&
            std::fabs(r.inlier_rmse - prev_rmse) < crit.relative_rmse) { h_meta[i].state = prk::kSkip; return false; }   // icp.cu:191-194
        float A[36], bb[6], E[16];
        for (int k = 0; k < 6; ++k) bb[k] = Ab[21 + k];
        int sh = 0;
        for (int y = 0; y < 6; ++y) for (int x = y; x < 6; ++x) { A[x + y * 6] = Ab[sh]; A[y + x * 6] = Ab[sh]; ++sh; }   // icp.cu:196-205
        prh::solve_666(A, bb, E);
        std::memcpy(h_meta[i].xform, E, sizeof(float) * 12);
        prh::mat4_mul(E, r.T, r.T);                                                       // icp.cu:212
        h_meta[i].state = prk::kRunWithTransform;
        return true;
    };
    // returns how many hypotheses of the group go on; < 0: a HIP error
    auto finish_group = [&](uint32_t grp, uint32_t it) -> int {
        if (!poll_flag) {
            if (hipError_t e = hipStreamSynchronize(group_stream(grp)); e != hipSuccess) { set_error("HIP error: %s", hipGetErrorString(e)); return -1; }
        } else {
            const uint32_t tag = it + 1u;
            uint32_t idle = 0;
            bool drained = false;
            while (h_flags[grp] != tag) {
                if (drained) { set_error("PR_SOLVE_HOST: the pass of iteration %u ended without completing its pose group", it); return -1; }
                if (++idle >= 20000u) {                              // every few hundred microseconds of fruitless polling: is the stream still alive?
                    idle = 0;
                    const hipError_t q = hipStreamQuery(group_stream(grp));
                    if (q == hipSuccess) drained = true;           // (one more look: the flag was stored before the kernel ended)
                    else if (q != hipErrorNotReady) { (void)hipGetLastError(); set_error("HIP error: %s", hipGetErrorString(q)); return -1; }
                }
                __builtin_ia32_pause();
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            // every row that was due carries the iteration's tag behind its sums (pass_deliver): a row without it was overtaken by the flag -- wait for the
            // stream (the end of the launch makes everything visible) instead of solving on stale sums
            bool all_in = true;
#if PR_HOST_ROW_TAG
            for (uint32_t i = group_begin(grp); i < group_begin(grp + 1) && all_in; ++i)
                if (h_meta[i].state != prk::kSkip && reinterpret_cast<volatile uint32_t *>(h_sums + (size_t)i * prk::kAccStride)[31] != tag) all_in = false;
#endif
            if (!all_in) {
                g_flag_overtook.fetch_add(1);
                if (hipError_t e = hipStreamSynchronize(group_stream(grp)); e != hipSuccess) { set_error("HIP error: %s", hipGetErrorString(e)); return -1; }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        int active = 0;
        for (uint32_t i = group_begin(grp); i < group_begin(grp + 1); ++i) if (h_meta[i].state != prk::kSkip && solve_one(i, it)) ++active;
        return active;
    };
    bool live[4] = { false, false, false, false };
    int rc_loop = PR_OK;
    for (uint32_t grp = 0; grp < n_groups && rc_loop == PR_OK; ++grp) {
        for (uint32_t i = group_begin(grp); i < group_begin(grp + 1); ++i) live[grp] |= (h_meta[i].state != prk::kSkip);
        if (live[grp]) rc_loop = enqueue_group(grp, 0);
    }
    for (uint32_t it = 0; it <= (uint32_t)crit.max_iteration && rc_loop == PR_OK && (live[0] || live[1] || live[2] || live[3]); ++it) {
        for (uint32_t grp = 0; grp < n_groups && rc_loop == PR_OK; ++grp) {
            if (!live[grp]) continue;
            const int active = finish_group(grp, it);
            if (active < 0) { rc_loop = PR_ERR_HIP; break; }
            live[grp] = active > 0;
            if (live[grp]) rc_loop = enqueue_group(grp, it + 1);                          // it + 1 <= max_iteration: the last iteration leaves nobody active
        }
    }
    for (uint32_t k = 1; k < n_groups; ++k) (void)hipStreamSynchronize(g->side[k - 1]);  // nothing of this call is left on a side stream, error or not
    if (poll_flag) (void)hipStreamSynchronize(g->stream);         // (a flag arrives before its pass has formally ended: nothing of this call runs on after it returns)
    trace_mark("icp_drive: host loop done");
    if (rc_loop != PR_OK) { (void)hipStreamSynchronize(g->stream); drain_spans(); return rc_loop; }
    if (results_host) std::memcpy(results_host, res, sizeof(pr_result) * P);
    if (results_dev) {
        // the records are PULLED from the pinned array by a kernel, not pushed by a copy command: round 6 traced the one-off stall of the host-solve
        // pipeline (one step of 6.5-8.7 ms among forty of 1.0 ms, PR_TRACE) to this 18 KB hipMemcpyAsync -- some batches into a helper thread's life
        // the runtime spends milliseconds inside it, on both helpers at once (its copy-engine path; the kernels around it take microseconds)
        void *res_dev = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&res_dev, res, 0));
        HIP_TRY(prk::launch_stage_words64(res_dev, results_dev, sizeof(pr_result) * P, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
    }
    drain_spans();
    return PR_OK;
}

}  // namespace prr

using namespace prr;

extern "C" {

int pr_icp_batch(pr_vec3 *clouds_dev, const uint32_t *offsets_host, uint32_t n_clouds, int scene_kind, const void *scene,
                 pr_criteria crit, pr_result *results_host)
{
    PR_ENTER();
    if (!offsets_host || (n_clouds && !results_host)) { set_error("pr_icp_batch: bad arguments"); return PR_ERR_INVALID; }
    if (n_clouds == 0) return PR_OK;
    if (!clouds_dev) {                                            // clouds without a single point come without an array (an empty device_vector):
        if (offsets_host[n_clouds] != offsets_host[0]) { set_error("pr_icp_batch: null cloud array"); return PR_ERR_INVALID; }
        for (uint32_t i = 0; i < n_clouds; ++i) {                 // count == 0 -> identity, fitness 0, rmse 0 (icp.cu:183), like any empty cloud
            identity16(results_host[i].T); results_host[i].fitness = 0.0f; results_host[i].inlier_rmse = 0.0f;
        }
        return PR_OK;
    }
    SceneSel sc;
    std::memset(static_cast<void *>(&sc), 0, sizeof sc);
    PR_TRY(make_scene(scene_kind, scene, /*want_packed=*/false, sc));
    std::vector<uint32_t> start(n_clouds), count(n_clouds);
    for (uint32_t i = 0; i < n_clouds; ++i) {
        if (offsets_host[i + 1] < offsets_host[i]) { set_error("pr_icp_batch: offsets must be non-decreasing"); return PR_ERR_INVALID; }
        start[i] = offsets_host[i]; count[i] = offsets_host[i + 1] - offsets_host[i];
    }
    // the hypothesis index is the y dimension of every launch of the loop (65 535 at most): longer lists run in pieces, one after the other
    // (the clouds are independent of each other, so the pieces are)
#ifndef PR_MAX_CLOUDS_PER_RUN
#define PR_MAX_CLOUDS_PER_RUN 32768                               // (a larger value in an experiment build exercises the launchers' own split)
#endif
    constexpr uint32_t kMaxCloudsPerRun = PR_MAX_CLOUDS_PER_RUN;
    for (uint32_t c0 = 0; c0 < n_clouds; c0 += kMaxCloudsPerRun) {
        const uint32_t nc = std::min(kMaxCloudsPerRun, n_clouds - c0);
        PR_TRY(icp_drive(clouds_dev, start.data() + c0, count.data() + c0, nc, sc, crit, results_host + c0, nullptr));
    }
    return PR_OK;
}

int pr_icp_proj(pr_vec3 *cloud_dev, uint32_t n_points, const pr_scene_proj *scene, pr_criteria crit, pr_result *result_out)
{
    const uint32_t off[2] = { 0, n_points };
    return pr_icp_batch(cloud_dev, off, 1, PR_SCENE_PROJ, scene, crit, result_out);
}

int pr_icp_nn(pr_vec3 *cloud_dev, uint32_t n_points, const pr_scene_nn *scene, pr_criteria crit, pr_result *result_out)
{
    const uint32_t off[2] = { 0, n_points };
    return pr_icp_batch(cloud_dev, off, 1, PR_SCENE_NN, scene, crit, result_out);
}

// Audit entry: what ONE correspondence pass contributes, point by point (29 floats each: the 21 upper-triangle products of J J^T row by
// row, the 6 products J r, r^2 and the inlier flag -- thrust__pcd2Ab, icp.h:128-209), with the pending update applied to the cloud first
// like the fused pass does.  A caller that adds the rows in point order reproduces the reference's single-thread sums (icp.cpp:139-148).
int pr_debug_contrib29(pr_vec3 *cloud_dev, uint32_t n_points, int scene_kind, const void *scene, const float *update16, int want_packed, float *contrib_host)
{
    PR_ENTER();
    if (n_points == 0) return PR_OK;
    if (!cloud_dev || !contrib_host) { set_error("pr_debug_contrib29: null buffer"); return PR_ERR_INVALID; }
    SceneSel sc;
    std::memset(static_cast<void *>(&sc), 0, sizeof sc);
    PR_TRY(make_scene(scene_kind, scene, want_packed != 0, sc));
    DevBuf out;
    PR_TRY(out.ensure(sizeof(float) * 29 * (size_t)n_points));
    hipError_t e;
    if (sc.kind == PR_SCENE_NN) e = prk::launch_contrib29_nn(cloud_dev, n_points, update16, sc.nn, out.as<float>(), g->stream);
    else if (sc.packed) e = prk::launch_contrib29_proj_packed(cloud_dev, n_points, update16, sc.pk, out.as<float>(), g->stream);
    else e = prk::launch_contrib29_proj_aos(cloud_dev, n_points, update16, sc.aos, out.as<float>(), g->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(contrib_host, out.p, sizeof(float) * 29 * (size_t)n_points, hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    note_write(cloud_dev, sizeof(pr_vec3) * (size_t)n_points);
    out.release();
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("pr_debug_contrib29: %s", hipGetErrorString(e)); return PR_ERR_HIP; }
    return PR_OK;
}

// work counters of the kd-tree search kernel (option "nn_count"): out[pass * 8 + k], k = queries, settled by the pixel window,
// handed to the tree, pyramid descents, tree nodes visited, leaves scanned, leaf points tested, window cells read; reading resets them
int pr_nn_counters(uint64_t *out, uint32_t passes)
{
    PR_ENTER();
    if (!out || passes == 0 || passes > kCounterPasses) { set_error("pr_nn_counters: bad arguments"); return PR_ERR_INVALID; }
    std::memset(out, 0, sizeof(uint64_t) * 8 * passes);
    if (!g->nn_counters.p) return PR_OK;
    HIP_TRY(hipStreamSynchronize(g->stream));
    HIP_TRY(hipMemcpy(out, g->nn_counters.p, sizeof(uint64_t) * 8 * passes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(g->nn_counters.p, 0, sizeof(unsigned long long) * 8 * kCounterPasses));
    return PR_OK;
}

}  // extern "C"
