// nn_search.hip -- Scene_nn::query (pcd_scene.h:60-136) as kernels of their own: search (keep-the-winner, pixel window), bound (descent + window), task walk over wide nodes, binary walk; pixel-grid construction
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#include "pr_launch.h"
#include "nn_query.h"

namespace prk {

// ================================================================================================
//  kd-tree scenes, split form: the SEARCH on its own.
//
//  The fused pass ties the search to the canonical reduction tree: lane t owns points 4t..4t+3 of a 1024-point step, so only
//  three of four queries can start from their neighbour's answer, and the unseeded ones -- a hundred node visits each while the
//  hypothesis is still centimetres off the surface -- set the pace of the first passes.  The winner of a query does not depend
//  on how the search is organised, so the search runs here in whatever order suits it and leaves the winners in `nn_prev`;
//  the pass that follows (icp_pass_kernel<SceneNNWinners>) gathers them in canonical order -- bit-identical sums.
//    * a lane walks a RUN of consecutive cloud points (image neighbours): every query but the first of a run starts from the
//      previous one's winner, whose distance is within microns of the answer (the step between neighbours is lateral);
//    * the previous pass' winner of the same point bounds the search as before;
//    * the first query of a run takes a bound from the scene points around its own pixel (grid_seed_bound);
//    * with a bound that tight the candidates are the handful of scene pixels around the query's pixel: grid_search settles the
//      query without touching the tree unless there is a tie (the tree's visiting order then decides, as in the reference).
//  The pending rigid update is applied (and written back) here, so the pass reads the cloud as it is.
// ================================================================================================
__global__ __launch_bounds__(256) void nn_search_kernel(IcpBatch b, SceneNNDev scene, uint32_t run)
{
    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];
    const int32_t st = pm.state;
    if (st == kSkip) return;
    const uint32_t n = pm.count;
    const uint32_t first = blockIdx.x * kBlockThreads * run;
    if (first >= n) return;

    float *cl = reinterpret_cast<float *>(b.cloud + pm.start);
    uint32_t *win = b.nn_prev + pm.start;
    const bool xf = (st == kRunWithTransform);                   // also: the previous pass left winners for this cloud
    float M[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) M[i] = xf ? pm.xform[i] : 0.0f;
    const float accept = scene.max_dist_diff * scene.max_dist_diff;

    // Two phases.  (1) Every lane takes one point per chunk (lane t of chunk k: point first + 256 k + t -- adjacent lanes hold
    // adjacent points), applies the pending update and tries the cheap way: the bound from the previous pass' winner and, if that
    // bound is a pixel or two wide, the exact window scan.  (2) What is left -- a few per cent of the points once aligned, most of
    // them on the first passes -- is packed into dense lanes (in point order, so neighbours stay neighbours) and only then pays for
    // the descent through the representative points and the tree search.  Without the packing four of five wavefronts ran a whole
    // tree search for two or three of their lanes.
    // the hypothesis' queue lives next to its winners (same indexing, so it can never overflow): (point, bound) per entry, filled
    // through this pass' counter of the hypothesis (two counters alternate between passes; nn_tree_kernel re-arms the idle one)
    uint2 *queue = b.nn_queue + pm.start;
    uint32_t *q_count = b.nn_qcount + kQCountStride * pose + (b.iter & 1u);
    __shared__ uint32_t wg_count, wg_base;
    __shared__ uint32_t wave_base[4];
    if (threadIdx.x == 0) wg_count = 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t j0 = first + threadIdx.x;
    NNCount cnt;
    uint32_t n_query = 0, n_window = 0, n_cells = 0, n_kept = 0;
    float *slk = b.nn_slack + pm.start;
    for (uint32_t k = 0; k < run; ++k) {
        const uint32_t j = j0 + k * kBlockThreads;
        const bool live = j < n;
        bool pending = false;
        float best = accept;
        if (live) {
            pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);
            // (round 6: the point's previous winner and slack are requested together with the point -- behind the write-back below they were two more
            // dependent trips to memory per chunk: the compiler may not move a load across a store it cannot tell apart.  0.88 -> 0.84 ms over the 21 passes,
            // configs[2] 53.9-54.1 k against 53.2-53.4 k poses/s.  Also fetching the NEXT chunk's three values ahead: 0.80 ms alone, 53.7 k pipelined -- not kept.)
            const uint32_t prev = xf ? win[j] : kNoPrev;
            const float slack_in = xf ? slk[j] : 0.0f;
            float x = q.x, y = q.y, z = q.z;
            bool still = false;                                  // has this point (nearly) kept the position its winner is from?
            float step_sq = 0.0f;
            if (xf) {                                            // icp.cu:142-153 transform_pcd_cuda
                const float tx = M[0] * x + M[1] * y + M[2]  * z + M[3];
                const float ty = M[4] * x + M[5] * y + M[6]  * z + M[7];
                const float tz = M[8] * x + M[9] * y + M[10] * z + M[11];
                step_sq = (tx - x) * (tx - x) + (ty - y) * (ty - y) + (tz - z) * (tz - z);
                still = step_sq <= PR_NN_STILL;
                x = tx; y = ty; z = tz;
                st_off<pr_vec3>(cl, j * 12u, pr_vec3{ x, y, z });
            }
            ++n_query;
            bool kept = false;
            if (prev != kNoPrev) {
                // KEEP THE WINNER WITHOUT SEARCHING.  `slack` is a lower bound on the distance from this point to every scene point
                // other than its winner, already reduced by every step the point has taken since the bound was established (a step
                // of length s changes any distance by at most s).  If the winner's distance -- computed here with the search's own
                // expression -- is below that bound by more than the float error of a squared distance (3e-7 relative; 1e-5 is
                // demanded), the winner is still the unique strict minimum, which is what the reference's search returns under any
                // visiting order.
                const pr_vec3 pw = ld_off<pr_vec3>(scene.pcd, prev * 12u);
                const float d2 = (x - pw.x) * (x - pw.x) + (y - pw.y) * (y - pw.y) + (z - pw.z) * (z - pw.z);
                const float slack = slack_in - margin_sqrt(step_sq) * 1.000002f;
                if (d2 < accept && margin_sqrt(d2) * 1.00001f < slack) { kept = true; slk[j] = slack; ++n_kept; }
                else { const float bnd = d2 * 1.000001f + 1e-30f; if (bnd < best) best = bnd; }              // = nn_seed_bound
            }
            if (!kept) {
                // Everything else goes to the bound kernel -- the pixel-window scan too (round 4).  Done here, a window scan ran for the few
                // lanes of a wavefront that needed it while the other lanes waited: with 1 % of the points in the window half of all
                // wavefronts paid the 400 instructions of a scan, with 25-85 % (passes 2-8) every one did.  Queued, the scans run in dense lanes.
                // The entry's bound carries two flags: its sign says "no descent needed" (a previous winner and a step below PR_NN_NODESCENT),
                // its lowest mantissa bit says "settle" (the point has stopped moving: take the widest cover the window holds).  Setting
                // or clearing that bit moves the bound by one ulp, far inside the 1e-6 by which a seed bound is inflated.
                pending = true;
                uint32_t bits = __float_as_uint(best) & ~1u;
                if (still && PR_NN_SETTLE) bits |= 1u;
                if (prev != kNoPrev && step_sq <= PR_NN_NODESCENT) bits |= 0x80000000u;
                best = __uint_as_float(bits);
            }
        }
        // one slot range per workgroup and chunk: waves in order, lanes in order -- the queue keeps the points' order
        const unsigned long long m = __ballot(pending);
        if (lane == 0) wave_base[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t total = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
            wg_base = total ? atomicAdd(q_count, total) : 0u;
        }
        __syncthreads();
        if (pending) {
            uint32_t slot = wg_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            for (uint32_t w2 = 0; w2 < wave; ++w2) slot += wave_base[w2];
            queue[slot] = make_uint2(j, __float_as_uint(best));
        }
        __syncthreads();
    }
    if (scene.counters) {                                        // instrumented runs only (option "nn_count")
        const uint32_t v[8] = { n_query, n_window, 0u, 0u, 0u, 0u, 0u, n_cells };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if ((threadIdx.x & 63u) == 0u && t) atomicAdd(&scene.counters[(size_t)b.iter * 8 + i], (unsigned long long)t);
        }
    }
}

// Second half of the search: the queued queries of every hypothesis in DENSE lanes (queue order = point order, so neighbouring lanes
// still hold neighbouring points).  A fixed number of workgroups per hypothesis walks the queue in chunks of 256; per-lane stacks in
// LDS as before.  Kept apart from the first half so that a handful of tree searches no longer pins a whole workgroup, its 32 KB of
// stacks and its place on the CU for fifteen dependent round trips while the other 250 lanes have long finished.
template <int kCode>
__global__ __launch_bounds__(256) void nn_tree_kernel(IcpBatch b, SceneNNDev scene)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];
    if (pm.state == kSkip) return;
    uint32_t *counts = b.nn_qcount + kQCountStride * pose;
    const uint32_t queued = counts[b.iter & 1u];
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[(b.iter + 1u) & 1u] = 0u;         // re-arm the counter the NEXT pass will fill
    if (blockIdx.x * kBlockThreads >= queued) return;
    int *stk_node = reinterpret_cast<int *>(lds_raw) + threadIdx.x;
    float *stk_lb = reinterpret_cast<float *>(lds_raw) + (size_t)(kCode & 0xff) * kBlockThreads + threadIdx.x;
    const float *cl = reinterpret_cast<const float *>(b.cloud + pm.start);
    uint32_t *win = b.nn_prev + pm.start;
    float *slk = b.nn_slack + pm.start;
    const uint2 *queue = b.nn_queue + pm.start;
    const float accept = scene.max_dist_diff * scene.max_dist_diff;
    NNCount cnt;
    uint32_t n_window = 0, n_tree = 0, n_pyramid = 0, n_cells = 0;
    for (uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x; i < queued; i += gridDim.x * kBlockThreads) {
        const uint2 e = queue[i];
        const uint32_t j = e.x;
        const bool still = (e.y & 0x80000000u) != 0u, settle = (e.y & 1u) != 0u;     // flags of the entry (nn_search_kernel)
        float best = __uint_as_float((e.y & 0x7fffffffu) | 1u);
        const pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);          // as nn_search_kernel stored it
        const float x = q.x, y = q.y, z = q.z;
        // A point that has hardly moved still has (nearly) its true neighbour as temporal seed: d_old - step <= d_new <= d_old + step.
        // Only queries without such a seed -- the first passes, while the updates are still millimetres -- pay for the descent
        // through the representative points.
        uint32_t w = kNoPrev;
        bool settled = false;
        float other = 0.0f;
        if (scene.grid) {
            float bsq = 0.0f, osq = 0.0f;
            if (!still) { grid_pyramid_bound(scene, x, y, z, best); ++n_pyramid; }
            settled = best < accept && grid_search(scene, x, y, z, best, w, &n_cells, &bsq, &osq, settle);
            if (settled) other = sqrtf(osq) * 0.99999f;
        }
        if (settled) ++n_window;
        else {
            Corr c;
            ++n_tree;
            if (!query_nn_stack_from<kCode, false>(scene, nullptr, stk_node, stk_lb, x, y, z, c, best, w, &cnt)) w = kNoPrev;
        }
        win[j] = w;
        slk[j] = other;                                           // the ordered binary walk does not report its runner-up: no shortcut next pass
    }
    if (scene.counters) {                                        // instrumented runs only (option "nn_count")
        const uint32_t v[8] = { 0u, n_window, n_tree, n_pyramid, cnt.nodes, cnt.leaves, cnt.leaf_points, n_cells };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if ((threadIdx.x & 63u) == 0u) atomicAdd(&scene.counters[(size_t)b.iter * 8 + i], (unsigned long long)t);
        }
    }
}

// The queued queries of every hypothesis, searched over the wide records as TASKS (see "wide records" above).  Two walks over the same
// records came first and both lost to the binary per-lane walk's 5.1 ms in pass 0 (256 hypotheses, 5.4 M tree searches): one query per
// lane (eight 16-byte loads per visit and lane: 6.8 ms, bound by the rate of divergent loads) and one query per group of eight / four
// lanes with a stack per group (6.6 / 6.0 ms: every lane of a wavefront pays for the node branch AND the leaf branch of every iteration
// and for the longest walk among its groups -- VALU-bound at a quarter of the lanes doing useful work).  So the walk is cut into
// uniform pieces: a NODE TASK = (query, wide node): four lanes test the node's eight boxes, two each; a LEAF TASK = (query, leaf): four
// lanes test its points, two each per round.  A wavefront owns 64 queries at a time and two LIFO task queues in LDS; a step pops sixteen
// tasks of ONE kind (one per group of four lanes), so all lanes run the same code, and pushes what the boxes admit.  What a query has
// found so far lives in LDS and is updated with LDS atomics: `bound` (float bits, atomicMin), `best` = (distance bits << 32 | point index)
// (64-bit atomicMin: smallest distance, lowest index among equals), `tied` (smallest distance two different points were seen at),
// `second` (smallest distance / box bound of everything that is not the best).  When both queues are empty all 64 queries are finished:
// exactly one point at the minimum -> that is the reference's answer under any visiting order; a tie (tied == minimum) or a queue
// overflow -> the ordered stackless walk repeats the query from the minimum, as the reference would resolve it.
// The bound (descent through the representative points) and the exact pixel window run in a kernel of their own, nn_bound_kernel, which
// hands what it cannot settle to nn_tree_wide_kernel through a second queue: the descent needs twice the registers of the task walk, and
// in one kernel (as first built: 3.6 ms for pass 0) it held the walk to four wavefronts per SIMD and three workgroup barriers per chunk.
constexpr uint32_t kTaskQCap = PR_WIDE_QCAP, kTaskLCap = PR_WIDE_LCAP;   // entries of the node / leaf task queue of a wavefront (a leaf queue is drained from 16 entries on)
constexpr uint32_t kNoIdx = 0xffffffffu;
// Queue 1 (nn_search_kernel's leftovers) -> bound + pixel window, one query per lane -> winners, or queue 2 (point, bound).
// Round 5: the kernel works in two phases per workgroup.  (A) every queued query: the window scans that need no descent -- `still` queries with their
// seed bound, "window first" for queries whose previous winner is near; what the window settles is done, a `still` query it does not settle goes to
// queue 2.  Everything that needs the DESCENT is deferred into a list in LDS and (B) descended 256 at a time in dense lanes.  Before, a wavefront paid
// for window scan AND descent whenever one of its lanes needed the descent -- in passes 2-4, where 13-46 % of the lanes do, that was every wavefront.
constexpr uint32_t kDeferCap = 512;                              // (at most 255 left over + 256 new entries)
__global__ __launch_bounds__(256, PR_BOUND_WAVES) void nn_bound_kernel(IcpBatch b, SceneNNDev scene)
{
    __shared__ uint32_t wave_n[4], wave_d[4], wg_base, s_ndefer;
    __shared__ uint2 s_defer[kDeferCap];                         // {point | tried << 30 | settle << 31, bound bits}
    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];
    if (pm.state == kSkip) return;
    uint32_t *counts = b.nn_qcount + kQCountStride * pose;
    const uint32_t queued = counts[b.iter & 1u];
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[(b.iter + 1u) & 1u] = 0u;         // re-arm the counter the NEXT pass' search kernel will fill
    if (blockIdx.x * kBlockThreads >= queued) return;
    const float *cl = reinterpret_cast<const float *>(b.cloud + pm.start);
    uint32_t *win = b.nn_prev + pm.start;
    float *slk = b.nn_slack + pm.start;
    const uint2 *queue = b.nn_queue + pm.start;
    uint2 *queue2 = b.nn_queue2 + pm.start;
    uint32_t *q2_count = counts + 2u + (b.iter & 1u);
    const float accept = scene.max_dist_diff * scene.max_dist_diff;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t n_window = 0, n_pyramid = 0, n_cells = 0;
    if (threadIdx.x == 0) s_ndefer = 0u;
    __syncthreads();
    // the lanes with `pending` append (point, bound) to queue 2, in lane order (called by every lane of the workgroup)
    auto push_q2 = [&](bool pending, uint32_t j, float bst) {
        const unsigned long long m = __ballot(pending);
        if (lane == 0) wave_n[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) { const uint32_t t = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3]; wg_base = t ? atomicAdd(q2_count, t) : 0u; }
        __syncthreads();
        if (pending) {
            uint32_t slot = wg_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            for (uint32_t w2 = 0; w2 < wave; ++w2) slot += wave_n[w2];
            queue2[slot] = make_uint2(j, __float_as_uint(bst));
        }
        __syncthreads();
    };
    // (B) deferred entries [first, first + n), n <= 256, one per lane: descent through the representative points, the window if it has not been
    // tried with this query, then the winner or queue 2
    auto descend = [&](uint32_t first, uint32_t n) {
        bool pending = false;
        uint32_t j = 0; float bst = 0.0f;
        if (threadIdx.x < n) {
            const uint2 e = s_defer[first + threadIdx.x];
            j = e.x & 0x3fffffffu;
            const bool tried = (e.x & 0x40000000u) != 0u, settle = (e.x & 0x80000000u) != 0u;
            bst = __uint_as_float(e.y);
            const pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);
            grid_pyramid_bound(scene, q.x, q.y, q.z, bst, b.iter >= 1u); ++n_pyramid;
            uint32_t w = kNoPrev; float bsq = 0.0f, osq = 0.0f;
            // (a query whose largest window held nothing within its reach, or a tie, cannot be settled by the window after a descent either)
            const bool done = !tried && bst < accept && grid_search(scene, q.x, q.y, q.z, bst, w, &n_cells, &bsq, &osq, settle);
            if (done) { ++n_window; win[j] = w; slk[j] = margin_sqrt(osq) * 0.99999f; }
            else pending = true;
        }
        push_q2(pending, j, bst);
    };
    for (uint32_t i0 = blockIdx.x * kBlockThreads; i0 < queued; i0 += gridDim.x * kBlockThreads) {
        const uint32_t i = i0 + threadIdx.x;
        bool pending = false, defer = false, tried = false, settle = false;
        uint32_t j = 0; float bst = 0.0f;
        if (i < queued) {
            const uint2 e = queue[i];
            j = e.x;
            const bool still = (e.y & 0x80000000u) != 0u;        // the sign carries "no descent needed", the lowest bit "settle" (nn_search_kernel)
            settle = (e.y & 1u) != 0u;
            bst = __uint_as_float((e.y & 0x7fffffffu) | 1u);     // (the flag bit set: the bound rounded UP by at most an ulp)
            pending = true;
            if (!scene.grid && !still) {
                // No pixel grid (a bare ICP call has no camera) and the query is new or has moved (its previous winner, centimetres away
                // now, is a loose bound): the task walk is order-free, so with a loose bound it would open a good part of the tree before
                // a leaf near the query tightens it.  The classic descent does that first: follow the split planes to the query's own leaf (8 bytes
                // per level) and take the nearest of its points as the seed -- an existing point's distance, inflated (nn_seed_bound).
                const pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);
                uint32_t cur = 0u;
                for (int guard = 0; guard < 64; ++guard) {
                    const uint2 d = scene.desc[cur];
                    const uint32_t tag = d.y >> 30;
                    if (tag == 3u) {
                        float dmin = FLT_MAX;
                        for (uint32_t i = d.x; i < (d.y & 0x3fffffffu); ++i) {
                            const float4 p = scene.pts[i];
                            const float d2 = (q.x - p.x) * (q.x - p.x) + (q.y - p.y) * (q.y - p.y) + (q.z - p.z) * (q.z - p.z);
                            dmin = d2 < dmin ? d2 : dmin;
                        }
                        const float bb = dmin * 1.000001f + 1e-30f;
                        if (bb < bst) bst = bb;
                        break;
                    }
                    const float c = (tag == 0u) ? q.x : ((tag == 1u) ? q.y : q.z);
                    const uint32_t c1 = d.y & 0x3fffffffu;
                    cur = (c - __uint_as_float(d.x) < 0.0f) ? c1 : c1 + 1u;
                }
            }
            if (scene.grid) {
                const pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);
                uint32_t w = kNoPrev; float bsq = 0.0f, osq = 0.0f;
                bool done = false;
                if (still) done = bst < accept && grid_search(scene, q.x, q.y, q.z, bst, w, &n_cells, &bsq, &osq, settle);
                else if (bst <= (b.iter <= 1u ? PR_NN_WINFIRST_EARLY : PR_NN_WINFIRST)) {
                    // WINDOW FIRST (round 5): the previous winner is near, so the neighbour is probably within the largest window's reach -- scan it
                    // without a tight bound (grid_search `full`) and spare the descent.  A failure (nothing within the covered radius, or a tie)
                    // means the window cannot settle this query after a descent either: it is descended (a bound for the tree) and goes to queue 2.
                    tried = true;
                    done = grid_search(scene, q.x, q.y, q.z, bst, w, &n_cells, &bsq, &osq, settle, true);
                    if (!done && bsq > 0.0f) { const float bb = bsq * 1.000001f + 1e-30f; if (bb < bst) bst = bb; }
                }
                if (done) { pending = false; ++n_window; win[j] = w; slk[j] = margin_sqrt(osq) * 0.99999f; }
                else if (!still) { pending = false; defer = true; }
            }
        }
        // the deferred queries of this chunk join the list (lane order); thread 0 alone advances the count, between two barriers
        {
            const unsigned long long md = __ballot(defer);
            if (lane == 0) wave_d[wave] = (uint32_t)__popcll(md);
            __syncthreads();
            const uint32_t base = s_ndefer;
            if (defer) {
                uint32_t slot = base + (uint32_t)__popcll(md & ((1ull << lane) - 1ull));
                for (uint32_t w2 = 0; w2 < wave; ++w2) slot += wave_d[w2];
                s_defer[slot] = make_uint2(j | (tried ? 0x40000000u : 0u) | (settle ? 0x80000000u : 0u), __float_as_uint(bst));
            }
            __syncthreads();
            if (threadIdx.x == 0) s_ndefer = base + wave_d[0] + wave_d[1] + wave_d[2] + wave_d[3];
        }
        push_q2(pending, j, bst);                                // (its barriers also publish the list and its count)
        const uint32_t nd = s_ndefer;
        if (nd >= kBlockThreads) {                               // workgroup-uniform
            descend(nd - kBlockThreads, kBlockThreads);
            if (threadIdx.x == 0) s_ndefer = nd - kBlockThreads;
            __syncthreads();
        }
    }
    for (uint32_t nd = s_ndefer; nd > 0u; ) {                    // what is left: fewer than 256 (workgroup-uniform: nobody writes s_ndefer any more)
        const uint32_t n = nd < kBlockThreads ? nd : kBlockThreads;
        descend(nd - n, n);
        nd -= n;
    }
    if (scene.counters) {                                        // instrumented runs only (option "nn_count")
        const uint32_t v[8] = { 0u, n_window, 0u, n_pyramid, 0u, 0u, 0u, n_cells };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if ((threadIdx.x & 63u) == 0u && t) atomicAdd(&scene.counters[(size_t)b.iter * 8 + i], (unsigned long long)t);
        }
    }
}
// minimum of two floats that are known not to be NaN: one v_min_f32 (fminf's IEEE minNum semantics cost a canonicalising v_max per operand)
__device__ __forceinline__ float min_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// exclusive prefix sum over the 64 lanes of a wavefront (and the total): in-row Hillis-Steele with DPP row shifts, row totals by readlane
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);      // row_shr:1 (lanes shifted in from outside the row read 0)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);      // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);      // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);      // row_shr:8
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 31),
                   t2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 47), t3 = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    x += (row > 0u ? t0 : 0u) + (row > 1u ? t1 : 0u) + (row > 2u ? t2 : 0u);
    total = t0 + t1 + t2 + t3;
    return x - v;
}
// two unsigned 16-bit lanes per register: saturating difference and maximum (v_pk_sub_u16 clamp, v_pk_max_u16)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
// Lower bounds of the squared distance from a query to the TWO boxes of a pair, in squared units of the wide records' frame.  The query
// enters as an interval [qd, qu] of whole units that holds its coordinate (both halves of a word carry the same value); per axis the
// distance is max(lo - qu, qd - hi, 0) -- saturating 16-bit differences, both boxes per instruction -- and the three are squared and
// added in float (the sum of three squares of 16-bit numbers does not fit 32 bits).  Every rounding is covered by the margins of the
// caller: qd + 1 < q < qu - 1, and the bound the result is compared with is inflated by 1e-6 and one squared unit.
// (VALU instructions per box: 10 against the 31 of the per-axis float dequantisation it replaces; it also frees the registers that let
// the walk run six wavefronts per SIMD.)
__device__ __forceinline__ void wide_pair_lb(uint32_t lox, uint32_t loy, uint32_t loz, uint32_t hix, uint32_t hiy, uint32_t hiz,
                                             uint32_t qdx, uint32_t qdy, uint32_t qdz, uint32_t qux, uint32_t quy, uint32_t quz, float &lb_a, float &lb_b)
{
    const uint32_t dx = pk_max_u16(pk_sub_sat_u16(lox, qux), pk_sub_sat_u16(qdx, hix));
    const uint32_t dy = pk_max_u16(pk_sub_sat_u16(loy, quy), pk_sub_sat_u16(qdy, hiy));
    const uint32_t dz = pk_max_u16(pk_sub_sat_u16(loz, quz), pk_sub_sat_u16(qdz, hiz));
    const float ax = (float)(dx & 0xffffu), ay = (float)(dy & 0xffffu), az = (float)(dz & 0xffffu);
    const float bx = (float)(dx >> 16), by = (float)(dy >> 16), bz = (float)(dz >> 16);
    lb_a = __builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax));
    lb_b = __builtin_fmaf(bz, bz, __builtin_fmaf(by, by, bx * bx));
}
// Queue 2 -> the task walk.  Wavefronts are independent (no workgroup barrier): each takes 64 queries at a time.  kLanes lanes share a task:
// 8 / kLanes boxes of a node, or 8 / kLanes points of a leaf per round, per lane; a step pops 64 / kLanes tasks of one kind.
// Box bounds travel through the queues in SQUARED UNITS of the wide records' frame (wide_pair_lb); a query's bound is converted when a task
// is popped (inflated by 1e-6, plus one unit), the runner-up bound when a step's skipped boxes are folded into `second` (deflated).
template <int kLanes, bool kCount>
__global__ __launch_bounds__(256, PR_WIDE_WAVES) void nn_tree_wide_kernel(IcpBatch b, SceneNNDev scene, uint32_t qbatch)
{
    static_assert(kLanes == 2, "the paired record layout (nn_wide_build_kernel, stage D) is made for two lanes per task");
    constexpr uint32_t kPer = 8u / kLanes, kTasks = 64u / kLanes;
    constexpr uint32_t kLeafPer = PR_WIDE_LEAF_PER;               // points of a leaf per lane and round
    __shared__ uint2 s_taskq[4][kTaskQCap + kTaskLCap];                          // per wavefront: node-task queue, leaf-task queue behind it; {reference, bound bits (low 6 bits cleared: rounded DOWN) | query slot}
    // (Round 5, VERDICT r04 item 1b: the per-query records as six arrays of words instead of 16-byte records -- a step's 32 slots then hit 32 different
    // banks -- left SQ_LDS_BANK_CONFLICT where it was, 5.13e7 against 5.16e7 cycles in pass 0, with a quarter more LDS instructions and 1 % less
    // throughput: the conflicts are not in these gathers but in the stores to the task queues and the atomics on best / second / tied.)
    __shared__ float4 s_q[4][64];                                                // query point | bound (float bits, lowered with atomicMin)
    __shared__ uint2 s_qq[4][64];                                                // the query in whole units of the wide records' frame: ux | uy << 16, uz | uz << 16
    __shared__ uint32_t s_second[4][64], s_tied[4][64], s_ovf[4][64], s_root[4][64];
    __shared__ unsigned long long s_best[4][64];
    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];
    if (pm.state == kSkip) return;
    uint32_t *counts = b.nn_qcount + kQCountStride * pose + 2u;
    const uint32_t queued = counts[b.iter & 1u];
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[(b.iter + 1u) & 1u] = 0u;         // re-arm the counter the NEXT pass' bound kernel will fill
    if (blockIdx.x * 4u * qbatch >= queued) return;               // qbatch = queries a wavefront takes at a time (64, or 16 when the whole launch has few)
    const float *cl = reinterpret_cast<const float *>(b.cloud + pm.start);
    uint32_t *win = b.nn_prev + pm.start;
    float *slk = b.nn_slack + pm.start;
    const uint2 *todo = b.nn_queue2 + pm.start;
    // (the wavefront's index is uniform, but only readfirstlane tells the compiler: with it the queue fill levels, the batch loop and every
    // branch on them live in scalar registers)
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c = lane % kLanes, grp = lane / kLanes;
    uint2 *nodeq = s_taskq[wave], *leafq = nodeq + kTaskQCap;
    float4 *qs = s_q[wave];
    uint2 *qq = s_qq[wave];
    uint32_t *second = s_second[wave], *tied = s_tied[wave], *ovf = s_ovf[wave], *root = s_root[wave];
    unsigned long long *best = s_best[wave];
    // the frame of the wide records: units per metre, the query's offset in units (minus the sixteenth of a unit that covers the rounding of
    // this very conversion), and the factors between squared metres and squared units -- each rounded to the safe side
    const float w_inv = 1.0f / scene.wscale;
    const float w_to_units2 = w_inv * w_inv * 1.000001f, w_to_m2 = scene.wscale * scene.wscale * 0.999999f;
    uint32_t n_tree = 0, n_nodes = 0, n_leaves = 0, n_leaf_points = 0;   // work counters of instrumented runs (kCount: option "nn_count"); the product instantiation carries none
    for (uint32_t base = (blockIdx.x * 4u + wave) * qbatch; base < queued; base += gridDim.x * 4u * qbatch) {
        const bool have_q = lane < qbatch && base + lane < queued;
        uint32_t nN = 0, nL = 0;                                    // fill levels of the two queues (wave-uniform)
        const uint2 mine = have_q ? todo[base + lane] : make_uint2(0u, 0u);    // this lane's query: (point, bound bits)
        uint32_t bound_now = mine.y;                                // the bound this lane's query was (last) started from
        {
            const pr_vec3 q = have_q ? ld_off<pr_vec3>(cl, mine.x * 12u) : pr_vec3{ 0.0f, 0.0f, 0.0f };
            qs[lane] = make_float4(q.x, q.y, q.z, __uint_as_float(mine.y));
            // The query in whole units of the frame, once per query: an interval [u, u + 3] that holds its coordinate with more than a unit to
            // spare on either side (the difference to the frame's origin is formed first -- exact to half an ulp of a number below the frame's
            // edge -- then scaled; a unit is at least an ulp of the largest coordinate, nn_frame_kernel refuses the frame otherwise: that is also
            // the slack a box's stored corners have against their real-number positions).  Negative and NaN values become 0.
            // (clamped in float BEFORE the conversion -- a float -> unsigned cast of a negative, NaN or > 2^32 value is undefined in C++; v_max_f32 with 0 returns 0 for NaN)
            const uint32_t ux = (uint32_t)__builtin_fminf(__builtin_fmaxf(__builtin_fmaf(q.x - scene.wmin[0], w_inv, -1.0625f), 0.0f), 65532.0f),
                           uy = (uint32_t)__builtin_fminf(__builtin_fmaxf(__builtin_fmaf(q.y - scene.wmin[1], w_inv, -1.0625f), 0.0f), 65532.0f),
                           uz = (uint32_t)__builtin_fminf(__builtin_fmaxf(__builtin_fmaf(q.z - scene.wmin[2], w_inv, -1.0625f), 0.0f), 65532.0f);
            qq[lane] = make_uint2(ux | (uy << 16), uz * 0x10001u);
            best[lane] = ((unsigned long long)mine.y << 32) | kNoIdx;
            second[lane] = 0x7f7fffffu; tied[lane] = 0xffffffffu; ovf[lane] = 0u; root[lane] = lane;
            if (kCount && have_q) ++n_tree;
        }
        // The walks of the 64 queries are started kTasks at a time, whenever the node queue runs low: all 64 root tasks at once would
        // spread three levels of every walk over the queues before the first leaf is reached (depth-first order keeps them short).
        // A query that lost a task to a full queue is walked again in a second round, alone with the other such queries and from the
        // minimum it did find (a handful of queries cannot fill the queues); only if that fails too does it go to the ordered walk.
        uint32_t n_q = (queued - base < qbatch) ? (queued - base) : qbatch;
        for (int round = 0; round < 2 && n_q; ++round) {
        uint32_t started = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while (nN | nL | (n_q - started)) {
            if (nN < kTasks && started < n_q) {                     // root tasks: wide node 0, bound 0.0f
                const uint32_t add = (n_q - started < kTasks) ? (n_q - started) : kTasks;
                if (lane < add) nodeq[nN + lane] = make_uint2(0u, root[started + lane]);
                nN += add; started += add;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const bool leaf_step = (nL >= kTasks) || (nN == 0u);
            const uint32_t avail = leaf_step ? nL : nN, k = avail < kTasks ? avail : kTasks;
            const bool active = grp < k;
            const uint2 e = active ? (leaf_step ? leafq : nodeq)[avail - 1u - grp] : make_uint2(0u, 0u);
            if (leaf_step) nL -= k; else nN -= k;
            const uint32_t q = e.y & 63u, ref = e.x;
            const float lb_in = __uint_as_float(e.y & ~63u);         // squared units, rounded down
            const float4 qp = qs[q];
            const float sx = qp.x, sy = qp.y, sz = qp.z, bnd = qp.w;
            uint32_t *bound_q = reinterpret_cast<uint32_t *>(&qs[q]) + 3;
            const float bnd_u = __builtin_fmaf(bnd, w_to_units2, 1.0f);   // the query's bound in squared units, rounded up generously
            const bool alive = active && lb_in <= bnd_u;
            float sec_u = (active && !alive) ? lb_in : FLT_MAX;      // what this lane rules out: boxes (squared units) ...
            float sec_l = FLT_MAX;                                   // ... and points (squared metres)
            if (!leaf_step) {
                // ---------------- node tasks: lane c of a group tests the four slots of half c (two pairs, see nn_wide_build_kernel stage D)
                // The record is loaded by every lane (a lane without a live task reads wide node 0 and masks the outcome with `alive`): no
                // zero-initialised copy of the record, no branch around the loads.
                uint4 r[4];
                {
                    const uint4 *rec = scene.wide + (size_t)(alive ? ref : 0u) * 8u + 4u * c;
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) r[i] = rec[i];
                    if (kCount && alive && c == 0u) ++n_nodes;
                }
                // the query's interval [qd, qd + 3] per axis, both halves of a word carrying the same value (two byte permutes of the stored pair)
                const uint2 qw = qq[q];
                const uint32_t qdx = __builtin_amdgcn_perm(qw.x, qw.x, 0x01000100u), qdy = __builtin_amdgcn_perm(qw.x, qw.x, 0x03020302u), qdz = qw.y;
                float lb[kPer];
                wide_pair_lb(r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, qdx, qdy, qdz, qdx + 0x30003u, qdy + 0x30003u, qdz + 0x30003u, lb[0], lb[1]);
                wide_pair_lb(r[1].z, r[1].w, r[2].x, r[2].y, r[2].z, r[2].w, qdx, qdy, qdz, qdx + 0x30003u, qdy + 0x30003u, qdz + 0x30003u, lb[2], lb[3]);
                const uint32_t refs[kPer] = { r[3].x, r[3].y, r[3].z, r[3].w };
                // Round 5: counts packed in one word (internal | leaf << 16), ONE prefix scan, and -- the common case, decided per step on the
                // totals -- a straight run of stores without a capacity test per slot: the wavefront's two queues are ONE array (node tasks
                // first, leaf tasks behind them), so a child's place is a single index whatever its kind.  (Tried first: numbering the
                // children slot by slot with ballots + v_mbcnt -- 151 VALU per step instead of 180, but 150 SALU instead of 54; the scalar
                // unit is shared by the four SIMDs of a CU and the walk already issues 0.45 scalar instructions per vector one.)
                uint32_t cnt = 0u;
                bool keep[kPer], leaf[kPer];
#pragma unroll
                for (uint32_t i = 0; i < kPer; ++i) {
                    const bool v = alive && refs[i] != kWideEmpty;
                    keep[i] = v && lb[i] <= bnd_u;
                    leaf[i] = (int32_t)refs[i] < 0;                    // kWideLeaf = the sign bit (an empty slot reads as a leaf: never kept)
                    sec_u = min_f32(sec_u, (v && !keep[i]) ? lb[i] : FLT_MAX);      // (a select, not a branch: the asm min cannot be if-converted)
                    cnt += keep[i] ? (leaf[i] ? 0x10000u : 1u) : 0u;
                }
                uint32_t tot = 0;
                const uint32_t ex = wave_excl_scan(cnt, tot);
                const uint32_t totI = tot & 0xffffu, totL = tot >> 16;
                if (nN + totI <= kTaskQCap && nL + totL <= kTaskLCap) {
                    uint32_t pI = nN + (ex & 0xffffu), pL = kTaskQCap + nL + (ex >> 16);
#pragma unroll
                    for (uint32_t i = 0; i < kPer; ++i) {
                        if (keep[i]) nodeq[leaf[i] ? pL : pI] = make_uint2(refs[i], (__float_as_uint(lb[i]) & ~63u) | q);
                        pI += (keep[i] && !leaf[i]) ? 1u : 0u; pL += (keep[i] && leaf[i]) ? 1u : 0u;
                    }
                    nN += totI; nL += totL;
                } else {
                    // a queue would overflow in this step: every slot checks its place; a child that does not fit marks its query (walked again)
                    uint32_t pI = nN + (ex & 0xffffu), pL = nL + (ex >> 16);
                    bool lost = false;
#pragma unroll 1
                    for (uint32_t i = 0; i < kPer; ++i) {
                        const uint32_t pos = leaf[i] ? pL : pI, cap = leaf[i] ? kTaskLCap : kTaskQCap;
                        if (keep[i]) { if (pos < cap) (leaf[i] ? leafq : nodeq)[pos] = make_uint2(refs[i], (__float_as_uint(lb[i]) & ~63u) | q); else lost = true; }
                        pL += (keep[i] && leaf[i]) ? 1u : 0u; pI += (keep[i] && !leaf[i]) ? 1u : 0u;
                    }
                    if (lost) ovf[q] = 1u;
                    nN += totI; if (nN > kTaskQCap) nN = kTaskQCap;
                    nL += totL; if (nL > kTaskLCap) nL = kTaskLCap;
                }
            } else {
                // ---------------- leaf tasks: kLeafPer points per lane and round, all loaded before any is looked at.  A lane first settles its own
                // points in registers -- minimum, which one, and the runner-up -- and only the minimum goes to the query's record in LDS: ONE
                // conditional block per round.  (Per point, as first built, "rare: a point that may be the minimum" was rare per lane but not per
                // wavefront: with 256 point tests per step some lane took the branch -- an LDS atomic round trip and twenty instructions -- for
                // nearly every point slot; and with four points per lane a leaf of nine or ten, one in three, cost every step a second round.)
                const uint32_t first = ref & kWideFirstMask, cnt = alive ? ((ref >> 27) & 15u) : 0u;
                if (kCount && alive && c == 0u) { ++n_leaves; n_leaf_points += cnt; }
                for (uint32_t kb = 0; kb < cnt; kb += kLanes * kLeafPer) {
                    const uint32_t ka = kb + kLeafPer * c;
                    // kLeafPer consecutive records from ONE address (immediate offsets): a short leaf reads on into its neighbour's points, or into
                    // the 16 records of padding behind the last leaf -- masked by `ka + h < cnt` below
                    const float4 *lp = scene.pts + (first + ka);
                    float4 pt[kLeafPer];
#pragma unroll
                    for (uint32_t h = 0; h < kLeafPer; ++h) pt[h] = lp[h];
                    float m = FLT_MAX, s2 = FLT_MAX;                   // the lane's smallest distance and the smallest of its other points
                    uint32_t hm = 0u;
                    const float2v sxy{ sx, sy };
#pragma unroll
                    for (uint32_t h = 0; h < kLeafPer; ++h) {
                        // pcd_scene.h:88-91: (dx*dx + dy*dy) + dz*dz -- x and y side by side in packed instructions (each element rounded as in the scalar form)
                        const float2v dxy = sxy - float2v{ pt[h].x, pt[h].y };
                        const float2v qxy = dxy * dxy;
                        const float dz = sz - pt[h].z;
                        const float e2 = (qxy.x + qxy.y) + dz * dz;
                        const float d2 = (ka + h < cnt) ? e2 : FLT_MAX;
                        const bool lt = d2 < m;                         // strict: of equal points the first keeps the slot, the other shows up in s2
                        s2 = min_f32(s2, lt ? m : d2);
                        hm = lt ? h : hm;
                        m = lt ? d2 : m;
                    }
                    sec_l = min_f32(sec_l, s2);                        // none of them is this query's answer (an equal one: recorded as a tie below)
                    if (m <= bnd) {                                    // the lane's minimum may be the query's (bnd is finite: m is a real point's distance)
                        const uint32_t idx = first + ka + hm, db = __float_as_uint(m);
                        const unsigned long long key = ((unsigned long long)db << 32) | idx;
                        const unsigned long long old = atomicMin(&best[q], key);
                        const uint32_t old_d = (uint32_t)(old >> 32), old_i = (uint32_t)old;
                        if (s2 == m || (old_d == db && old_i != idx && old_i != kNoIdx)) atomicMin(&tied[q], db);
                        if (key < old) { if (old_i != kNoIdx) sec_l = min_f32(sec_l, __uint_as_float(old_d)); if (m < bnd) atomicMin(bound_q, db); }
                        else sec_l = min_f32(sec_l, m);
                    } else sec_l = min_f32(sec_l, m);
                }
            }
            nN = (uint32_t)__builtin_amdgcn_readfirstlane((int)nN); nL = (uint32_t)__builtin_amdgcn_readfirstlane((int)nL);     // wave-uniform by construction
            sec_l = min_f32(sec_l, sec_u < FLT_MAX ? sec_u * w_to_m2 : FLT_MAX);   // boxes: squared units -> squared metres, rounded down
            float sec_g = sec_l;
            if (kLanes >= 2) sec_g = min_f32(sec_l, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sec_l), 0xB1, 0xf, 0xf, true)));          // quad_perm [1,0,3,2]
            if (kLanes == 4) sec_g = min_f32(sec_g, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sec_g), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
            if (c == 0u && sec_g < FLT_MAX) atomicMin(&second[q], __float_as_uint(sec_g));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
            n_q = 0u;
            if (round == 0) {                                       // queries that lost a task: reset, tighten, list for the second round
                const bool lost = have_q && ovf[lane] != 0u;
                const unsigned long long ml = __ballot(lost);
                if (ml) {
                    if (lost) {
                        const unsigned long long key = best[lane];
                        const uint32_t m_bits = (uint32_t)(key >> 32), idx = (uint32_t)key;
                        uint32_t bb = mine.y;                         // (positive float bits order like the floats)
                        if (idx != kNoIdx && m_bits < bb) { const uint32_t t = __float_as_uint(__uint_as_float(m_bits) * 1.000001f + 1e-30f); if (t < bb) bb = t; }   // = nn_seed_bound
                        reinterpret_cast<uint32_t *>(&qs[lane])[3] = bb;
                        best[lane] = ((unsigned long long)bb << 32) | kNoIdx;
                        second[lane] = 0x7f7fffffu; tied[lane] = 0xffffffffu; ovf[lane] = 0u;
                        root[__popcll(ml & ((1ull << lane) - 1ull))] = lane;
                        bound_now = bb;
                    }
                    n_q = (uint32_t)__builtin_amdgcn_readfirstlane((int)__popcll(ml));
                }
            }
        }
        // all 64 queries of this wavefront are finished: one lane per query delivers
        if (have_q) {
            const unsigned long long key = best[lane];
            const uint32_t m_bits = (uint32_t)(key >> 32), idx = (uint32_t)key, b0 = bound_now, j = mine.x;
            const bool found = idx != kNoIdx && m_bits < b0;
            if (ovf[lane] != 0u || (found && tied[lane] == m_bits)) {
                // a tie or a dropped task: the ordered walk, from the minimum found (= nn_seed_bound: an existing point's distance)
                const float bnd = found ? fminf(__uint_as_float(b0), __uint_as_float(m_bits) * 1.000001f + 1e-30f) : __uint_as_float(b0);
                const float4 qp = qs[lane];
                win[j] = query_nn_bounded(scene, qp.x, qp.y, qp.z, bnd);
                slk[j] = 0.0f;                                      // an ordered walk does not report its runner-up: no shortcut next pass
            } else if (found) { win[j] = idx; slk[j] = sqrtf(__uint_as_float(second[lane])) * 0.99999f; }
            else { win[j] = kNoPrev; slk[j] = 0.0f; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (kCount && scene.counters) {
        uint32_t n_pyr = 0u;
        const uint32_t v[8] = { 0u, 0u, n_tree, n_pyr, n_nodes, n_leaves, n_leaf_points, 0u };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if ((threadIdx.x & 63u) == 0u && t) atomicAdd(&scene.counters[(size_t)b.iter * 8 + i], (unsigned long long)t);
        }
    }
}

// scene points -> grid cells.  cell_idx starts at -1; a second claim on a cell, or a point outside the image, clears info[0].
__global__ __launch_bounds__(256) void nn_grid_claim_kernel(const pr_vec3 *__restrict__ pcd, uint32_t n_points, SceneNNDev g, int32_t *__restrict__ cell_idx,
                                                            uint32_t *__restrict__ info)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_points) return;
    const pr_vec3 p = pcd[i];
    float u, v;
    grid_project(g, p.x, p.y, p.z, u, v);
    if (!(p.z > 0 && u >= 0.0f && u < (float)g.gw && v >= 0.0f && v < (float)g.gh)) { info[0] = 0u; return; }
    const int cell = (int)floorf(v) * (int)g.gw + (int)floorf(u);
    if (atomicCAS(&cell_idx[cell], -1, (int)i) != -1) info[0] = 0u;
}
// one coarser level: cell (bx, by) takes the occupied child cell nearest its centre (children = 4 x 4 cells of the finer level)
__global__ __launch_bounds__(256) void nn_grid_coarsen_kernel(const float4 *__restrict__ fine, int fw, int fh, float4 *__restrict__ coarse, int cw, int ch)
{
    const int c = (int)(blockIdx.x * 256 + threadIdx.x);
    if (c >= cw * ch) return;
    const int bx = c % cw, by = c / cw;
    float4 pick = make_float4(1e30f, 1e30f, 1e30f, __int_as_float(-1));
    int pick_d = 1 << 30;
    for (int dy = 0; dy < 4; ++dy)
        for (int dx = 0; dx < 4; ++dx) {
            const int x = bx * 4 + dx, y = by * 4 + dy;
            if (x >= fw || y >= fh) continue;
            const float4 v = fine[(size_t)y * fw + x];
            const int dist = (2 * dx - 3) * (2 * dx - 3) + (2 * dy - 3) * (2 * dy - 3);      // squared distance to the block centre, in half cells
            if (__float_as_int(v.w) >= 0 && dist < pick_d) { pick = v; pick_d = dist; }
        }
    coarse[c] = pick;
}
__global__ __launch_bounds__(256) void nn_grid_fill_kernel(const pr_vec3 *__restrict__ pcd, const int32_t *__restrict__ cell_idx, uint32_t n_cells,
                                                           float4 *__restrict__ grid)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    const int i = cell_idx[c];
    grid[c] = (i >= 0) ? make_float4(pcd[i].x, pcd[i].y, pcd[i].z, __int_as_float(i)) : make_float4(1e30f, 1e30f, 1e30f, __int_as_float(-1));
}

hipError_t launch_nn_search(const IcpBatch &b, const SceneNNDev &sc, uint32_t n_poses, uint32_t max_points, uint32_t run, hipStream_t s, hipEvent_t *marks)
{
    if (n_poses == 0 || max_points == 0) return hipSuccess;
    if (!sc.rec32 || (sc.stack_depth != 16 && sc.stack_depth != 24) || !b.nn_prev || !b.nn_slack || !b.nn_queue || !b.nn_queue2 || !b.nn_qcount) return hipErrorInvalidValue;
    if (max_points >= (1u << 30)) return hipErrorInvalidValue;        // nn_bound_kernel keeps two flags in the top bits of a point's index (a cloud of 2^30 points is 12 GiB)
    if (run == 0) run = 1;
    if (run > 8) run = 8;
    const uint32_t per_block = kBlockThreads * run;
    const uint32_t gx = (max_points + per_block - 1) / per_block;
    // workgroups per hypothesis walking its queue: 8 when there are hundreds of hypotheses, more for a handful (a single cloud -- the
    // reference's own ICP() call -- would otherwise run its tree searches on 8 CUs of 256)
    uint32_t want = (uint32_t)PR_TREE_GX;
    if (n_poses * want < 1024u) want = (1024u + n_poses - 1u) / n_poses;
    const uint32_t tree_gx = gx < want ? gx : want;
    for (uint32_t p0 = 0; p0 < n_poses; p0 += 32768) {
        const uint32_t np = (n_poses - p0 < 32768) ? (n_poses - p0) : 32768;
        IcpBatch bb = b;
        bb.meta += p0; bb.nn_qcount += kQCountStride * p0;
        hipLaunchKernelGGL(nn_search_kernel, dim3(gx, np), dim3(kBlockThreads), 0, s, bb, sc, run);
        if (marks) (void)hipEventRecord(marks[0], s);
        if (!sc.wide && marks) (void)hipEventRecord(marks[1], s);
        if (sc.wide) {
            hipLaunchKernelGGL(nn_bound_kernel, dim3(tree_gx, np), dim3(kBlockThreads), 0, s, bb, sc);
            if (marks) (void)hipEventRecord(marks[1], s);
            // a wavefront of the task walk takes 64 queries at a time -- unless the whole launch has too few to fill the chip that way (a single
            // cloud: 26 k queries = 103 workgroups of 4 x 64): then 16 at a time in four times as many workgroups
            uint32_t qbatch = 64u, walk_gx = tree_gx;
            if ((size_t)np * max_points < (size_t)64 * 4 * 1024) { qbatch = 16u; walk_gx = (max_points + 4u * qbatch - 1u) / (4u * qbatch); if (walk_gx * np > 4096u) walk_gx = (4096u + np - 1u) / np; if (walk_gx < tree_gx) walk_gx = tree_gx; }
            if (sc.counters) hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_tree_wide_kernel<PR_WIDE_LANES, true>), dim3(walk_gx, np), dim3(kBlockThreads), 0, s, bb, sc, qbatch);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_tree_wide_kernel<PR_WIDE_LANES, false>), dim3(walk_gx, np), dim3(kBlockThreads), 0, s, bb, sc, qbatch);
        }
        else if (sc.stack_depth == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_tree_kernel<16 + 0x100>), dim3(tree_gx, np), dim3(kBlockThreads), (size_t)16 * kBlockThreads * 8, s, bb, sc);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_tree_kernel<24 + 0x100>), dim3(tree_gx, np), dim3(kBlockThreads), (size_t)24 * kBlockThreads * 8, s, bb, sc);
        if (marks) (void)hipEventRecord(marks[2], s);
    }
    return hipGetLastError();
}

hipError_t launch_build_nn_grid(const pr_vec3 *pcd, uint32_t n_points, uint32_t gw, uint32_t gh, float fx, float fy, float cx, float cy,
                                int32_t *cell_idx, float4 *grid, uint32_t *info, hipStream_t s)
{
    const uint32_t cells = gw * gh;
    if (cells == 0 || n_points == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(cell_idx, 0xff, sizeof(int32_t) * cells, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(info, 1, sizeof(uint32_t), s);                 // non-zero: usable so far
    if (e != hipSuccess) return e;
    SceneNNDev g{};
    g.gw = gw; g.gh = gh; g.gfx = fx; g.gfy = fy; g.gcx = cx; g.gcy = cy;
    hipLaunchKernelGGL(nn_grid_claim_kernel, dim3((n_points + 255) / 256), dim3(256), 0, s, pcd, n_points, g, cell_idx, info);
    hipLaunchKernelGGL(nn_grid_fill_kernel, dim3((cells + 255) / 256), dim3(256), 0, s, pcd, cell_idx, cells, grid);
    int fw = (int)gw, fh = (int)gh;
    float4 *fine = grid;
    for (int level = 0; level < 3; ++level) {
        const int cw = (fw + 3) / 4, ch = (fh + 3) / 4;
        float4 *coarse = fine + (size_t)fw * fh;
        hipLaunchKernelGGL(nn_grid_coarsen_kernel, dim3((uint32_t)((cw * ch + 255) / 256)), dim3(256), 0, s, fine, fw, fh, coarse, cw, ch);
        fine = coarse; fw = cw; fh = ch;
    }
    return hipGetLastError();
}
size_t nn_grid_cells(uint32_t gw, uint32_t gh)
{
    size_t total = (size_t)gw * gh, w = gw, h = gh;
    for (int level = 0; level < 3; ++level) { w = (w + 3) / 4; h = (h + 3) / 4; total += w * h; }
    return total;
}

}  // namespace prk
