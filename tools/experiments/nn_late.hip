// NOT PART OF THE LIBRARY (not in pose_refine_amd/build.py): round 6's fused late pass, kept as the record of an experiment that lost.
// Measured on MI355X (profiles/r06/README.md, "late passes as one kernel"): bit-identical to the four-kernel pass (tests + oracle), but 119-180 us per
// late pass against ~120 us for search + bound + walk + winners pass; even with its ordered walks skipped (timing only) 90-100 us: phases A + C move the
// same 64 B per point through L2 as the search kernel and the winners pass together and do it at lower occupancy.  configs[2] 43.2 k against 53.2 k poses/s.
// nn_late.hip -- a LATE correspondence pass against a kd-tree scene as ONE kernel: pending update + keep-the-winner test + (for what the test does not
// settle) pixel window / descent / ordered walk + the canonical 29-term reduce.  Replaces the four launches of a pass (nn_search_kernel -> nn_bound_kernel ->
// nn_tree_wide_kernel -> icp_pass_kernel<SceneNNWinners>) once the hypotheses have nearly stopped moving.
// Scene_nn::query: pcd_scene.h:60-136; thrust__pcd2Ab: icp.h:128-209; transform_pcd_cuda: icp.cu:142-153.
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
//
// Why one kernel.  From pass 6 on of the 21 (profiles/r05/nn_work_counters.md) 75-99.9 % of the queries keep their previous winner without a search (the
// carried runner-up bound `slack`), 24 -> 0.1 % are settled by the pixel window and < 1 % need the tree, yet the pass still cost four dependent launches:
// search 40 us + bound 12-25 us + walk 28-40 us (fifteen dependent steps for a handful of queries) + winners pass 36 us = ~120 us, 1.8 ms of a 5.9 ms
// group-step.  Here a workgroup does for its own 3072-point block what those kernels did for the batch:
//   (A) per 1024-point step, the canonical lane mapping of the pass (lane t: points t, t + 256, t + 512, t + 768): apply the pending update and write it
//       back, gather the previous winner (point + normal, all four of a lane in flight), run nn_search_kernel's keep test -- same expressions -- and
//       ACCUMULATE the kept points straight away, in the lane's canonical order, for as long as none of the lane's points has failed the test;
//       a point that fails goes to a list in LDS (its index and its seed bound with nn_search_kernel's two flags), and the lane stops accumulating;
//   (B) the list in dense lanes: nn_bound_kernel's logic per query -- window with the seed bound for a point that has stopped, the largest window first
//       for one whose winner is near, the descent through the representative points otherwise -- and, for what the window cannot settle (a tie, an
//       empty window), the ordered stackless walk from the bound (query_nn_bounded: the reference's own visiting order, so ties resolve as in
//       pcd_scene.h:79-119); winners and runner-up bounds go to memory exactly as the four-kernel path leaves them;
//   (C) the lanes that stopped in (A) resume at their first unsettled point: the winners pass' step (re-read point, winner, gather) from there on.
// Every winner is the reference's answer whichever way it was found (a unique strict minimum, or the ordered walk), and every lane adds its points in
// index order into the same tree (vb_reduce): sums bit-identical to the four-kernel path and to oracle/pose_oracle.c:sum29_canonical.
#include "pr_launch.h"
#include "icp_accumulate.h"
#include "icp_solve_device.h"

namespace prk {

// one 1024-point step of the winners pass for lanes that resume in phase C: points [i_from, cnt) of the lane
template <bool kScoreOnly>
__device__ __forceinline__ void late_resume_step(Acc29 &acc, const float *cl, const uint32_t *win, const SceneNNDev &scene, uint32_t j0, uint32_t cnt, uint32_t i_from)
{
    float p[12];
    uint32_t w[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const bool on = i >= i_from && i < cnt;
        const uint32_t j = on ? j0 + i * kBlockThreads : j0;
        const pr_vec3 v = ld_off<pr_vec3>(cl, j * 12u);
        p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z;
        w[i] = on ? win[j] : kNoPrev;
    }
    float4 d[4]; float nx[4], ny[4], nz[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t at = (w[i] != kNoPrev) ? w[i] : 0u;
        d[i] = scene.pts[at];
        const float *nn = reinterpret_cast<const float *>(scene.normal + at);
        nx[i] = nn[0]; ny[i] = nn[1]; nz[i] = nn[2];
    }
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        if (w[i] != kNoPrev) {
            Corr c; c.dx = d[i].x; c.dy = d[i].y; c.dz = d[i].z; c.nx = nx[i]; c.ny = ny[i]; c.nz = nz[i];
            if constexpr (kScoreOnly) accumulate_score(acc, p[3 * i], p[3 * i + 1], p[3 * i + 2], c); else accumulate(acc, p[3 * i], p[3 * i + 1], p[3 * i + 2], c);
        }
    }
}

// phases A-C for one virtual block; the lane's 29 sums come back in acc_out (canonical order)
template <bool kScoreOnly>
__device__ __forceinline__ void late_block(float (&acc_out)[29], float *cl, uint32_t *win, float *slk, uint32_t n, uint32_t first, uint32_t steps, bool xf,
                                           const float (&M)[12], const SceneNNDev &scene, uint32_t iter, uint2 *s_list, uint32_t *s_count)
{
    Acc29 acc;
    acc_clear(acc);
    const float accept = scene.max_dist_diff * scene.max_dist_diff;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t stop = 0xffffffffu;                                 // step << 2 | point of the lane's first unsettled point (0xffffffff: none so far)
    if (threadIdx.x == 0) *s_count = 0u;
    __syncthreads();
    // ---------------- (A)
    for (uint32_t s = 0; s < steps; ++s) {
        const uint32_t j0 = first + s * kPointsPerStep + threadIdx.x;
        if (j0 >= n) break;                                      // (this lane; lanes of a wavefront leave together except in the cloud's last step)
        const uint32_t left = (n - j0 + kBlockThreads - 1u) / kBlockThreads, cnt = left < kPointsPerLane ? left : kPointsPerLane;
        const uint32_t last = n - 1u;
        float p[12], step_sq[4];
        uint32_t w[4]; float sl[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            const uint32_t j = j0 + i * kBlockThreads, jc = j < last ? j : last;
            const pr_vec3 v = ld_off<pr_vec3>(cl, jc * 12u);
            p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z;
            w[i] = (xf && i < cnt) ? win[jc] : kNoPrev;             // (no pending update = no previous pass: nothing to keep)
            sl[i] = slk[jc];
        }
        float4 d[4]; float nx[4], ny[4], nz[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            const uint32_t at = (w[i] != kNoPrev) ? w[i] : 0u;
            d[i] = scene.pts[at];
            const float *nn = reinterpret_cast<const float *>(scene.normal + at);
            nx[i] = nn[0]; ny[i] = nn[1]; nz[i] = nn[2];
        }
        if (xf) {                                                // icp.cu:142-153 transform_pcd_cuda: ((m0*x + m1*y) + m2*z) + m3, as nn_search_kernel applies it
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
                const float tx = M[0] * x + M[1] * y + M[2]  * z + M[3];
                const float ty = M[4] * x + M[5] * y + M[6]  * z + M[7];
                const float tz = M[8] * x + M[9] * y + M[10] * z + M[11];
                step_sq[i] = (tx - x) * (tx - x) + (ty - y) * (ty - y) + (tz - z) * (tz - z);
                p[3 * i] = tx; p[3 * i + 1] = ty; p[3 * i + 2] = tz;
                if (i < cnt) st_off<pr_vec3>(cl, (j0 + i * kBlockThreads) * 12u, pr_vec3{ tx, ty, tz });
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) step_sq[i] = 0.0f;
        }
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            const uint32_t j = j0 + i * kBlockThreads;
            const float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
            bool pending = i < cnt;
            float best = accept;
            if (pending && w[i] != kNoPrev) {
                // KEEP THE WINNER WITHOUT SEARCHING: nn_search_kernel's test, expression by expression (nn_search.hip)
                const float d2 = (x - d[i].x) * (x - d[i].x) + (y - d[i].y) * (y - d[i].y) + (z - d[i].z) * (z - d[i].z);
                const float slack = sl[i] - margin_sqrt(step_sq[i]) * 1.000002f;
                if (d2 < accept && margin_sqrt(d2) * 1.00001f < slack) {
                    pending = false;
                    slk[j] = slack;
                    if (stop == 0xffffffffu) {
                        Corr c; c.dx = d[i].x; c.dy = d[i].y; c.dz = d[i].z; c.nx = nx[i]; c.ny = ny[i]; c.nz = nz[i];
                        if constexpr (kScoreOnly) accumulate_score(acc, x, y, z, c); else accumulate(acc, x, y, z, c);
                    }
                } else { const float bnd = d2 * 1.000001f + 1e-30f; if (bnd < best) best = bnd; }        // = nn_seed_bound
            }
            if (pending) {
                if (stop == 0xffffffffu) stop = (s << 2) | i;
                uint32_t bits = __float_as_uint(best) & ~1u;      // the two flags of a queue entry (nn_search_kernel): sign = no descent needed, low bit = settle
                if (step_sq[i] <= PR_NN_STILL && xf && PR_NN_SETTLE) bits |= 1u;
                if (w[i] != kNoPrev && step_sq[i] <= PR_NN_NODESCENT) bits |= 0x80000000u;
                best = __uint_as_float(bits);
            }
            // the list's order is immaterial (every entry is resolved on its own): one LDS atomic per wavefront and point slot
            const unsigned long long m = __ballot(pending);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                uint32_t base = 0;
                if ((int)lane == leader) base = atomicAdd(s_count, (uint32_t)__popcll(m));
                base = (uint32_t)__shfl((int)base, leader);
                if (pending) s_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2(j, __float_as_uint(best));
            }
        }
    }
    __syncthreads();                                             // the list, and this workgroup's cloud stores, are visible to all its wavefronts
    // ---------------- (B)
    const uint32_t n_list = *s_count;
    for (uint32_t k = threadIdx.x; k < n_list; k += kBlockThreads) {
        const uint2 e = s_list[k];
        const uint32_t j = e.x;
        const bool still = (e.y & 0x80000000u) != 0u, settle = (e.y & 1u) != 0u;
        float bst = __uint_as_float((e.y & 0x7fffffffu) | 1u);   // (the flag bit set: the bound rounded UP by at most an ulp)
        const pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);
        uint32_t wn = kNoPrev;
        float other = 0.0f;
        bool done = false;
        if (scene.grid) {
            float bsq = 0.0f, osq = 0.0f;
            bool tried = false;
            if (still) done = bst < accept && grid_search(scene, q.x, q.y, q.z, bst, wn, nullptr, &bsq, &osq, settle);
            else if (bst <= (iter <= 1u ? PR_NN_WINFIRST_EARLY : PR_NN_WINFIRST)) {                      // window first (nn_bound_kernel)
                tried = true;
                done = grid_search(scene, q.x, q.y, q.z, bst, wn, nullptr, &bsq, &osq, settle, true);
                if (!done && bsq > 0.0f) { const float bb = bsq * 1.000001f + 1e-30f; if (bb < bst) bst = bb; }
            }
            if (!done && !still) {
                grid_pyramid_bound(scene, q.x, q.y, q.z, bst, iter >= 1u);
                if (!tried) done = bst < accept && grid_search(scene, q.x, q.y, q.z, bst, wn, nullptr, &bsq, &osq, settle);
            }
            if (done) other = margin_sqrt(osq) * 0.99999f;
        }
#if defined(PR_LATE_EXPERIMENT) && PR_LATE_EXPERIMENT == 1
        if (!done) { wn = kNoPrev; other = 0.0f; }                 // TIMING ONLY (wrong results): what the kernel costs without its ordered walks
#else
        if (!done) { wn = query_nn_bounded(scene, q.x, q.y, q.z, bst); other = 0.0f; }
#endif     // the reference's own order: ties as pcd_scene.h:79-119 resolves them; no runner-up
        win[j] = wn;
        slk[j] = other;
    }
    __syncthreads();                                             // winners of the list are in memory for the lanes that resume
    // ---------------- (C)
    if (stop != 0xffffffffu) {
        const uint32_t s0 = stop >> 2;
        for (uint32_t s = s0; s < steps; ++s) {
            const uint32_t j0 = first + s * kPointsPerStep + threadIdx.x;
            if (j0 >= n) break;
            const uint32_t left = (n - j0 + kBlockThreads - 1u) / kBlockThreads, cnt = left < kPointsPerLane ? left : kPointsPerLane;
            late_resume_step<kScoreOnly>(acc, cl, win, scene, j0, cnt, s == s0 ? (stop & 3u) : 0u);
        }
    }
    acc_export(acc, acc_out);
}

__global__ __launch_bounds__(256, PR_LATE_WAVES) void nn_late_pass_kernel(IcpBatch b, SceneNNDev scene)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];      // the block's list: (point, bound bits) per entry, at most one per point
    __shared__ float wsum[4][kAccStride];
    __shared__ uint32_t s_count;

    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];
    const int32_t st = pm.state;
    if (st == kSkip) return;
    const uint32_t n = pm.count;
    const uint32_t ppb = b.steps * kPointsPerStep;
    if ((uint64_t)blockIdx.x * ppb >= n) return;
    float *cl = reinterpret_cast<float *>(b.cloud + pm.start);
    uint32_t *win = b.nn_prev + pm.start;
    float *slk = b.nn_slack + pm.start;
    const bool xf = (st == kRunWithTransform);
    float M[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) M[i] = xf ? pm.xform[i] : 0.0f;
    uint2 *s_list = reinterpret_cast<uint2 *>(lds_raw);
    const uint32_t used = (n + ppb - 1) / ppb;
    for (uint32_t vb = blockIdx.x; vb < used; vb += gridDim.x) {
        if (vb != blockIdx.x) __syncthreads();                   // wsum and the list of the previous trip have been read
        float acc[29];
        float t;
        if (b.score_only) { late_block<true>(acc, cl, win, slk, n, vb * ppb, b.steps, xf, M, scene, b.iter, s_list, &s_count); t = vb_reduce<true>(acc, wsum); }
        else { late_block<false>(acc, cl, win, slk, n, vb * ppb, b.steps, xf, M, scene, b.iter, s_list, &s_count); t = vb_reduce(acc, wsum); }
        if (pass_deliver(b, pose, vb, used, n, t)) return;
    }
}

// LDS a workgroup needs for its list; 0 when a block is too large for the fused form (the caller then keeps the four-kernel pass)
size_t nn_late_lds_bytes(uint32_t steps) { const size_t bytes = (size_t)steps * kPointsPerStep * sizeof(uint2); return bytes <= 64u * 1024u ? bytes : 0; }

hipError_t launch_nn_late_pass(const IcpBatch &b, const SceneNNDev &sc, uint32_t n_poses, hipStream_t s)
{
    if (n_poses == 0 || b.nblk == 0) return hipSuccess;
    const size_t lds = nn_late_lds_bytes(b.steps);
    if (!lds || !b.nn_prev || !b.nn_slack) return hipErrorInvalidValue;
    for (uint32_t p0 = 0; p0 < n_poses; p0 += 32768) {
        const uint32_t np = (n_poses - p0 < 32768) ? (n_poses - p0) : 32768;
        IcpBatch bb = b;
        bb.meta += p0;
        bb.partial += (size_t)p0 * b.nblk * kAccStride;
        if (bb.st) bb.st += p0;
        if (bb.arrive) bb.arrive += p0;
        if (bb.sums_out) bb.sums_out += (size_t)p0 * kAccStride;
        hipLaunchKernelGGL(nn_late_pass_kernel, dim3(b.grid_x ? b.grid_x : b.nblk, np), dim3(kBlockThreads), lds, s, bb, sc);
    }
    return hipGetLastError();
}

}  // namespace prk
