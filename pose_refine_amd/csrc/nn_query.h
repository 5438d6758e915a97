// nn_query.h -- Scene_nn::query (pcd_scene.h:60-136): the ordered walks (stackless, per-lane stack), wide-record helpers of the task walk, and the pixel grid of a scene made from a depth image
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#pragma once
#include "proj_query.h"

namespace prk {

// Scene_nn::query pcd_scene.h:60-136 -- same stackless near-first traversal, same strict '<' on
// leaf points and '<=' on the bound, but the bound is the FAR CHILD's own tight box instead of
// the current node's box.  That prunes a superset of what the reference prunes and can only skip
// subtrees whose every point is farther than the current best, so winner and distance are
// identical (DESIGN.md "kd-tree bound").
// dequantisation of a compact-record box coordinate (nn_records32_kernel): one multiply, one add
__device__ __forceinline__ float nn_deq(uint32_t q, float qmin, float qscale) { return qmin + (float)q * qscale; }
__device__ __forceinline__ float box_dist_sq(float sx, float sy, float sz, const float4 lo, const float4 hi)
{
    float lb = 0;
    if (sx < lo.x) lb += (lo.x - sx) * (lo.x - sx); else if (sx > hi.x) lb += (hi.x - sx) * (hi.x - sx);
    if (sy < lo.y) lb += (lo.y - sy) * (lo.y - sy); else if (sy > hi.y) lb += (hi.y - sy) * (hi.y - sy);
    if (sz < lo.z) lb += (lo.z - sz) * (lo.z - sz); else if (sz > hi.z) lb += (hi.z - sz) * (hi.z - sz);
    return lb;
}

template <bool kUseLds>
__device__ __forceinline__ bool query_nn(const SceneNNDev &s, const int4 *lds_topo, float sx, float sy, float sz, Corr &c)
{
    int cur = 0, prev = -1, best_i = 0;
    bool climbing = false;
    float best = FLT_MAX;
    while (cur >= 0) {
        const int4 t = (kUseLds && (uint32_t)cur < s.lds_nodes) ? lds_topo[cur] : s.topo[cur];
        const int parent = (t.w & 0x3fffffff) - 1;
        const bool leaf = t.z < 0;
        if (!climbing && leaf) {
            for (int i = t.x; i < t.y; ++i) {
                const float4 p = s.pts[i];
                const float d2 = (sx - p.x) * (sx - p.x) + (sy - p.y) * (sy - p.y) + (sz - p.z) * (sz - p.z);
                if (d2 < best) { best = d2; best_i = i; }
            }
            climbing = true; prev = cur; cur = parent;
            continue;
        }
        const int dim = (int)((uint32_t)t.w >> 30);
        const float q = (dim == 0) ? sx : ((dim == 1) ? sy : sz);
        const float diff = q - __int_as_float(t.x);
        const int near_c = (diff < 0) ? t.y : t.z;
        const int far_c  = (diff < 0) ? t.z : t.y;
        if (!climbing) { prev = cur; cur = near_c; continue; }
        if (prev == near_c) {
            const float lb = box_dist_sq(sx, sy, sz, s.bmin[far_c], s.bmax[far_c]);
            if (lb <= best) { prev = cur; cur = far_c; climbing = false; continue; }
        }
        prev = cur; cur = parent;
    }
    if (!(best < s.max_dist_diff * s.max_dist_diff)) return false;
    const float *d = reinterpret_cast<const float *>(s.pcd + best_i);
    const float *n = reinterpret_cast<const float *>(s.normal + best_i);
    c.dx = d[0]; c.dy = d[1]; c.dz = d[2]; c.nx = n[0]; c.ny = n[1]; c.nz = n[2];
    return true;
}

// Stack variant of the same search.  The order in which leaves are visited is the reference's
// near-first depth-first order: a far child is parked on a per-lane stack (LDS, [entry][lane] so the
// 64 lanes of a wavefront hit 64 different banks) together with the distance of the query to its
// box, and re-tested against the current best when it is popped -- exactly the test the stackless
// walk makes when it climbs back to the parent (pcd_scene.h:115).  Each internal node is fetched
// once, as one 64-byte record that already contains both children's boxes.
//   record = { split_v | left, child1 | right, child2 | -1, dim,  c1.min.xyz c1.max.xyz  c2.min.xyz c2.max.xyz }
constexpr int kLeafBatch = PR_LEAF_BATCH;
constexpr uint32_t kNoPrev = 0xffffffffu;

// Bound a search may start from when scene point `seed` is known to exist: its distance, inflated by one part in a million so
// that the point itself -- or an equal one visited earlier -- is still found by the strict '<' of the search.  The bound only
// removes subtrees and points that are strictly farther than an existing point; winner, distance and tie-break are those of the
// unseeded search.
__device__ __forceinline__ void nn_seed_bound(const SceneNNDev &s, float sx, float sy, float sz, uint32_t seed, float &best)
{
    const pr_vec3 p = s.pcd[seed != kNoPrev ? seed : 0u];
    if (seed != kNoPrev) {
        const float d2 = (sx - p.x) * (sx - p.x) + (sy - p.y) * (sy - p.y) + (sz - p.z) * (sz - p.z);
        const float b = d2 * 1.000001f + 1e-30f;
        if (b < best) best = b;
    }
}

// kCode = stack entries per lane (16 / 24), + 0x100 when the scene's compact 32-byte records are used.
// best_init: the bound the search starts from (<= max_dist_diff^2, see PR_NN_BOUNDED; tightened by nn_seed_bound).
// work counters of the search (SURVEY 8d "count its own visits"): per lane, summed into SceneNNDev::counters when that is set
struct NNCount { uint32_t nodes = 0, leaves = 0, leaf_points = 0; };
template <int kCode, bool kWantCorr = true>
__device__ __forceinline__ bool query_nn_stack_from(const SceneNNDev &s, const float4 *lds_rec, int *stk_node, float *stk_lb, float sx, float sy, float sz, Corr &c,
                                                    float best_init, uint32_t &winner, NNCount *cnt = nullptr)
{
    constexpr int kDepth = kCode & 0xff;
    constexpr bool kCompact = (kCode & 0x100) != 0;
    int cur = 0, sp = 0, best_i = -1;                            // -1: no point below the starting bound yet
    float best = best_init;
    winner = kNoPrev;
    for (;;) {
        float4 h, b0, b1, b2;
        if (cnt) cnt->nodes++;
        if constexpr (kCompact) {
            // compact records: half the bytes through the L1 (the kernel is bound by the texture-addresser / L1 rate of its
            // divergent loads, not by latency or HBM); only the far child's box is decoded
            // While the bound is wide (first passes, first descent) the whole 32-byte record is fetched at once.  Once it is
            // tight, 8 bytes (split | child | dim) are enough for most nodes: every point of the far side lies beyond the
            // split plane (left_max <= split <= right_min, pcd_scene.cpp:140-160), so (q - split)^2 is a lower bound of its
            // distance in the same float arithmetic, and only if that bound does not exceed `best` is the far box needed.
            const bool wide = best > PR_NN_WIDE_BOUND;
            uint4 A, B;
            if (wide) { A = s.rec32[(size_t)cur * 2]; B = s.rec32[(size_t)cur * 2 + 1]; }
            else { const uint2 d = s.desc[cur]; A = make_uint4(d.x, d.y, 0u, 0u); B = make_uint4(0u, 0u, 0u, 0u); }
            const uint32_t tag = A.y >> 30;
            if (tag == 3u) { h = make_float4(__uint_as_float(A.x), __uint_as_float(A.y & 0x3fffffffu), __int_as_float(-1), 0.0f); b0 = b1 = b2 = h; }
            else {
                const float q = (tag == 0u) ? sx : ((tag == 1u) ? sy : sz);
                const float diff = q - __uint_as_float(A.x);
                const bool left_near = diff < 0;
                const uint32_t c1 = A.y & 0x3fffffffu;
                const int near_c = (int)(left_near ? c1 : c1 + 1u), far_c = (int)(left_near ? c1 + 1u : c1);
                if (diff * diff <= best && sp < kDepth) {
                    if (!wide) { A = s.rec32[(size_t)cur * 2]; B = s.rec32[(size_t)cur * 2 + 1]; }
                    const uint32_t u0 = left_near ? B.y : A.z, u1 = left_near ? B.z : A.w, u2 = left_near ? B.w : B.x;
                    const float4 lo = make_float4(nn_deq(u0 & 0xffffu, s.qmin[0], s.qscale[0]), nn_deq(u0 >> 16, s.qmin[1], s.qscale[1]),
                                                  nn_deq(u1 & 0xffffu, s.qmin[2], s.qscale[2]), 0.0f);
                    const float4 hi = make_float4(nn_deq(u1 >> 16, s.qmin[0], s.qscale[0]), nn_deq(u2 & 0xffffu, s.qmin[1], s.qscale[1]),
                                                  nn_deq(u2 >> 16, s.qmin[2], s.qscale[2]), 0.0f);
                    const float lb = box_dist_sq(sx, sy, sz, lo, hi);
                    if (lb <= best) { stk_node[sp * kBlockThreads] = far_c; stk_lb[sp * kBlockThreads] = lb; ++sp; }
                }
                // While the bound is wide the record is in registers anyway: the near child is entered only if its own box is
                // within the bound (the reference walks into it unconditionally and finds nothing there).
                if (wide) {
                    const uint32_t v0 = left_near ? A.z : B.y, v1 = left_near ? A.w : B.z, v2 = left_near ? B.x : B.w;
                    const float4 nlo = make_float4(nn_deq(v0 & 0xffffu, s.qmin[0], s.qscale[0]), nn_deq(v0 >> 16, s.qmin[1], s.qscale[1]),
                                                   nn_deq(v1 & 0xffffu, s.qmin[2], s.qscale[2]), 0.0f);
                    const float4 nhi = make_float4(nn_deq(v1 >> 16, s.qmin[0], s.qscale[0]), nn_deq(v2 & 0xffffu, s.qmin[1], s.qscale[1]),
                                                   nn_deq(v2 >> 16, s.qmin[2], s.qscale[2]), 0.0f);
                    if (!(box_dist_sq(sx, sy, sz, nlo, nhi) <= best)) {
                        bool found = false;
                        while (sp > 0) {
                            --sp;
                            if (stk_lb[sp * kBlockThreads] <= best) { cur = stk_node[sp * kBlockThreads]; found = true; break; }
                        }
                        if (!found) break;
                        continue;
                    }
                }
                cur = near_c;
                continue;
            }
        } else
        // 64-byte records; the leading (top-level) ones may be staged in LDS
        if ((uint32_t)cur < s.lds_nodes) {
            const float4 *q = lds_rec + (size_t)cur * 4;
            h = q[0]; b0 = q[1]; b1 = q[2]; b2 = q[3];
        } else {
            const float4 *q = s.rec + (size_t)cur * 4;
            h = q[0]; b0 = q[1]; b1 = q[2]; b2 = q[3];
        }
        const int hz = __float_as_int(h.z);
        if (hz < 0) {                                            // leaf: points [left, right)
            const int lo = __float_as_int(h.x), hi = __float_as_int(h.y);
            if (cnt) { cnt->leaves++; cnt->leaf_points += (uint32_t)(hi - lo); }
            // four point loads are issued before the first compare (one memory round trip per batch); indices past
            // the end are clamped to the last point, whose repeated distance can never pass the strict '<' again,
            // so the in-order compares keep the reference's first-occurrence winner
#pragma unroll 1
            for (int i = lo; i < hi; i += kLeafBatch) {
                float d2[kLeafBatch];
                int idx[kLeafBatch];
#pragma unroll
                for (int k = 0; k < kLeafBatch; ++k) {
                    idx[k] = (i + k < hi) ? (i + k) : (hi - 1);
                    const pr_vec3 p = s.pcd[idx[k]];                 // 12-byte points: a quarter fewer bytes through the L1 than the padded copy
                    d2[k] = (sx - p.x) * (sx - p.x) + (sy - p.y) * (sy - p.y) + (sz - p.z) * (sz - p.z);
                }
#pragma unroll
                for (int k = 0; k < kLeafBatch; ++k)
                    if (d2[k] < best) { best = d2[k]; best_i = idx[k]; }
            }
            bool found = false;
            while (sp > 0) {
                --sp;
                if (stk_lb[sp * kBlockThreads] <= best) { cur = stk_node[sp * kBlockThreads]; found = true; break; }
            }
            if (!found) break;
        } else {
            const int dim = __float_as_int(h.w);
            const float q = (dim == 0) ? sx : ((dim == 1) ? sy : sz);
            const float diff = q - h.x;
            const bool left_near = diff < 0;
            const int near_c = left_near ? __float_as_int(h.y) : hz;
            const int far_c  = left_near ? hz : __float_as_int(h.y);
            const float4 lo = left_near ? make_float4(b1.z, b1.w, b2.x, 0.f) : make_float4(b0.x, b0.y, b0.z, 0.f);
            const float4 hi = left_near ? make_float4(b2.y, b2.z, b2.w, 0.f) : make_float4(b0.w, b1.x, b1.y, 0.f);
            const float lb = box_dist_sq(sx, sy, sz, lo, hi);
            if (lb <= best && sp < kDepth) { stk_node[sp * kBlockThreads] = far_c; stk_lb[sp * kBlockThreads] = lb; ++sp; }
            cur = near_c;
        }
    }
    if (!(best < s.max_dist_diff * s.max_dist_diff) || best_i < 0) return false;
    winner = (uint32_t)best_i;
    if constexpr (kWantCorr) {
        const float4 d = s.pts[best_i];                          // {x,y,z,0} copy of pcd[best_i]
        const float *n = reinterpret_cast<const float *>(s.normal + best_i);
        c.dx = d.x; c.dy = d.y; c.dz = d.z; c.nx = n[0]; c.ny = n[1]; c.nz = n[2];
    }
    return true;
}

// the search started from the acceptance bound and, for compact records, from up to two known scene points:
//   seed  = the previous pass' winner of this cloud point (temporal),
//   seed2 = the winner of the cloud point this lane handled just before (its neighbour in the image, 1-2 mm away): on
//           the first passes, when the cloud is still centimetres off the surface, that neighbour's answer is a far
//           tighter bound than anything the descent finds early
template <int kCode>
__device__ __forceinline__ bool query_nn_stack(const SceneNNDev &s, const float4 *lds_rec, int *stk_node, float *stk_lb, float sx, float sy, float sz, Corr &c,
                                               uint32_t seed, uint32_t seed2, uint32_t &winner)
{
    // a winner is only accepted below max_dist_diff^2 (pcd_scene.h query tail), so the search can start from that bound:
    // subtrees and points at or beyond it could only produce a neighbour the final test rejects
    float best = s.max_dist_diff * s.max_dist_diff;
    if constexpr ((kCode & 0x100) != 0) { nn_seed_bound(s, sx, sy, sz, seed, best); nn_seed_bound(s, sx, sy, sz, seed2, best); }
    return query_nn_stack_from<kCode>(s, lds_rec, stk_node, stk_lb, sx, sy, sz, c, best, winner);
}

// ---- wide records: eight subtree boxes per 128-byte line, searched as tasks of four lanes (nn_tree_wide_kernel) -----------------
// A per-lane walk pays for every (lane, load instruction) pair: 44 records of 32 bytes per query while a hypothesis is still
// centimetres off the surface, each lane on a line of its own.  (Measured first: the same wide nodes walked one query per lane --
// 15 node visits instead of 44, but eight 16-byte loads per visit and lane: pass 0 went from 5.2 to 6.8 ms.)  The wide form is made
// for cooperative access instead: a WIDE NODE is one 128-byte line of eight 16-byte slots {box as 6 x uint16 in the root-box
// frame, rounded OUTWARDS and verified with nn_deq_fma | reference}, one slot per descendant of a binary node (the frontier the
// builder reaches by repeatedly opening the largest box, about three binary levels).  A group of four adjacent lanes handles one
// (query, node): lane c loads slots 2c and 2c+1 -- the group's loads are ONE coalesced line -- and tests those two boxes.  A leaf
// reference carries (first point, count): the group reads the leaf's points from the padded point array, two 16-byte points per lane.
//   reference: kWideLeaf | count << 27 | first point   (leaf, count 1..15)   |   index of a wide node   |   kWideEmpty
// The search is ORDER-FREE: every admitted child becomes a task of its own.  That finds the minimum squared distance m and every
// point attaining it, because a box is only skipped when its lower bound EXCEEDS the query's bound (>= m).  The reference's answer (pcd_scene.h:60-136) is the first point at distance m in ITS visiting order: if exactly one
// point attains m, that is the answer under any order (the argument the pixel window already rests on); if several do, or a stack
// overflows, the query is repeated by the ordered stackless walk (query_nn_bounded), started from m.  `second`, a lower bound on the
// squared distance of every scene point other than the winner (skipped boxes count with their bound), lets the following passes
// keep this winner without searching (nn_search_kernel) -- the ordered walks cannot report it.
constexpr uint32_t kWideLeaf = 0x80000000u, kWideEmpty = 0xffffffffu;
constexpr uint32_t kWideMaxLeafPoints = 15u, kWideFirstMask = 0x07ffffffu;
__device__ __forceinline__ float nn_deq_fma(uint32_t q, float qmin, float qscale) { return __builtin_fmaf((float)q, qscale, qmin); }
__device__ __forceinline__ float wide_box_lb(float sx, float sy, float sz, uint32_t u0, uint32_t u1, uint32_t u2, const SceneNNDev &s)
{
    const float lox = nn_deq_fma(u0 & 0xffffu, s.qmin[0], s.qscale[0]), loy = nn_deq_fma(u0 >> 16, s.qmin[1], s.qscale[1]);
    const float loz = nn_deq_fma(u1 & 0xffffu, s.qmin[2], s.qscale[2]), hix = nn_deq_fma(u1 >> 16, s.qmin[0], s.qscale[0]);
    const float hiy = nn_deq_fma(u2 & 0xffffu, s.qmin[1], s.qscale[1]), hiz = nn_deq_fma(u2 >> 16, s.qmin[2], s.qscale[2]);
    // = box_dist_sq: per axis (lo - q)^2 below the box, (hi - q)^2 == (q - hi)^2 above it, 0 inside
    const float dx = fmaxf(fmaxf(lox - sx, sx - hix), 0.0f), dy = fmaxf(fmaxf(loy - sy, sy - hiy), 0.0f), dz = fmaxf(fmaxf(loz - sz, sz - hiz), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}
// Scene_nn::query pcd_scene.h:60-136 as query_nn above, started from a bound (an existing point's distance, inflated: nn_seed_bound) and
// reporting the winner's index: the ordered walk ties and overflows of the wide search fall back to.  No LDS.
__device__ __forceinline__ uint32_t query_nn_bounded(const SceneNNDev &s, float sx, float sy, float sz, float best_init)
{
    int cur = 0, prev = -1, best_i = -1;
    bool climbing = false;
    float best = best_init;
    while (cur >= 0) {
        const int4 t = s.topo[cur];
        const int parent = (t.w & 0x3fffffff) - 1;
        if (!climbing && t.z < 0) {
            for (int i = t.x; i < t.y; ++i) {
                const float4 p = s.pts[i];
                const float d2 = (sx - p.x) * (sx - p.x) + (sy - p.y) * (sy - p.y) + (sz - p.z) * (sz - p.z);
                if (d2 < best) { best = d2; best_i = i; }
            }
            climbing = true; prev = cur; cur = parent;
            continue;
        }
        const int dim = (int)((uint32_t)t.w >> 30);
        const float q = (dim == 0) ? sx : ((dim == 1) ? sy : sz);
        const float diff = q - __int_as_float(t.x);
        const int near_c = (diff < 0) ? t.y : t.z;
        const int far_c  = (diff < 0) ? t.z : t.y;
        if (!climbing) { prev = cur; cur = near_c; continue; }
        if (prev == near_c) {
            const float lb = box_dist_sq(sx, sy, sz, s.bmin[far_c], s.bmax[far_c]);
            if (lb <= best) { prev = cur; cur = far_c; climbing = false; continue; }
        }
        prev = cur; cur = parent;
    }
    return (best_i >= 0 && best < s.max_dist_diff * s.max_dist_diff) ? (uint32_t)best_i : kNoPrev;
}

// ---- pixel grid of a kd-tree scene ---------------------------------------------------------------------------------------
// A Scene_nn is made from a depth image (pcd_scene.cpp:10-29), so its points are the pixels of that image: cell (px, py) of the
// grid holds the point that projects into it.  With a valid upper bound B on the squared nearest-neighbour distance (from a
// seed: an existing scene point), every scene point closer than sqrt(B) projects into a small pixel window around the query's
// own pixel, and scanning that window IS the exact search -- except for the tie-break between equidistant points, which the
// reference resolves by traversal order: a tie (or an empty window, or a window too large to pay) hands the query to the
// kd-tree search.  A unique strict minimum is the reference's winner under any visiting order.
//   window: u = x/z*fx + cx.  For |p - q| <= r and z_q - r > 0:  |u_p - u_q| <= fx * r * (z_q + |x_q|) / (z_q * (z_q - r)),
//   and |floor(a) - floor(b)| <= ceil(|a - b|); r carries a 1e-4 relative margin and the window another 1e-3 px for rounding.
__device__ __forceinline__ void grid_project(const SceneNNDev &s, float x, float y, float z, float &u, float &v)
{
    u = x / z * s.gfx + s.gcx + 0.5f;
    v = y / z * s.gfy + s.gcy + 0.5f;
}
constexpr int kGridMaxW = PR_GRID_MAXW;                                     // windows up to 5 x 5 cells; larger ones go to the tree
// half-widths of the pixel window that holds every scene point closer than sqrt(bound); false when it exceeds kGridMaxW
__device__ __forceinline__ bool grid_window(const SceneNNDev &s, float sx, float sy, float sz, float bound, int &wx, int &wy)
{
    const float r = margin_sqrt(bound) * 1.0001f;
    if (!(sz - r > 0.25f * sz)) return false;                    // also NaN and points at or behind the camera
    const float k = r * margin_rcp(sz * (sz - r)) * 1.000001f;
    const float du = s.gfx * k * (sz + fabsf(sx)), dv = s.gfy * k * (sz + fabsf(sy));
    if (!(du <= (float)kGridMaxW - 1e-3f && dv <= (float)kGridMaxW - 1e-3f)) return false;
    wx = (int)ceilf(du + 1e-3f); wy = (int)ceilf(dv + 1e-3f);
    return true;
}
// Coarse-to-fine descent through the representative points: the nearest of ALL 64 x 64-block representatives, then the nearest
// 16 x 16-block representative in and around that block (the block's 4 x 4 children plus the row and column before them: PR_RING_W / PR_RING_OFF), then 4 x 4 blocks, then pixels.
// Every point met is an existing scene point, so its distance is a valid bound (nn_seed_bound's argument); the descent itself
// decides nothing.  On the test.cpp scene it lands on the true nearest neighbour for 70-85 % of the queries of a hypothesis that
// starts centimetres off the surface and within a few points of it for the rest (9 + 3 x 25 distance evaluations) -- which
// turns the tree search that follows from "find the neighbour" into "confirm it".
__device__ __forceinline__ void grid_ring_min(const float4 *__restrict__ level, int lw, int lh, int x0, int y0, int nb, float sx, float sy, float sz,
                                              float &dmin, int &bx, int &by)
{
    // The descent is bound by its instruction count, not by its loads (SQ counters, profiles/r04): so the clamping is done once per column and
    // row (12 clamps instead of 72), a cell's address is one 32-bit add on top of a scalar base, and the arg-min is kept as ONE index
    // (same visiting order -- rows, then columns -- and the same strict '<' as before: the same minimum and the same winner).
    // kRingRows rows (kW cells each) are in flight at a time.
    constexpr int kRingRows = PR_RING_ROWS, kW = PR_RING_W;
    uint32_t xo[kW], yo[kW];
#pragma unroll
    for (int d = 0; d < kW; ++d) {
        xo[d] = (uint32_t)min(max(x0 + d, 0), lw - 1) * 16u;
        yo[d] = (uint32_t)(min(max(y0 + min(d, nb - 1), 0), lh - 1) * lw) * 16u;
    }
    float best = FLT_MAX;
    int kbest = 0;
#pragma unroll
    for (int dy0 = 0; dy0 < kW; dy0 += kRingRows) {
        if (dy0 >= nb) break;
        float4 c[kRingRows][kW];
#pragma unroll
        for (int r = 0; r < kRingRows; ++r)
#pragma unroll
            for (int dx = 0; dx < kW; ++dx) c[r][dx] = ld_off<float4>(level, yo[min(dy0 + r, kW - 1)] + xo[dx]);
#pragma unroll
        for (int r = 0; r < kRingRows; ++r)
#pragma unroll
            for (int dx = 0; dx < kW; ++dx) {
                if (dy0 + r >= kW) continue;
                const float d2 = (sx - c[r][dx].x) * (sx - c[r][dx].x) + (sy - c[r][dx].y) * (sy - c[r][dx].y) + (sz - c[r][dx].z) * (sz - c[r][dx].z);
                const bool lt = d2 < best;
                best = lt ? d2 : best; kbest = lt ? (dy0 + r) * kW + dx : kbest;
            }
    }
    const int ky = kbest / kW;
    dmin = best;
    bx = min(max(x0 + (kbest - ky * kW), 0), lw - 1);
    by = min(max(y0 + min(ky, nb - 1), 0), lh - 1);
}
// Round 5: the rings of the descent as SELECTION KEYS.  The descent decides nothing -- it only has to land on a near scene point, whose distance
// (inflated) is the bound -- so its arithmetic is free to be approximate: a cell's key is its squared distance with the products contracted into
// FMAs (x and y side by side in packed instructions), the low six mantissa bits replaced by the cell's number in the ring; the ring's minimum
// is one unsigned minimum per cell (v_min3_u32: two cells per instruction).  6.5 VALU instructions per cell instead of 12 (three subtractions,
// three multiplications, two additions, a compare and two selects), and the ring is clamped as a whole (its origin moved inside the level),
// not cell by cell.  The key of the landing cell understates that cell's exact distance (the search's own expression, pcd_scene.h:88-91) by at
// most the truncation (2^-17 relative) plus the rounding differences of three contracted products (< 3e-7): the bound taken from it is
// inflated by 2e-5 instead of nn_seed_bound's 1e-6 -- a tenth of a micron at 15 mm, far inside the 5-micron units of the wide records.
constexpr uint32_t kRingKeyMask = ~63u;
__device__ __forceinline__ uint32_t ring_key(const float4 c, const float2v sxy, float sz, uint32_t idx)
{
    const float2v dxy = float2v{ c.x, c.y } - sxy;
    const float2v qxy = dxy * dxy;
    const float dz = c.z - sz;
    const float d2 = __builtin_fmaf(dz, dz, qxy.x) + qxy.y;          // empty cells hold huge coordinates: +inf (0x7f800000), above every real key
    return (__float_as_uint(d2) & kRingKeyMask) | idx;
}
// the kW x kW ring whose first cell is (x0, y0), moved inside the level; returns the smallest key (index bits cleared) and that cell
template <int kW>
__device__ __forceinline__ uint32_t grid_ring_key(const float4 *__restrict__ level, int lw, int lh, int x0, int y0, const float2v sxy, float sz, int &bx, int &by)
{
    static_assert(kW * kW <= 64, "a ring's cell number has six bits in the key");
    constexpr int kRingRows = PR_RING_ROWS;
    const int xc = min(max(x0, 0), lw - kW), yc = min(max(y0, 0), lh - kW);          // (the caller guarantees lw, lh >= kW)
    const uint32_t pitch = (uint32_t)lw * 16u;
    uint32_t row = (uint32_t)(yc * lw + xc) * 16u, best = 0xffffffffu;
#pragma unroll
    for (int dy0 = 0; dy0 < kW; dy0 += kRingRows) {
        float4 c[kRingRows][kW];
#pragma unroll
        for (int r = 0; r < kRingRows; ++r)
#pragma unroll
            for (int dx = 0; dx < kW; ++dx)
                if (dy0 + r < kW) c[r][dx] = ld_off<float4>(level, row + (uint32_t)r * pitch + (uint32_t)dx * 16u);
#pragma unroll
        for (int r = 0; r < kRingRows; ++r)
#pragma unroll
            for (int dx = 0; dx < kW; ++dx)
                if (dy0 + r < kW) best = min(best, ring_key(c[r][dx], sxy, sz, (uint32_t)((dy0 + r) * kW + dx)));
        row += (uint32_t)kRingRows * pitch;
#if PR_RING_STAGED
        // the next rows' loads stay behind this group's keys: with nothing between them the compiler issues all kW x kW loads of a ring at once
        // (their addresses no longer depend on anything) and the kernel needs 101 VGPRs instead of 72 -- four wavefronts per SIMD instead of seven
        asm volatile("" : "+v"(best) : : "memory");
#endif
    }
    const int k = (int)(best & 63u), ky = k / kW;
    bx = xc + (k - ky * kW); by = yc + ky;
    return best & kRingKeyMask;
}
__device__ __forceinline__ void grid_pyramid_bound(const SceneNNDev &s, float sx, float sy, float sz, float &best, bool late = false)
{
    const int w4 = ((int)s.gw + 3) / 4, h4 = ((int)s.gh + 3) / 4, w16 = (w4 + 3) / 4, h16 = (h4 + 3) / 4, w64 = (w16 + 3) / 4, h64 = (h16 + 3) / 4;
    float dmin = FLT_MAX, dall;
    int bx = 0, by = 0;
    // Round 5: the descent starts AT THE QUERY'S OWN PROJECTION -- a ring of 5 x 5 blocks of 16 x 16 pixels centred on it (3 x 3 when `late`: from
    // pass 1 on the cloud is within millimetres of the surface, PR_DESCENT_LATE) -- instead of choosing one of the 3 x 3 blocks of 64 x 64 pixels
    // around it first: 9 to 25 cells fewer, and a BETTER landing (the chosen 64-block's children did not always hold the neighbour of a query near
    // the block's edge): same box, bound 1.58 -> 1.46 ms and walk 2.94 -> 2.80 ms per group-step, configs[2] 50.9 -> 53.2 k poses/s.  A ring that
    // holds no scene point at all (the query projects far off the object) falls through to the blocks of 64 x 64 pixels below.
    if (w16 >= PR_RING16_W && h16 >= PR_RING16_W && w16 >= PR_RING_W && h16 >= PR_RING_W) {
        float u, v;
        grid_project(s, sx, sy, sz, u, v);
        if (u > -1e6f && u < 1e6f && v > -1e6f && v < 1e6f) {
            const float2v sxy{ sx, sy };
            uint32_t k = 0xffffffffu;
            const int pu = (int)floorf(u), pv = (int)floorf(v);
            if (late && PR_DESCENT_LATE == 2) {                  // straight to the 4 x 4-pixel blocks around the projection
                k = grid_ring_key<PR_RING_W>(s.pyr4, w4, h4, (pu >> 2) - PR_RING_W / 2, (pv >> 2) - PR_RING_W / 2, sxy, sz, bx, by);
            } else {
                if (late && PR_DESCENT_LATE == 1) k = grid_ring_key<3>(s.pyr16, w16, h16, (pu >> 4) - 1, (pv >> 4) - 1, sxy, sz, bx, by);
                else k = grid_ring_key<PR_RING16_W>(s.pyr16, w16, h16, (pu >> 4) - PR_RING16_W / 2, (pv >> 4) - PR_RING16_W / 2, sxy, sz, bx, by);
                if (k < 0x7f800000u) k = min(k, grid_ring_key<PR_RING_W>(s.pyr4, w4, h4, bx * 4 - PR_RING_OFF, by * 4 - PR_RING_OFF, sxy, sz, bx, by));
            }
            if (k < 0x7f800000u) {
                k = min(k, grid_ring_key<PR_RING_W>(s.grid, (int)s.gw, (int)s.gh, bx * 4 - PR_RING_OFF, by * 4 - PR_RING_OFF, sxy, sz, bx, by));
                // (a further ring of pixels centred on the landing pixel: walk pass 0 1195 -> 1146 us, bound pass 0 275 -> 355 us -- not kept)
                const float bk = __uint_as_float(k) * 1.00002f + 1e-30f;     // the landing cell's exact distance is below this (see ring_key)
                if (bk < best) best = bk;
                return;
            }
        }
    }
    {   // the 3 x 3 blocks of 64 x 64 pixels around the query's own projection first: a hypothesis within a few centimetres / degrees of the
        // scene pose has its neighbour there; only a query that finds nothing there looks at all blocks
        float u, v;
        grid_project(s, sx, sy, sz, u, v);
        if (u > -1e6f && u < 1e6f && v > -1e6f && v < 1e6f) {
            const int cx = min(max((int)floorf(u) >> 6, 0), w64 - 1), cy = min(max((int)floorf(v) >> 6, 0), h64 - 1);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int x = min(max(cx + k % 3 - 1, 0), w64 - 1), y = min(max(cy + k / 3 - 1, 0), h64 - 1);
                const float4 c = s.pyr64[y * w64 + x];
                const float d2 = (sx - c.x) * (sx - c.x) + (sy - c.y) * (sy - c.y) + (sz - c.z) * (sz - c.z);
                if (d2 < dmin) { dmin = d2; bx = x; by = y; }
            }
        }
    }
    if (!(dmin < 1.0e20f))
    for (int i = 0; i < w64 * h64; ++i) {                          // wave-uniform addresses: every lane reads the same few cache lines
        const float4 c = s.pyr64[i];
        const float d2 = (sx - c.x) * (sx - c.x) + (sy - c.y) * (sy - c.y) + (sz - c.z) * (sz - c.z);
        if (d2 < dmin) { dmin = d2; bx = i % w64; by = i / w64; }
    }
    dall = dmin;
    if (w16 >= PR_RING_W && h16 >= PR_RING_W) {                      // (wave-uniform: every level holds a whole ring -- any frame of 320 x 320 pixels or more)
        const float2v sxy{ sx, sy };
        uint32_t k = grid_ring_key<PR_RING_W>(s.pyr16, w16, h16, bx * 4 - PR_RING_OFF, by * 4 - PR_RING_OFF, sxy, sz, bx, by);
        k = min(k, grid_ring_key<PR_RING_W>(s.pyr4, w4, h4, bx * 4 - PR_RING_OFF, by * 4 - PR_RING_OFF, sxy, sz, bx, by));
        k = min(k, grid_ring_key<PR_RING_W>(s.grid, (int)s.gw, (int)s.gh, bx * 4 - PR_RING_OFF, by * 4 - PR_RING_OFF, sxy, sz, bx, by));
        const float bk = __uint_as_float(k) * 1.00002f + 1e-30f;     // the landing cell's exact distance is below this (see above); +inf / NaN keys give no bound
        const float b0 = dall * 1.000001f + 1e-30f;
        if (b0 < best) best = b0;
        if (bk < best) best = bk;
        return;
    }
    float d;
    grid_ring_min(s.pyr16, w16, h16, bx * 4 - PR_RING_OFF, by * 4 - PR_RING_OFF, PR_RING_W, sx, sy, sz, d, bx, by);   dall = fminf(dall, d);
    grid_ring_min(s.pyr4, w4, h4, bx * 4 - PR_RING_OFF, by * 4 - PR_RING_OFF, PR_RING_W, sx, sy, sz, d, bx, by);       dall = fminf(dall, d);
    grid_ring_min(s.grid, (int)s.gw, (int)s.gh, bx * 4 - PR_RING_OFF, by * 4 - PR_RING_OFF, PR_RING_W, sx, sy, sz, d, bx, by);   dall = fminf(dall, d);
    const float b = dall * 1.000001f + 1e-30f;                       // empty cells hold huge coordinates: inf
    if (b < best) best = b;
}
// `full` (round 5, "window first"): the caller has no tight bound yet -- only a previous winner a few millimetres away.  The LARGEST window is
// scanned without one: the radius it covers (`rc`, the formula of `settle`) takes the bound's place, and whatever minimum the scan finds below
// rc^2 is the global one (every closer point projects into the window) -- exact under the same uniqueness rule, without the 84-cell descent
// through the representative points that would otherwise have to tighten the bound first.  When nothing lies within rc the call fails and the
// query takes the usual way (descent, tree); `*best_sq` then carries the nearest point the window did hold (a valid bound), if any.
__device__ __forceinline__ bool grid_search(const SceneNNDev &s, float sx, float sy, float sz, float bound, uint32_t &winner, uint32_t *cells = nullptr,
                                            float *best_sq = nullptr, float *other_sq = nullptr, bool settle = false, bool full = false)
{
    int wx, wy;
    float cover_full = 0.0f;
    if (full) {
        const float W = (float)kGridMaxW - 4e-3f;
        const float rx = W * sz * sz * margin_rcp(s.gfx * (sz + fabsf(sx)) + W * sz), ry = W * sz * sz * margin_rcp(s.gfy * (sz + fabsf(sy)) + W * sz);
        const float rc = fminf(rx, ry) * 0.999f;                   // (grid_window below VERIFIES that the window of rc fits: rc itself may be approximate)
        if (!(rc > 0.0f) || !grid_window(s, sx, sy, sz, rc * rc, wx, wy)) return false;
        const float c2 = rc * rc * 0.9999f;
        if (c2 < bound) bound = c2;                                // only points strictly inside the covered radius can win
        cover_full = rc * rc;                                      // (the margin: everything the window holds; wx, wy are its window)
    }
    // The window has to hold every point closer than sqrt(bound) for the search to be exact.  When a slightly larger window still
    // fits it is taken instead: the extra ring costs a few cells and tells how far the RUNNER-UP is (other_sq), which is what lets
    // the following passes keep this winner without searching (nn_search_kernel).  `settle`: the point has (nearly) stopped moving,
    // so whatever margin is found will last for the rest of the loop -- the radius is then not sqrt(bound) + half a millimetre but the
    // largest the 5 x 5 window covers (du(r) = fx (z + |x|) r / (z (z - r)) <= W  <=>  r <= W z^2 / (fx (z + |x|) + W z)): with the
    // fixed pad a point whose neighbour is more than half a millimetre away (a quarter of them: the depth image is in whole
    // millimetres) never got a margin at all and went through the window in every pass.
    float cover = bound;
    if (full) cover = cover_full;
    else if (other_sq) {
        const float rb = margin_sqrt(bound) * 1.000001f;
        float rc = rb + PR_NN_COVER_PAD;
        if (settle) {
            const float W = (float)kGridMaxW - 4e-3f;
            const float rx = W * sz * sz * margin_rcp(s.gfx * (sz + fabsf(sx)) + W * sz), ry = W * sz * sz * margin_rcp(s.gfy * (sz + fabsf(sy)) + W * sz);
            rc = fminf(rx, ry) * 0.999f;                           // (approximate is fine: grid_window verifies the window of whatever radius is taken)
        }
        if (rc > rb && grid_window(s, sx, sy, sz, rc * rc, wx, wy)) cover = rc * rc;
        else if (settle) { rc = rb + PR_NN_COVER_PAD; if (grid_window(s, sx, sy, sz, rc * rc, wx, wy)) cover = rc * rc; }
    }
    if (!full && cover == bound && !grid_window(s, sx, sy, sz, bound, wx, wy)) return false;
    float u, v;
    grid_project(s, sx, sy, sz, u, v);
    if (!(u > -1e6f && u < 1e6f && v > -1e6f && v < 1e6f)) return false;
    const int cx0 = (int)floorf(u), cy0 = (int)floorf(v);
    const int x0 = max(cx0 - wx, 0), x1 = min(cx0 + wx, (int)s.gw - 1);
    const int y0 = max(cy0 - wy, 0), y1 = min(cy0 + wy, (int)s.gh - 1);
    if (x0 > x1 || y0 > y1) return false;
    if (cells) *cells += (uint32_t)((x1 - x0 + 1) * (y1 - y0 + 1));
    float best = bound, second = cover;                           // `second`: smallest squared distance of any scene point other than the winner,
    int best_i = -1, ties = 0;                                    // capped at what the window covers (points outside it are at least that far)
    if (wx <= 1 && wy <= 1) {
        // the common case once aligned (a bound below one pixel): the 3 x 3 cells in ONE round trip instead of one per row
        float4 c[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) c[i] = s.grid[(size_t)min(y0 + i / 3, y1) * s.gw + min(x0 + i % 3, x1)];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (x0 + i % 3 > x1 || y0 + i / 3 > y1) continue;    // (clamped repeats must not count as ties)
            const float d2 = (sx - c[i].x) * (sx - c[i].x) + (sy - c[i].y) * (sy - c[i].y) + (sz - c[i].z) * (sz - c[i].z);
            if (d2 < best) { if (best_i >= 0) second = best; best = d2; best_i = __float_as_int(c[i].w); ties = 0; }
            else if (d2 == best && best_i >= 0) ++ties;
            else if (d2 < second) second = d2;
        }
        if (best_i < 0 || ties != 0) { if (full && best_sq && best_i >= 0) *best_sq = best; return false; }     // (a tie: still an existing point's distance)
        winner = (uint32_t)best_i;
        if (best_sq) *best_sq = best;
        if (other_sq) *other_sq = second;
        return true;
    }
    constexpr int kWinRows = PR_WIN_ROWS;                           // rows of the window in flight at a time (one round trip per group)
    for (int y = y0; y <= y1; y += kWinRows) {
        float4 c[kWinRows][2 * kGridMaxW + 1];
#pragma unroll
        for (int r = 0; r < kWinRows; ++r) {
            const float4 *row = s.grid + (size_t)min(y + r, y1) * s.gw;
#pragma unroll
            for (int i = 0; i < 2 * kGridMaxW + 1; ++i) c[r][i] = row[min(x0 + i, x1)];
        }
#pragma unroll
        for (int r = 0; r < kWinRows; ++r)
#pragma unroll
            for (int i = 0; i < 2 * kGridMaxW + 1; ++i) {
                if (y + r > y1 || x0 + i > x1) continue;             // (clamped repeats must not count as ties)
                const float d2 = (sx - c[r][i].x) * (sx - c[r][i].x) + (sy - c[r][i].y) * (sy - c[r][i].y) + (sz - c[r][i].z) * (sz - c[r][i].z);
                if (d2 < best) { if (best_i >= 0) second = best; best = d2; best_i = __float_as_int(c[r][i].w); ties = 0; }
                else if (d2 == best && best_i >= 0) ++ties;
                else if (d2 < second) second = d2;
            }
    }
    if (best_i < 0 || ties != 0) { if (full && best_sq && best_i >= 0) *best_sq = best; return false; }
    winner = (uint32_t)best_i;
    if (best_sq) *best_sq = best;
    if (other_sq) *other_sq = second;
    return true;
}
// a first bound for a query nothing is known about yet: the nearest of the scene points in the 3 x 3 cells around its own pixel
__device__ __forceinline__ void grid_seed_bound(const SceneNNDev &s, float sx, float sy, float sz, float &best)
{
    float u, v;
    grid_project(s, sx, sy, sz, u, v);
    if (!(u > -1e6f && u < 1e6f && v > -1e6f && v < 1e6f)) return;
    const int cx0 = (int)floorf(u), cy0 = (int)floorf(v);
    float4 c[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int x = min(max(cx0 + (i % 3) - 1, 0), (int)s.gw - 1), y = min(max(cy0 + (i / 3) - 1, 0), (int)s.gh - 1);
        c[i] = s.grid[(size_t)y * s.gw + x];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float d2 = (sx - c[i].x) * (sx - c[i].x) + (sy - c[i].y) * (sy - c[i].y) + (sz - c[i].z) * (sz - c[i].z);
        const float b = d2 * 1.000001f + 1e-30f;                  // empty cells hold huge coordinates: b = inf
        if (b < best) best = b;
    }
}

}  // namespace prk
