// nn_build.hip -- traversal records derived from the reference's Node_kdtree array: topology + tight boxes, 64- and 32-byte records, 128-byte wide records
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#include "pr_launch.h"
#include "nn_query.h"

namespace prk {

__global__ __launch_bounds__(256) void nn_accel_kernel(const pr_kdnode *__restrict__ nodes, uint32_t n_nodes,
                                                       const pr_vec3 *__restrict__ pcd, uint32_t n_points,
                                                       int4 *__restrict__ topo, float4 *__restrict__ bmin,
                                                       float4 *__restrict__ bmax, float4 *__restrict__ pts)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_points) pts[i] = make_float4(pcd[i].x, pcd[i].y, pcd[i].z, 0.0f);
    if (i >= n_nodes) return;
    const pr_kdnode nd = nodes[i];
    const bool leaf = (nd.child1 < 0 || nd.child2 < 0);              // pcd_scene.h:21-24
    const int pw = ((nd.parent + 1) & 0x3fffffff) | (int)((uint32_t)(nd.split_dim & 3) << 30);
    if (leaf) {
        float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for (int k = nd.left; k < nd.right; ++k) {
            const float c3[3] = { pcd[k].x, pcd[k].y, pcd[k].z };
            for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], c3[d]); hi[d] = fmaxf(hi[d], c3[d]); }
        }
        topo[i] = make_int4(nd.left, nd.right, -1, pw);
        bmin[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        bmax[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    } else {
        // .w of the two box records: the "size" (squared box diagonal; -1: a leaf) of child1 / child2 -- what nn_wide_open_kernel orders its frontier by,
        // so that opening a node is ONE round of loads keyed by the node (it was the node's links, then the children's links and boxes)
        float csz[2] = { -1.0f, -1.0f };
        const int ch[2] = { nd.child1, nd.child2 };
        for (int c = 0; c < 2; ++c)
            if ((uint32_t)ch[c] < n_nodes) {
                const pr_kdnode k = nodes[ch[c]];
                if (!(k.child1 < 0 || k.child2 < 0)) {
                    const float dx = k.bbox[1] - k.bbox[0], dy = k.bbox[3] - k.bbox[2], dz = k.bbox[5] - k.bbox[4];
                    csz[c] = dx * dx + dy * dy + dz * dz;
                }
            }
        topo[i] = make_int4(__float_as_int(nd.split_v), nd.child1, nd.child2, pw);
        bmin[i] = make_float4(nd.bbox[0], nd.bbox[2], nd.bbox[4], csz[0]);
        bmax[i] = make_float4(nd.bbox[1], nd.bbox[3], nd.bbox[5], csz[1]);
    }
}

// 64-byte traversal records for the stack variant + the depth of the tree (longest root-to-node path)
__global__ __launch_bounds__(256) void nn_records_kernel(const int4 *__restrict__ topo, const float4 *__restrict__ bmin,
                                                         const float4 *__restrict__ bmax, uint32_t n_nodes,
                                                         float4 *__restrict__ rec, uint32_t *__restrict__ max_depth)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    const int4 t = topo[i];
    float4 r0, r1 = make_float4(0, 0, 0, 0), r2 = r1, r3 = r1;
    if (t.z < 0) r0 = make_float4(__int_as_float(t.x), __int_as_float(t.y), __int_as_float(-1), 0.0f);
    else {
        const int dim = (int)((uint32_t)t.w >> 30);
        r0 = make_float4(__int_as_float(t.x), __int_as_float(t.y), __int_as_float(t.z), __int_as_float(dim));
        const int y = ((uint32_t)t.y < n_nodes) ? t.y : 0, z = ((uint32_t)t.z < n_nodes) ? t.z : 0;   // malformed links are reported below
        const float4 a0 = bmin[y], a1 = bmax[y], c0 = bmin[z], c1 = bmax[z];
        r1 = make_float4(a0.x, a0.y, a0.z, a1.x);
        r2 = make_float4(a1.y, a1.z, c0.x, c0.y);
        r3 = make_float4(c0.z, c1.x, c1.y, c1.z);
    }
    rec[(size_t)i * 4] = r0; rec[(size_t)i * 4 + 1] = r1; rec[(size_t)i * 4 + 2] = r2; rec[(size_t)i * 4 + 3] = r3;
    uint32_t depth = 0;
    for (int p = (t.w & 0x3fffffff) - 1; p >= 0 && depth < 4096; p = (topo[p].w & 0x3fffffff) - 1) ++depth;
    // The stack size is derived from this depth, which follows the PARENT links, while the search follows the CHILD links: the
    // nodes are the caller's, so the two are cross-checked.  Any disagreement reports an impossible depth and the scene is
    // searched with the reference's stackless walk instead (which uses both kinds of link exactly like pcd_scene.h:60-136).
    if (t.z >= 0) {
        const bool in_range = (uint32_t)t.y < n_nodes && (uint32_t)t.z < n_nodes && t.y > (int)i && t.z > (int)i;
        if (!in_range || (topo[t.y].w & 0x3fffffff) - 1 != (int)i || (topo[t.z].w & 0x3fffffff) - 1 != (int)i) depth = 0x7fffffffu;
    }
    atomicMax(max_depth, depth);
}

// Compact traversal records: 32 bytes per node.
//   word 0: split value (internal) | first point (leaf)
//   word 1: child1 | dim << 30 (internal; child2 = child1 + 1, as KDTree_cpu::build_tree appends children pairwise) | end point | 3 << 30 (leaf)
//   words 2..7: the boxes of child1 and child2 as 12 uint16 {lo.x lo.y lo.z hi.x hi.y hi.z} x 2, quantised in the frame of
//   the root box and rounded OUTWARDS (checked with the very dequantisation arithmetic the query uses).  A looser box can
//   only make the search visit a subtree it could have skipped -- a subtree whose every point is farther than the current
//   best -- so winners, distances and tie-breaks are those of the exact boxes.
__global__ void nn_frame_kernel(const float4 *__restrict__ bmin, const float4 *__restrict__ bmax, uint32_t *__restrict__ info, float wide_margin)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float lo[3] = { bmin[0].x, bmin[0].y, bmin[0].z }, hi[3] = { bmax[0].x, bmax[0].y, bmax[0].z };
    for (int a = 0; a < 3; ++a) {
        float sc = (hi[a] - lo[a]) / 65535.0f * 1.000001f;
        if (!(sc > 1e-30f)) sc = 1e-30f;
        info[2 + a] = __float_as_uint(lo[a]);
        info[5 + a] = __float_as_uint(sc);
    }
    // the wide records use ONE scale for the three axes (the longest edge of the root box): a box test then is integer differences in
    // that unit and one sum of squares, no per-axis dequantisation (nn_tree_wide_kernel)
    // The frame reaches `wide_margin` (the scene's acceptance radius) beyond the root box on every side: a query is clamped into the frame
    // before its box tests, which would make every box look nearer than it is for a query outside -- hypotheses that start centimetres in
    // front of the surface are outside the ROOT box along z -- and a query beyond the margin has no neighbour within the radius anyway.
    if (!(wide_margin >= 0.0f && wide_margin < 1e30f)) wide_margin = 0.0f;
    float edge = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]) + 2.0f * wide_margin;
    float wsc = edge / 65535.0f * 1.000001f;
    if (!(wsc > 1e-30f)) wsc = 1e-30f;
    for (int a = 0; a < 3; ++a) info[16 + a] = __float_as_uint(lo[a] - wide_margin);      // ([12] is the scene fingerprint's word)
    info[19] = __float_as_uint(wsc);
    // The integer box test of the walk relies on a unit being at least one ulp of the largest coordinate (then a stored corner sits within
    // half a unit of its real-number position and the query's interval, a unit wider on both sides, covers it): a scene far from the origin
    // of its coordinate system with a tiny extent does not qualify and keeps the binary walk (info[20] = 0).
    float cmax = 0.0f;
    for (int a = 0; a < 3; ++a) cmax = fmaxf(cmax, fmaxf(fabsf(lo[a] - wide_margin), fabsf(hi[a] + wide_margin)));
    info[20] = (cmax * 1.1920929e-7f <= wsc && cmax < 1e30f) ? 1u : 0u;
    info[1] = 1u;
}
__global__ __launch_bounds__(256) void nn_records32_kernel(const int4 *__restrict__ topo, const float4 *__restrict__ bmin,
                                                           const float4 *__restrict__ bmax, uint32_t n_nodes, uint4 *__restrict__ rec32,
                                                           uint2 *__restrict__ desc, uint32_t *__restrict__ info)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    const int4 t = topo[i];
    uint32_t w[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    bool ok = true;
    if (t.z < 0) { w[0] = (uint32_t)t.x; w[1] = (uint32_t)t.y | (3u << 30); ok = ((uint32_t)t.y < (1u << 30)); }
    else {
        const uint32_t dim = (uint32_t)t.w >> 30;
        w[0] = (uint32_t)t.x; w[1] = (uint32_t)t.y | (dim << 30);
        ok = (t.z == t.y + 1) && ((uint32_t)t.y < (1u << 30)) && dim < 3 && (uint32_t)t.z < n_nodes;
        // the 8-byte descent skips a far side on (q - split)^2 alone, which needs left_max <= split <= right_min
        // (pcd_scene.cpp:135 builds trees that way; other trees take the exact 64-byte records)
        if (ok) {
            const float lmax = (dim == 0) ? bmax[t.y].x : ((dim == 1) ? bmax[t.y].y : bmax[t.y].z);
            const float rmin = (dim == 0) ? bmin[t.z].x : ((dim == 1) ? bmin[t.z].y : bmin[t.z].z);
            const float split = __int_as_float(t.x);
            if (!(lmax <= split && split <= rmin)) ok = false;
        }
        uint32_t q[12];
        for (int c = 0; c < 2; ++c) {
            const int ch = ok ? (c ? t.z : t.y) : 0;
            const float lo[3] = { bmin[ch].x, bmin[ch].y, bmin[ch].z }, hi[3] = { bmax[ch].x, bmax[ch].y, bmax[ch].z };
            for (int a = 0; a < 3; ++a) {
                const float qmin = __uint_as_float(info[2 + a]), qs = __uint_as_float(info[5 + a]);
                float fl = floorf((lo[a] - qmin) / qs) - 1.0f;
                uint32_t ql = fl > 0.0f ? (fl < 65535.0f ? (uint32_t)fl : 65535u) : 0u;
                while (ql > 0 && !(nn_deq(ql, qmin, qs) <= lo[a])) --ql;
                if (!(nn_deq(ql, qmin, qs) <= lo[a])) ok = false;
                float fh = ceilf((hi[a] - qmin) / qs) + 1.0f;
                uint32_t qh = fh > 0.0f ? (fh < 65535.0f ? (uint32_t)fh : 65535u) : 0u;
                while (qh < 65535u && !(nn_deq(qh, qmin, qs) >= hi[a])) ++qh;
                if (!(nn_deq(qh, qmin, qs) >= hi[a])) ok = false;
                q[c * 6 + a] = ql; q[c * 6 + 3 + a] = qh;
            }
        }
        for (int k = 0; k < 6; ++k) w[2 + k] = q[2 * k] | (q[2 * k + 1] << 16);
    }
    rec32[(size_t)i * 2] = make_uint4(w[0], w[1], w[2], w[3]);
    rec32[(size_t)i * 2 + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    desc[i] = make_uint2(w[0], w[1]);
    if (!ok) info[1] = 0u;                                       // any node that does not fit: the 64-byte records are used instead
}

// ---- wide records (query_nn_wide) ---------------------------------------------------------------------------------------------
// Box of a binary node as 6 uint16 in the root-box frame, rounded outwards and checked with the dequantisation the query uses.
__device__ __forceinline__ bool wide_quant_box(const float4 lo4, const float4 hi4, const uint32_t *__restrict__ info, uint32_t (&u)[3])
{
    const float lo[3] = { lo4.x, lo4.y, lo4.z }, hi[3] = { hi4.x, hi4.y, hi4.z };
    uint32_t q[6];
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
        const float qmin = __uint_as_float(info[16 + a]), qs = __uint_as_float(info[19]);     // the wide records' frame: one scale for all axes
        const float fl = floorf((lo[a] - qmin) / qs) - 1.0f;
        uint32_t ql = fl > 0.0f ? (fl < 65535.0f ? (uint32_t)fl : 65535u) : 0u;
        while (ql > 0 && !(nn_deq_fma(ql, qmin, qs) <= lo[a])) --ql;
        if (!(nn_deq_fma(ql, qmin, qs) <= lo[a])) ok = false;
        const float fh = ceilf((hi[a] - qmin) / qs) + 1.0f;
        uint32_t qh = fh > 0.0f ? (fh < 65535.0f ? (uint32_t)fh : 65535u) : 0u;
        while (qh < 65535u && !(nn_deq_fma(qh, qmin, qs) >= hi[a])) ++qh;
        if (!(nn_deq_fma(qh, qmin, qs) >= hi[a])) ok = false;
        q[a] = ql; q[3 + a] = qh;
    }
    u[0] = q[0] | (q[1] << 16); u[1] = q[2] | (q[3] << 16); u[2] = q[4] | (q[5] << 16);
    return ok;
}
// The wide nodes are built level by level, two launches per level over all wide nodes of the level (round 5: ONE workgroup looping over the
// levels took 1.1 ms for a frame-filling scene's 61 k binary nodes; the scene is prepared once per frame when the scene changes per frame):
// (A) `open`: every wide node opens its binary root into a frontier of up to eight descendants -- repeatedly the internal frontier node
// with the longest box diagonal -- and counts the internal ones (also summed per chunk of 1 024 wide nodes), (B + C) `number`: an exclusive
// scan of those counts numbers the next level's wide nodes (deterministic: a wide node's index does not depend on timing) and the records'
// references are rewritten; then (D) `layout`, once the last level is through.  `wq[k]` = binary root of wide node k, `cnt[k]` = its count.
// The level's range lives in a control record double-buffered by level parity ({begin, end, bad}; level 0 is {0, 1, frame unusable}); a tree
// whose links are out of order (child index <= parent index) or that does not fit sets `bad`, and info[8] stays 0.  The host launches eight
// levels and the layout kernel before its one synchronisation; info[8] = 2 tells it that the tree is deeper (it launches eight more).
constexpr uint32_t kWideChunk = 1024;
struct WideLevel { uint32_t begin, end, bad, pad; };
__device__ __forceinline__ WideLevel wide_level_state(const WideLevel *__restrict__ ctrl, uint32_t level, const uint32_t *__restrict__ info)
{
    if (level == 0u) return WideLevel{ 0u, 1u, (info[1] == 1u && info[20] == 1u) ? 0u : 1u, 0u };
    return ctrl[level & 1u];
}
__global__ __launch_bounds__(256) void nn_wide_open_kernel(const int4 *__restrict__ topo, const float4 *__restrict__ bmin, const float4 *__restrict__ bmax, uint32_t n_nodes,
                                                           uint4 *__restrict__ wide, const uint32_t *__restrict__ wq, uint32_t *__restrict__ cnt, const WideLevel *__restrict__ ctrl,
                                                           uint32_t *__restrict__ bad_flag, uint32_t *__restrict__ chunk_sum /* this level's */, uint32_t level, uint32_t n_points, const uint32_t *__restrict__ info)
{
    const WideLevel st = wide_level_state(ctrl, level, info);
    if (st.bad || st.begin >= st.end) return;
    // EIGHT LANES per wide node, one per slot of its frontier: (node, size) live in registers, the pick is a reduction over the eight, and the
    // boxes of the eight slots are quantised side by side (one thread per wide node kept the frontier in scratch memory -- a dynamically
    // indexed array -- and worked through the slots one after the other: 32 us for the single wide node of level 0)
    const uint32_t lane = threadIdx.x & 63u, g = lane & 7u, gbase = lane & ~7u;
    bool bad = false;
    for (uint32_t k0 = st.begin + blockIdx.x * 32u; k0 < st.end; k0 += gridDim.x * 32u) {       // (workgroup-uniform trip count)
        const uint32_t k = k0 + threadIdx.x / 8u;
        const bool live = k < st.end;
        uint32_t node = 0u; float size = -1.0f; uint32_t nf = live ? 1u : 8u;
        if (live && g == 0u) {
            node = level == 0u ? 0u : wq[k];
            size = (topo[node].z < 0) ? -1.0f : FLT_MAX;           // the root of a wide node is always opened (unless the whole tree is one leaf)
        }
        for (;;) {
            // the first of the largest sizes among the slots in use ("fsz[i] > fsz[pick]": strictly larger replaces)
            float best = (g < nf && size >= 0.0f) ? size : -1.0f; uint32_t arg = g;
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                const float b2 = __shfl_xor(best, off); const uint32_t a2 = __shfl_xor(arg, off);
                if (b2 > best || (b2 == best && a2 < arg)) { best = b2; arg = a2; }
            }
            const bool going = nf < 8u && best >= 0.0f;
            if (!__any(going)) break;
            int4 t = make_int4(0, 0, 0, 0); float s1 = -1.0f, s2 = -1.0f; uint32_t okl = 1u;
            if (going && g == arg) {
                t = topo[node]; s1 = bmin[node].w; s2 = bmax[node].w;          // the children's sizes (nn_accel_kernel): -1 for a leaf (never opened)
                okl = ((uint32_t)t.y < n_nodes && (uint32_t)t.z < n_nodes && (uint32_t)t.y > node && (uint32_t)t.z > node) ? 1u : 0u;
                if (!okl) { bad = true; size = -1.0f; } else { node = (uint32_t)t.y; size = s1; }
            }
            const uint32_t src = gbase + arg;
            const uint32_t ok_b = __shfl(okl, src), c2 = (uint32_t)__shfl(t.z, src); const float sz2 = __shfl(s2, src);
            if (going && ok_b) { if (g == nf) { node = c2; size = sz2; } ++nf; }
        }
        uint32_t internal = 0u;
        if (live) {                                                   // slot g = {box, reference}; internal children carry their BINARY id until they are numbered
            uint32_t u[3] = { 0u, 0u, 0u }, ref = kWideEmpty;
            if (g < nf) {
                const int4 t = topo[node];
                if (!wide_quant_box(bmin[node], bmax[node], info, u)) bad = true;
                if (t.z < 0) {
                    const int lo = t.x, hi = t.y;
                    if (lo >= 0 && hi > lo && (uint32_t)(hi - lo) <= kWideMaxLeafPoints && (uint32_t)lo <= kWideFirstMask && (uint32_t)hi <= n_points)
                        ref = kWideLeaf | ((uint32_t)(hi - lo) << 27) | (uint32_t)lo;
                    else if (hi != lo) bad = true;                   // (an empty leaf stays an empty slot)
                } else { ref = node; internal = 1u; if (node >= 0x7fffffffu) bad = true; }
            }
            wide[(size_t)k * 8 + g] = make_uint4(u[0], u[1], u[2], ref);
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) internal += __shfl_xor(internal, off);
        if (live && g == 0u) cnt[k] = internal;
        // one addition per wavefront and chunk (eight consecutive wide nodes: one chunk, at a boundary two)
        uint32_t chunk = (k - st.begin) / kWideChunk, left = (live && g == 0u) ? internal : 0u;
        for (;;) {
            const unsigned long long todo = __ballot(left != 0u);
            if (!todo) break;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)chunk, (int)__builtin_ctzll(todo));
            uint32_t mine = (left != 0u && chunk == c) ? left : 0u;
            if (chunk == c) left = 0u;
            for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
            if (lane == 0u) atomicAdd(&chunk_sum[c], mine);
        }
    }
    if (bad) *bad_flag = 1u;
}
__global__ __launch_bounds__(1024) void nn_wide_number_kernel(uint4 *__restrict__ wide, uint32_t cap_wide, uint32_t *__restrict__ wq, const uint32_t *__restrict__ cnt,
                                                              const WideLevel *__restrict__ ctrl, WideLevel *__restrict__ ctrl_next, const uint32_t *__restrict__ bad_flag,
                                                              const uint32_t *__restrict__ sums, uint32_t *__restrict__ sums_next, uint32_t level, const uint32_t *__restrict__ info)
{
    __shared__ uint32_t s_a[16], s_b[16];
    const WideLevel st = wide_level_state(ctrl, level, info);
    const bool bad = st.bad || *bad_flag != 0u;
    if (bad || st.begin >= st.end) { if (blockIdx.x == 0 && threadIdx.x == 0) *ctrl_next = WideLevel{ st.begin, st.end, bad ? 1u : 0u, 0u }; return; }
    const uint32_t span = st.end - st.begin, n_chunk = (span + kWideChunk - 1u) / kWideChunk, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunk; chunk += gridDim.x) {
        uint32_t before = 0u, all = 0u;
        for (uint32_t i = threadIdx.x; i < n_chunk; i += 1024) { const uint32_t c = sums[i]; all += c; if (i < chunk) before += c; }
        for (int off = 32; off > 0; off >>= 1) { before += __shfl_xor(before, off); all += __shfl_xor(all, off); }
        __syncthreads();
        if (lane == 0u) { s_a[wave] = before; s_b[wave] = all; }
        __syncthreads();
        before = 0u; all = 0u;
        for (uint32_t w = 0; w < 16; ++w) { before += s_a[w]; all += s_b[w]; }
        __syncthreads();
        const uint32_t k = st.begin + chunk * kWideChunk + threadIdx.x;
        const uint32_t mine = k < st.end ? cnt[k] : 0u;
        uint32_t incl = mine;
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        if (lane == 63u) s_a[wave] = incl;
        __syncthreads();
        uint32_t excl = incl - mine;
        for (uint32_t w = 0; w < wave; ++w) excl += s_a[w];
        const bool fits = (size_t)st.end + all <= (size_t)cap_wide;
        if (k < st.end && fits) {                                    // number the internal children: wide node k's go to end + (counts before k) ...
            uint32_t next = st.end + before + excl;
            for (int c = 0; c < 8; ++c) {
                uint4 r = wide[(size_t)k * 8 + c];
                if (r.w != kWideEmpty && !(r.w & kWideLeaf)) { wq[next] = r.w; r.w = next; ++next; wide[(size_t)k * 8 + c] = r; }
            }
        }
        if (chunk == 0u) {
            if (threadIdx.x == 0) *ctrl_next = WideLevel{ st.end, fits ? st.end + all : st.end, fits ? 0u : 1u, 0u };
            for (uint32_t i = threadIdx.x; i < (all + kWideChunk - 1u) / kWideChunk; i += 1024) sums_next[i] = 0u;       // the next level's chunk sums
        }
    }
}
// (D) the layout the task walk reads (kWidePaired): a wide node is two 64-byte halves, one per lane of a task; a half holds its four
// slots as two PAIRS -- per pair six words {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z}, each word = first slot | second slot << 16, so that
// one packed 16-bit instruction works on both boxes -- followed by the four references.
__global__ __launch_bounds__(256) void nn_wide_layout_kernel(uint4 *__restrict__ wide, const WideLevel *__restrict__ ctrl, uint32_t levels, uint32_t *__restrict__ info)
{
    const WideLevel st = wide_level_state(ctrl, levels, info);     // (levels >= 1: the record the last `number` launch wrote)
    const bool done = st.begin >= st.end, ok = !st.bad && done;
    if (blockIdx.x == 0 && threadIdx.x == 0) { info[8] = ok ? 1u : ((!st.bad && !done) ? 2u : 0u); info[9] = st.end; }
    if (!ok) return;
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < st.end; k += gridDim.x * 256) {
        uint4 sl[8];
        for (int c = 0; c < 8; ++c) sl[c] = wide[(size_t)k * 8 + c];
        for (int h = 0; h < 2; ++h) {
            uint32_t w[16];
            for (int pr = 0; pr < 2; ++pr) {
                const uint4 A = sl[4 * h + 2 * pr], B = sl[4 * h + 2 * pr + 1];
                // a slot: x = lo.x | lo.y << 16, y = lo.z | hi.x << 16, z = hi.y | hi.z << 16
                const uint32_t a6[6] = { A.x & 0xffffu, A.x >> 16, A.y & 0xffffu, A.y >> 16, A.z & 0xffffu, A.z >> 16 };
                const uint32_t b6[6] = { B.x & 0xffffu, B.x >> 16, B.y & 0xffffu, B.y >> 16, B.z & 0xffffu, B.z >> 16 };
                for (int f = 0; f < 6; ++f) w[6 * pr + f] = a6[f] | (b6[f] << 16);
            }
            for (int c = 0; c < 4; ++c) w[12 + c] = sl[4 * h + c].w;
            for (int v = 0; v < 4; ++v) wide[(size_t)k * 8 + 4 * h + v] = make_uint4(w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]);
        }
    }
}

// wide levels [first, first + count), then the layout; scratch = nn_wide_scratch_words(n_nodes) words
hipError_t launch_nn_wide_levels(const int4 *topo, const float4 *bmin, const float4 *bmax, uint32_t n_nodes, uint32_t n_points, uint4 *wide, uint32_t *scratch,
                                 uint32_t *info, uint32_t first, uint32_t count, hipStream_t s)
{
    const uint32_t cap = (uint32_t)nn_wide_capacity(n_nodes), chunks = (cap / kWideChunk + 3u) & ~1u;
    uint32_t *wq = scratch, *cnt = scratch + cap, *sums = scratch + ((2u * cap + 3u) & ~3u), *bad_flag = sums + 2u * chunks;      // (the control records 16-byte aligned: chunks is even)
    WideLevel *ctrl = reinterpret_cast<WideLevel *>(bad_flag + 4);
    if (first == 0u) {
        const hipError_t e = hipMemsetAsync(sums, 0, (2u * chunks + 4u + 8u) * sizeof(uint32_t), s);      // chunk sums, flag, control records
        if (e != hipSuccess) return e;
    }
    const uint32_t open_groups = std::min<uint32_t>((cap + 31u) / 32u, 2048u), number_groups = std::min<uint32_t>(chunks, 128u);
    for (uint32_t level = first; level < first + count; ++level) {
        // (wide level l holds at most 8^l nodes: the first levels' launches bring one workgroup, not 2 048 that find nothing to do)
        const uint32_t level_nodes = level < 8u ? std::min<uint32_t>(1u << (3u * level), cap) : cap;
        const uint32_t og = std::min<uint32_t>((level_nodes + 31u) / 32u, open_groups), ng = std::min<uint32_t>(level_nodes / kWideChunk + 1u, number_groups);
        uint32_t *mine = sums + (size_t)(level & 1u) * chunks, *other = sums + (size_t)((level + 1u) & 1u) * chunks;
        hipLaunchKernelGGL(nn_wide_open_kernel, dim3(og), dim3(256), 0, s, topo, bmin, bmax, n_nodes, wide, wq, cnt, ctrl, bad_flag, mine, level, n_points, info);
        hipLaunchKernelGGL(nn_wide_number_kernel, dim3(ng), dim3(1024), 0, s, wide, cap, wq, cnt, ctrl, ctrl + ((level + 1u) & 1u), bad_flag, mine, other, level, info);
    }
    hipLaunchKernelGGL(nn_wide_layout_kernel, dim3(open_groups), dim3(256), 0, s, wide, ctrl, first + count, info);
    return hipGetLastError();
}

hipError_t launch_build_nn_accel(const pr_kdnode *nodes, uint32_t n_nodes, const pr_vec3 *pcd, uint32_t n_points,
                                 int4 *topo, float4 *bmin, float4 *bmax, float4 *pts, float4 *rec, uint4 *rec32, uint2 *desc, uint32_t *info, hipStream_t s,
                                 float wide_margin)
{
    const uint32_t m = (n_nodes > n_points) ? n_nodes : n_points;
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(nn_accel_kernel, dim3((m + 255) / 256), dim3(256), 0, s, nodes, n_nodes, pcd, n_points, topo, bmin, bmax, pts);
    hipError_t e = hipMemsetAsync(info, 0, 24 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nn_records_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, topo, bmin, bmax, n_nodes, rec, info);
    hipLaunchKernelGGL(nn_frame_kernel, dim3(1), dim3(64), 0, s, bmin, bmax, info, wide_margin);
    hipLaunchKernelGGL(nn_records32_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, topo, bmin, bmax, n_nodes, rec32, desc, info);
    return hipGetLastError();
}

}  // namespace prk
