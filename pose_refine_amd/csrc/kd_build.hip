// kd_build.hip -- KDTree_cpu::build_tree (pcd_scene.cpp:45-184) on the device, level by level, bit-identical nodes and permutation
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#include "pr_launch.h"

namespace prk {

// ================================================================================================
//  kd-tree build on the device -- the reference's level-order build (pcd_scene.cpp:45-184), a whole LEVEL at a time.
//
//  The reference walks a node's points in sequence: box (first occurrence wins among equal extremes), widest axis, cut at the box's
//  midpoint, then a stable two-ended partition (the left part keeps its order, the right part is filled from the end, the k-th point ON
//  the cut goes left iff k is even), split value = mean of the largest left and the smallest right coordinate.  Every one of these is a
//  function of a point's RANK among the points of its node, so a level is four launches over ALL positions / all nodes of the level,
//  whatever their number (round 5: the first build gave each node ONE workgroup -- the root's 300 k points took 1 200 trips of a single
//  workgroup -- and the host read the level's size back before every level: 5.9 ms for a frame-filling scene, 1.2 ms for the bench's):
//    count    per tile of positions: how many points since the last node start are below / on their node's cut
//    scatter  per tile: carry from the tiles before it (look-back over the tile records), ranks by a segmented scan, every point to its
//             place in the other index buffer; the children's boxes and the parent's two split candidates as 64-bit minima of
//             (order-preserving value bits, position) -- the position breaks ties as the sequential loop does and names the point whose
//             coordinate is stored, so the sign of a zero survives.  Keys are folded per lane, per wavefront where it holds one node, then
//             in an LDS table per tile; only the table goes to memory (atomics on ONE address complete one after the other, about 100 ns
//             each: 1 200 wavefronts lowering the root's children's keys directly took 126 us)
//    finish   per node of the level: split value, the two children's records, how many children of each 1 024-chunk split again
//    plan     per child: its child slots (chunk base + rank: children are appended pairwise in node order), box, axis, cut
//  The host launches twelve levels, reads 32 bytes back, and goes on four levels at a time until the build reports it is done; launches
//  past the last level return at once.  The control record is double-buffered by level parity (plan writes the next level's).
// ================================================================================================
constexpr uint32_t kKdPer = PR_KD_PER;                              // consecutive positions per thread of the count / scatter passes
constexpr uint32_t kKdTile = 256u * kKdPer;                         // positions per workgroup
constexpr uint32_t kKdTable = 256;                                  // nodes of the level a tile's LDS key table holds (beyond: straight to memory)
constexpr uint32_t kKdChunk = 1024;                                 // children per plan workgroup
constexpr unsigned long long kKdNoKey = ~0ull;
static_assert(kKdPer == 4 || kKdPer == 8, "a thread loads its positions as int4");

__device__ __forceinline__ float axis_coord(const pr_vec3 &p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }
// float -> unsigned, order-preserving; -0 and +0 give the same key (they compare equal in the reference's "if (v > best)")
__device__ __forceinline__ uint32_t kd_ord(float v) { const uint32_t u = __float_as_uint(v + 0.0f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
// (a NaN never replaces an extreme in the sequential loops: no key)
__device__ __forceinline__ unsigned long long kd_key_min(float v, uint32_t pos) { return v == v ? (((unsigned long long)kd_ord(v) << 32) | pos) : ~0ull; }     // smallest value, then first position
__device__ __forceinline__ unsigned long long kd_key_max(float v, uint32_t pos) { return v == v ? (((unsigned long long)(~kd_ord(v)) << 32) | pos) : ~0ull; }  // largest value, then first position
__device__ __forceinline__ unsigned long long min_u64(unsigned long long a, unsigned long long b) { return b < a ? b : a; }
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k)
{
    for (int off = 32; off > 0; off >>= 1) k = min_u64(k, (unsigned long long)__shfl_xor((long long)k, off));
    return k;
}
__device__ __forceinline__ void kd_blank(pr_kdnode &b) { b.parent = b.child1 = b.child2 = -1; b.split_v = 0.0f; b.split_dim = 0; b.left = 0; b.right = 0; for (int k = 0; k < 6; ++k) b.bbox[k] = 0.0f; }

// identity permutation, every position in the root; the root's box as one partial key set per workgroup (no atomics: 1 200 on one address)
__global__ __launch_bounds__(256) void kd_init_kernel(int *__restrict__ idx, int *__restrict__ owner, uint32_t n, const pr_vec3 *__restrict__ pcd,
                                                      unsigned long long *__restrict__ root_part)
{
    __shared__ unsigned long long s_k[4][6];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long key[6];
    for (int a = 0; a < 6; ++a) key[a] = kKdNoKey;
    if (i < n) {
        idx[i] = (int)i; owner[i] = 0;
        const pr_vec3 p = pcd[i];
        for (int a = 0; a < 3; ++a) { const float c = axis_coord(p, a); key[2 * a] = kd_key_min(c, i); key[2 * a + 1] = kd_key_max(c, i); }
    }
    for (int a = 0; a < 6; ++a) { const unsigned long long k = wave_min_u64(key[a]); if ((threadIdx.x & 63u) == 0u) s_k[threadIdx.x >> 6][a] = k; }
    __syncthreads();
    if (threadIdx.x < 6) root_part[(size_t)blockIdx.x * 6 + threadIdx.x] = min_u64(min_u64(s_k[0][threadIdx.x], s_k[1][threadIdx.x]), min_u64(s_k[2][threadIdx.x], s_k[3][threadIdx.x]));
}
__global__ __launch_bounds__(1024) void kd_root_kernel(pr_kdnode *__restrict__ nodes, uint32_t n, int max_leaf, const unsigned long long *__restrict__ root_part, uint32_t n_part,
                                                       unsigned long long *__restrict__ bbkeys, uint32_t *__restrict__ chunk_cnt)
{
    __shared__ unsigned long long s_k[16][6];
    unsigned long long key[6];
    for (int a = 0; a < 6; ++a) key[a] = kKdNoKey;
    for (uint32_t i = threadIdx.x; i < n_part; i += 1024) for (int a = 0; a < 6; ++a) key[a] = min_u64(key[a], root_part[(size_t)i * 6 + a]);
    for (int a = 0; a < 6; ++a) { const unsigned long long k = wave_min_u64(key[a]); if ((threadIdx.x & 63u) == 0u) s_k[threadIdx.x >> 6][a] = k; }
    __syncthreads();
    if (threadIdx.x < 6) { unsigned long long k = kKdNoKey; for (int w = 0; w < 16; ++w) k = min_u64(k, s_k[w][threadIdx.x]); bbkeys[threadIdx.x] = k; }
    if (threadIdx.x == 0) { pr_kdnode b; kd_blank(b); b.right = (int)n; nodes[0] = b; chunk_cnt[0] = (int)n > max_leaf ? 1u : 0u; }
}

// what a position of the permutation is to its node in this level
// (arrays of scalars, filled through plain assignments: as an array of records copied with a conditional expression the whole set lived in scratch memory, 196 bytes per lane)
struct KdTileIn { int id[kKdPer], o[kKdPer], left[kKdPer], right[kKdPer], child[kKdPer]; bool split[kKdPer], head[kKdPer], lt[kKdPer], tie[kKdPer]; float v[kKdPer]; };
__device__ __forceinline__ void kd_classify(const KdLevelNode *__restrict__ lv, const pr_vec3 *__restrict__ pcd, const int *__restrict__ idx,
                                            const int *__restrict__ owner, uint32_t n, uint32_t k0, KdTileIn &t)
{
    if (k0 + kKdPer - 1u < n) {
#pragma unroll
        for (uint32_t q = 0; q < kKdPer; q += 4) {
            const int4 a = *reinterpret_cast<const int4 *>(idx + k0 + q), b = *reinterpret_cast<const int4 *>(owner + k0 + q);
            t.id[q] = a.x; t.id[q + 1] = a.y; t.id[q + 2] = a.z; t.id[q + 3] = a.w; t.o[q] = b.x; t.o[q + 1] = b.y; t.o[q + 2] = b.z; t.o[q + 3] = b.w;
        }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < kKdPer; ++j) { const bool in = k0 + j < n; t.id[j] = in ? idx[k0 + j] : 0; t.o[j] = in ? owner[k0 + j] : -1; }
    }
    int axis = 0; float cut = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < kKdPer; ++j) {
        const bool owned = t.o[j] >= 0;
        if (j == 0 || t.o[j] != t.o[j - 1]) {                         // (consecutive positions of one node: one record)
            int4 r = make_int4(0, 0, -1, 0); float c = 0.0f;
            if (owned) { r = *reinterpret_cast<const int4 *>(&lv[t.o[j]]); c = lv[t.o[j]].cut; }       // left, right, child, axis | cut
            t.left[j] = r.x; t.right[j] = r.y; t.child[j] = r.z; axis = r.w; cut = c;
        } else { t.left[j] = t.left[j - 1]; t.right[j] = t.right[j - 1]; t.child[j] = t.child[j - 1]; }
        t.split[j] = owned && t.child[j] >= 0;
        const float v = t.split[j] ? axis_coord(pcd[t.id[j]], axis) : 0.0f;
        t.v[j] = v;
        t.lt[j] = t.split[j] && v < cut; t.tie[j] = t.split[j] && v == cut;
        t.head[j] = t.split[j] && (int)(k0 + j) == t.left[j];
    }
}
// segmented sums over the tile, in position order: (h, a) = (a node starts in the span, packed counts since the last start -- below the cut in
// the low, on the cut in the high half).  combine(left, right) = (hl | hr, hr ? ar : al + ar).
__device__ __forceinline__ void kd_seg_wave_scan(uint32_t &h, uint32_t &a)
{
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t h2 = __shfl_up(h, off), a2 = __shfl_up(a, off);
        if (lane >= off) { if (!h) a += a2; h |= h2; }
    }
}

// ---- count: one record per tile = packed counts behind the tile's last node start | bit 31: a node starts in the tile
__global__ __launch_bounds__(256) void kd_tile_count_kernel(const KdCtrl *__restrict__ ctrl, const KdLevelNode *__restrict__ lv, const pr_vec3 *__restrict__ pcd,
                                                            const int *__restrict__ idx, const int *__restrict__ owner, uint32_t n, uint32_t *__restrict__ tile_agg,
                                                            unsigned long long *__restrict__ bbkeys, unsigned long long *__restrict__ lrkeys, uint32_t *__restrict__ chunk_cnt)
{
    __shared__ uint32_t s_h[4], s_a[4];
    if (ctrl->done) return;
    {   // what this level's scatter and finish passes accumulate into: six keys per child, two per node of the level, one count per chunk of children
        const uint32_t n_level = ctrl->hi - ctrl->lo, n_child = ctrl->next - ctrl->hi, gt = blockIdx.x * 256 + threadIdx.x, gn = gridDim.x * 256;
        for (uint32_t i = gt; i < n_child * 6u; i += gn) bbkeys[i] = kKdNoKey;
        for (uint32_t i = gt; i < n_level * 2u; i += gn) lrkeys[i] = kKdNoKey;
        for (uint32_t i = gt; i < (n_child + kKdChunk - 1u) / kKdChunk; i += gn) chunk_cnt[i] = 0u;
    }
    KdTileIn t;
    kd_classify(lv, pcd, idx, owner, n, blockIdx.x * kKdTile + kKdPer * threadIdx.x, t);
    uint32_t h = 0u, a = 0u;
#pragma unroll
    for (uint32_t j = 0; j < kKdPer; ++j) { if (t.head[j]) { h = 1u; a = 0u; } a += (t.lt[j] ? 1u : 0u) + (t.tie[j] ? 0x10000u : 0u); }
    kd_seg_wave_scan(h, a);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 63u) { s_h[wave] = h; s_a[wave] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t th = s_h[0], ta = s_a[0];
        for (uint32_t w = 1; w < 4; ++w) { ta = s_h[w] ? s_a[w] : ta + s_a[w]; th |= s_h[w]; }
        tile_agg[blockIdx.x] = ta | (th << 31);                     // (at most 2 048 of either kind: 12 bits each)
    }
}

// ---- scatter
// a lane's keys of ONE node of the level: the two split candidates and the two children's boxes
struct KdKeys { unsigned long long lr[2], bb[2][6]; };
__device__ __forceinline__ void kd_keys_clear(KdKeys &k) { k.lr[0] = k.lr[1] = kKdNoKey; for (int c = 0; c < 2; ++c) for (int q = 0; q < 6; ++q) k.bb[c][q] = kKdNoKey; }
// lower node `o`'s slots by `k`: in the tile's table if the node is one of the kKdTable it holds, else in memory
__device__ __forceinline__ void kd_keys_flush(const KdKeys &k, int o, int o_base, uint32_t child0, unsigned long long (*table)[14],
                                              unsigned long long *__restrict__ bbkeys, unsigned long long *__restrict__ lrkeys)
{
    const uint32_t rel = (uint32_t)(o - o_base);
    if (rel < kKdTable) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (k.lr[c] != kKdNoKey) atomicMin(&table[rel][c], k.lr[c]);
#pragma unroll
            for (int q = 0; q < 6; ++q) if (k.bb[c][q] != kKdNoKey) atomicMin(&table[rel][2 + 6 * c + q], k.bb[c][q]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (k.lr[c] != kKdNoKey) atomicMin(&lrkeys[2u * (uint32_t)o + c], k.lr[c]);
#pragma unroll
            for (int q = 0; q < 6; ++q) if (k.bb[c][q] != kKdNoKey) atomicMin(&bbkeys[6u * (child0 + c) + q], k.bb[c][q]);
        }
    }
}
__global__ __launch_bounds__(256) void kd_tile_scatter_kernel(const KdCtrl *__restrict__ ctrl, const KdLevelNode *__restrict__ lv, const pr_vec3 *__restrict__ pcd,
                                                              const int *__restrict__ idx, const int *__restrict__ owner, uint32_t n, const uint32_t *__restrict__ tile_agg,
                                                              int *__restrict__ idx_out, int *__restrict__ owner_out, uint32_t *__restrict__ left_total,
                                                              unsigned long long *__restrict__ bbkeys, unsigned long long *__restrict__ lrkeys)
{
    __shared__ uint32_t s_h[4], s_a[4];
    __shared__ int s_last[4], s_base[4];
    __shared__ uint32_t s_lt[4], s_tie[4];
    __shared__ unsigned long long s_table[kKdTable][14];              // per node of the level in this tile: lr[2], child 0's six, child 1's six
    if (ctrl->done) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, tile = blockIdx.x;
    const uint32_t first_child = ctrl->hi;                          // node id of the next level's first node
    for (uint32_t i = threadIdx.x; i < kKdTable * 14u; i += 256) (&s_table[0][0])[i] = kKdNoKey;
    // 1. carry: points below / on the cut between the last node start before this tile and the tile
    uint32_t carry_lt = 0u, carry_tie = 0u;
    {
        int last = -1;
        for (uint32_t i = threadIdx.x; i < tile; i += 256) if (tile_agg[i] >> 31) last = (int)i;
        for (int off = 32; off > 0; off >>= 1) { const int o2 = __shfl_xor(last, off); last = o2 > last ? o2 : last; }
        if (lane == 0u) s_last[wave] = last;
        __syncthreads();
        last = s_last[0];
        for (uint32_t w = 1; w < 4; ++w) last = s_last[w] > last ? s_last[w] : last;
        uint32_t lt = 0u, tie = 0u;
        for (uint32_t i = (last < 0 ? 0u : (uint32_t)last) + threadIdx.x; i < tile; i += 256) { const uint32_t r = tile_agg[i]; lt += r & 0xffffu; tie += (r >> 16) & 0x7fffu; }
        for (int off = 32; off > 0; off >>= 1) { lt += __shfl_xor(lt, off); tie += __shfl_xor(tie, off); }
        if (lane == 0u) { s_lt[wave] = lt; s_tie[wave] = tie; }
        __syncthreads();
        carry_lt = s_lt[0] + s_lt[1] + s_lt[2] + s_lt[3]; carry_tie = s_tie[0] + s_tie[1] + s_tie[2] + s_tie[3];
    }
    // 2. ranks inside the tile
    const uint32_t k0 = tile * kKdTile + kKdPer * threadIdx.x;
    KdTileIn t;
    kd_classify(lv, pcd, idx, owner, n, k0, t);
    uint32_t h = 0u, a = 0u;
    int o_min = 0x7fffffff;                                         // the level's nodes are numbered in position order: the first splitting one of the tile has the smallest number
#pragma unroll
    for (uint32_t j = 0; j < kKdPer; ++j) {
        if (t.head[j]) { h = 1u; a = 0u; }
        a += (t.lt[j] ? 1u : 0u) + (t.tie[j] ? 0x10000u : 0u);
        if (t.split[j] && t.o[j] < o_min) o_min = t.o[j];
    }
    kd_seg_wave_scan(h, a);
    for (int off = 32; off > 0; off >>= 1) { const int o2 = __shfl_xor(o_min, off); o_min = o2 < o_min ? o2 : o_min; }
    if (lane == 63u) { s_h[wave] = h; s_a[wave] = a; }
    if (lane == 0u) s_base[wave] = o_min;
    uint32_t xh = __shfl_up(h, 1), xa = __shfl_up(a, 1);           // exclusive: what the lanes before this one hold
    if (lane == 0u) { xh = 0u; xa = 0u; }
    __syncthreads();                                                // (also: the table is cleared)
    {
        uint32_t ph = 0u, pa = 0u;                                  // the wavefronts before this one, folded left to right
        for (uint32_t w = 0; w < wave; ++w) { pa = s_h[w] ? s_a[w] : pa + s_a[w]; ph |= s_h[w]; }
        if (!xh) xa += pa;
        xh |= ph;
    }
    int o_base = s_base[0];
    for (uint32_t w = 1; w < 4; ++w) o_base = s_base[w] < o_base ? s_base[w] : o_base;
    // 3. places, and the keys: folded in the lane while the node stays the same, per wavefront if it holds ONE node, then into the table
    KdKeys keys;
    kd_keys_clear(keys);
    int o_cur = -1; uint32_t child_cur = 0u;
    uint32_t run_h = xh, run = xa;
#pragma unroll
    for (uint32_t j = 0; j < kKdPer; ++j) {
        const uint32_t k = k0 + j;
        if (k >= n) break;
        const int id = t.id[j], o = t.o[j], n_left = t.left[j], n_right = t.right[j];
        if (!t.split[j]) { idx_out[k] = id; owner_out[k] = -1; continue; }
        if (t.head[j]) { run_h = 1u; run = 0u; }
        const uint32_t lt_before = (run & 0xffffu) + (run_h ? 0u : carry_lt), tie_before = (run >> 16) + (run_h ? 0u : carry_tie);
        run += (t.lt[j] ? 1u : 0u) + (t.tie[j] ? 0x10000u : 0u);
        // pcd_scene.cpp:113-133: the switch starts on and flips at every point on the cut BEFORE the test -- the k-th such point goes left iff k is even
        const bool goes_left = t.lt[j] || (t.tie[j] && ((tie_before + 1u) & 1u) == 0u);
        const uint32_t left_before = lt_before + (tie_before >> 1), right_before = (k - (uint32_t)n_left) - left_before;
        const uint32_t dest = goes_left ? (uint32_t)n_left + left_before : (uint32_t)n_right - 1u - right_before;
        const uint32_t side = goes_left ? 0u : 1u, child0 = (uint32_t)t.child[j] - first_child;
        idx_out[dest] = id; owner_out[dest] = (int)(child0 + side);
        if ((int)k == n_right - 1) left_total[o] = left_before + (goes_left ? 1u : 0u);
        if (o != o_cur) {
            if (o_cur >= 0) { kd_keys_flush(keys, o_cur, o_base, child_cur, s_table, bbkeys, lrkeys); kd_keys_clear(keys); }
            o_cur = o; child_cur = child0;
        }
        const pr_vec3 pt = pcd[id];
        const unsigned long long lr = goes_left ? kd_key_max(t.v[j], k) : kd_key_min(t.v[j], k);     // "below" / "above" of the reference's loop, in the OLD order
        keys.lr[0] = min_u64(keys.lr[0], goes_left ? lr : kKdNoKey); keys.lr[1] = min_u64(keys.lr[1], goes_left ? kKdNoKey : lr);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float c = axis_coord(pt, q);
            const unsigned long long kmn = kd_key_min(c, dest), kmx = kd_key_max(c, dest);      // the children's boxes, in the NEW order
            keys.bb[0][2 * q] = min_u64(keys.bb[0][2 * q], goes_left ? kmn : kKdNoKey); keys.bb[0][2 * q + 1] = min_u64(keys.bb[0][2 * q + 1], goes_left ? kmx : kKdNoKey);
            keys.bb[1][2 * q] = min_u64(keys.bb[1][2 * q], goes_left ? kKdNoKey : kmn); keys.bb[1][2 * q + 1] = min_u64(keys.bb[1][2 * q + 1], goes_left ? kKdNoKey : kmx);
        }
    }
    {
        // every lane of the wavefront on the same node (all wavefronts of the upper levels): fold across the lanes first
        const int o_first = __builtin_amdgcn_readfirstlane(o_cur);
        const bool same = o_cur == o_first && o_cur >= 0 && t.o[0] == o_cur;      // (the lane never changed node: its keys are all of o_cur)
        if (__all(same)) {
#pragma unroll
            for (int c = 0; c < 2; ++c) { keys.lr[c] = wave_min_u64(keys.lr[c]); for (int q = 0; q < 6; ++q) keys.bb[c][q] = wave_min_u64(keys.bb[c][q]); }
            if (lane == 0u) kd_keys_flush(keys, o_cur, o_base, child_cur, s_table, bbkeys, lrkeys);
        } else if (o_cur >= 0) kd_keys_flush(keys, o_cur, o_base, child_cur, s_table, bbkeys, lrkeys);
    }
    __syncthreads();
    // 4. the table to memory: entries that were touched
    for (uint32_t i = threadIdx.x; i < kKdTable * 14u; i += 256) {
        const unsigned long long k = (&s_table[0][0])[i];
        if (k == kKdNoKey) continue;
        const uint32_t rel = i / 14u, slot = i % 14u, o = (uint32_t)o_base + rel;
        if (slot < 2u) atomicMin(&lrkeys[2u * o + slot], k);
        else { const uint32_t child0 = (uint32_t)lv[o].child - first_child; atomicMin(&bbkeys[6u * child0 + (slot - 2u)], k); }      // (slot - 2 = 6 * side + q: the two children are consecutive)
    }
}

// ---- finish: the level that was just partitioned -- split value, the children's records, how many children per chunk split again
__global__ __launch_bounds__(256) void kd_finish_kernel(const KdCtrl *__restrict__ ctrl, pr_kdnode *__restrict__ nodes, int max_leaf, const KdLevelNode *__restrict__ lv,
                                                        const pr_vec3 *__restrict__ pcd, const int *__restrict__ idx_old, const uint32_t *__restrict__ left_total,
                                                        const unsigned long long *__restrict__ lrkeys, uint32_t *__restrict__ chunk_cnt)
{
    if (ctrl->done) return;
    const uint32_t lo = ctrl->lo, hi = ctrl->hi;
    const uint32_t n_level = hi - lo;
    for (uint32_t o0 = blockIdx.x * 256; o0 < n_level; o0 += gridDim.x * 256) {      // (workgroup-uniform trip count: the wavefront folds its chunk counts below)
        const uint32_t o = o0 + threadIdx.x;
        uint32_t grow = 0u, chunk = 0xffffffffu;
        if (o < n_level) {
            const KdLevelNode nd = lv[o];
            if (nd.child >= 0) {
                const unsigned long long kl = lrkeys[2u * o], kr = lrkeys[2u * o + 1u];
                const float below = kl == kKdNoKey ? -FLT_MAX : axis_coord(pcd[idx_old[(uint32_t)kl]], nd.axis);
                const float above = kr == kKdNoKey ? FLT_MAX : axis_coord(pcd[idx_old[(uint32_t)kr]], nd.axis);
                nodes[lo + o].split_v = (below + above) / 2;         // pcd_scene.cpp:135
                const int head = nd.left + (nd.right > nd.left ? (int)left_total[o] : 0);
                pr_kdnode c1, c2;
                kd_blank(c1); kd_blank(c2);
                c1.parent = (int)(lo + o); c1.left = nd.left; c1.right = head;
                c2.parent = (int)(lo + o); c2.left = head; c2.right = nd.right;
                nodes[nd.child] = c1; nodes[nd.child + 1] = c2;
                grow = (head - nd.left > max_leaf ? 1u : 0u) + (nd.right - head > max_leaf ? 1u : 0u);
                chunk = ((uint32_t)nd.child - hi) / kKdChunk;          // (a pair of children never straddles a chunk: both numbers even)
            }
        }
        // one addition per wavefront and chunk (a wavefront's 64 parents have consecutive children: one chunk, at a boundary two) -- additions to
        // ONE address complete one after the other: 512 parents of a chunk adding on their own took 50 us
        for (;;) {
            const unsigned long long todo = __ballot(grow != 0u);
            if (!todo) break;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)chunk, (int)__builtin_ctzll(todo));
            uint32_t mine = (grow != 0u && chunk == c) ? grow : 0u;
            if (chunk == c) grow = 0u;
            for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
            if ((threadIdx.x & 63u) == 0u) atomicAdd(&chunk_cnt[c], mine);
        }
    }
}

// ---- plan: the nodes [hi, next) -- child slots to those that hold more than max_leaf points, pairwise in node order; their box, axis and cut.
// Writes the NEXT level's control record.  `first`: the root's plan (no scatter pass went before it)
__global__ __launch_bounds__(1024) void kd_plan_kernel(const KdCtrl *__restrict__ ctrl, KdCtrl *__restrict__ ctrl_next, pr_kdnode *__restrict__ nodes, uint32_t cap,
                                                       uint32_t max_level, int max_leaf, KdLevelNode *__restrict__ lv_next, const pr_vec3 *__restrict__ pcd,
                                                       const int *__restrict__ idx_new, const unsigned long long *__restrict__ bbkeys, const uint32_t *__restrict__ chunk_cnt, int first)
{
    __shared__ uint32_t s_wave[16], s_sum[16];
    const KdCtrl now = *ctrl;
    if (now.done) { if (blockIdx.x == 0 && threadIdx.x == 0) *ctrl_next = now; return; }
    const uint32_t hi = now.hi, next = now.next, n_child = next - hi, n_chunk = (n_child + kKdChunk - 1u) / kKdChunk;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunk; chunk += gridDim.x) {
        // children that split in the chunks before this one (and, for the workgroup that writes the control record, in all of them)
        uint32_t before_chunks = 0u, all_chunks = 0u;
        for (uint32_t i = threadIdx.x; i < n_chunk; i += 1024) { const uint32_t c = chunk_cnt[i]; all_chunks += c; if (i < chunk) before_chunks += c; }
        for (int off = 32; off > 0; off >>= 1) { before_chunks += __shfl_xor(before_chunks, off); all_chunks += __shfl_xor(all_chunks, off); }
        __syncthreads();                                            // (the arrays' previous readers are done)
        if (lane == 0u) { s_wave[wave] = before_chunks; s_sum[wave] = all_chunks; }
        __syncthreads();
        before_chunks = 0u; all_chunks = 0u;
        for (uint32_t w = 0; w < 16; ++w) { before_chunks += s_wave[w]; all_chunks += s_sum[w]; }
        __syncthreads();
        const uint32_t j = chunk * kKdChunk + threadIdx.x, i = hi + j;
        int L = 0, R = 0;
        if (j < n_child) { L = nodes[i].left; R = nodes[i].right; }
        const bool split = (j < n_child) && (R - L > max_leaf);
        const unsigned long long m = __ballot(split);
        if (lane == 0u) s_wave[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) rank += s_wave[w];
        if (j < n_child) {
            const uint32_t c = next + 2u * (before_chunks + rank);
            const bool ok = split && c + 2u <= cap && j < max_level;
            KdLevelNode nd;
            nd.left = L; nd.right = R; nd.child = ok ? (int)c : -1; nd.axis = 0; nd.cut = 0.0f; nd.pad0 = nd.pad1 = nd.pad2 = 0;
            if (ok) {
                float bmin[3], bmax[3];
                for (int a = 0; a < 3; ++a) {                        // the stored extremes are the coordinates OF the first points that reach them
                    const unsigned long long k_lo = bbkeys[6u * j + 2u * a], k_hi = bbkeys[6u * j + 2u * a + 1u];
                    bmin[a] = k_lo == kKdNoKey ? FLT_MAX : axis_coord(pcd[idx_new[(uint32_t)k_lo]], a);
                    bmax[a] = k_hi == kKdNoKey ? -FLT_MAX : axis_coord(pcd[idx_new[(uint32_t)k_hi]], a);
                }
                int axis = 0; float cut = 0.0f, widest = -FLT_MAX;
                for (int a = 0; a < 3; ++a) {                         // first strictly widest axis, box midpoint (pcd_scene.cpp:96-110)
                    const float extent = bmax[a] - bmin[a];
                    if (extent > widest) { widest = extent; axis = a; cut = (bmin[a] + bmax[a]) / 2; }
                }
                nd.axis = axis; nd.cut = cut;
                pr_kdnode &out = nodes[i];
                for (int a = 0; a < 3; ++a) { out.bbox[2 * a] = bmin[a]; out.bbox[2 * a + 1] = bmax[a]; }
                out.split_dim = axis; out.child1 = (int)c; out.child2 = (int)c + 1;
            }
            if (j < max_level) lv_next[j] = nd;
        }
        if (chunk == 0u && threadIdx.x == 0) {
            const uint32_t after = next + 2u * all_chunks;
            KdCtrl o = now;
            o.lo = hi; o.hi = next; o.next = after;
            if (!first) o.levels = now.levels + 1u;
            if (after > cap) { o.error = 1u; o.done = 1u; }                                                          // 1: the caller's node array is too small
            else if (n_child > max_level || after - next > max_level) { o.error = 2u; o.done = 1u; }               // 2: a level wider than the work arrays were sized for       // (the next level's arrays hold max_level nodes)
            else if (after == next) o.done = 1u;                     // no node of this level splits: done (pcd_scene.cpp:166-168)
            *ctrl_next = o;
        }
    }
}

// the points in tree order; the permutation is in the buffer the last scatter pass wrote
__global__ __launch_bounds__(256) void kd_permute_kernel(const pr_vec3 *__restrict__ pcd, const pr_vec3 *__restrict__ nrm, const KdCtrl *__restrict__ ctrl,
                                                         const int *__restrict__ idx_even, const int *__restrict__ idx_odd,
                                                         uint32_t n, pr_vec3 *__restrict__ pcd_out, pr_vec3 *__restrict__ nrm_out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int *idx = (ctrl->levels & 1u) ? idx_odd : idx_even;
    pcd_out[i] = pcd[idx[i]]; nrm_out[i] = nrm[idx[i]];
}

size_t kd_work_bytes(uint32_t n, uint32_t cap, int max_leaf, KdWork *w)
{
    const size_t n8 = ((size_t)n + 7) & ~(size_t)7, tiles = ((size_t)n + kKdTile - 1) / kKdTile, parts = ((size_t)n + 255) / 256;
    // nodes of one level: the children of disjoint nodes that hold more than max_leaf points each (max_leaf < 1 -- every point a node that splits -- twice as many)
    const uint32_t max_level = (uint32_t)std::min<size_t>(cap, (max_leaf >= 1 ? (size_t)n : 2 * (size_t)n) + 2);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_ctrl = take(2 * sizeof(KdCtrl)), o_idx0 = take(4 * n8), o_idx1 = take(4 * n8), o_own0 = take(4 * n8), o_own1 = take(4 * n8),
                 o_lv0 = take(sizeof(KdLevelNode) * (size_t)max_level), o_lv1 = take(sizeof(KdLevelNode) * (size_t)max_level),
                 o_bb = take(48 * (size_t)max_level + 48), o_lr = take(16 * (size_t)max_level), o_lt = take(4 * (size_t)max_level), o_agg = take(4 * tiles),
                 o_chunk = take(4 * ((size_t)max_level / kKdChunk + 2)), o_part = take(48 * parts);
    if (w && w->base) {
        char *b = static_cast<char *>(w->base);
        w->ctrl[0] = reinterpret_cast<KdCtrl *>(b + o_ctrl); w->ctrl[1] = w->ctrl[0] + 1;
        w->idx[0] = reinterpret_cast<int *>(b + o_idx0); w->idx[1] = reinterpret_cast<int *>(b + o_idx1);
        w->owner[0] = reinterpret_cast<int *>(b + o_own0); w->owner[1] = reinterpret_cast<int *>(b + o_own1);
        w->lv[0] = reinterpret_cast<KdLevelNode *>(b + o_lv0); w->lv[1] = reinterpret_cast<KdLevelNode *>(b + o_lv1);
        w->bbkeys = reinterpret_cast<unsigned long long *>(b + o_bb); w->lrkeys = reinterpret_cast<unsigned long long *>(b + o_lr);
        w->left_total = reinterpret_cast<uint32_t *>(b + o_lt); w->tile_agg = reinterpret_cast<uint32_t *>(b + o_agg);
        w->chunk_cnt = reinterpret_cast<uint32_t *>(b + o_chunk); w->root_part = reinterpret_cast<unsigned long long *>(b + o_part);
        w->max_level = max_level;
    }
    return off;
}

hipError_t launch_kd_init(const KdWork &w, pr_kdnode *nodes, uint32_t cap, const pr_vec3 *pcd, uint32_t n, int max_leaf, hipStream_t s)
{
    const KdCtrl init = { 0u, 0u, 1u, 0u, 0u, 0u, 0u, 0u };           // "level" [0, 0) whose children are the root: the first plan decides whether the root splits
    const hipError_t e = hipMemcpyAsync(w.ctrl[1], &init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    const uint32_t parts = (n + 255) / 256;
    hipLaunchKernelGGL(kd_init_kernel, dim3(parts), dim3(256), 0, s, w.idx[0], w.owner[0], n, pcd, w.root_part);
    hipLaunchKernelGGL(kd_root_kernel, dim3(1), dim3(1024), 0, s, nodes, n, max_leaf, w.root_part, parts, w.bbkeys, w.chunk_cnt);
    hipLaunchKernelGGL(kd_plan_kernel, dim3(1), dim3(1024), 0, s, w.ctrl[1], w.ctrl[0], nodes, cap, w.max_level, max_leaf, w.lv[0], pcd, w.idx[0], w.bbkeys, w.chunk_cnt, 1);
    return hipGetLastError();
}
// level `level` (0 = the root's split): reads ctrl / idx / owner / lv [level & 1], writes the other ones
hipError_t launch_kd_level(const KdWork &w, pr_kdnode *nodes, uint32_t cap, const pr_vec3 *pcd, uint32_t n, int max_leaf, uint32_t level, hipStream_t s)
{
    const uint32_t a = level & 1u, b = a ^ 1u, tiles = (n + kKdTile - 1) / kKdTile;
    // (level l holds at most 2^l nodes and twice as many children: the upper levels' finish and plan launches are one workgroup each -- under load a launch waits
    //  for wave slots for every workgroup it brings, whether it has work or not)
    const uint32_t level_nodes = level < 20u ? std::min<uint32_t>(1u << level, w.max_level) : w.max_level, level_children = std::min<uint32_t>(2u * level_nodes, w.max_level);
    const uint32_t node_groups = std::min<uint32_t>((level_nodes + 255u) / 256u, 512u), chunk_groups = std::min<uint32_t>(level_children / kKdChunk + 1u, 128u);
    hipLaunchKernelGGL(kd_tile_count_kernel, dim3(tiles), dim3(256), 0, s, w.ctrl[a], w.lv[a], pcd, w.idx[a], w.owner[a], n, w.tile_agg, w.bbkeys, w.lrkeys, w.chunk_cnt);
    hipLaunchKernelGGL(kd_tile_scatter_kernel, dim3(tiles), dim3(256), 0, s, w.ctrl[a], w.lv[a], pcd, w.idx[a], w.owner[a], n, w.tile_agg, w.idx[b], w.owner[b],
                       w.left_total, w.bbkeys, w.lrkeys);
    hipLaunchKernelGGL(kd_finish_kernel, dim3(node_groups), dim3(256), 0, s, w.ctrl[a], nodes, max_leaf, w.lv[a], pcd, w.idx[a], w.left_total, w.lrkeys, w.chunk_cnt);
    hipLaunchKernelGGL(kd_plan_kernel, dim3(chunk_groups), dim3(1024), 0, s, w.ctrl[a], w.ctrl[b], nodes, cap, w.max_level, max_leaf, w.lv[b], pcd, w.idx[b], w.bbkeys, w.chunk_cnt, 0);
    return hipGetLastError();
}
hipError_t launch_kd_permute(const KdWork &w, uint32_t levels_launched, const pr_vec3 *pcd, const pr_vec3 *nrm, uint32_t n, pr_vec3 *pcd_out, pr_vec3 *nrm_out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kd_permute_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pcd, nrm, w.ctrl[levels_launched & 1u], w.idx[0], w.idx[1], n, pcd_out, nrm_out);
    return hipGetLastError();
}

}  // namespace prk
