// kd_build.hip -- KDTree_cpu::build_tree (pcd_scene.cpp:45-184) on the device, level by level, bit-identical nodes and permutation
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#include "pr_launch.h"

namespace prk {

// ================================================================================================
//  kd-tree build on the device -- the reference's level-order build (pcd_scene.cpp:45-184) level by level:
//  one small kernel hands out child slots to the nodes of the level that split (children are appended pairwise in node order),
//  one workgroup per node then computes the box, picks the widest axis, and performs the stable two-ended partition with the
//  alternating tie rule through block scans (left part keeps its order, right part is filled from the end, the k-th tie goes
//  left iff k is even).  Same float operations, so nodes and permutation are bit-identical to the CPU build.
// ================================================================================================
__global__ __launch_bounds__(256) void kd_init_kernel(pr_kdnode *__restrict__ nodes, uint32_t cap, int *__restrict__ idx, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) idx[i] = (int)i;
    if (i < cap) {
        pr_kdnode b;
        b.parent = b.child1 = b.child2 = -1; b.split_v = 0.0f; b.split_dim = 0; b.left = 0; b.right = 0;
        for (int k = 0; k < 6; ++k) b.bbox[k] = 0.0f;
        if (i == 0) b.right = (int)n;
        nodes[i] = b;
    }
}

// ctrl = {level_lo, level_hi, count}.  Nodes [lo,hi) that hold more than max_leaf points get child slots count + 2*rank.
__global__ __launch_bounds__(256) void kd_level_plan_kernel(const pr_kdnode *__restrict__ nodes, uint32_t *__restrict__ ctrl, int max_leaf,
                                                            int *__restrict__ child_of, uint32_t cap)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t running;
    const uint32_t lo = ctrl[0], hi = ctrl[1];
    if (threadIdx.x == 0) running = ctrl[2];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = lo; base < hi; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const bool split = (i < hi) && (nodes[i].right - nodes[i].left > max_leaf);
        const unsigned long long m = __ballot(split);
        const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = 0, total = 0;
        for (uint32_t w = 0; w < 4; ++w) { if (w < wave) off += wsum[w]; total += wsum[w]; }
        if (i < hi) {
            const uint32_t c = running + 2 * (off + before);
            child_of[i - lo] = (split && c + 2 <= cap) ? (int)c : -1;
        }
        __syncthreads();
        if (threadIdx.x == 0) running += 2 * total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { ctrl[3] = running; }                   // count after this level (may exceed cap: host checks)
}

__device__ __forceinline__ float axis_coord(const pr_vec3 &p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }

// The CPU build keeps an extreme with "if (v > best) best = v" while walking the points in order, i.e. the FIRST occurrence
// among equal values survives -- observable only through the sign of a zero, but the node records are compared bit for bit.
// (value, sequence position) pairs reproduce that under any reduction order.
struct Ext { float v; int k; };
__device__ __forceinline__ Ext ext_max(Ext a, Ext b) { return (b.v > a.v || (b.v == a.v && b.k < a.k)) ? b : a; }
__device__ __forceinline__ Ext ext_min(Ext a, Ext b) { return (b.v < a.v || (b.v == a.v && b.k < a.k)) ? b : a; }
__device__ __forceinline__ Ext ext_shfl(Ext a, int off) { Ext r; r.v = __shfl_xor(a.v, off); r.k = __shfl_xor(a.k, off); return r; }

__global__ __launch_bounds__(256) void kd_level_split_kernel(pr_kdnode *__restrict__ nodes, const uint32_t *__restrict__ ctrl,
                                                             const int *__restrict__ child_of, const pr_vec3 *__restrict__ pcd,
                                                             int *__restrict__ idx, int *__restrict__ scratch)
{
    __shared__ float red[4][8];
    __shared__ int redk[4][8];
    __shared__ uint32_t wcnt[4][3];
    __shared__ uint32_t run[3];
    __shared__ float s_split;
    __shared__ int s_axis;
    const uint32_t node = ctrl[0] + blockIdx.x;
    const int child = child_of[blockIdx.x];
    if (child < 0) return;
    const int L = nodes[node].left, R = nodes[node].right;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // 1. box of the node's points (first occurrence wins among equal extremes, like the sequential CPU loop)
    const int kNone = 0x7fffffff;
    Ext mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a].v = FLT_MAX; mn[a].k = kNone; mx[a].v = -FLT_MAX; mx[a].k = kNone; }
    for (int k = L + (int)threadIdx.x; k < R; k += 256) {
        const pr_vec3 p = pcd[idx[k]];
        const float c3[3] = { p.x, p.y, p.z };
#pragma unroll
        for (int a = 0; a < 3; ++a) { Ext e; e.v = c3[a]; e.k = k; if (e.v > mx[a].v) mx[a] = e; if (e.v < mn[a].v) mn[a] = e; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) { mn[a] = ext_min(mn[a], ext_shfl(mn[a], off)); mx[a] = ext_max(mx[a], ext_shfl(mx[a], off)); }
    if (lane == 0) for (int a = 0; a < 3; ++a) { red[wave][a] = mn[a].v; redk[wave][a] = mn[a].k; red[wave][3 + a] = mx[a].v; redk[wave][3 + a] = mx[a].k; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bmin[3], bmax[3];
        for (int a = 0; a < 3; ++a) {
            Ext lo{ red[0][a], redk[0][a] }, hi{ red[0][3 + a], redk[0][3 + a] };
            for (int w = 1; w < 4; ++w) { lo = ext_min(lo, Ext{ red[w][a], redk[w][a] }); hi = ext_max(hi, Ext{ red[w][3 + a], redk[w][3 + a] }); }
            bmin[a] = lo.v; bmax[a] = hi.v;
        }
        int axis = 0; float cut = 0.0f, widest = -FLT_MAX;
        for (int a = 0; a < 3; ++a) {                             // first strictly widest axis, box midpoint (pcd_scene.cpp:96-110)
            const float extent = bmax[a] - bmin[a];
            if (extent > widest) { widest = extent; axis = a; cut = (bmin[a] + bmax[a]) / 2; }
        }
        s_axis = axis; s_split = cut;
        pr_kdnode &nd = nodes[node];
        for (int a = 0; a < 3; ++a) { nd.bbox[2 * a] = bmin[a]; nd.bbox[2 * a + 1] = bmax[a]; }
        nd.split_dim = axis; nd.child1 = child; nd.child2 = child + 1;
        run[0] = run[1] = run[2] = 0;
    }
    __syncthreads();
    const int axis = s_axis; const float cut = s_split;

    // 2. stable two-ended partition, 256 points at a time
    Ext left_max{ -FLT_MAX, kNone }, right_min{ FLT_MAX, kNone };
    for (int base = L; base < R; base += 256) {
        const int k = base + (int)threadIdx.x;
        const bool live = k < R;
        int id = 0; float v = 0.0f;
        if (live) { id = idx[k]; v = axis_coord(pcd[id], axis); }
        const bool tie = live && (v == cut);
        const unsigned long long mt = __ballot(tie);
        if (lane == 0) wcnt[wave][0] = (uint32_t)__popcll(mt);
        __syncthreads();
        uint32_t tie_before = run[0] + (uint32_t)__popcll(mt & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) tie_before += wcnt[w][0];
        // the toggle starts true and flips at every tie before the test: the k-th tie (k = tie_before + 1) goes left iff k is even
        const bool goes_left = live && (v < cut || (tie && ((tie_before + 1u) % 2u == 0u)));
        const bool goes_right = live && !goes_left;
        const unsigned long long ml = __ballot(goes_left), mr = __ballot(goes_right);
        if (lane == 0) { wcnt[wave][1] = (uint32_t)__popcll(ml); wcnt[wave][2] = (uint32_t)__popcll(mr); }
        __syncthreads();
        uint32_t lb = run[1] + (uint32_t)__popcll(ml & ((1ull << lane) - 1ull));
        uint32_t rb = run[2] + (uint32_t)__popcll(mr & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) { lb += wcnt[w][1]; rb += wcnt[w][2]; }
        if (goes_left) { scratch[L + (int)lb] = id; if (v > left_max.v) { left_max.v = v; left_max.k = k; } }
        if (goes_right) { scratch[R - 1 - (int)rb] = id; if (v < right_min.v) { right_min.v = v; right_min.k = k; } }
        __syncthreads();
        if (threadIdx.x == 0) for (uint32_t w = 0; w < 4; ++w) { run[0] += wcnt[w][0]; run[1] += wcnt[w][1]; run[2] += wcnt[w][2]; }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) { left_max = ext_max(left_max, ext_shfl(left_max, off)); right_min = ext_min(right_min, ext_shfl(right_min, off)); }
    if (lane == 0) { red[wave][6] = left_max.v; redk[wave][6] = left_max.k; red[wave][7] = right_min.v; redk[wave][7] = right_min.k; }
    __syncthreads();
    for (int k = L + (int)threadIdx.x; k < R; k += 256) idx[k] = scratch[k];
    if (threadIdx.x == 0) {
        Ext lmx{ red[0][6], redk[0][6] }, rmn{ red[0][7], redk[0][7] };
        for (int w = 1; w < 4; ++w) { lmx = ext_max(lmx, Ext{ red[w][6], redk[w][6] }); rmn = ext_min(rmn, Ext{ red[w][7], redk[w][7] }); }
        nodes[node].split_v = (lmx.v + rmn.v) / 2;                // pcd_scene.cpp:135
        const int head = L + (int)run[1];
        pr_kdnode &c1 = nodes[child], &c2 = nodes[child + 1];
        c1.parent = (int)node; c1.left = L; c1.right = head;
        c2.parent = (int)node; c2.left = head; c2.right = R;
    }
}

__global__ __launch_bounds__(256) void kd_advance_kernel(uint32_t *ctrl) { if (threadIdx.x == 0 && blockIdx.x == 0) { ctrl[0] = ctrl[1]; ctrl[1] = ctrl[3]; ctrl[2] = ctrl[3]; } }

__global__ __launch_bounds__(256) void kd_permute_kernel(const pr_vec3 *__restrict__ pcd, const pr_vec3 *__restrict__ nrm, const int *__restrict__ idx,
                                                         uint32_t n, pr_vec3 *__restrict__ pcd_out, pr_vec3 *__restrict__ nrm_out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    pcd_out[i] = pcd[idx[i]]; nrm_out[i] = nrm[idx[i]];
}

hipError_t launch_kd_init(pr_kdnode *nodes, uint32_t cap, int *idx, uint32_t n, uint32_t *ctrl, hipStream_t s)
{
    const uint32_t m = cap > n ? cap : n;
    hipLaunchKernelGGL(kd_init_kernel, dim3((m + 255) / 256), dim3(256), 0, s, nodes, cap, idx, n);
    const uint32_t init[4] = { 0, 1, 1, 1 };
    return hipMemcpyAsync(ctrl, init, sizeof init, hipMemcpyHostToDevice, s);
}
hipError_t launch_kd_level(pr_kdnode *nodes, uint32_t *ctrl, int max_leaf, int *child_of, uint32_t cap, uint32_t level_nodes,
                           const pr_vec3 *pcd, int *idx, int *scratch, bool plan_only, hipStream_t s)
{
    if (plan_only) { hipLaunchKernelGGL(kd_level_plan_kernel, dim3(1), dim3(256), 0, s, nodes, ctrl, max_leaf, child_of, cap); return hipGetLastError(); }
    if (level_nodes) hipLaunchKernelGGL(kd_level_split_kernel, dim3(level_nodes), dim3(256), 0, s, nodes, ctrl, child_of, pcd, idx, scratch);
    hipLaunchKernelGGL(kd_advance_kernel, dim3(1), dim3(64), 0, s, ctrl);
    return hipGetLastError();
}
hipError_t launch_kd_permute(const pr_vec3 *pcd, const pr_vec3 *nrm, const int *idx, uint32_t n, pr_vec3 *pcd_out, pr_vec3 *nrm_out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kd_permute_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pcd, nrm, idx, n, pcd_out, nrm_out);
    return hipGetLastError();
}

}  // namespace prk
