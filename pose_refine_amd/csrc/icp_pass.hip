// icp_pass.hip -- THE hot kernel: thrust::transform_reduce(thrust__pcd2Ab<Scene>) (icp.cu:170-172) fused with transform_pcd_cuda of the previous iteration (icp.cu:142-153), and the second reduction stage / device solve (icp.cu:178-212)
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#include "pr_launch.h"
#include "icp_accumulate.h"
#include "icp_solve_device.h"

namespace prk {

// ================================================================================================
//  THE hot kernel: pending transform + correspondence + 29-term transform-reduce, all hypotheses
//  of a batch in one launch.  grid = (workgroups per hypothesis, hypotheses), 256 lanes.
// ================================================================================================
template <class Scene, bool kNN, int kStack = 0>
__global__ __launch_bounds__(256, PR_PASS_WAVES) void icp_pass_kernel(IcpBatch b, Scene scene)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float wsum[4][kAccStride];

    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];                           // uniform address: one 64-byte scalar load
    const int32_t st = pm.state;
    if (st == kSkip) return;
    const uint32_t n = pm.count;
    const uint32_t ppb = b.steps * kPointsPerStep;
    if ((uint64_t)blockIdx.x * ppb >= n) return;

    const int4 *lds_topo = nullptr;
    int *stk_node = nullptr; float *stk_lb = nullptr;
    if constexpr (kNN && kStack == 0) {
        int4 *dst = reinterpret_cast<int4 *>(lds_raw);
        for (uint32_t i = threadIdx.x; i < scene.lds_nodes; i += kBlockThreads) dst[i] = scene.topo[i];
        __syncthreads();
        lds_topo = dst;
    }
    if constexpr (kNN && kStack > 0) {                           // per-lane stacks: [entry][lane], then the staged records
        stk_node = reinterpret_cast<int *>(lds_raw) + threadIdx.x;
        stk_lb = reinterpret_cast<float *>(lds_raw) + (size_t)(kStack & 0xff) * kBlockThreads + threadIdx.x;
        float4 *recs = reinterpret_cast<float4 *>(lds_raw + (size_t)(kStack & 0xff) * kBlockThreads * 8);
        if constexpr ((kStack & 0x100) == 0) for (uint32_t i = threadIdx.x; i < scene.lds_nodes * 4; i += kBlockThreads) recs[i] = scene.rec[i];
        __syncthreads();
        lds_topo = reinterpret_cast<const int4 *>(recs);
    }
    if constexpr (kNN && kStack == -1) scene.winner += pm.start;  // winners are indexed like the cloud

    float *cl = reinterpret_cast<float *>(b.cloud + pm.start);
    // a pending update that nn_search_kernel has already applied to the cloud must not be applied again
    const bool xf = (st == kRunWithTransform) && !b.pre_transformed;   // also: not the first pass of this cloud (a seed exists)
    uint32_t *nn_prev = (kNN && kStack >= 0 && b.nn_prev) ? b.nn_prev + pm.start : nullptr;
    float M[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) M[i] = xf ? pm.xform[i] : 0.0f;

    // One virtual workgroup of the canonical tree per trip.  The grid normally holds one workgroup per 2048-point block; the
    // asynchronous fused path sizes its grid from the previous batch's cloud sizes (the current ones are still on the device),
    // so a workgroup may have to take more than one block -- partial sums are per virtual block either way.
    const uint32_t used = (n + ppb - 1) / ppb;
    for (uint32_t vb = blockIdx.x; vb < used; vb += gridDim.x) {
        if (vb != blockIdx.x) __syncthreads();                   // wsum of the previous trip has been read
        float acc[29];
#pragma unroll
        for (int i = 0; i < 29; ++i) acc[i] = 0.0f;
        float t;
        if (b.score_only) {                                      // uniform: the final pass needs sums 27 and 28 only
            vb_accumulate<Scene, kNN, kStack, true>(acc, cl, n, vb * ppb, b.steps, xf, M, scene, lds_topo, stk_node, stk_lb, nn_prev, xf);
            t = vb_reduce<true>(acc, wsum);
        } else {
            vb_accumulate<Scene, kNN, kStack>(acc, cl, n, vb * ppb, b.steps, xf, M, scene, lds_topo, stk_node, stk_lb, nn_prev, xf);
            t = vb_reduce(acc, wsum);
        }
        if (pass_deliver(b, pose, vb, used, n, t)) return;
    }
}

// second stage: workgroup sums added sequentially in workgroup order, starting from 0
__device__ __forceinline__ float sum_partials(const float *partial, uint32_t pose, uint32_t nblk, uint32_t used, uint32_t comp)
{
    // the loads of a chunk are independent and issued together; only the additions are ordered
    float total = 0.0f;
    const float *p = partial + (size_t)pose * nblk * kAccStride + comp;
    for (uint32_t g0 = 0; g0 < used; g0 += 16) {
        float v[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) v[k] = (g0 + k < used) ? p[(size_t)(g0 + k) * kAccStride] : 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) if (g0 + k < used) total += v[k];
    }
    return total;
}

__global__ __launch_bounds__(64) void icp_finalize_kernel(const float *__restrict__ partial, const PoseMeta *__restrict__ meta,
                                                          uint32_t nblk, uint32_t ppb, float *__restrict__ sums)
{
    const uint32_t pose = blockIdx.x;
    if (meta[pose].state == kSkip) return;
    const uint32_t n = meta[pose].count;
    const uint32_t used = (n + ppb - 1) / ppb;
    if (threadIdx.x < kAccStride)
        sums[(size_t)pose * kAccStride + threadIdx.x] = (threadIdx.x < 29) ? sum_partials(partial, pose, nblk, used, threadIdx.x) : 0.0f;
}

// PR_SOLVE_DEVICE: the per-iteration host logic of icp.cu:178-212 for one hypothesis per wavefront
__global__ __launch_bounds__(64) void icp_finalize_solve_kernel(const float *__restrict__ partial, PoseMeta *__restrict__ meta,
                                                                uint32_t nblk, uint32_t ppb, DevIcpState *__restrict__ st,
                                                                pr_criteria crit, uint32_t iter)
{
    const uint32_t pose = blockIdx.x;
    if (meta[pose].state == kSkip) return;
    const uint32_t n = meta[pose].count;
    const uint32_t used = (n + ppb - 1) / ppb;
    DevIcpState s = st[pose];
    float total = 0.0f;
    if (threadIdx.x < 29) total = sum_partials(partial, pose, nblk, used, threadIdx.x);
    float E[16];
    const bool finished = pose_iteration_wave(total, n, s, crit, iter, E);
    if (threadIdx.x != 0) return;
    if (finished) { s.done = 1; meta[pose].state = kSkip; }
    else {
#pragma unroll
        for (int i = 0; i < 12; ++i) meta[pose].xform[i] = E[i];
        meta[pose].state = kRunWithTransform;
    }
    st[pose] = s;
}

__global__ void pack_results_kernel(const DevIcpState *__restrict__ st, pr_result *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pr_result r;
    for (int k = 0; k < 16; ++k) r.T[k] = st[i].T[k];
    r.inlier_rmse = st[i].rmse; r.fitness = st[i].fitness;
    out[i] = r;
}
// results to the device buffer and, in the same launch, results / cloud sizes to pinned host staging (stores over the host link)
__global__ void pack_export_kernel(const DevIcpState *__restrict__ st, pr_result *__restrict__ out, const uint32_t *__restrict__ counts,
                                   uint32_t *__restrict__ host_counts, pr_result *__restrict__ host_results, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pr_result r;
    for (int k = 0; k < 16; ++k) r.T[k] = st[i].T[k];
    r.inlier_rmse = st[i].rmse; r.fitness = st[i].fitness;
    out[i] = r;
    host_counts[i] = counts[i];
    if (host_results) host_results[i] = r;
}
// small host-staged inputs (poses, pixel boxes) pulled into device memory by a kernel: an SDMA copy of 20 KB costs ~20 us of latency
__global__ void stage_words_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

// the same in 8-byte words, for records whose total is no multiple of 16 (72-byte results): never a byte beyond the destination
__global__ void stage_words64_kernel(const uint2 *__restrict__ src, uint2 *__restrict__ dst, uint32_t n8)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n8) dst[i] = src[i];
}

template <class Scene, bool kNN, int kStack = 0>
static hipError_t launch_pass(const IcpBatch &b, const Scene &sc, uint32_t n_poses, size_t lds_bytes, hipStream_t s)
{
    if (n_poses == 0 || b.nblk == 0) return hipSuccess;
    for (uint32_t p0 = 0; p0 < n_poses; p0 += 32768) {
        const uint32_t np = (n_poses - p0 < 32768) ? (n_poses - p0) : 32768;
        IcpBatch bb = b;
        bb.meta += p0;
        bb.partial += (size_t)p0 * b.nblk * kAccStride;
        if (bb.st) bb.st += p0;                                  // everything indexed by the hypothesis moves with the piece
        if (bb.arrive) bb.arrive += p0;
        if (bb.sums_out) bb.sums_out += (size_t)p0 * kAccStride;
        if (bb.nn_qcount) bb.nn_qcount += kQCountStride * (size_t)p0;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(icp_pass_kernel<Scene, kNN, kStack>), dim3(b.grid_x ? b.grid_x : b.nblk, np), dim3(kBlockThreads), lds_bytes, s, bb, sc);
    }
    return hipGetLastError();
}
hipError_t launch_icp_pass_proj_aos(const IcpBatch &b, const SceneProjAoS &sc, uint32_t n_poses, hipStream_t s)
{ return launch_pass<SceneProjAoS, false>(b, sc, n_poses, 0, s); }
// (Round 5 measured the two tables colf / rowf of the packed scene staged in LDS per workgroup -- two of a point's three gathers off the texture
// addresser: 0.84 against 0.82 ms per 21 launches, 272 / 263 k against 277 / 276 k poses/s on one box.  Not kept.)
hipError_t launch_icp_pass_proj_packed(const IcpBatch &b, const SceneProjPacked &sc, uint32_t n_poses, hipStream_t s)
{ return launch_pass<SceneProjPacked, false>(b, sc, n_poses, 0, s); }
hipError_t launch_icp_pass_nn(const IcpBatch &b, const SceneNNDev &sc, uint32_t n_poses, hipStream_t s)
{
    if (sc.stack_depth == 16 && sc.rec32) return launch_pass<SceneNNDev, true, 16 + 0x100>(b, sc, n_poses, (size_t)16 * kBlockThreads * 8, s);
    if (sc.stack_depth == 24 && sc.rec32) return launch_pass<SceneNNDev, true, 24 + 0x100>(b, sc, n_poses, (size_t)24 * kBlockThreads * 8, s);
    if (sc.stack_depth == 16) return launch_pass<SceneNNDev, true, 16>(b, sc, n_poses, (size_t)16 * kBlockThreads * 8 + (size_t)sc.lds_nodes * 64, s);
    if (sc.stack_depth == 24) return launch_pass<SceneNNDev, true, 24>(b, sc, n_poses, (size_t)24 * kBlockThreads * 8 + (size_t)sc.lds_nodes * 64, s);
    return launch_pass<SceneNNDev, true, 0>(b, sc, n_poses, (size_t)sc.lds_nodes * sizeof(int4), s);
}

hipError_t launch_icp_pass_nn_winners(const IcpBatch &b, const SceneNNWinners &sc, uint32_t n_poses, hipStream_t s)
{ return launch_pass<SceneNNWinners, true, -1>(b, sc, n_poses, 0, s); }

hipError_t launch_icp_finalize(const float *partial, const PoseMeta *meta, uint32_t nblk,
                               uint32_t steps, float *sums, uint32_t n_poses, hipStream_t s)
{
    if (n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(icp_finalize_kernel, dim3(n_poses), dim3(64), 0, s, partial, meta, nblk, steps * kPointsPerStep, sums);
    return hipGetLastError();
}
hipError_t launch_icp_finalize_solve(const float *partial, PoseMeta *meta, uint32_t nblk,
                                     uint32_t steps, DevIcpState *st, pr_criteria crit, uint32_t iter,
                                     uint32_t n_poses, hipStream_t s)
{
    if (n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(icp_finalize_solve_kernel, dim3(n_poses), dim3(64), 0, s, partial, meta, nblk,
                       steps * kPointsPerStep, st, crit, iter);
    return hipGetLastError();
}
hipError_t launch_pack_export(const DevIcpState *st, pr_result *out, const uint32_t *counts, uint32_t *host_counts, pr_result *host_results,
                              uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_export_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st, out, counts, host_counts, host_results, n);
    return hipGetLastError();
}
hipError_t launch_stage_words(const void *src_host_mapped, void *dst, size_t bytes, hipStream_t s)
{
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    if (n16 == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_words_kernel, dim3((n16 + 255) / 256), dim3(256), 0, s, static_cast<const uint4 *>(src_host_mapped), static_cast<uint4 *>(dst), n16);
    return hipGetLastError();
}
// 4-byte words, either direction (device -> pinned host too: small read-backs without a copy command)
__global__ void copy_words32_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
hipError_t launch_copy_words32(const void *src, void *dst, uint32_t n_words, hipStream_t s)
{
    if (n_words == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_words32_kernel, dim3((n_words + 255) / 256), dim3(256), 0, s, static_cast<const uint32_t *>(src), static_cast<uint32_t *>(dst), n_words);
    return hipGetLastError();
}
hipError_t launch_stage_words64(const void *src_host_mapped, void *dst, size_t bytes, hipStream_t s)
{
    if (bytes % 8) return hipErrorInvalidValue;
    const uint32_t n8 = (uint32_t)(bytes / 8);
    if (n8 == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_words64_kernel, dim3((n8 + 255) / 256), dim3(256), 0, s, static_cast<const uint2 *>(src_host_mapped), static_cast<uint2 *>(dst), n8);
    return hipGetLastError();
}
hipError_t launch_pack_results(const DevIcpState *st, pr_result *out, uint32_t n_poses, hipStream_t s)
{
    if (n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_results_kernel, dim3((n_poses + 255) / 256), dim3(256), 0, s, st, out, n_poses);
    return hipGetLastError();
}

}  // namespace prk
