// icp_flow.hip -- persistent dataflow form of the ICP loop (option icp_flow, off by default: measured, bit-identical, not faster -- DESIGN.md)
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#include "pr_launch.h"
#include "icp_accumulate.h"
#include "icp_solve_device.h"

namespace prk {

// ================================================================================================
//  Dataflow ICP: ALL iterations of ALL hypotheses in one persistent launch.
//
//  The multi-launch loop serialises [pass over every pose] -> [solve of every pose] 21 times although pose i only ever
//  waits for pose i: at 256 poses about a third of each pass and all of each solve launch is dependency latency.  Here
//  every resident workgroup owns a fixed, strided set of virtual workgroups (pose, g) of the canonical tree and walks
//  them iteration by iteration; the last workgroup to deliver a partial sum for a pose runs that pose's finalize + 6x6
//  solve and publishes the update, everyone else picks it up when it next touches the pose.  Sums, solve and results are
//  bit-identical to the multi-launch path (same tree, same code).
//
//  Cross-workgroup data (partials, PoseMeta, DevIcpState, counters) moves with system-scope (sc0 sc1) accesses, i.e.
//  through memory, never through a possibly stale L1/L2 line; a producer drains its stores (s_waitcnt vmcnt(0)) before
//  the arrival atomic / ready flag that publishes them (cdna_hip_programming.md G16, valid form "sc0 sc1 stores and loads
//  both sides").  A cloud block is only ever touched by its owning workgroup, so plain accesses are fine there.
//  Every spin is bounded; a timeout raises the abort flag and the whole grid drains.
// ================================================================================================

template <class Scene, bool kNN, int kStack>
__global__ __launch_bounds__(256, 5) void icp_flow_kernel(FlowArgs a, Scene scene)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float wsum[4][kAccStride];
    __shared__ float Ab[kAccStride];
    __shared__ uint32_t sm_meta[16];
    __shared__ int sm_go;

    const int4 *lds_topo = nullptr;
    int *stk_node = nullptr; float *stk_lb = nullptr;
    if constexpr (kNN && kStack == 0) {
        int4 *dst = reinterpret_cast<int4 *>(lds_raw);
        for (uint32_t i = threadIdx.x; i < scene.lds_nodes; i += kBlockThreads) dst[i] = scene.topo[i];
        __syncthreads();
        lds_topo = dst;
    }
    if constexpr (kNN && kStack > 0) {
        stk_node = reinterpret_cast<int *>(lds_raw) + threadIdx.x;
        stk_lb = reinterpret_cast<float *>(lds_raw) + (size_t)(kStack & 0xff) * kBlockThreads + threadIdx.x;
        float4 *recs = reinterpret_cast<float4 *>(lds_raw + (size_t)(kStack & 0xff) * kBlockThreads * 8);
        if constexpr ((kStack & 0x100) == 0) for (uint32_t i = threadIdx.x; i < scene.lds_nodes * 4; i += kBlockThreads) recs[i] = scene.rec[i];
        __syncthreads();
        lds_topo = reinterpret_cast<const int4 *>(recs);
    }

    const uint32_t ppb = a.steps * kPointsPerStep;
    constexpr uint32_t kSpinLimit = 1u << 22;

    for (uint32_t it = 0; it <= (uint32_t)a.crit.max_iteration; ++it) {
        for (uint32_t v = blockIdx.x; v < a.n_vbs; v += gridDim.x) {
            const uint2 d = a.vb_desc[v];                          // {pose, g}: uniform
            const uint32_t pose = d.x, g = d.y;
            if (threadIdx.x == 0) {
                int go = 1;
                uint32_t spins = 0;
                while (ld_sys_u32(&a.ready[pose]) < it) {          // the solve that closes iteration it-1 of this pose
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > kSpinLimit || ((spins & 255u) == 0 && ld_sys_u32(a.abort_flag) != 0)) { st_sys_u32(a.abort_flag, 1u); go = -1; break; }
                }
                sm_go = go;
            }
            __syncthreads();
            if (sm_go < 0) return;
            if (threadIdx.x < 16) sm_meta[threadIdx.x] = ld_sys_u32(reinterpret_cast<const uint32_t *>(a.meta + pose) + threadIdx.x);
            __syncthreads();
            const uint32_t start = sm_meta[0], n = sm_meta[1];
            const int32_t st = (int32_t)sm_meta[2];
            const uint32_t first = g * ppb;
            if (st != kSkip && first < n) {
                const bool xf = (st == kRunWithTransform);
                float M[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) M[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(sm_meta[4 + i]));
                float acc[29];
#pragma unroll
                for (int i = 0; i < 29; ++i) acc[i] = 0.0f;
                float *cl = reinterpret_cast<float *>(a.cloud + start);
                vb_accumulate<Scene, kNN, kStack>(acc, cl, n, first, a.steps, xf, M, scene, lds_topo, stk_node, stk_lb);
                const float t = vb_reduce(acc, wsum);
                const uint32_t used = (n + ppb - 1) / ppb;
                if (threadIdx.x < 29) st_sys_f32(&a.partial[((size_t)pose * a.nblk + g) * kAccStride + threadIdx.x], t);
                if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // wave 0 issued the stores
                if (threadIdx.x == 0) {
                    const uint32_t ticket = atomicAdd(&a.arrive[pose], 1u);
                    sm_go = (ticket + 1u == used * (it + 1u)) ? 2 : 1;
                }
                __syncthreads();
                if (sm_go == 2) {                                  // last arrival of this iteration: finalize + solve
                    if (threadIdx.x < 29) {
                        float total = 0.0f;
                        const float *pp = a.partial + (size_t)pose * a.nblk * kAccStride + threadIdx.x;
                        for (uint32_t k = 0; k < used; ++k) total += ld_sys_f32(pp + (size_t)k * kAccStride);
                        Ab[threadIdx.x] = total;
                    }
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        DevIcpState s;
                        uint32_t *sw = reinterpret_cast<uint32_t *>(&s);
                        uint32_t *gw = reinterpret_cast<uint32_t *>(a.st + pose);
#pragma unroll
                        for (uint32_t i = 0; i < sizeof(DevIcpState) / 4; ++i) sw[i] = ld_sys_u32(gw + i);
                        float E[16];
                        const bool finished = pose_iteration(Ab, n, s, a.crit, it, E);
                        uint32_t *mw = reinterpret_cast<uint32_t *>(a.meta + pose);
                        if (finished) { s.done = 1; st_sys_u32(mw + 2, (uint32_t)kSkip); }
                        else {
#pragma unroll
                            for (int i = 0; i < 12; ++i) st_sys_f32(reinterpret_cast<float *>(mw + 4) + i, E[i]);
                            st_sys_u32(mw + 2, (uint32_t)kRunWithTransform);
                        }
#pragma unroll
                        for (uint32_t i = 0; i < sizeof(DevIcpState) / 4; ++i) st_sys_u32(gw + i, sw[i]);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        st_sys_u32(&a.ready[pose], finished ? 0xffffffffu : it + 1u);
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <class Scene, bool kNN, int kStack>
static hipError_t launch_flow_t(const FlowArgs &a, const Scene &sc, size_t lds_bytes, uint32_t n_cus, hipStream_t s, uint32_t *grid_out)
{
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(icp_flow_kernel<Scene, kNN, kStack>),
                                                                (int)kBlockThreads, lds_bytes);
    if (e != hipSuccess) return e;
    if (per_cu < 1) return hipErrorInvalidValue;
    if (per_cu > 8) per_cu = 8;
    // every workgroup must be resident (they wait on each other): grid <= what one wave of dispatch can hold
    uint32_t grid = n_cus * (uint32_t)per_cu;
    if (grid > a.n_vbs) grid = a.n_vbs;
    if (grid_out) *grid_out = grid;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(icp_flow_kernel<Scene, kNN, kStack>), dim3(grid), dim3(kBlockThreads), lds_bytes, s, a, sc);
    return hipGetLastError();
}
hipError_t launch_icp_flow_proj_aos(const FlowArgs &a, const SceneProjAoS &sc, uint32_t n_cus, hipStream_t s, uint32_t *grid_out)
{ return launch_flow_t<SceneProjAoS, false, 0>(a, sc, 0, n_cus, s, grid_out); }
hipError_t launch_icp_flow_proj_packed(const FlowArgs &a, const SceneProjPacked &sc, uint32_t n_cus, hipStream_t s, uint32_t *grid_out)
{ return launch_flow_t<SceneProjPacked, false, 0>(a, sc, 0, n_cus, s, grid_out); }
hipError_t launch_icp_flow_nn(const FlowArgs &a, const SceneNNDev &sc, uint32_t n_cus, hipStream_t s, uint32_t *grid_out)
{
    if (sc.stack_depth == 16) return launch_flow_t<SceneNNDev, true, 16>(a, sc, (size_t)16 * kBlockThreads * 8 + (size_t)sc.lds_nodes * 64, n_cus, s, grid_out);
    if (sc.stack_depth == 24) return launch_flow_t<SceneNNDev, true, 24>(a, sc, (size_t)24 * kBlockThreads * 8 + (size_t)sc.lds_nodes * 64, n_cus, s, grid_out);
    return launch_flow_t<SceneNNDev, true, 0>(a, sc, (size_t)sc.lds_nodes * sizeof(int4), n_cus, s, grid_out);
}

}  // namespace prk
