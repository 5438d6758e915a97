// icp_debug.hip -- audit entry of the correspondence pass: the 29 per-point terms of thrust__pcd2Ab (icp.h:128-209), point by point.
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
//
// The product kernel (icp_pass.hip) adds these terms in its own fixed tree; a test that adds them SEQUENTIALLY on the host reproduces the
// summation order of the reference with one thread (icp.cpp:139-148) and with it the reference's inlier counts digit for digit.  One lane
// per point, the same device functions as the pass: the pending update in the pass' operand order, query(), accumulate() from zero.
#include "pr_launch.h"
#include "icp_accumulate.h"

namespace prk {

struct Update12 { float m[12]; };

template <class Scene, bool kNN>
__global__ __launch_bounds__(256) void contrib29_kernel(pr_vec3 *__restrict__ cloud, uint32_t n, int apply, Update12 u, Scene scene, float *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    pr_vec3 p = cloud[i];
    if (apply) {                                                 // transform_pcd_cuda icp.cu:142-153, operand order of the fused pass: ((m0*x + m1*y) + m2*z) + m3
        const float *M = u.m;
        const float x = p.x, y = p.y, z = p.z;
        p.x = M[0] * x + M[1] * y + M[2] * z + M[3];
        p.y = M[4] * x + M[5] * y + M[6] * z + M[7];
        p.z = M[8] * x + M[9] * y + M[10] * z + M[11];
        cloud[i] = p;
    }
    Corr c;
    bool ok;
    if constexpr (kNN) ok = query_nn<false>(scene, nullptr, p.x, p.y, p.z, c);      // the reference's own stackless ordered walk
    else ok = query(scene, p.x, p.y, p.z, c);
    Acc29 acc;
    acc_clear(acc);
    if (ok) accumulate(acc, p.x, p.y, p.z, c);
    float t[29];
    acc_export(acc, t);
#pragma unroll
    for (int k = 0; k < 29; ++k) out[(size_t)i * 29 + k] = t[k];
}

template <class Scene, bool kNN>
static hipError_t launch_contrib29_t(pr_vec3 *cloud, uint32_t n, const float *update12, const Scene &sc, float *out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    Update12 u{};
    if (update12) for (int k = 0; k < 12; ++k) u.m[k] = update12[k];
    hipLaunchKernelGGL(HIP_KERNEL_NAME(contrib29_kernel<Scene, kNN>), dim3((n + 255) / 256), dim3(256), 0, s, cloud, n, update12 ? 1 : 0, u, sc, out);
    return hipGetLastError();
}
hipError_t launch_contrib29_proj_aos(pr_vec3 *cloud, uint32_t n, const float *update12, const SceneProjAoS &sc, float *out, hipStream_t s)
{ return launch_contrib29_t<SceneProjAoS, false>(cloud, n, update12, sc, out, s); }
hipError_t launch_contrib29_proj_packed(pr_vec3 *cloud, uint32_t n, const float *update12, const SceneProjPacked &sc, float *out, hipStream_t s)
{ return launch_contrib29_t<SceneProjPacked, false>(cloud, n, update12, sc, out, s); }
hipError_t launch_contrib29_nn(pr_vec3 *cloud, uint32_t n, const float *update12, const SceneNNDev &sc, float *out, hipStream_t s)
{ return launch_contrib29_t<SceneNNDev, true>(cloud, n, update12, sc, out, s); }

}  // namespace prk
