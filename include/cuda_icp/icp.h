// icp.h -- cuda_icp:: API of cuda_icp/icp.h:16-120 over the C ABI (HIP build: the reference's
// CUDA_ON configuration).  ICP_Point2Plane / depth2cloud dispatch to the device versions.
#pragma once
#include <cmath>
#include <cstdint>
#include <utility>

#include "geometry.h"
#include "scene/depth_scene/depth_scene.h"
#include "scene/pcd_scene/pcd_scene.h"

namespace cuda_icp {

using V3f_holder = device_vector_holder<Vec3f>;

struct RegistrationResult {                       // icp.h:26-36 (72 bytes == pr_result)
    RegistrationResult(const Mat4x4f &t = Mat4x4f::identity()) : transformation_(t), inlier_rmse_(0.0f), fitness_(0.0f) {}
    Mat4x4f transformation_;
    float inlier_rmse_;
    float fitness_;
};
static_assert(sizeof(RegistrationResult) == sizeof(pr_result), "RegistrationResult layout");

struct ICPConvergenceCriteria {                   // icp.h:38-50
    ICPConvergenceCriteria(float relative_fitness = 1e-5f, float relative_rmse = 1e-5f, int max_iteration = 30)
        : relative_fitness_(relative_fitness), relative_rmse_(relative_rmse), max_iteration_(max_iteration) {}
    float relative_fitness_, relative_rmse_;
    int max_iteration_;
};

inline Mat4x4f eigen_slover_666(float *A, float *b)           // icp.h:54 / icp.cpp:29-45
{
    pr_mat4 T; pr_solve_666(A, b, &T); return Mat4x4f(T.m);
}

namespace detail {
inline pr_criteria c(const ICPConvergenceCriteria &k) { return pr_criteria{ k.relative_fitness_, k.relative_rmse_, k.max_iteration_ }; }
inline device_vector_holder<Vec3f> adopt(pr_vec3 *p, uint32_t n)
{ device_vector_holder<Vec3f> h; h.__gpu_memory = reinterpret_cast<Vec3f *>(p); h.__size = n; h.valid = true; return h; }
}  // namespace detail

// icp.cu:256-291: DEVICE depth pointer in, device cloud out (row-major pixel order, metres)
inline device_vector_holder<Vec3f> depth2cloud_cuda(int32_t *depth, uint32_t width, uint32_t height, Mat3x3f &K, uint32_t stride = 1, uint32_t tl_x = 0, uint32_t tl_y = 0)
{ pr_vec3 *c = nullptr; uint32_t n = 0; pose_refine_detail::must(pr_depth2cloud_i32(depth, width, height, K.data(), stride, tl_x, tl_y, &c, &n), "pr_depth2cloud_i32"); return detail::adopt(c, n); }
inline device_vector_holder<Vec3f> depth2cloud_cuda(uint16_t *depth, uint32_t width, uint32_t height, Mat3x3f &K, uint32_t stride = 1, uint32_t tl_x = 0, uint32_t tl_y = 0)
{ pr_vec3 *c = nullptr; uint32_t n = 0; pose_refine_detail::must(pr_depth2cloud_u16(depth, width, height, K.data(), stride, tl_x, tl_y, &c, &n), "pr_depth2cloud_u16"); return detail::adopt(c, n); }

// icp.cu:156-223: mutates model_pcd in place; the scene is passed by value like the reference does
inline RegistrationResult ICP_Point2Plane_cuda(device_vector_holder<Vec3f> &model_pcd, const Scene_projective scene,
                                               const ICPConvergenceCriteria criteria = ICPConvergenceCriteria())
{
    RegistrationResult r; pr_scene_proj s = scene.c_view();
    pose_refine_detail::must(pr_icp_proj(reinterpret_cast<pr_vec3 *>(model_pcd.data()), (uint32_t)model_pcd.size(), &s, detail::c(criteria), reinterpret_cast<pr_result *>(&r)), "pr_icp_proj");
    return r;
}
inline RegistrationResult ICP_Point2Plane_cuda(device_vector_holder<Vec3f> &model_pcd, const Scene_nn scene,
                                               const ICPConvergenceCriteria criteria = ICPConvergenceCriteria())
{
    RegistrationResult r; pr_scene_nn s = scene.c_view();
    pose_refine_detail::must(pr_icp_nn(reinterpret_cast<pr_vec3 *>(model_pcd.data()), (uint32_t)model_pcd.size(), &s, detail::c(criteria), reinterpret_cast<pr_result *>(&r)), "pr_icp_nn");
    return r;
}

// ---- CPU twins (cuda_icp/icp.cpp:73-188, icp.h:125-215): part of the reference API (test.cpp:72,129).  Plain host code
// over host pointers; single-threaded sums in index order (= the reference with OMP_NUM_THREADS=1, which is the only
// deterministic configuration of its OpenMP reduction).  Never used as a fallback by the device path.
typedef vec<29, float> Vec29f;

template <class Scene> struct thrust__pcd2Ab {             // icp.h:128-209: per-point 29-term contribution
    Scene __scene;
    explicit thrust__pcd2Ab(Scene scene) : __scene(scene) {}
    Vec29f operator()(const Vec3f &src) const
    {
        Vec29f out;
        Vec3f dst, n; bool valid = false;
        __scene.query(src, dst, n, valid);
        if (!valid) return out;
        const float ex = dst.x - src.x, ey = dst.y - src.y, ez = dst.z - src.z;
        const float r = ex * n.x + ey * n.y + ez * n.z;
        const float J[6] = { n.z * src.y - n.y * src.z, n.x * src.z - n.z * src.x, n.y * src.x - n.x * src.y, n.x, n.y, n.z };
        int k = 0;
        for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) out[k++] = J[a] * J[b];
        for (int a = 0; a < 6; ++a) out[21 + a] = J[a] * r;
        out[27] = ex * ex + ey * ey + ez * ez;
        out[28] = 1;
        return out;
    }
};
struct thrust__plus { Vec29f operator()(const Vec29f &a, const Vec29f &b) const { return a + b; } };

template <class T>
std::vector<Vec3f> depth2cloud_cpu(T *depth, uint32_t width, uint32_t height, Mat3x3f &K, uint32_t stride = 1, uint32_t tl_x = 0, uint32_t tl_y = 0)
{
    std::vector<Vec3f> cloud;
    for (uint32_t y = 0; y < height / stride; ++y)
        for (uint32_t x = 0; x < width / stride; ++x) {
            const T d = depth[x * stride + y * stride * width];
            if (d <= 0) continue;
            const float z = d / 1000.0f;
            cloud.push_back(Vec3f((x + tl_x - K[0][2]) / K[0][0] * z, (y + tl_y - K[1][2]) / K[1][1] * z, z));
        }
    return cloud;
}

template <class Scene>
RegistrationResult ICP_Point2Plane_cpu(std::vector<Vec3f> &model_pcd, const Scene scene, const ICPConvergenceCriteria criteria = ICPConvergenceCriteria())
{
    RegistrationResult result, backup;
    thrust__pcd2Ab<Scene> per_point(scene);
    for (uint32_t iter = 0; iter <= (uint32_t)criteria.max_iteration_; ++iter) {
        Vec29f Ab;
        for (const Vec3f &p : model_pcd) Ab += per_point(p);
        backup = result;
        const float count = Ab[28], total_error = Ab[27];
        if (count == 0) return result;
        result.fitness_ = count / model_pcd.size();
        result.inlier_rmse_ = std::sqrt(total_error / count);
        if (iter == (uint32_t)criteria.max_iteration_) return result;
        if (std::abs(result.fitness_ - backup.fitness_) < criteria.relative_fitness_ &&
            std::abs(result.inlier_rmse_ - backup.inlier_rmse_) < criteria.relative_rmse_) return result;
        float A[36], b[6];
        for (int i = 0; i < 6; ++i) b[i] = Ab[21 + i];
        int k = 0;
        for (int y = 0; y < 6; ++y) for (int x = y; x < 6; ++x) { A[x + y * 6] = Ab[k]; A[y + x * 6] = Ab[k]; ++k; }
        const Mat4x4f E = eigen_slover_666(A, b);
        for (Vec3f &p : model_pcd) {                          // icp.cpp:47-59 transform_pcd
            const float nx = E[0][0] * p.x + E[0][1] * p.y + E[0][2] * p.z + E[0][3];
            const float ny = E[1][0] * p.x + E[1][1] * p.y + E[1][2] * p.z + E[1][3];
            const float nz = E[2][0] * p.x + E[2][1] * p.y + E[2][2] * p.z + E[2][3];
            p = Vec3f(nx, ny, nz);
        }
        result.transformation_ = E * result.transformation_;
    }
    return result;
}

template <typename... Params> V3f_holder depth2cloud(Params &&...p) { return depth2cloud_cuda(std::forward<Params>(p)...); }                 // icp.h:102-110
template <typename... Params> RegistrationResult ICP_Point2Plane(Params &&...p) { return ICP_Point2Plane_cuda(std::forward<Params>(p)...); }  // icp.h:112-120

}  // namespace cuda_icp
