// icp.h -- cuda_icp:: API of cuda_icp/icp.h:16-120 over the C ABI (HIP build: the reference's
// CUDA_ON configuration).  ICP_Point2Plane / depth2cloud dispatch to the device versions.
#pragma once
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

template <typename... Params> V3f_holder depth2cloud(Params &&...p) { return depth2cloud_cuda(std::forward<Params>(p)...); }                 // icp.h:102-110
template <typename... Params> RegistrationResult ICP_Point2Plane(Params &&...p) { return ICP_Point2Plane_cuda(std::forward<Params>(p)...); }  // icp.h:112-120

}  // namespace cuda_icp
