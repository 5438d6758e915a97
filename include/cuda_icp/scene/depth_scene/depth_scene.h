// depth_scene.h -- ::Scene_projective (cuda_icp/scene/depth_scene/depth_scene.h:7-48): a non-owning view
// {width, height, max_dist_diff, K, pcd_ptr, normal_ptr} over user-owned buffers.  72 bytes, same
// field order as the reference, bit-compatible with pr_scene_proj of the C ABI.
#pragma once
#include <iostream>
#include "../common.h"

struct Scene_projective {
    size_t width = 640, height = 480;
    float max_dist_diff = 0.1f;
    Mat3x3f K;
    Vec3f *pcd_ptr = nullptr;
    Vec3f *normal_ptr = nullptr;

    void init_Scene_projective_cpu(cv::Mat &scene_depth, Mat3x3f &scene_K, std::vector<Vec3f> &pcd_buffer, std::vector<Vec3f> &normal_buffer,
                                   size_t width_ = 640, size_t height_ = 480, float max_dist_diff_ = 0.1f)
    {
        assert(scene_depth.type() == CV_16U || scene_depth.type() == CV_32S);
        K = scene_K; width = width_; height = height_; max_dist_diff = max_dist_diff_;
        pcd_buffer.assign(width * height, Vec3f()); normal_buffer.assign(width * height, Vec3f());
        pose_refine_detail::must(pr_scene_proj_prepare(scene_depth.data, scene_depth.type() == CV_32S, K.data(), width, height,
                                                       reinterpret_cast<pr_vec3 *>(pcd_buffer.data()), reinterpret_cast<pr_vec3 *>(normal_buffer.data())),
                                 "pr_scene_proj_prepare");
        pcd_ptr = pcd_buffer.data(); normal_ptr = normal_buffer.data();
    }
    // depth_scene.cu:3-20 = CPU preparation + two uploads (7.4 MB).  Here the depth image goes up (1.2 MB) and back-projection + normals run as one kernel:
    // the same arrays, bit for bit (tests/test_scene_prep_gpu.py).  -DPOSE_REFINE_CPU_SCENE_PREP keeps the reference's route.
    void init_Scene_projective_cuda(cv::Mat &scene_depth, Mat3x3f &scene_K, device_vector_holder<Vec3f> &pcd_buffer,
                                    device_vector_holder<Vec3f> &normal_buffer, size_t width_ = 640, size_t height_ = 480, float max_dist_diff_ = 0.1f)
    {
#ifdef POSE_REFINE_CPU_SCENE_PREP
        std::vector<Vec3f> p, n;
        init_Scene_projective_cpu(scene_depth, scene_K, p, n, width_, height_, max_dist_diff_);
        pcd_buffer.upload(p); normal_buffer.upload(n);
        pcd_ptr = pcd_buffer.data(); normal_ptr = normal_buffer.data();
#else
        bool is32 = false;
        device_vector_holder<unsigned char> depth_dev;
        pose_refine_detail::upload_depth(scene_depth, width_, height_, depth_dev, is32);      // type / size checked at run time; pitched or wider images row by row
        if (is32) init_Scene_projective_device(reinterpret_cast<int32_t *>(depth_dev.__gpu_memory), scene_K, pcd_buffer, normal_buffer, width_, height_, max_dist_diff_);
        else init_Scene_projective_device(reinterpret_cast<uint16_t *>(depth_dev.__gpu_memory), scene_K, pcd_buffer, normal_buffer, width_, height_, max_dist_diff_);
#endif
    }
    // the same initialisation with the depth image already on the device (T = int32_t or uint16_t);
    // back-projection and normals run as one kernel, bit-identical to the CPU preparation.
    template <class T>
    void init_Scene_projective_device(T *scene_depth_dev, Mat3x3f &scene_K, device_vector_holder<Vec3f> &pcd_buffer,
                                      device_vector_holder<Vec3f> &normal_buffer, size_t width_ = 640, size_t height_ = 480, float max_dist_diff_ = 0.1f)
    {
        static_assert(sizeof(T) == 4 || sizeof(T) == 2, "depth must be int32 or uint16");
        K = scene_K; width = width_; height = height_; max_dist_diff = max_dist_diff_;
        pcd_buffer.__malloc(width * height); normal_buffer.__malloc(width * height);
        pose_refine_detail::must(pr_scene_proj_prepare_dev(scene_depth_dev, sizeof(T) == 4, K.data(), width, height,
                                                           reinterpret_cast<pr_vec3 *>(pcd_buffer.data()), reinterpret_cast<pr_vec3 *>(normal_buffer.data())),
                                 "pr_scene_proj_prepare_dev");
        pcd_ptr = pcd_buffer.data(); normal_ptr = normal_buffer.data();
    }
    // depth_scene.h:29-48 (host evaluation; only meaningful when the pointers are host pointers)
    void query(const Vec3f &src, Vec3f &dst, Vec3f &nrm, bool &valid) const
    {
        Vec3i q = pcd2dep(src, K);
        valid = false;
        if (q.x < 0 || q.y < 0 || (size_t)q.x >= width || (size_t)q.y >= height) return;
        size_t idx = q.x + q.y * width;
        dst = pcd_ptr[idx];
        if (dst.z <= 0 || std__abs(src.z - dst.z) > max_dist_diff) return;
        valid = true; nrm = normal_ptr[idx];
    }
    pr_scene_proj c_view() const
    {
        pr_scene_proj s;
        s.width = width; s.height = height; s.max_dist_diff = max_dist_diff;
        for (int i = 0; i < 9; ++i) s.K[i] = K.data()[i];
        s.pcd = reinterpret_cast<const pr_vec3 *>(pcd_ptr);
        s.normal = reinterpret_cast<const pr_vec3 *>(normal_ptr);
        return s;
    }
};
static_assert(sizeof(Scene_projective) == 72, "Scene_projective is passed by value; keep the reference's size");
