// pose_renderer.h -- ::PoseRenderer (pose_renderer.h:9-32, pose_renderer.cpp:3-63): keeps the triangles of one model on
// the device, converts cv::Mat poses, renders a batch and returns cv::Mat depth / mask images.  Header-only over the
// adapters; `down_sample` keeps the reference's behaviour (renders at width/down_sample with the full-size projection).
#pragma once
#include <string>
#include <vector>

#include "cuda_renderer/renderer.h"

class PoseRenderer {
public:
    cv::Mat K;
    int width = 0, height = 0;
    cuda_renderer::Model model;
    cuda_renderer::device_vector_holder<cuda_renderer::Model::Triangle> tris;
    cuda_renderer::Model::mat4x4 proj_mat;

    explicit PoseRenderer(std::string model_path, cv::Mat /*depth*/ = cv::Mat(), cv::Mat /*K*/ = cv::Mat()) : model(model_path) { tris.upload(model.tris); }
    void set_K_width_height(cv::Mat K_, int width_, int height_)
    {
        assert(K_.type() == CV_32F);
        K = K_; width = width_; height = height_;
        proj_mat = cuda_renderer::compute_proj(K, width, height);
    }
    template <typename F> auto render_what(F f, std::vector<cv::Mat> &init_poses, float down_sample = 1)
    {
        const int w = int(width / down_sample), h = int(height / down_sample);
        std::vector<cuda_renderer::Model::mat4x4> mat4_v(init_poses.size());
        for (size_t i = 0; i < init_poses.size(); ++i) mat4_v[i].init_from_cv(init_poses[i]);
        auto depths = cuda_renderer::render(tris, mat4_v, w, h, proj_mat);
        return f(depths, w, h, init_poses.size());
    }
    std::vector<cv::Mat> render_depth(std::vector<cv::Mat> &poses, float down_sample = 1) { return render_what(cuda_renderer::raw2depth_uint16_cuda, poses, down_sample); }
    std::vector<cv::Mat> render_mask(std::vector<cv::Mat> &poses, float down_sample = 1) { return render_what(cuda_renderer::raw2mask_uint8_cuda, poses, down_sample); }
    std::vector<std::vector<cv::Mat>> render_depth_mask(std::vector<cv::Mat> &poses, float down_sample = 1) { return render_what(cuda_renderer::raw2depth_mask_cuda, poses, down_sample); }
};
