// drop_in_test.cpp -- the GPU half of the reference's end-to-end driver (test.cpp:22-46,143-172)
// written against the adapter headers with the reference's own names.  OpenCV / Open3D specifics
// of test.cpp (cv::Mat_ initialisers, helper.h viewers, the Open3D comparison, the CUDA warm-up
// calls) are the only lines that differ.  Prints one JSON object that tests/test_drop_in_cpp.py
// compares with the CPU oracle.
#include <cmath>
#include <cstdio>
#include <string>

#include "cuda_icp/icp.h"
#include "cuda_renderer/renderer.h"
#include "pose_renderer.h"

static void matmul3(const float *a, const float *b, float *c)
{   // cv::Mat CV_32F product: double accumulation, float result
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += (double)a[i * 3 + k] * b[k * 3 + j]; c[i * 3 + j] = (float)s; }
}

template <class Scene> static void report(const char *name, const cuda_icp::RegistrationResult &r, bool last)
{
    std::printf("\"%s\": {\"fitness\": %.9g, \"rmse\": %.9g, \"T\": [", name, r.fitness_, r.inlier_rmse_);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) std::printf("%.9g%s", r.transformation_[i][j], (i == 3 && j == 3) ? "" : ", ");
    std::printf("]}%s\n", last ? "" : ",");
}

int main(int argc, char **argv)
{
    std::string prefix = argc > 1 ? argv[1] : "tests/golden/";
    if (pr_init(0) != PR_OK) { std::fprintf(stderr, "%s\n", pr_last_error()); return 2; }   // replaces cudaFree(0)/cublasCreate (test.cpp:12-20)
    int width = 640, height = 480;
    cuda_renderer::Model model(prefix + "obj_06.ply");

    float Kd[9] = { 572.4114f, 0.0f, 325.2611f, 0.0f, 573.57043f, 242.04899f, 0.0f, 0.0f, 1.0f };
    cv::Mat K(3, 3, CV_32F, Kd);
    auto proj = cuda_renderer::compute_proj(K, width, height);

    float R_ren[9] = { 0.34768538f, 0.93761126f, 0.00000000f, 0.70540612f, -0.26157897f, -0.65877056f, -0.61767070f, 0.22904489f, -0.75234390f };
    float t_ren[3] = { 0.0f, 0.0f, 300.0f }, t_ren2[3] = { 20.0f, 20.0f, 320.0f };
    float a = 10.0f / 180.0f * 3.14f;
    float Rx[9] = { 1, 0, 0, 0, std::cos(a), -std::sin(a), 0, std::sin(a), std::cos(a) };
    float Ry[9] = { std::cos(a), 0, std::sin(a), 0, 1, 0, -std::sin(a), 0, std::cos(a) };
    float Rz[9] = { std::cos(a), -std::sin(a), 0, std::sin(a), std::cos(a), 0, 0, 0, 1 };
    float Rzy[9], Rzyx[9], R_ren2[9];
    matmul3(Rz, Ry, Rzy); matmul3(Rzy, Rx, Rzyx); matmul3(Rzyx, R_ren, R_ren2);

    cuda_renderer::Model::mat4x4 mat4, mat4_2;
    mat4.init_from_ptr(R_ren, t_ren);
    mat4_2.init_from_ptr(R_ren2, t_ren2);
    std::vector<cuda_renderer::Model::mat4x4> mat4_v = { mat4, mat4_2 };

    // scene depth = render of pose 2, brought to the host (test.cpp:50,75 uses render_cpu for this)
    std::vector<int> depth_host = cuda_renderer::render_cuda(model.tris, mat4_v, width, height, proj);
    cv::Mat scene_depth(height, width, CV_32S, depth_host.data() + width * height);

    Mat3x3f K_((float *)K.data);
    auto depth_cuda = cuda_renderer::render_cuda_keep_in_gpu(model.tris, mat4_v, width, height, proj);    // test.cpp:143

    long long pr_depth_sum[2], pr_mask_px[2];
    {   // PoseRenderer (pose_renderer.cpp:16-63): cv::Mat poses in, uint16 depth + uint8 mask out
        PoseRenderer pr(prefix + "obj_06.ply");
        pr.set_K_width_height(K, width, height);
        float P0[16], P1[16];
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) { P0[r * 4 + c] = R_ren[r * 3 + c]; P1[r * 4 + c] = R_ren2[r * 3 + c]; } P0[r * 4 + 3] = t_ren[r]; P1[r * 4 + 3] = t_ren2[r]; }
        P0[12] = P0[13] = P0[14] = 0; P0[15] = 1; P1[12] = P1[13] = P1[14] = 0; P1[15] = 1;
        std::vector<cv::Mat> poses = { cv::Mat(4, 4, CV_32F, P0), cv::Mat(4, 4, CV_32F, P1) };
        auto dm = pr.render_depth_mask(poses);
        long long ds[2] = { 0, 0 }, ms[2] = { 0, 0 };
        for (int i = 0; i < 2; ++i) for (int k = 0; k < width * height; ++k) { ds[i] += dm[i][0].ptr<uint16_t>()[k]; ms[i] += dm[i][1].ptr<uint8_t>()[k]; }
        for (int i = 0; i < 2; ++i) { pr_depth_sum[i] = ds[i]; pr_mask_px[i] = ms[i] / 255; }
    }
    std::printf("{\n\"n_triangles\": %zu,\n", model.tris.size());
    std::printf("\"pose_renderer\": {\"depth_sum\": [%lld, %lld], \"mask_px\": [%lld, %lld]},\n", pr_depth_sum[0], pr_depth_sum[1], pr_mask_px[0], pr_mask_px[1]);
    long long sum0 = 0, sum1 = 0;
    for (int i = 0; i < width * height; ++i) { sum0 += depth_host[i]; sum1 += depth_host[width * height + i]; }
    std::printf("\"depth_sum\": [%lld, %lld],\n", sum0, sum1);

    {
        auto pcd1_cuda = cuda_icp::depth2cloud_cuda(depth_cuda.data(), width, height, K_);               // test.cpp:153
        std::printf("\"cloud_points\": %zu,\n", pcd1_cuda.size());
        Scene_projective scene;
        device_vector_holder<::Vec3f> pcd_buffer_cuda, normal_buffer_cuda;
        scene.init_Scene_projective_cuda(scene_depth, K_, pcd_buffer_cuda, normal_buffer_cuda);          // test.cpp:161-163
        auto result_cuda = cuda_icp::ICP_Point2Plane_cuda(pcd1_cuda, scene);                             // test.cpp:172
        report<Scene_projective>("proj_default", result_cuda, false);
    }
    {
        auto pcd1_cuda = cuda_icp::depth2cloud(depth_cuda.data(), width, height, K_);
        Scene_nn scene;
        KDTree_cuda kdtree_cuda;
        scene.init_Scene_nn_cuda(scene_depth, K_, kdtree_cuda);                                          // test.cpp:165-166
        auto result_cuda = cuda_icp::ICP_Point2Plane(pcd1_cuda, scene, cuda_icp::ICPConvergenceCriteria(0.f, 0.f, 20));
        report<Scene_nn>("nn_fixed20", result_cuda, true);
    }
    std::printf("}\n");
    return 0;
}
