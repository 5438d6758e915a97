// cpu_twins_test.cpp -- the CPU half of the reference's end-to-end driver (test.cpp:48-129): render_cpu,
// depth2cloud_cpu, init_Scene_*_cpu, ICP_Point2Plane_cpu through the adapter headers.  No GPU needed.
#include <cmath>
#include <cstdio>
#include <string>

#include "cuda_icp/icp.h"
#include "cuda_renderer/renderer.h"

static void matmul3(const float *a, const float *b, float *c)
{ for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += (double)a[i * 3 + k] * b[k * 3 + j]; c[i * 3 + j] = (float)s; } }

static void report(const char *name, const cuda_icp::RegistrationResult &r, bool last)
{
    std::printf("\"%s\": {\"fitness\": %.9g, \"rmse\": %.9g, \"T\": [", name, r.fitness_, r.inlier_rmse_);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) std::printf("%.9g%s", r.transformation_[i][j], (i == 3 && j == 3) ? "" : ", ");
    std::printf("]}%s\n", last ? "" : ",");
}

int main(int argc, char **argv)
{
    std::string prefix = argc > 1 ? argv[1] : "tests/golden/";
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
    mat4.init_from_ptr(R_ren, t_ren); mat4_2.init_from_ptr(R_ren2, t_ren2);
    std::vector<cuda_renderer::Model::mat4x4> mat4_v = { mat4, mat4_2 };

    std::vector<int> depth_cpu = cuda_renderer::render_cpu(model.tris, mat4_v, width, height, proj);     // test.cpp:50
    long long sum0 = 0, sum1 = 0;
    for (int i = 0; i < width * height; ++i) { sum0 += depth_cpu[i]; sum1 += depth_cpu[width * height + i]; }
    cuda_renderer::Model::ROI roi = { 160, 80, 320, 240 };
    std::vector<int> roi_cpu = cuda_renderer::render_cpu(model.tris, mat4_v, width, height, proj, roi);
    long long sumr = 0; for (int v : roi_cpu) sumr += v;
    std::printf("{\n\"depth_sum\": [%lld, %lld], \"roi_sum\": %lld,\n", sum0, sum1, sumr);

    Mat3x3f K_((float *)K.data);
    std::vector<::Vec3f> pcd1 = cuda_icp::depth2cloud_cpu(depth_cpu.data(), width, height, K_);          // test.cpp:72
    std::printf("\"cloud_points\": %zu,\n", pcd1.size());
    cv::Mat scene_depth(height, width, CV_32S, depth_cpu.data() + width * height);
    {
        std::vector<::Vec3f> cloud = pcd1;
        Scene_projective scene;
        std::vector<::Vec3f> pcd_buffer, normal_buffer;
        scene.init_Scene_projective_cpu(scene_depth, K_, pcd_buffer, normal_buffer);
        report("proj_default", cuda_icp::ICP_Point2Plane_cpu(cloud, scene), false);
    }
    {
        std::vector<::Vec3f> cloud = pcd1;
        Scene_nn scene;
        KDTree_cpu kdtree_cpu;
        scene.init_Scene_nn_cpu(scene_depth, K_, kdtree_cpu);                                            // test.cpp:84-86
        std::printf("\"kd_nodes\": %zu,\n", kdtree_cpu.nodes.size());
        report("nn_default", cuda_icp::ICP_Point2Plane_cpu(cloud, scene), true);                         // test.cpp:129
    }
    std::printf("}\n");
    return 0;
}
