// scene_init_time.cpp -- how long the reference-named scene initialisations take through the adapters (include/cuda_icp/scene/...):
// init_Scene_projective_cuda / init_Scene_nn_cuda on a host cv::Mat, device route (default) against the reference's CPU route
// (-DPOSE_REFINE_CPU_SCENE_PREP), for the bench's scene and a frame-filling one.
//   g++ -std=c++14 -O2 -Iinclude tools/scene_init_time.cpp -o /tmp/sit -Lpose_refine_amd/lib -lpose_refine_hip -Wl,-rpath,$PWD/pose_refine_amd/lib && /tmp/sit tests/golden/
#include <chrono>
#include <cmath>
#include <cstdio>
#include <string>
#include "cuda_icp/icp.h"
#include "cuda_renderer/renderer.h"

static double ms_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }

int main(int argc, char **argv)
{
    std::string prefix = argc > 1 ? argv[1] : "tests/golden/";
    if (pr_init(0) != PR_OK) { std::fprintf(stderr, "%s\n", pr_last_error()); return 2; }
    const int width = 640, height = 480;
    cuda_renderer::Model model(prefix + "obj_06.ply");
    float Kd[9] = { 572.4114f, 0.0f, 325.2611f, 0.0f, 573.57043f, 242.04899f, 0.0f, 0.0f, 1.0f };
    cv::Mat K(3, 3, CV_32F, Kd);
    auto proj = cuda_renderer::compute_proj(K, width, height);
    float R[9] = { 0.34768538f, 0.93761126f, 0.0f, 0.70540612f, -0.26157897f, -0.65877056f, -0.61767070f, 0.22904489f, -0.75234390f }, t[3] = { 0.0f, 0.0f, 300.0f };
    cuda_renderer::Model::mat4x4 m; m.init_from_ptr(R, t);
    std::vector<cuda_renderer::Model::mat4x4> mv = { m };
    std::vector<int> object = cuda_renderer::render_cuda(model.tris, mv, width, height, proj);
    std::vector<int> full(object);
    for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) if (full[y * width + x] == 0) full[y * width + x] = (int)(900 + 0.05 * x + 0.03 * y + 3.0 * std::sin(x / 17.0) * std::cos(y / 23.0));
    Mat3x3f K_((float *)K.data);
#ifdef POSE_REFINE_CPU_SCENE_PREP
    const char *route = "CPU preparation + uploads (the reference's route)";
#else
    const char *route = "depth image up, preparation on the device";
#endif
    for (int which = 0; which < 2; ++which) {
        cv::Mat depth(height, width, CV_32S, which ? full.data() : object.data());
        double best_p = 1e30, best_n = 1e30; size_t npts = 0, nnodes = 0;
        for (int rep = 0; rep < 6; ++rep) {
            {
                Scene_projective scene; device_vector_holder<::Vec3f> pcd, nrm;
                auto t0 = std::chrono::steady_clock::now();
                scene.init_Scene_projective_cuda(depth, K_, pcd, nrm);
                pr_sync();
                best_p = std::min(best_p, ms_since(t0));
            }
            {
                Scene_nn scene; KDTree_cuda kd;
                auto t0 = std::chrono::steady_clock::now();
                scene.init_Scene_nn_cuda(depth, K_, kd);
                pr_sync();
                best_n = std::min(best_n, ms_since(t0));
                npts = kd.pcd_buffer.size(); nnodes = kd.nodes.size();
            }
        }
        std::printf("%-28s %s: init_Scene_projective_cuda %.2f ms, init_Scene_nn_cuda %.2f ms (%zu points, %zu nodes)\n", which ? "object + wall (frame filled)" : "object alone", route, best_p, best_n, npts, nnodes);
    }
    return 0;
}
