// edge_cases_test.cpp -- the adapters with what an application can hand them at its edges: no hypotheses, a render in which nothing is
// visible, clouds without points, default-constructed holders.  The reference answers with empty vectors / the identity (icp.cu:183);
// a failing call would end this process through the adapters' error exit.  Prints one JSON object.
#include <cstdio>
#include <string>
#include <vector>

#include "cuda_icp/icp.h"
#include "cuda_renderer/renderer.h"
#include "pose_renderer.h"

int main(int argc, char **argv)
{
    std::string prefix = argc > 1 ? argv[1] : "tests/golden/";
    if (pr_init(0) != PR_OK) { std::fprintf(stderr, "%s\n", pr_last_error()); return 2; }
    const int width = 640, height = 480;
    cuda_renderer::Model model(prefix + "obj_06.ply");
    float Kd[9] = { 572.4114f, 0.0f, 325.2611f, 0.0f, 573.57043f, 242.04899f, 0.0f, 0.0f, 1.0f };
    cv::Mat K(3, 3, CV_32F, Kd);
    auto proj = cuda_renderer::compute_proj(K, width, height);
    Mat3x3f K_((float *)K.data);

    // no hypotheses
    std::vector<cuda_renderer::Model::mat4x4> none;
    std::vector<int> d0 = cuda_renderer::render_cuda(model.tris, none, width, height, proj);
    auto g0 = cuda_renderer::render_cuda_keep_in_gpu(model.tris, none, width, height, proj);
    // a hypothesis far off to the side: nothing visible
    float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t_off[3] = { 5.0e6f, 0.0f, 300.0f }, t_on[3] = { 0.0f, 0.0f, 300.0f };
    cuda_renderer::Model::mat4x4 off, on;
    off.init_from_ptr(R, t_off); on.init_from_ptr(R, t_on);
    std::vector<cuda_renderer::Model::mat4x4> two = { off, on };
    auto g2 = cuda_renderer::render_cuda_keep_in_gpu(model.tris, two, width, height, proj);
    auto empty_cloud = cuda_icp::depth2cloud_cuda(g2.data(), width, height, K_);                    // image 0: empty
    auto full_cloud = cuda_icp::depth2cloud_cuda(g2.data() + (size_t)width * height, width, height, K_);
    std::vector<int> scene_host = cuda_renderer::render_cuda(model.tris, two, width, height, proj);
    cv::Mat scene_depth(height, width, CV_32S, scene_host.data() + (size_t)width * height);

    Scene_projective sp;
    device_vector_holder<::Vec3f> pcd_buf, nrm_buf;
    sp.init_Scene_projective_cuda(scene_depth, K_, pcd_buf, nrm_buf);
    Scene_nn sn; KDTree_cuda kd;
    sn.init_Scene_nn_cuda(scene_depth, K_, kd);

    device_vector_holder<::Vec3f> never_filled;                                                     // default-constructed: no memory at all
    auto r1 = cuda_icp::ICP_Point2Plane_cuda(empty_cloud, sp);
    auto r2 = cuda_icp::ICP_Point2Plane_cuda(empty_cloud, sn);
    auto r3 = cuda_icp::ICP_Point2Plane_cuda(never_filled, sp);
    auto r4 = cuda_icp::ICP_Point2Plane_cuda(never_filled, sn);
    auto r5 = cuda_icp::ICP_Point2Plane_cuda(full_cloud, sp);

    PoseRenderer pr(prefix + "obj_06.ply");
    pr.set_K_width_height(K, width, height);
    std::vector<cv::Mat> no_poses;
    auto dm = pr.render_depth_mask(no_poses);

    auto ident = [](const cuda_icp::RegistrationResult &r) { bool ok = true; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) ok = ok && r.transformation_[i][j] == (i == j ? 1.0f : 0.0f); return ok && r.fitness_ == 0.0f && r.inlier_rmse_ == 0.0f; };
    std::printf("{\"render_none\": %zu, \"keep_none\": %zu, \"empty_cloud\": %zu, \"full_cloud\": %zu, \"identity\": [%d, %d, %d, %d], \"full_fitness\": %.6f, \"pose_renderer_none\": %zu}\n",
                d0.size(), g0.size(), empty_cloud.size(), full_cloud.size(), (int)ident(r1), (int)ident(r2), (int)ident(r3), (int)ident(r4), r5.fitness_, dm.size());
    return 0;
}
