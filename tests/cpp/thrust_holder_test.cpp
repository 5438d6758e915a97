// thrust_holder_test.cpp -- compiled by hipcc (rocThrust on the include path): the holders expose thrust::device_ptr like the
// reference's, so the thrust::copy lines of cuda_renderer/test.cpp:90,135 and pose_renderer.cpp:12 build and run unchanged.
// Also exercises the fill constructor (renderer.h:169) and move assignment.
#include <cstdio>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>

#include "cuda_icp/icp.h"
#include "cuda_renderer/renderer.h"

#ifndef POSE_REFINE_HAVE_THRUST
#error "this test must be compiled by hipcc with rocThrust available"
#endif

int main()
{
    if (pr_init(0) != PR_OK) { std::fprintf(stderr, "%s\n", pr_last_error()); return 2; }
    cuda_renderer::device_vector_holder<int> depth(1000, 77);                       // holder(size, init)
    std::vector<int> back(depth.size());
    thrust::copy(depth.begin_thr(), depth.end_thr(), back.begin());                 // cuda_renderer/test.cpp:90
    for (int v : back) if (v != 77) { std::printf("FAIL fill\n"); return 1; }
    cuda_renderer::device_vector_holder<int> other(10, 5);
    other = std::move(depth);                                                       // move assignment
    if (other.size() != 1000 || depth.size() != 0) { std::printf("FAIL move\n"); return 1; }
    std::vector<int> src(1000);
    for (int i = 0; i < 1000; ++i) src[i] = i * 3;
    thrust::copy(src.begin(), src.end(), other.begin_thr());                        // host -> device through the same members
    back = other.download();
    for (int i = 0; i < 1000; ++i) if (back[i] != i * 3) { std::printf("FAIL h2d\n"); return 1; }
    ::device_vector_holder<Vec3f> cloud(16, Vec3f(1.f, 2.f, 3.f));                  // cuda_icp holder, same members (common.h:30-33)
    std::vector<Vec3f> pts(16);
    thrust::copy(cloud.begin_thr(), cloud.end_thr(), pts.begin());
    if (pts[7].y != 2.f) { std::printf("FAIL vec3\n"); return 1; }
    // a scene array edited IN PLACE through the pointers the holders hand out -- the reference's own upload idiom,
    // thrust::copy(host..., buf.begin_thr()) (depth_scene.cu:12-16), and a raw pointer kept from an earlier data() call written through later
    // with a plain hipMemcpy -- must be seen by the next ICP call: the reference reads the arrays at every call (depth_scene.h:29-48).
    // Expected answer: a second pair of holders, freshly allocated, with the same content.
    {
        const int W = 64, H = 48;
        float Kd[9] = { 80.0f, 0.0f, 32.0f, 0.0f, 80.0f, 24.0f, 0.0f, 0.0f, 1.0f };
        Mat3x3f K(Kd);
        auto frame = [&](int shift) { std::vector<int32_t> d((size_t)W * H); for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) d[(size_t)y * W + x] = 500 + shift + (x * 3 + y * 2) % 40; return d; };
        std::vector<int32_t> d0 = frame(0), d1 = frame(6);
        cv::Mat m0(H, W, CV_32S, d0.data()), m1(H, W, CV_32S, d1.data());
        std::vector<Vec3f> p1, n1, cl;
        Scene_projective host1; host1.init_Scene_projective_cpu(m1, K, p1, n1, W, H);
        for (int y = 4; y < H - 4; ++y) for (int x = 4; x < W - 4; ++x) { const float z = (503 + (x * 3 + y * 2) % 40) / 1000.0f; cl.push_back(Vec3f((x - Kd[2]) / Kd[0] * z, (y - Kd[5]) / Kd[4] * z, z)); }
        auto run = [&](Scene_projective &sc) { ::device_vector_holder<Vec3f> c; c.upload(cl); return cuda_icp::ICP_Point2Plane_cuda(c, sc, cuda_icp::ICPConvergenceCriteria(0.f, 0.f, 5)); };
        auto same = [](const cuda_icp::RegistrationResult &a, const cuda_icp::RegistrationResult &b) { return std::memcmp(&a, &b, sizeof a) == 0; };
        ::device_vector_holder<Vec3f> pa, na, pb, nb;
        Scene_projective sa, sb;
        sa.init_Scene_projective_cuda(m0, K, pa, na, W, H);
        sb.init_Scene_projective_cuda(m1, K, pb, nb, W, H);
        const auto want1 = run(sb), got0 = run(sa);                                    // (caches of both scenes are built here)
        if (same(want1, got0)) { std::printf("FAIL the two frames give the same answer\n"); return 1; }
        Vec3f *kept = pa.data();                                                        // a raw pointer, taken BEFORE the calls that follow
        thrust::copy(p1.begin(), p1.end(), pa.begin_thr());                             // frame 1 into scene a's arrays, the reference's idiom
        thrust::copy(n1.begin(), n1.end(), na.begin_thr());
        if (!same(run(sa), want1)) { std::printf("FAIL edit through begin_thr\n"); return 1; }
        std::vector<Vec3f> p0, n0; Scene_projective host0; host0.init_Scene_projective_cpu(m0, K, p0, n0, W, H);
        if (hipMemcpy(kept, p0.data(), p0.size() * sizeof(Vec3f), hipMemcpyHostToDevice) != hipSuccess) { std::printf("FAIL hipMemcpy\n"); return 1; }   // nothing announced
        if (hipMemcpy(sa.normal_ptr, n0.data(), n0.size() * sizeof(Vec3f), hipMemcpyHostToDevice) != hipSuccess) { std::printf("FAIL hipMemcpy\n"); return 1; }
        if (!same(run(sa), got0)) { std::printf("FAIL edit through a kept raw pointer\n"); return 1; }
    }
    std::printf("OK\n");
    return 0;
}
