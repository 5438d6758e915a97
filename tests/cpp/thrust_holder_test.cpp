// thrust_holder_test.cpp -- compiled by hipcc (rocThrust on the include path): the holders expose thrust::device_ptr like the
// reference's, so the thrust::copy lines of cuda_renderer/test.cpp:90,135 and pose_renderer.cpp:12 build and run unchanged.
// Also exercises the fill constructor (renderer.h:169) and move assignment.
#include <cstdio>
#include <vector>

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
    std::printf("OK\n");
    return 0;
}
