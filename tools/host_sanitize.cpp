// build: g++ -std=c++17 -O1 -g -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ipose_refine_amd/csrc tools/host_sanitize.cpp pose_refine_amd/csrc/pr_host.cpp -o host_sanitize;  ./host_sanitize [rounds] [seed]
// Sanitizer harness for the host-side C ABI of pr_host.cpp: random and extreme inputs through scene preparation, kd-tree build, solver, proj.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>
#include "pose_refine.h"
namespace prh { void set_error(const char *, ...) {} }
int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(argc > 2 ? atoi(argv[2]) : 1);
    auto U = [&](double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); };
    const float specials[] = { 0.0f, -0.0f, 1.0f, -1.0f, NAN, INFINITY, -INFINITY, 1e30f, -1e30f, 1e-38f, 65535.0f, 2000.0f };
    long errs = 0, oks = 0;
    for (int r = 0; r < rounds; ++r) {
        const int W = 1 + rng() % 120, H = 1 + rng() % 90;
        float K[9] = { (float)U(10, 200), 0, W / 2.0f, 0, (float)U(10, 200), H / 2.0f, 0, 0, 1 };
        if (rng() % 5 == 0) K[rng() % 9] = specials[rng() % 12];
        const bool i32 = rng() & 1;
        std::vector<int32_t> d32((size_t)W * H); std::vector<uint16_t> d16((size_t)W * H);
        for (size_t i = 0; i < d32.size(); ++i) {
            int32_t v = (rng() % 4) ? (int32_t)(300 + rng() % 2500) : 0;
            if (rng() % 50 == 0) { const int32_t ex[] = { -1, -2147483647 - 1, 2147483647, 65535, 65536, 70000, 1999, 2000, 2001, 49, 50 }; v = ex[rng() % 11]; }
            d32[i] = v; d16[i] = (uint16_t)std::min(65535, std::max(0, v));
        }
        std::vector<pr_vec3> pcd((size_t)W * H), nrm((size_t)W * H);
        (pr_scene_proj_prepare(i32 ? (const void *)d32.data() : (const void *)d16.data(), i32, K, (size_t)W, (size_t)H, pcd.data(), nrm.data()) == PR_OK ? oks : errs)++;
        (pr_get_normal(d16.data(), W, H, K, nrm.data()) == PR_OK ? oks : errs)++;
        const int ml = 1 + rng() % 20;
        std::vector<pr_kdnode> nodes(2 * (size_t)W * H + 1);
        uint32_t np = 0, nn = 0;
        (pr_scene_nn_prepare(i32 ? (const void *)d32.data() : (const void *)d16.data(), i32, K, W, H, ml, pcd.data(), nrm.data(), nodes.data(), nodes.size(), &np, &nn) == PR_OK ? oks : errs)++;
        // kd-tree build on arbitrary points (duplicates, lattice, a few non-finite ones)
        const size_t n = rng() % 400;
        std::vector<pr_vec3> p(n), q(n);
        for (size_t i = 0; i < n; ++i) { p[i] = pr_vec3{ (float)std::round(U(-5, 5)) / 4, (float)std::round(U(-5, 5)) / 4, (float)U(0, 1) }; q[i] = pr_vec3{ 0, 0, 1 }; }
        if (n && rng() % 4 == 0) { pr_vec3 &v = p[rng() % n]; (&v.x)[rng() % 3] = specials[rng() % 12]; }
        std::vector<pr_kdnode> kn(2 * n + 1);
        uint32_t cnt = 0;
        (pr_kdtree_build(p.data(), q.data(), n, ml, kn.data(), kn.size(), &cnt) == PR_OK ? oks : errs)++;
        // solver and projection on random / extreme numbers
        float A[36], b[6]; pr_mat4 T;
        for (int i = 0; i < 36; ++i) A[i] = (rng() % 10) ? (float)U(-100, 100) : specials[rng() % 12];
        for (int i = 0; i < 6; ++i) b[i] = (rng() % 10) ? (float)U(-1, 1) : specials[rng() % 12];
        pr_solve_666(A, b, &T);
        pr_mat4 P; pr_compute_proj(K, W, H, (float)U(0.1, 20), (float)U(100, 1e5), &P);
        uint32_t f = 0, c = 0; pr_shard_range(rng(), rng() % 9, 1 + rng() % 9, &f, &c);
    }
    printf("rounds %d, calls ok %ld, refused %ld\n", rounds, oks, errs);
    return 0;
}
