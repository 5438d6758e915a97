// build: g++ -std=c++17 -O1 -g -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ipose_refine_amd/csrc tools/mesh_sanitize.cpp pose_refine_amd/csrc/pr_host.cpp -o mesh_asan;  PR_MESH_HARNESS=./mesh_asan python tools/fuzz_mesh_files.py
#include <cstdio>
#include <vector>
#include "pose_refine.h"
int main(int argc, char **argv)
{
    int ok = 0, refused = 0;
    for (int i = 1; i < argc; ++i) {
        size_t nt = 0, nv = 0;
        if (pr_mesh_count(argv[i], &nt, &nv) != PR_OK) { ++refused; continue; }
        std::vector<pr_triangle> t(nt); std::vector<pr_vec3> v(nv); std::vector<int32_t> f(3 * nt);
        float lo[3], hi[3];
        if (pr_mesh_load(argv[i], t.data(), nt, &nt, v.data(), nv, &nv, f.data(), lo, hi) != PR_OK) { ++refused; continue; }
        ++ok;
    }
    printf("ok %d refused %d\n", ok, refused);
    return 0;
}
#include <cstdarg>
namespace prh { void set_error(const char *, ...) {} }
