// geometry_ref.cpp -- driver around the reference's OWN header, compiled from where it lies:
//     g++ -std=c++14 -O2 -ffp-contract=off -I/root/reference/cuda_icp oracle/ref_driver/geometry_ref.cpp -o oracle/_ref/geometry_ref
// (oracle/Makefile target `ref`).  cuda_icp/geometry.h without CUDA_ON includes only <cmath> <vector> <cassert> <iostream>, so
// it is the one piece of the reference this image can build without stand-ins (icp.cpp, renderer.cpp and the scene sources need
// OpenCV / Eigen / assimp).  This file is the repo's own code; nothing of the reference is copied.
//
// Emits one JSON object: for seeded float inputs, the bit patterns of
//   mat4x4 * mat4x4      (geometry.h:292-298 -- what `result.transformation_ = extrinsic * result.transformation_` uses, icp.cu:212)
//   mat4x4 * vec4, mat3x3 * vec3, vec * vec (dot), cross, transpose, identity, embed / proj, vec3i(vec3f) rounding,
//   det / cofactor / get_minor / adjugate / invert_transpose / invert (geometry.h:164-179,222-262)
// tests/golden/geometry_h.json is this output; tests compare the oracle, the C++ adapters and the solver's mat4_mul with it.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "geometry.h"

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
template <size_t R, size_t C> static void dump(const char *key, const mat<R, C, float> &m, bool last = false)
{
    std::printf("\"%s\": [", key);
    for (size_t i = 0; i < R; ++i) for (size_t j = 0; j < C; ++j) std::printf("%u%s", bits(m[i][j]), (i + 1 == R && j + 1 == C) ? "" : ", ");
    std::printf("]%s", last ? "" : ", ");
}
template <size_t D> static void dumpv(const char *key, const vec<D, float> &v, bool last = false)
{
    std::printf("\"%s\": [", key);
    for (size_t i = 0; i < D; ++i) std::printf("%u%s", bits(v[i]), i + 1 == D ? "" : ", ");
    std::printf("]%s", last ? "" : ", ");
}

int main()
{
    std::mt19937 rng(20240928u);
    auto uni = [&](float lo, float hi) { return lo + (hi - lo) * (float)(rng() >> 8) * (1.0f / 16777216.0f); };
    std::printf("{\"_provenance\": \"bit patterns computed by /root/reference/cuda_icp/geometry.h itself (compiled verbatim, no CUDA_ON) through oracle/ref_driver/geometry_ref.cpp; regenerate with `make -C oracle ref fixtures`\",\n\"cases\": [\n");
    const int n_cases = 24;
    for (int c = 0; c < n_cases; ++c) {
        float a[16], b[16], v4[4], v3a[3], v3b[3];
        const bool rigid_like = c < 12;                              // near-rigid transforms (what ICP multiplies) and generic matrices
        for (int i = 0; i < 16; ++i) { a[i] = rigid_like ? uni(-1.f, 1.f) : uni(-1000.f, 1000.f); b[i] = rigid_like ? uni(-1.f, 1.f) : uni(-3.f, 3.f); }
        if (rigid_like) { a[12] = a[13] = a[14] = 0; a[15] = 1; b[12] = b[13] = b[14] = 0; b[15] = 1; a[3] = uni(-.05f, .05f); a[7] = uni(-.05f, .05f); a[11] = uni(-.05f, .05f); }
        for (int i = 0; i < 4; ++i) v4[i] = uni(-2.f, 2.f);
        for (int i = 0; i < 3; ++i) { v3a[i] = uni(-0.5f, 0.5f); v3b[i] = uni(-0.5f, 0.5f); }
        Mat4x4f A(a), B(b);
        Mat3x3f K(a);                                                 // first 9 values, row-major
        Vec4f V4; for (int i = 0; i < 4; ++i) V4[i] = v4[i];
        Vec3f P(v3a[0], v3a[1], v3a[2]), Q(v3b[0], v3b[1], v3b[2]);
        std::printf("{");
        dump("A", A); dump("B", B); dumpv("v4", V4); dumpv("p", P); dumpv("q", Q);
        dump("A_mul_B", A * B);
        dump("A_transpose", A.transpose());
        dumpv("A_mul_v4", A * V4);
        dumpv("K_mul_p", K * P);
        dumpv("cross_pq", cross(P, Q));
        dumpv("p_plus_q", P + Q); dumpv("p_minus_q", P - Q); dumpv("p_times_s", P * v4[0]); dumpv("p_over_s", P / v4[1]);
        dumpv("embed4_p", embed<4>(P)); dumpv("proj3_v4", proj<3>(V4));
        dump("K_adjugate", K.adjugate()); dump("K_invert_transpose", K.invert_transpose()); dump("K_invert", K.invert());
        dump("A_adjugate", A.adjugate()); dump("A_invert", A.invert()); dump("A_minor_1_2", A.get_minor(1, 2));
        dumpv("A_col2", A.col(2));
        std::printf("\"det_K\": %u, \"det_A\": %u, \"cofactor_A_2_1\": %u, ", bits(K.det()), bits(A.det()), bits(A.cofactor(2, 1)));
        Vec3f scaled = P * 1000.0f;
        Vec3i rounded(scaled);                                        // vec<3,int>(vec<3,float>): int(v + .5f)
        std::printf("\"vec3i_of_1000p\": [%d, %d, %d], ", rounded.x, rounded.y, rounded.z);
        std::printf("\"dot_pq\": %u, \"norm_p\": %u", bits(P * Q), bits(P.norm()));
        std::printf("}%s\n", c + 1 == n_cases ? "" : ",");
    }
    Mat4x4f I = Mat4x4f::identity();
    std::printf("],\n");
    dump("identity4", I, true);
    std::printf("}\n");
    return 0;
}
