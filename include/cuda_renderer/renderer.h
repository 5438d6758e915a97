// renderer.h -- cuda_renderer:: API of cuda_renderer/renderer.h:25-248 over the C ABI.
// Model keeps the public members the path uses (tris, vertices, faces, nested PODs); the assimp
// scene graph members are gone (own PLY / OBJ / glTF readers behind pr_mesh_load; the node walk of recursive_render happens there).
#pragma once
#include <cassert>
#include <cstring>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "pose_refine.h"
#include "pose_refine/cv_compat.h"
#include "pose_refine/thrust_compat.h"

namespace cuda_renderer {

class Model {
public:
    struct int3 { int v0, v1, v2; };
    struct ROI { int x, y, width, height; };
    struct float3 { float x, y, z; };
    struct Triangle { float3 v0, v1, v2; };
    struct mat4x4 {                               // row-major a0..d3, identity by default (renderer.h:69-141)
        float a0 = 1, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 1, b2 = 0, b3 = 0, c0 = 0, c1 = 0, c2 = 1, c3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 1;
        void t() { std::swap(a1, b0); std::swap(a2, c0); std::swap(a3, d0); std::swap(b2, c1); std::swap(b3, d1); std::swap(c3, d2); }
        void init_from_ptr(const float *d) { float *m = &a0; for (int i = 0; i < 16; ++i) m[i] = d[i]; }
        void init_from_ptr(const float *R, const float *tr) { a0 = R[0]; a1 = R[1]; a2 = R[2]; a3 = tr[0]; b0 = R[3]; b1 = R[4]; b2 = R[5]; b3 = tr[1]; c0 = R[6]; c1 = R[7]; c2 = R[8]; c3 = tr[2]; }
        void init_from_cv(const cv::Mat &pose) { assert(pose.type() == CV_32F); init_from_ptr(pose.ptr<float>()); }
        void init_from_cv(const cv::Mat &R, const cv::Mat &tr) { assert(R.type() == CV_32F && tr.type() == CV_32F); init_from_ptr(R.ptr<float>(), tr.ptr<float>()); d0 = d1 = d2 = 0; d3 = 1; }
    };
    Model() {}
    ~Model() {}
    explicit Model(const std::string &fileName) { LoadModel(fileName); }
    void LoadModel(const std::string &fileName)              // renderer.cpp:16-58 (PLY / OBJ / glTF through pr_mesh_load instead of assimp)
    {
        size_t nt = 0, nv = 0;
        if (pr_mesh_count(fileName.c_str(), &nt, &nv) != PR_OK) { std::cerr << pr_last_error() << std::endl; std::exit(1); }
        tris.resize(nt); faces.resize(nt); vertices.resize(nv);
        if (pr_mesh_load(fileName.c_str(), reinterpret_cast<pr_triangle *>(tris.data()), nt, &nt, reinterpret_cast<pr_vec3 *>(vertices.data()), nv, &nv,
                         reinterpret_cast<int32_t *>(faces.data()), &bbox_min.x, &bbox_max.x) != PR_OK) { std::cerr << pr_last_error() << std::endl; std::exit(1); }
        std::cout << "load model success    " << std::endl;
        std::cout << "face(triangles) nums: " << faces.size() << std::endl;
        std::cout << "       vertices nums: " << vertices.size() << std::endl;
        if (faces.size() > 10000) std::cout << "you may want tools like meshlab to simplify models to speed up rendering" << std::endl;
        std::cout << "------------------------------------\n" << std::endl;
    }
    // wanted data (renderer.h:143-147); bbox_* are aiVector3D in the reference: same .x / .y / .z access
    std::vector<Triangle> tris;
    std::vector<float3> vertices;
    std::vector<int3> faces;
    float3 bbox_min{ 0, 0, 0 }, bbox_max{ 0, 0, 0 };
};
static_assert(sizeof(Model::int3) == 12 && sizeof(Model::float3) == sizeof(pr_vec3), "POD layouts");
static_assert(sizeof(Model::Triangle) == sizeof(pr_triangle) && sizeof(Model::mat4x4) == sizeof(pr_mat4) && sizeof(Model::ROI) == sizeof(pr_roi), "POD layouts");

// renderer.h:161-187 / renderer.cu:15-50.  Move-only (the reference returns it by value relying on copy elision).
// begin_thr() / end_thr() (renderer.h:175-177) return thrust::device_ptr<T>: they exist when this header is compiled by hipcc with
// rocThrust on the include path, so `thrust::copy(h.begin_thr(), h.end_thr(), host.begin())` (cuda_renderer/test.cpp:90,135,
// pose_renderer.cpp:12) builds unchanged there; a plain host compiler has no Thrust and uses upload() / download(), which are the
// same two copies.
template <typename T> class device_vector_holder {
public:
    T *__gpu_memory = nullptr; size_t __size = 0; bool valid = false;
    device_vector_holder() {}
    explicit device_vector_holder(size_t n) { __malloc(n); }
    device_vector_holder(size_t n, T init) { __malloc(n); fill(init); }               // renderer.h:169, renderer.cu:29-34 thrust::fill
    device_vector_holder(const device_vector_holder &) = delete;
    device_vector_holder &operator=(const device_vector_holder &) = delete;
    device_vector_holder(device_vector_holder &&o) noexcept : __gpu_memory(o.__gpu_memory), __size(o.__size), valid(o.valid) { o.__gpu_memory = nullptr; o.valid = false; o.__size = 0; }
    device_vector_holder &operator=(device_vector_holder &&o) noexcept
    { if (this != &o) { __free(); __gpu_memory = o.__gpu_memory; __size = o.__size; valid = o.valid; o.__gpu_memory = nullptr; o.__size = 0; o.valid = false; } return *this; }
    ~device_vector_holder() { __free(); }
    // A non-const accessor hands out a pointer the caller may WRITE through (thrust::copy(..., begin_thr()), a kernel of its own), and the
    // library caches forms derived from scene arrays by address: every such hand-out is announced (pr_invalidate: a host-side generation
    // bump), the const overloads are for reading.  A pointer that is kept and written through LATER is caught by the full-array fingerprint
    // every synchronous ICP / refine call compares (pose_refine.h "Caches"): the reference reads the arrays at every call.
    T *data() { touched(); return __gpu_memory; }
    T *begin() { touched(); return __gpu_memory; }
    T *end() { touched(); return __gpu_memory + __size; }
    const T *data() const { return __gpu_memory; }
    const T *begin() const { return __gpu_memory; }
    const T *end() const { return __gpu_memory + __size; }
    size_t size() const { return __size; }
#ifdef POSE_REFINE_HAVE_THRUST
    thrust::device_ptr<T> data_thr() { touched(); return thrust::device_ptr<T>(__gpu_memory); }
    thrust::device_ptr<T> begin_thr() { touched(); return thrust::device_ptr<T>(__gpu_memory); }
    thrust::device_ptr<T> end_thr() { touched(); return thrust::device_ptr<T>(__gpu_memory + __size); }
    thrust::device_ptr<const T> data_thr() const { return thrust::device_ptr<const T>(__gpu_memory); }
    thrust::device_ptr<const T> begin_thr() const { return thrust::device_ptr<const T>(__gpu_memory); }
    thrust::device_ptr<const T> end_thr() const { return thrust::device_ptr<const T>(__gpu_memory + __size); }
#endif
    void __malloc(size_t n) { if (valid) __free(); void *p = nullptr; if (pr_malloc(&p, n * sizeof(T)) != PR_OK) { std::cerr << pr_last_error() << std::endl; std::exit(1); } __gpu_memory = static_cast<T *>(p); __size = n; valid = true; }
    void __free() { if (valid) { pr_free(__gpu_memory); valid = false; __size = 0; __gpu_memory = nullptr; } }
    void upload(const std::vector<T> &h) { if (__size != h.size()) __malloc(h.size()); if (!h.empty()) pr_memcpy_h2d(__gpu_memory, h.data(), h.size() * sizeof(T)); }
    std::vector<T> download() const { std::vector<T> h(__size); if (__size) pr_memcpy_d2h(h.data(), __gpu_memory, __size * sizeof(T)); return h; }
private:
    void touched() { if (valid && __size) pr_invalidate(__gpu_memory, __size * sizeof(T)); }
    void fill(T init)
    {
        if (sizeof(T) == 4) { int32_t bits; std::memcpy(&bits, &init, 4); if (pr_fill_i32(reinterpret_cast<int32_t *>(__gpu_memory), __size, bits) != PR_OK) { std::cerr << pr_last_error() << std::endl; std::exit(1); } }
        else { std::vector<T> h(__size, init); upload(h); }
    }
};
using Int_holder = device_vector_holder<int>;

inline std::vector<Model::mat4x4> mat_to_compact_4x4(const std::vector<cv::Mat> &poses)
{ std::vector<Model::mat4x4> out(poses.size()); for (size_t i = 0; i < poses.size(); ++i) out[i].init_from_cv(poses[i]); return out; }

inline Model::mat4x4 compute_proj(const cv::Mat &K, int width, int height, float near = 10, float far = 10000)   // renderer.cpp:161-185
{ assert(K.type() == CV_32F); Model::mat4x4 p; pr_compute_proj(K.ptr<float>(), width, height, near, far, reinterpret_cast<pr_mat4 *>(&p)); return p; }

namespace detail {
inline void must(int rc) { if (rc != PR_OK) { std::cerr << "pose_refine: " << pr_last_error() << std::endl; std::exit(rc); } }   // renderer.cu:4-12
inline size_t out_pixels(size_t w, size_t h, const Model::ROI &r) { return (r.width > 0 && r.height > 0) ? (size_t)r.width * r.height : w * h; }
inline pr_roi roi(const Model::ROI &r) { return pr_roi{ r.x, r.y, r.width, r.height }; }
}  // namespace detail

// renderer.cu:306-336 (device-resident triangles)
inline device_vector_holder<int> render_cuda_keep_in_gpu(device_vector_holder<Model::Triangle> &tris, const std::vector<Model::mat4x4> &poses,
                                                         size_t width, size_t height, const Model::mat4x4 &proj_mat, const Model::ROI roi = { 0, 0, 0, 0 })
{
    device_vector_holder<int> out(poses.size() * detail::out_pixels(width, height, roi));
    detail::must(pr_render(reinterpret_cast<const pr_triangle *>(tris.__gpu_memory), tris.size(), reinterpret_cast<const pr_mat4 *>(poses.data()), poses.size(),
                           width, height, reinterpret_cast<const pr_mat4 *>(&proj_mat), detail::roi(roi), out.data()));
    return out;
}
// renderer.cu:269-303 (host triangles: uploaded on every call like the reference)
inline device_vector_holder<int> render_cuda_keep_in_gpu(const std::vector<Model::Triangle> &tris, const std::vector<Model::mat4x4> &poses,
                                                         size_t width, size_t height, const Model::mat4x4 &proj_mat, const Model::ROI roi = { 0, 0, 0, 0 })
{ device_vector_holder<Model::Triangle> d; d.upload(tris); return render_cuda_keep_in_gpu(d, poses, width, height, proj_mat, roi); }

// renderer.cu:189-267
inline std::vector<int32_t> render_cuda(device_vector_holder<Model::Triangle> &tris, const std::vector<Model::mat4x4> &poses,
                                        size_t width, size_t height, const Model::mat4x4 &proj_mat, const Model::ROI roi = { 0, 0, 0, 0 })
{
    std::vector<int32_t> out(poses.size() * detail::out_pixels(width, height, roi));
    detail::must(pr_render_to_host(reinterpret_cast<const pr_triangle *>(tris.__gpu_memory), tris.size(), reinterpret_cast<const pr_mat4 *>(poses.data()), poses.size(),
                                   width, height, reinterpret_cast<const pr_mat4 *>(&proj_mat), detail::roi(roi), out.data()));
    return out;
}
inline std::vector<int32_t> render_cuda(const std::vector<Model::Triangle> &tris, const std::vector<Model::mat4x4> &poses,
                                        size_t width, size_t height, const Model::mat4x4 &proj_mat, const Model::ROI roi = { 0, 0, 0, 0 })
{ device_vector_holder<Model::Triangle> d; d.upload(tris); return render_cuda(d, poses, width, height, proj_mat, roi); }

// ---- CPU twin of the renderer (cuda_renderer/renderer.cpp:190-298): part of the reference API (test.cpp:50 renders the
// scene with it).  Plain host code, same arithmetic as the device raster; never used as a fallback by the device path.
namespace detail {
inline Model::float3 mul3(const Model::mat4x4 &m, const Model::float3 &v)          // renderer.h:296-303 mat_mul_v
{ return { m.a0 * v.x + m.a1 * v.y + m.a2 * v.z + m.a3, m.b0 * v.x + m.b1 * v.y + m.b2 * v.z + m.b3, m.c0 * v.x + m.c1 * v.y + m.c2 * v.z + m.c3 }; }
inline float pick_max(float a, float b) { return (a > b) ? a : b; }
inline float pick_min(float a, float b) { return (a < b) ? a : b; }
inline float half_cross(const float *A, const float *B, const float *C) { return 0.5f * ((C[0] - A[0]) * (B[1] - A[1]) - (B[0] - A[0]) * (C[1] - A[1])); }
inline void raster_triangle_cpu(const Model::float3 clip[3], const float zc[3], int32_t *img, size_t width, size_t height, const Model::ROI &roi)
{
    float s[3][2];
    for (int i = 0; i < 3; ++i) {
        s[i][0] = clip[i].x / zc[i] * width / 2.0f + width / 2.0f;
        s[i][1] = clip[i].y / zc[i] * height / 2.0f + height / 2.0f;
    }
    float lo[2] = { 3.402823466e+38f, 3.402823466e+38f }, hi[2] = { -3.402823466e+38f, -3.402823466e+38f };
    float cmin[2] = { 0, 0 }, cmax[2] = { float(width - 1), float(height - 1) };
    size_t out_w = width;
    if (roi.width > 0 && roi.height > 0) {
        cmin[0] = roi.x; cmin[1] = height - 1 - (roi.y + roi.height - 1);
        cmax[0] = (roi.x + roi.width) - 1; cmax[1] = height - 1 - roi.y;
        out_w = roi.width;
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) {
        lo[j] = pick_max(cmin[j], pick_min(lo[j], s[i][j]));
        hi[j] = pick_min(cmax[j], pick_max(hi[j], s[i][j]));
    }
    const float area = half_cross(s[0], s[1], s[2]);
    if (!(area != 0.0f)) return;                                   // zero-area triangles are skipped (DESIGN.md deviations)
    const float inv = 1 / area;
    if (!(lo[0] + 0.5f >= 0.0f) || !(lo[1] + 0.5f >= 0.0f)) return; // NaN boxes draw nothing
    for (size_t y = size_t(lo[1] + 0.5f); y <= hi[1]; ++y)
        for (size_t x = size_t(lo[0] + 0.5f); x <= hi[0]; ++x) {
            float P[2] = { float(x), float(y) };
            const float beta = half_cross(s[0], P, s[2]) * inv, gamma = half_cross(s[0], s[1], P) * inv, alpha = 1.0f - beta - gamma;
            if (alpha < -0.0f || beta < -0.0f || gamma < -0.0f || alpha > 1.0f || beta > 1.0f || gamma > 1.0f) continue;
            const float frag = (alpha + beta + gamma) / (alpha / zc[0] + beta / zc[1] + gamma / zc[2]);
            const int32_t d = int32_t(frag + 0.5f);
            int32_t &cell = img[(x - roi.x) + (height - 1 - y - roi.y) * out_w];
            if (d < cell) cell = d;
        }
}
}  // namespace detail

inline std::vector<int32_t> render_cpu(const std::vector<Model::Triangle> &tris, const std::vector<Model::mat4x4> &poses, size_t width,
                                       size_t height, const Model::mat4x4 &proj_mat, const Model::ROI roi = { 0, 0, 0, 0 })
{
    const size_t px = detail::out_pixels(width, height, roi);
    std::vector<int32_t> depth(poses.size() * px, 2147483647);
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (long i = 0; i < (long)poses.size(); ++i)
        for (const auto &t : tris) {
            const Model::float3 cam[3] = { detail::mul3(poses[i], t.v0), detail::mul3(poses[i], t.v1), detail::mul3(poses[i], t.v2) };
            const float zc[3] = { cam[0].z, cam[1].z, cam[2].z };
            const Model::float3 clip[3] = { detail::mul3(proj_mat, cam[0]), detail::mul3(proj_mat, cam[1]), detail::mul3(proj_mat, cam[2]) };
            detail::raster_triangle_cpu(clip, zc, depth.data() + (size_t)i * px, width, height, roi);
        }
    for (auto &d : depth) if (d == 2147483647) d = 0;
    return depth;
}

// ---- raw depth stack -> cv::Mat conversions (renderer.cu:354-439, renderer.cpp:300-366) ------------------------------------
namespace detail {
inline std::vector<cv::Mat> split_stack(const void *flat, size_t elem, int type, size_t width, size_t height, size_t poses)
{
    std::vector<cv::Mat> out(poses);
    for (size_t i = 0; i < poses; ++i) {
        out[i] = cv::Mat((int)height, (int)width, type);
        std::memcpy(out[i].data, static_cast<const unsigned char *>(flat) + i * width * height * elem, width * height * elem);
    }
    return out;
}
}  // namespace detail
inline std::vector<cv::Mat> raw2depth_uint16_cuda(device_vector_holder<int> &raw, size_t width, size_t height, size_t pose_size)
{
    assert(raw.size() == width * height * pose_size);
    std::vector<uint16_t> d(raw.size());
    detail::must(pr_raw2depth_mask(raw.data(), raw.size(), d.data(), nullptr));
    return detail::split_stack(d.data(), 2, CV_16U, width, height, pose_size);
}
inline std::vector<cv::Mat> raw2mask_uint8_cuda(device_vector_holder<int> &raw, size_t width, size_t height, size_t pose_size)
{
    assert(raw.size() == width * height * pose_size);
    std::vector<uint8_t> m(raw.size());
    detail::must(pr_raw2depth_mask(raw.data(), raw.size(), nullptr, m.data()));
    return detail::split_stack(m.data(), 1, CV_8U, width, height, pose_size);
}
inline std::vector<std::vector<cv::Mat>> raw2depth_mask_cuda(device_vector_holder<int> &raw, size_t width, size_t height, size_t pose_size)
{
    assert(raw.size() == width * height * pose_size);
    std::vector<uint16_t> d(raw.size()); std::vector<uint8_t> m(raw.size());
    detail::must(pr_raw2depth_mask(raw.data(), raw.size(), d.data(), m.data()));
    auto dv = detail::split_stack(d.data(), 2, CV_16U, width, height, pose_size);
    auto mv = detail::split_stack(m.data(), 1, CV_8U, width, height, pose_size);
    std::vector<std::vector<cv::Mat>> out(pose_size);
    for (size_t i = 0; i < pose_size; ++i) out[i] = { dv[i], mv[i] };
    return out;
}
inline std::vector<cv::Mat> raw2depth_uint16_cpu(std::vector<int32_t> &raw, size_t width, size_t height, size_t pose_size)
{
    std::vector<uint16_t> d(raw.size());
    for (size_t i = 0; i < raw.size(); ++i) d[i] = uint16_t(raw[i]);
    return detail::split_stack(d.data(), 2, CV_16U, width, height, pose_size);
}
inline std::vector<cv::Mat> raw2mask_uint8_cpu(std::vector<int32_t> &raw, size_t width, size_t height, size_t pose_size)
{
    std::vector<uint8_t> m(raw.size());
    for (size_t i = 0; i < raw.size(); ++i) m[i] = (raw[i] > 0) ? 255 : 0;
    return detail::split_stack(m.data(), 1, CV_8U, width, height, pose_size);
}
inline std::vector<std::vector<cv::Mat>> raw2depth_mask_cpu(std::vector<int32_t> &raw, size_t width, size_t height, size_t pose_size)
{
    auto dv = raw2depth_uint16_cpu(raw, width, height, pose_size); auto mv = raw2mask_uint8_cpu(raw, width, height, pose_size);
    std::vector<std::vector<cv::Mat>> out(pose_size);
    for (size_t i = 0; i < pose_size; ++i) out[i] = { dv[i], mv[i] };
    return out;
}

template <typename... Params> Int_holder render(Params &&...p) { return render_cuda_keep_in_gpu(std::forward<Params>(p)...); }          // renderer.h:230-238
template <typename... Params> std::vector<int32_t> render_host(Params &&...p) { return render_cuda(std::forward<Params>(p)...); }      // renderer.h:240-248

}  // namespace cuda_renderer
