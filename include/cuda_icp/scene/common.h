// common.h -- device_vector_holder<T> and the per-point helpers of cuda_icp/scene/common.h:16-81,
// on top of the C ABI (pr_malloc / pr_free / pr_memcpy_*).
#pragma once
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../geometry.h"
#include "pose_refine.h"
#include "pose_refine/cv_compat.h"
#include "pose_refine/thrust_compat.h"

namespace pose_refine_detail {
inline void must(int rc, const char *what)
{   // the reference prints and exit()s on device errors (renderer.cu:4-12 gpuErrchk)
    if (rc != PR_OK) { std::fprintf(stderr, "pose_refine: %s failed: %s\n", what, pr_last_error()); std::exit(rc); }
}
}  // namespace pose_refine_detail

// RAII device buffer (common.h:16-44).  The reference returns it by value relying on copy elision;
// here copies are deleted and moves are provided -- `auto x = f();` is unchanged.
template <typename T> class device_vector_holder {
public:
    T *__gpu_memory = nullptr;
    size_t __size = 0;
    bool valid = false;
    device_vector_holder() {}
    explicit device_vector_holder(size_t n) { __malloc(n); }
    device_vector_holder(size_t n, T init) { __malloc(n); fill(init); }
    device_vector_holder(const device_vector_holder &) = delete;
    device_vector_holder &operator=(const device_vector_holder &) = delete;
    device_vector_holder(device_vector_holder &&o) noexcept : __gpu_memory(o.__gpu_memory), __size(o.__size), valid(o.valid) { o.__gpu_memory = nullptr; o.__size = 0; o.valid = false; }
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
#ifdef POSE_REFINE_HAVE_THRUST                                   // common.h:30-33; hipcc + rocThrust only (pose_refine/thrust_compat.h)
    thrust::device_ptr<T> data_thr() { touched(); return thrust::device_ptr<T>(__gpu_memory); }
    thrust::device_ptr<T> begin_thr() { touched(); return thrust::device_ptr<T>(__gpu_memory); }
    thrust::device_ptr<T> end_thr() { touched(); return thrust::device_ptr<T>(__gpu_memory + __size); }
    thrust::device_ptr<const T> data_thr() const { return thrust::device_ptr<const T>(__gpu_memory); }
    thrust::device_ptr<const T> begin_thr() const { return thrust::device_ptr<const T>(__gpu_memory); }
    thrust::device_ptr<const T> end_thr() const { return thrust::device_ptr<const T>(__gpu_memory + __size); }
#endif
    void __malloc(size_t n)
    {
        if (valid) __free();
        void *p = nullptr;
        pose_refine_detail::must(pr_malloc(&p, n * sizeof(T)), "pr_malloc");
        __gpu_memory = static_cast<T *>(p); __size = n; valid = true;
    }
    void __free() { if (valid) { pr_free(__gpu_memory); __gpu_memory = nullptr; valid = false; __size = 0; } }
    // replaces thrust::copy(host.begin(), host.end(), holder.begin_thr()) and the reverse
    void upload(const std::vector<T> &h) { if (__size != h.size()) __malloc(h.size()); if (!h.empty()) pose_refine_detail::must(pr_memcpy_h2d(__gpu_memory, h.data(), h.size() * sizeof(T)), "pr_memcpy_h2d"); }
    std::vector<T> download() const { std::vector<T> h(__size); if (__size) pose_refine_detail::must(pr_memcpy_d2h(h.data(), __gpu_memory, __size * sizeof(T)), "pr_memcpy_d2h"); return h; }
private:
    void touched() { if (valid && __size) pr_invalidate(__gpu_memory, __size * sizeof(T)); }
    void fill(T init)
    {
        if (sizeof(T) == 4) { int32_t bits; std::memcpy(&bits, &init, 4); pose_refine_detail::must(pr_fill_i32(reinterpret_cast<int32_t *>(__gpu_memory), __size, bits), "pr_fill_i32"); }
        else { std::vector<T> h(__size, init); upload(h); }
    }
};

namespace pose_refine_detail {
// A host depth image (CV_16U or CV_32S: depth_scene.cpp:11-12, pcd_scene.cpp:6-7 assert exactly that) as `height` dense rows of `width` values on
// the device.  The reference's CPU route indexes the image with at<>(r, c), i.e. through the Mat's own row pitch, and reads only rows < height,
// cols < width: an image that is wider or taller than the scene it is used for, or a non-continuous Mat (an ROI view), is copied row by row with
// the source pitch.  Any other element type, or an image smaller than the scene, ends the program with a message (the reference asserts /
// reads out of bounds) -- checked at run time, NDEBUG or not.
inline void upload_depth(const cv::Mat &depth, size_t width, size_t height, device_vector_holder<unsigned char> &out, bool &is32)
{
    if (depth.type() != CV_16U && depth.type() != CV_32S) { std::fprintf(stderr, "pose_refine: the scene depth image must be CV_16U or CV_32S (type %d given)\n", depth.type()); std::exit(1); }
    is32 = depth.type() == CV_32S;
    const size_t el = is32 ? 4 : 2;
    if ((size_t)depth.cols < width || (size_t)depth.rows < height) {
        std::fprintf(stderr, "pose_refine: the scene depth image is %d x %d, smaller than the %zu x %zu it is used as: pass width / height of the image\n", depth.cols, depth.rows, width, height);
        std::exit(1);
    }
    out.__malloc(width * height * el);
    const size_t pitch = (size_t)depth.step;
    if ((size_t)depth.cols == width && depth.isContinuous() && pitch == width * el) { must(pr_memcpy_h2d(out.__gpu_memory, depth.data, width * height * el), "pr_memcpy_h2d"); return; }
    std::vector<unsigned char> dense(width * height * el);
    for (size_t r = 0; r < height; ++r) std::memcpy(dense.data() + r * width * el, depth.data + r * pitch, width * el);
    must(pr_memcpy_h2d(out.__gpu_memory, dense.data(), dense.size()), "pr_memcpy_h2d");
}
}  // namespace pose_refine_detail

// common.h:47-61
template <class T> inline Vec3f dep2pcd(size_t x, size_t y, T dep, Mat3x3f &K, size_t tl_x = 0, size_t tl_y = 0)
{
    if (dep == 0) return Vec3f(0, 0, 0);
    float z = dep / 1000.0f;
    return Vec3f((x + tl_x - K[0][2]) / K[0][0] * z, (y + tl_y - K[1][2]) / K[1][1] * z, z);
}
// common.h:63-73
inline Vec3i pcd2dep(const Vec3f &p, const Mat3x3f &K, size_t tl_x = 0, size_t tl_y = 0)
{
    return Vec3i(int(p.x / p.z * K[0][0] + K[0][2] - tl_x + 0.5f), int(p.y / p.z * K[1][1] + K[1][2] - tl_y + 0.5f), int(p.z * 1000.0f + 0.5f));
}
template <typename T> inline T std__abs(T v) { return (v > 0) ? v : (-v); }
template <typename T> inline T pow2(T v) { return v * v; }

// common.cpp:17-107 (CPU in the reference too)
inline std::vector<Vec3f> get_normal(const cv::Mat &depth, const Mat3x3f &K)
{
    assert(depth.type() == CV_16U || depth.type() == CV_32S);
    const size_t n = (size_t)depth.rows * depth.cols;
    std::vector<uint16_t> d16(n);
    if (depth.type() == CV_32S) { const int32_t *s = depth.ptr<int32_t>(); for (size_t i = 0; i < n; ++i) d16[i] = (uint16_t)(s[i] < 0 ? 0 : (s[i] > 65535 ? 65535 : s[i])); }
    else { const uint16_t *s = depth.ptr<uint16_t>(); d16.assign(s, s + n); }
    std::vector<Vec3f> out(n);
    pose_refine_detail::must(pr_get_normal(d16.data(), depth.cols, depth.rows, K.data(), reinterpret_cast<pr_vec3 *>(out.data())), "pr_get_normal");
    return out;
}
