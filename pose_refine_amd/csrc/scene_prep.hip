// scene_prep.hip -- scene preparation on the device: get_normal + dep2pcd (common.h:47-61, depth_scene.cpp / pcd_scene.cpp:10-29), raw2depth / raw2mask (renderer.cu:338-439), the packed projective record, the sampled fingerprint of caller-owned arrays
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#include "pr_launch.h"

namespace prk {

// ================================================================================================
//  scene preparation and raw-depth conversions on the device
// ================================================================================================
template <typename T> __device__ __forceinline__ uint16_t depth_as_u16(T d);
template <> __device__ __forceinline__ uint16_t depth_as_u16<uint16_t>(uint16_t d) { return d; }
template <> __device__ __forceinline__ uint16_t depth_as_u16<int32_t>(int32_t d) { return (uint16_t)(d < 0 ? 0 : (d > 65535 ? 65535 : d)); }  // cv saturate_cast

// init_Scene_projective_cpu (depth_scene.cpp:3-35) per pixel: dep2pcd (common.h:47-61) + get_normal (common.cpp:17-107:
// 8 taps at radius 5, 64-bit integer normal equations, |delta| < 50 and depth < 2000 gates, zero in a 5-pixel border).
template <typename T>
__global__ __launch_bounds__(256) void scene_proj_prepare_kernel(const T *__restrict__ depth, uint32_t W, uint32_t H, float fx, float fy,
                                                                 float cx, float cy, pr_vec3 *__restrict__ pcd, pr_vec3 *__restrict__ normal)
{
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t at = (size_t)y * W + x;
    const T raw = depth[at];
    pr_vec3 p = { 0.0f, 0.0f, 0.0f };
    // CV_32S is read through at<uint32_t> (depth_scene.cpp:26), CV_16U as is
    const float dv = (sizeof(T) == 4) ? (float)(uint32_t)raw : (float)(int)raw;
    if (raw != 0) {
        const float z = dv / 1000.0f;
        p.x = ((float)x - cx) / fx * z;
        p.y = ((float)y - cy) / fy * z;
        p.z = z;
    }
    pcd[at] = p;

    pr_vec3 nrm = { 0.0f, 0.0f, 0.0f };
    const int R = 5;
    if ((int)y >= R && (int)y < (int)H - R - 1 && (int)x >= R && (int)x < (int)W - R - 1) {
        const long long d0 = depth_as_u16<T>(raw);
        if (d0 < 2000) {
            long long sxx = 0, sxy = 0, syy = 0, bx = 0, by = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ox = (k == 0 || k == 3 || k == 5) ? -R : ((k == 1 || k == 6) ? 0 : R);
                const int oy = (k < 3) ? -R : ((k < 5) ? 0 : R);
                const long long delta = (long long)depth_as_u16<T>(depth[at + ox + (long long)oy * W]) - d0;
                const long long ad = delta < 0 ? -delta : delta;
                if (ad < 50) { sxx += ox * ox; sxy += ox * oy; syy += oy * oy; bx += ox * delta; by += oy * delta; }
            }
            const long long det = sxx * syy - sxy * sxy;
            const long long gx = syy * bx - sxy * by;
            const long long gy = -sxy * bx + sxx * by;
            const float nx = fx * (float)gx, ny = fy * (float)gy, nz = (float)(-det * d0);
            const float len = sqrtf(nx * nx + ny * ny + nz * nz);
            if (len > 0) { const float inv = 1.0f / len; nrm.x = nx * inv; nrm.y = ny * inv; nrm.z = nz * inv; }
        }
    }
    normal[at] = nrm;
}

// raw2depth_uint16 / raw2mask_uint8 / raw2depth_mask (renderer.cu:338-439): uint16_t(x) truncation, mask = x>0 ? 255 : 0
__global__ __launch_bounds__(256) void raw2depth_mask_kernel(const int32_t *__restrict__ raw, size_t n, uint16_t *__restrict__ depth16,
                                                             uint8_t *__restrict__ mask8)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            const int4 v = *reinterpret_cast<const int4 *>(raw + i);
            if (depth16) { ushort4 d; d.x = (uint16_t)v.x; d.y = (uint16_t)v.y; d.z = (uint16_t)v.z; d.w = (uint16_t)v.w; *reinterpret_cast<ushort4 *>(depth16 + i) = d; }
            if (mask8) { uchar4 m; m.x = v.x > 0 ? 255 : 0; m.y = v.y > 0 ? 255 : 0; m.z = v.z > 0 ? 255 : 0; m.w = v.w > 0 ? 255 : 0; *reinterpret_cast<uchar4 *>(mask8 + i) = m; }
        } else {
            for (size_t k = i; k < n; ++k) { if (depth16) depth16[k] = (uint16_t)raw[k]; if (mask8) mask8[k] = raw[k] > 0 ? 255 : 0; }
        }
    }
}

// NN scene gather (pcd_scene.cpp:10-29): depth -> uint16 (saturating), valid pixels row-major, dep2pcd of the uint16 value
template <typename T>
__global__ __launch_bounds__(256) void nn_gather_count_kernel(const T *__restrict__ depth, uint32_t W, uint32_t H, uint32_t *__restrict__ row_count)
{
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= H) return;
    uint32_t cnt = 0;
    for (uint32_t x0 = 0; x0 < W; x0 += 64) { const uint32_t x = x0 + lane; cnt += (uint32_t)__popcll(__ballot(x < W && depth_as_u16<T>(depth[(size_t)row * W + x]) > 0)); }
    if (lane == 0) row_count[row] = cnt;
}
template <typename T>
__global__ __launch_bounds__(256) void nn_gather_emit_kernel(const T *__restrict__ depth, uint32_t W, uint32_t H, float fx, float fy, float cx, float cy,
                                                             const pr_vec3 *__restrict__ normal_full, const uint32_t *__restrict__ row_off,
                                                             pr_vec3 *__restrict__ pcd, pr_vec3 *__restrict__ nrm)
{
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= H) return;
    uint32_t done = row_off[row];
    for (uint32_t x0 = 0; x0 < W; x0 += 64) {
        const uint32_t x = x0 + lane;
        uint16_t d = 0;
        if (x < W) d = depth_as_u16<T>(depth[(size_t)row * W + x]);
        const bool v = d > 0;
        const unsigned long long m = __ballot(v);
        if (v) {
            const uint32_t k = done + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            const float z = (float)(int)d / 1000.0f;
            pr_vec3 p; p.x = ((float)x - cx) / fx * z; p.y = ((float)row - cy) / fy * z; p.z = z;
            pcd[k] = p; nrm[k] = normal_full[(size_t)row * W + x];
        }
        done += (uint32_t)__popcll(m);
    }
}

// ================================================================================================
//  scene repacking
// ================================================================================================
// colf[x] = ((float)(x + tl_x) - cx)/fx, rowf[y] = ((float)(y + tl_y) - cy)/fy: the factors dep2pcd (common.h:47-61) multiplies z with.
// The packed record drops pcd.x / pcd.y and the query rebuilds them as colf*z, rowf*z -- exact only for a pcd buffer that WAS
// built by dep2pcd with this K and offset.  The buffers are the caller's (public members in the reference), so every pixel that
// can take part in a match (z > 0) is checked here; a single deviation clears *exact and the caller's arrays are used as they are.
__global__ __launch_bounds__(256) void pack_proj_scene_kernel(const pr_vec3 *__restrict__ pcd, const pr_vec3 *__restrict__ normal,
                                                              float4 *__restrict__ rec, size_t n, float *__restrict__ colf,
                                                              float *__restrict__ rowf, uint32_t width, uint32_t height,
                                                              float fx, float fy, float cx, float cy, uint32_t tl_x, uint32_t tl_y,
                                                              uint32_t *__restrict__ exact)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < width) colf[i] = ((float)(i + tl_x) - cx) / fx;
    if (i < height) rowf[i] = ((float)(i + tl_y) - cy) / fy;
    if (i >= n) return;
    const pr_vec3 p = pcd[i];
    rec[i] = make_float4(normal[i].x, normal[i].y, normal[i].z, p.z);
    if (p.z > 0) {
        const uint32_t x = (uint32_t)(i % width), y = (uint32_t)(i / width);
        const float ex = ((float)((size_t)x + tl_x) - cx) / fx * p.z, ey = ((float)((size_t)y + tl_y) - cy) / fy * p.z;
        if (!(__float_as_uint(ex) == __float_as_uint(p.x) && __float_as_uint(ey) == __float_as_uint(p.y))) *exact = 0u;
    }
}

// ---- sampled fingerprint of caller-owned scene arrays -------------------------------------------------------------------------
// The packed projective scene and the kd-tree search records are cached by the ADDRESS of the arrays they were derived from.  Writes that
// go through this library drop them; a write the library cannot see (a caller's kernel, a raw hipMemcpy, an allocator handing the address
// out again) would leave a stale cache behind.  Every asynchronous batch therefore re-reads 4096 words of each source array, spread over
// the whole array, and compares their hash with the one taken when the cache was built: a frame that changed as a whole cannot pass, and
// the batch is then repeated with fresh caches (refine_wait, like a stale model box).  A sampled check is not a proof -- an edit confined
// to words it does not look at still needs pr_invalidate, as pose_refine.h says.
__global__ __launch_bounds__(256) void scene_fingerprint_kernel(const uint32_t *__restrict__ a, unsigned long long na, const uint32_t *__restrict__ b,
                                                                unsigned long long nb, const uint32_t *__restrict__ c, unsigned long long nc,
                                                                uint32_t *__restrict__ expected, uint32_t *__restrict__ flag, int check)
{
    __shared__ uint32_t part[4];
    uint32_t h = fingerprint_lane(a, na, b, nb, c, nc);
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t v = part[0] + part[1] + part[2] + part[3];
        if (check) { if (*expected != v) *flag = 1u; }
        else *expected = v;
    }
}
hipError_t launch_scene_fingerprint(const void *a, size_t a_bytes, const void *b, size_t b_bytes, const void *c, size_t c_bytes,
                                    uint32_t *expected, uint32_t *flag, bool check, hipStream_t s)
{
    hipLaunchKernelGGL(scene_fingerprint_kernel, dim3(1), dim3(256), 0, s, static_cast<const uint32_t *>(a), (unsigned long long)(a_bytes / 4),
                       static_cast<const uint32_t *>(b), (unsigned long long)(b_bytes / 4), static_cast<const uint32_t *>(c), (unsigned long long)(c_bytes / 4),
                       expected, flag, check ? 1 : 0);
    return hipGetLastError();
}

// The same hash over EVERY word of the arrays (order-free sum over workgroups, atomicAdd into *sum, which the caller zeroes): what the
// synchronous entry points -- the ones the C++ adapters of the reference's API call -- compare on a cache hit, so that an in-place edit of
// any word of a scene array is seen although the caller never announced it (the reference reads the arrays at every call,
// depth_scene.h:29-48).  7.4 MB of a 640 x 480 projective scene: ~5 us.
__global__ __launch_bounds__(256) void scene_fingerprint_full_kernel(const uint32_t *__restrict__ a, unsigned long long na, const uint32_t *__restrict__ b,
                                                                     unsigned long long nb, const uint32_t *__restrict__ c, unsigned long long nc,
                                                                     uint32_t *__restrict__ sum)
{
    __shared__ uint32_t part[4];
    const uint32_t *arr[3] = { a, b, c };
    const unsigned long long len[3] = { na, nb, nc };
    uint32_t h = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!arr[k]) continue;
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < len[k]; i += stride)
            h += (arr[k][i] ^ (uint32_t)i) * 2654435761u + (uint32_t)k + (uint32_t)(i >> 32);
    }
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);
}
hipError_t launch_scene_fingerprint_full(const void *a, size_t a_bytes, const void *b, size_t b_bytes, const void *c, size_t c_bytes, uint32_t *sum, hipStream_t s)
{
    const size_t words = a_bytes / 4 + b_bytes / 4 + c_bytes / 4;
    const uint32_t grid = (uint32_t)std::max<size_t>(1, std::min<size_t>(1024, (words + 256 * 16 - 1) / (256 * 16)));
    hipError_t e = hipMemsetAsync(sum, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scene_fingerprint_full_kernel, dim3(grid), dim3(256), 0, s, static_cast<const uint32_t *>(a), (unsigned long long)(a_bytes / 4),
                       static_cast<const uint32_t *>(b), (unsigned long long)(b_bytes / 4), static_cast<const uint32_t *>(c), (unsigned long long)(c_bytes / 4), sum);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_nn_gather(const T *depth, uint32_t W, uint32_t H, float fx, float fy, float cx, float cy, const pr_vec3 *normal_full,
                            uint32_t *row_count, uint32_t *row_off, uint32_t *count, pr_vec3 *pcd, pr_vec3 *nrm, bool emit, hipStream_t s)
{
    if (!emit) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_gather_count_kernel<T>), dim3((H + 3) / 4), dim3(256), 0, s, depth, W, H, row_count);
        { hipError_t e = launch_d2c_scan(row_count, H, row_off, count, 1, s); if (e != hipSuccess) return e; }
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_gather_emit_kernel<T>), dim3((H + 3) / 4), dim3(256), 0, s, depth, W, H, fx, fy, cx, cy, normal_full, row_off, pcd, nrm);
    }
    return hipGetLastError();
}
template hipError_t launch_nn_gather<int32_t>(const int32_t *, uint32_t, uint32_t, float, float, float, float, const pr_vec3 *, uint32_t *, uint32_t *, uint32_t *, pr_vec3 *, pr_vec3 *, bool, hipStream_t);
template hipError_t launch_nn_gather<uint16_t>(const uint16_t *, uint32_t, uint32_t, float, float, float, float, const pr_vec3 *, uint32_t *, uint32_t *, uint32_t *, pr_vec3 *, pr_vec3 *, bool, hipStream_t);

template <typename T>
hipError_t launch_scene_proj_prepare(const T *depth, uint32_t W, uint32_t H, float fx, float fy, float cx, float cy, pr_vec3 *pcd, pr_vec3 *normal, hipStream_t s)
{
    if (W == 0 || H == 0) return hipSuccess;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scene_proj_prepare_kernel<T>), dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, depth, W, H, fx, fy, cx, cy, pcd, normal);
    return hipGetLastError();
}
template hipError_t launch_scene_proj_prepare<int32_t>(const int32_t *, uint32_t, uint32_t, float, float, float, float, pr_vec3 *, pr_vec3 *, hipStream_t);
template hipError_t launch_scene_proj_prepare<uint16_t>(const uint16_t *, uint32_t, uint32_t, float, float, float, float, pr_vec3 *, pr_vec3 *, hipStream_t);
hipError_t launch_raw2depth_mask(const int32_t *raw, size_t n, uint16_t *depth16, uint8_t *mask8, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(raw2depth_mask_kernel, dim3(cap_grid((n + 1023) / 1024)), dim3(256), 0, s, raw, n, depth16, mask8);
    return hipGetLastError();
}

// *exact (device word) is set non-zero and cleared by the kernel when the pcd buffer is not what dep2pcd would have produced
hipError_t launch_pack_proj_scene(const pr_vec3 *pcd, const pr_vec3 *normal, float4 *rec, size_t n, float *colf, float *rowf,
                                  uint32_t width, uint32_t height, float fx, float fy, float cx, float cy, uint32_t tl_x, uint32_t tl_y,
                                  uint32_t *exact, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(exact, 1, sizeof(uint32_t), s);          // non-zero = exact so far
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pack_proj_scene_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, pcd, normal, rec, n, colf, rowf,
                       width, height, fx, fy, cx, cy, tl_x, tl_y, exact);
    return hipGetLastError();
}

}  // namespace prk
