// d2c.hip -- depth2mask / exclusive_scan / depth2cloud (cuda_icp/icp.cu:228-291): rows counted, scanned per image, emitted row-major
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#include "pr_launch.h"

namespace prk {

// depth2cloud emit restricted to the hypothesis' pixel box (the fused path never reads outside it)
__global__ __launch_bounds__(256) void d2c_emit_box_kernel(const int32_t *__restrict__ depth, uint32_t width, uint32_t height,
                                                           const int4 *__restrict__ bbox, float fx, float fy, float cx, float cy,
                                                           const uint32_t *__restrict__ row_count, const uint32_t *__restrict__ row_off,
                                                           pr_vec3 *__restrict__ cloud, size_t cloud_stride, const PoseMeta *__restrict__ meta,
                                                           const uint32_t *__restrict__ box_off)
{
    const uint32_t lane = threadIdx.x & 63;
    const int4 bb = bbox[blockIdx.y];
    // where this hypothesis' cloud starts: packed behind the one before it (fused asynchronous path: PoseMeta::start from d2c_pack_starts_kernel), or at a fixed stride
    const size_t start = meta ? (size_t)meta[blockIdx.y].start : (size_t)blockIdx.y * cloud_stride;
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t row = blockIdx.x * kBoxRowsPerBlock + (threadIdx.x >> 6) * 4 + r;
        if (row >= height) return;
        if (row_count[(size_t)blockIdx.y * height + row] == 0) continue;
        const int32_t *line = depth + ((size_t)blockIdx.y * height + row) * width;
        if (box_off) {                                           // packed boxes (fill_box_kernel): the box's own pitch, origin at its top-left pixel; rows outside it hold nothing
            const int r0 = (int)height - 1 - bb.w;
            if ((int)row < r0 || (int)row > (int)height - 1 - bb.y) continue;
            line = depth + box_off[blockIdx.y] + ((ptrdiff_t)row - r0) * (ptrdiff_t)max(bb.z - bb.x + 1, 0) - (ptrdiff_t)bb.x;
        }
        pr_vec3 *out = cloud + start + row_off[(size_t)blockIdx.y * height + row];
        uint32_t done = 0;
        for (int x0 = bb.x; x0 <= bb.z; x0 += 512) {             // 8 independent loads in flight per lane
            int32_t dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int x = x0 + 64 * j + (int)lane; dv[j] = (x <= bb.z) ? line[x] : 0; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int x = x0 + 64 * j + (int)lane;
                const int32_t d = dv[j];
                const bool v = d > 0 && d != INT_MAX;
                const unsigned long long m = __ballot(v);
                if (v) {
                    const uint32_t k = done + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    const float z = d / 1000.0f;
                    pr_vec3 p;
                    p.x = ((float)(uint32_t)x - cx) / fx * z;
                    p.y = ((float)row - cy) / fy * z;
                    p.z = z;
                    out[k] = p;
                }
                done += (uint32_t)__popcll(m);
            }
        }
    }
}

// ================================================================================================
//  depth -> cloud: count per row, scan rows per image, emit in row-major order
// ================================================================================================
template <typename T> __device__ __forceinline__ bool depth_valid(T d, bool empty_intmax)
{
    return d > 0 && !(empty_intmax && (long long)d == (long long)INT_MAX);
}

// one wavefront per grid row; lanes sweep the row 64 pixels at a time
template <typename T>
__global__ __launch_bounds__(256) void d2c_count_kernel(const T *__restrict__ depth, size_t img_stride, uint32_t width,
                                                        uint32_t gw, uint32_t gh, uint32_t stride, bool empty_intmax,
                                                        uint32_t *__restrict__ row_count)
{
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (row >= gh) return;
    const T *line = depth + (size_t)blockIdx.y * img_stride + (size_t)row * stride * width;
    uint32_t cnt = 0;
    for (uint32_t x0 = 0; x0 < gw; x0 += 64) {
        const uint32_t x = x0 + lane;
        const bool v = (x < gw) && depth_valid(line[(size_t)x * stride], empty_intmax);
        cnt += (uint32_t)__popcll(__ballot(v));
    }
    if (lane == 0) row_count[(size_t)blockIdx.y * gh + row] = cnt;
}

// one workgroup per image: exclusive scan of the row counts (serial carry over chunks of 256 rows)
// exclusive scan of one image's row counts (256 lanes, Hillis-Steele per 256-row chunk); returns the total in every lane
__device__ __forceinline__ uint32_t row_scan_block(const uint32_t *__restrict__ rc, uint32_t *__restrict__ ro, uint32_t gh, uint32_t *buf, uint32_t *carry)
{
    if (threadIdx.x == 0) *carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < gh; base += 256) {
        const uint32_t r = base + threadIdx.x;
        const uint32_t v = (r < gh) ? rc[r] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 256; off <<= 1) {           // Hillis-Steele inclusive scan
            uint32_t t = (threadIdx.x >= off) ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (r < gh) ro[r] = *carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) *carry += buf[255];
        __syncthreads();
    }
    return *carry;
}
__global__ __launch_bounds__(256) void d2c_scan_kernel(const uint32_t *__restrict__ row_count, uint32_t gh,
                                                       uint32_t *__restrict__ row_off, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t buf[256];
    __shared__ uint32_t carry;
    const uint32_t total = row_scan_block(row_count + (size_t)blockIdx.x * gh, row_off + (size_t)blockIdx.x * gh, gh, buf, &carry);
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
// the same scan, and the start state of the hypothesis' ICP written from the count while it is at hand (asynchronous fused
// path: the host never sees the cloud sizes before the loop; icp.h:29-31 identity / zero result, an empty cloud is finished
// before it starts, icp.cu:183)
__global__ __launch_bounds__(256) void d2c_scan_init_kernel(const uint32_t *__restrict__ row_count, uint32_t gh, uint32_t *__restrict__ row_off,
                                                            uint32_t *__restrict__ counts, PoseMeta *__restrict__ meta, DevIcpState *__restrict__ st,
                                                            uint32_t *__restrict__ arrive, uint32_t cloud_stride)
{
    __shared__ uint32_t buf[256];
    __shared__ uint32_t carry;
    const uint32_t i = blockIdx.x;
    const uint32_t c = row_scan_block(row_count + (size_t)i * gh, row_off + (size_t)i * gh, gh, buf, &carry);
    if (threadIdx.x == 0) {
        counts[i] = c;
        arrive[i] = 0u;
        PoseMeta m;
        m.start = i * cloud_stride; m.count = c; m.state = c > 0 ? kRun : kSkip; m.pad = 0;       // (start: overwritten by d2c_pack_starts_kernel when the clouds are packed)
#pragma unroll
        for (int k = 0; k < 12; ++k) m.xform[k] = 0.0f;
        meta[i] = m;
    }
    if (threadIdx.x < 20) {                                      // DevIcpState: T[16], fitness, rmse, done, passes
        uint32_t w = 0;
        if (threadIdx.x < 16) w = (threadIdx.x % 5 == 0) ? __float_as_uint(1.0f) : 0u;
        else if (threadIdx.x == 18) w = c > 0 ? 0u : 1u;
        reinterpret_cast<uint32_t *>(st + i)[threadIdx.x] = w;
    }
}

// The clouds of a sub-batch PACKED one behind the other (each rounded up to kCloudAlign points = 128 bytes) instead of one per fixed stride of the
// largest pixel box: exclusive scan of the sizes, one workgroup.  With a stride the 256 clouds of a batch are 264 KB islands 1.2 MB apart, and
// how those islands fall on the memory system's interleave -- together with where the driver happened to put the two slots' buffers -- decided
// between 246 and 262 k poses/s for the same process (profiles/r04/README.md, "the spread between processes"); packed, the loop streams one dense range.
__global__ __launch_bounds__(256) void d2c_pack_starts_kernel(const uint32_t *__restrict__ counts, uint32_t n, PoseMeta *__restrict__ meta)
{
    __shared__ uint32_t buf[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < n) ? ((counts[i] + (kCloudAlign - 1u)) & ~(kCloudAlign - 1u)) : 0u;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 256; off <<= 1) {           // Hillis-Steele inclusive scan
            const uint32_t t = (threadIdx.x >= off) ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) meta[i].start = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += buf[255];
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void d2c_emit_kernel(const T *__restrict__ depth, size_t img_stride, uint32_t width,
                                                       uint32_t gw, uint32_t gh, uint32_t stride, uint32_t tl_x, uint32_t tl_y,
                                                       float fx, float fy, float cx, float cy, bool empty_intmax,
                                                       const uint32_t *__restrict__ row_count,
                                                       const uint32_t *__restrict__ row_off,
                                                       pr_vec3 *__restrict__ cloud, size_t cloud_stride)
{
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (row >= gh) return;
    if (row_count[(size_t)blockIdx.y * gh + row] == 0) return;    // wave-uniform: empty rows cost one load
    const T *line = depth + (size_t)blockIdx.y * img_stride + (size_t)row * stride * width;
    pr_vec3 *out = cloud + (size_t)blockIdx.y * cloud_stride + row_off[(size_t)blockIdx.y * gh + row];
    uint32_t done = 0;
    for (uint32_t x0 = 0; x0 < gw; x0 += 64) {
        const uint32_t x = x0 + lane;
        T d = 0;
        if (x < gw) d = line[(size_t)x * stride];
        const bool v = (x < gw) && depth_valid(d, empty_intmax);
        const unsigned long long m = __ballot(v);
        if (v) {
            const uint32_t k = done + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            // icp.cu:249-253: z = d/1000.f; x = (u + tl_x - cx)/fx*z; y = (v + tl_y - cy)/fy*z
            const float z = d / 1000.0f;
            pr_vec3 p;
            p.x = ((float)(x + tl_x) - cx) / fx * z;
            p.y = ((float)(row + tl_y) - cy) / fy * z;
            p.z = z;
            out[k] = p;
        }
        done += (uint32_t)__popcll(m);
    }
}

hipError_t launch_emit_box(const int32_t *depth, uint32_t n_poses, uint32_t width, uint32_t height, const int4 *bbox, float fx, float fy,
                           float cx, float cy, const uint32_t *row_count, const uint32_t *row_off, pr_vec3 *cloud, size_t cloud_stride,
                           hipStream_t s, const PoseMeta *meta, const uint32_t *box_off)
{
    if (n_poses == 0) return hipSuccess;
    for (uint32_t i0 = 0; i0 < n_poses; i0 += 32768) {
        const uint32_t ni = (n_poses - i0 < 32768) ? (n_poses - i0) : 32768;
        hipLaunchKernelGGL(d2c_emit_box_kernel, dim3((height + kBoxRowsPerBlock - 1) / kBoxRowsPerBlock, ni), dim3(256), 0, s, box_off ? depth : depth + (size_t)i0 * width * height, width, height,
                           bbox + i0, fx, fy, cx, cy, row_count + (size_t)i0 * height, row_off + (size_t)i0 * height,
                           meta ? cloud : cloud + (size_t)i0 * cloud_stride, cloud_stride, meta ? meta + i0 : nullptr, box_off ? box_off + i0 : nullptr);
    }
    return hipGetLastError();
}

template <typename T>
hipError_t launch_depth2cloud(const T *depth, uint32_t n_img, size_t img_stride, uint32_t width, uint32_t height,
                              uint32_t stride, uint32_t tl_x, uint32_t tl_y, float fx, float fy, float cx, float cy,
                              bool empty_intmax, uint32_t *row_count, uint32_t *row_off, uint32_t *counts,
                              pr_vec3 *cloud, size_t cloud_stride, bool emit, hipStream_t s)
{
    const uint32_t gw = width / stride, gh = height / stride;
    if (n_img == 0) return hipSuccess;
    if (gw == 0 || gh == 0) return hipMemsetAsync(counts, 0, sizeof(uint32_t) * n_img, s);
    for (uint32_t i0 = 0; i0 < n_img; i0 += 32768) {
        const uint32_t ni = (n_img - i0 < 32768) ? (n_img - i0) : 32768;
        const dim3 grid((gh + 3) / 4, ni);
        if (!emit) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(d2c_count_kernel<T>), grid, dim3(256), 0, s, depth + (size_t)i0 * img_stride,
                               img_stride, width, gw, gh, stride, empty_intmax, row_count + (size_t)i0 * gh);
            hipLaunchKernelGGL(d2c_scan_kernel, dim3(ni), dim3(256), 0, s, row_count + (size_t)i0 * gh, gh,
                               row_off + (size_t)i0 * gh, counts + i0);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(d2c_emit_kernel<T>), grid, dim3(256), 0, s, depth + (size_t)i0 * img_stride,
                               img_stride, width, gw, gh, stride, tl_x, tl_y, fx, fy, cx, cy, empty_intmax,
                               row_count + (size_t)i0 * gh, row_off + (size_t)i0 * gh,
                               cloud + (size_t)i0 * cloud_stride, cloud_stride);
        }
    }
    return hipGetLastError();
}
template hipError_t launch_depth2cloud<int32_t>(const int32_t *, uint32_t, size_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                                                float, float, float, float, bool, uint32_t *, uint32_t *, uint32_t *, pr_vec3 *,
                                                size_t, bool, hipStream_t);
template hipError_t launch_depth2cloud<uint16_t>(const uint16_t *, uint32_t, size_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                                                 float, float, float, float, bool, uint32_t *, uint32_t *, uint32_t *, pr_vec3 *,
                                                 size_t, bool, hipStream_t);

hipError_t launch_d2c_scan(const uint32_t *row_count, uint32_t gh, uint32_t *row_off, uint32_t *counts, uint32_t n_img, hipStream_t s)
{
    if (n_img == 0) return hipSuccess;
    hipLaunchKernelGGL(d2c_scan_kernel, dim3(n_img), dim3(256), 0, s, row_count, gh, row_off, counts);
    return hipGetLastError();
}
hipError_t launch_d2c_pack_starts(const uint32_t *counts, uint32_t n, PoseMeta *meta, hipStream_t s)
{
    if (n == 0 || !kCloudAlign) return hipSuccess;
    hipLaunchKernelGGL(d2c_pack_starts_kernel, dim3(1), dim3(256), 0, s, counts, n, meta);
    return hipGetLastError();
}
hipError_t launch_d2c_scan_init(const uint32_t *row_count, uint32_t gh, uint32_t *row_off, uint32_t *counts, uint32_t n_img,
                                PoseMeta *meta, DevIcpState *st, uint32_t *arrive, uint32_t cloud_stride, hipStream_t s)
{
    if (n_img == 0) return hipSuccess;
    hipLaunchKernelGGL(d2c_scan_init_kernel, dim3(n_img), dim3(256), 0, s, row_count, gh, row_off, counts, meta, st, arrive, cloud_stride);
    if (kCloudAlign) hipLaunchKernelGGL(d2c_pack_starts_kernel, dim3(1), dim3(256), 0, s, counts, n_img, meta);
    return hipGetLastError();
}

}  // namespace prk
