// pr_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hot path.
//
//   raster_kernel          <- render_triangle + rasterization        cuda_renderer/renderer.cu:83-187
//   depth2cloud kernels    <- depth2mask / exclusive_scan / depth2cloud   cuda_icp/icp.cu:228-291
//   icp_pass_kernel<Scene> <- thrust::transform_reduce(thrust__pcd2Ab<Scene>) fused with
//                             transform_pcd_cuda of the previous iteration   icp.cu:142-153,170-172, icp.h:128-209
//   icp_finalize*          <- second reduction stage (+ optional device-side solve, icp.cu:181-212)
//
// All float arithmetic follows the operand order of the reference expression by expression and
// the TU is compiled with -ffp-contract=off, so every per-element value is bit-identical to the
// CPU restatement; sums are bit-identical because the reduction tree is fixed ("canonical tree").
// No MFMA, no Thrust/rocPRIM, wave64 DPP reductions, 16-byte vector memory operations.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <float.h>
#include <atomic>
#include <stdlib.h>

#include "pr_internal.h"

#define PR_HD __host__ __device__
#include "pr_solver.inl"

namespace prk {

// ------------------------------------------------------------------------------------------------
// conversions with the semantics the reference CPU build has on x86-64 (cvttss2si): out-of-range
// and NaN inputs give INT_MIN.  v_cvt_i32_f32 saturates instead, so the range test is explicit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int f2i_x86(float v)
{
    return (v > -2147483904.0f && v < 2147483648.0f) ? (int)v : INT_MIN;
}
// size_t(float) for loop starts that are later compared against a bound < 2^24: anything
// outside (-1, 2^24) can never satisfy "(float)p <= bound", so it maps to "no iterations".
__device__ __forceinline__ int loop_start(float v)
{
    return (v > -1.0f && v < 16777216.0f) ? (int)v : INT_MAX;
}

__device__ __forceinline__ float sel_max(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float sel_min(float a, float b) { return (a < b) ? a : b; }

// ================================================================================================
//  fill / max2zero
// ================================================================================================
__global__ __launch_bounds__(256) void fill_i32_kernel(int32_t *dst, size_t n, int32_t v)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) *reinterpret_cast<int4 *>(dst + i) = make_int4(v, v, v, v);
    if (i < n) for (size_t k = i; k < n && k < i + 4; ++k) dst[k] = v;
}

// renderer.cu:71-80 max2zero_functor over the whole stack of images
__global__ __launch_bounds__(256) void max2zero_kernel(int32_t *d, size_t n)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) {
        int4 v = *reinterpret_cast<int4 *>(d + i);
        v.x = (v.x == INT_MAX) ? 0 : v.x; v.y = (v.y == INT_MAX) ? 0 : v.y;
        v.z = (v.z == INT_MAX) ? 0 : v.z; v.w = (v.w == INT_MAX) ? 0 : v.w;
        *reinterpret_cast<int4 *>(d + i) = v;
    }
    if (i < n) for (size_t k = i; k < n && k < i + 4; ++k) if (d[k] == INT_MAX) d[k] = 0;
}

// ================================================================================================
//  triangle raster: one lane = one (triangle, hypothesis); int32 atomicMin resolves depth
// ================================================================================================
__device__ __forceinline__ float area2(float ax, float ay, float bx, float by, float cx, float cy)
{   // renderer.h:315-318 calculateSignedArea
    return 0.5f * ((cx - ax) * (by - ay) - (bx - ax) * (cy - ay));
}

// Per-triangle screen-space setup shared by the raster kernels: model + projection + viewport
// transform (renderer.cu:159-186, :90-98), clamped pixel box (:100-122), signed area (renderer.h:315-333).
struct TriSetup {
    float px[3], py[3], w3[3];
    float base_inv;
    int x0, y0, nx, ny;          // first pixel column/row of the loops of renderer.cu:124-125 and their trip counts
};
__device__ __forceinline__ int trip_count(int first, float hi)
{   // number of iterations of  for (p = first; (float)p <= hi; ++p)  with first in [0, 2^24) or INT_MAX
    if (first == INT_MAX || !(hi >= (float)first)) return 0;
    return (int)floorf(hi) - first + 1;
}
__device__ __forceinline__ void tri_setup(const float *__restrict__ tv, const float *__restrict__ M, const pr_mat4 &proj,
                                          uint32_t width, uint32_t height, float cmin0, float cmin1, float cmax0, float cmax1,
                                          TriSetup &t)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = tv[3 * k], y = tv[3 * k + 1], z = tv[3 * k + 2];
        // model transform (renderer.h:296-303 mat_mul_v, rows a,b,c)
        const float lx = M[0] * x + M[1] * y + M[2] * z + M[3];
        const float ly = M[4] * x + M[5] * y + M[6] * z + M[7];
        const float lz = M[8] * x + M[9] * y + M[10] * z + M[11];
        t.w3[k] = lz;                                            // renderer.cu:177-183 last_row
        // projection transform: only x and y of the result are used downstream
        const float cxp = proj.m[0] * lx + proj.m[1] * ly + proj.m[2] * lz + proj.m[3];
        const float cyp = proj.m[4] * lx + proj.m[5] * ly + proj.m[6] * lz + proj.m[7];
        // viewport (renderer.cu:90-98)
        t.px[k] = cxp / lz * (float)width / 2.0f + (float)width / 2.0f;
        t.py[k] = cyp / lz * (float)height / 2.0f + (float)height / 2.0f;
    }
    float lo0 = FLT_MAX, lo1 = FLT_MAX, hi0 = -FLT_MAX, hi1 = -FLT_MAX;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo0 = sel_max(cmin0, sel_min(lo0, t.px[k]));  hi0 = sel_min(cmax0, sel_max(hi0, t.px[k]));
        lo1 = sel_max(cmin1, sel_min(lo1, t.py[k]));  hi1 = sel_min(cmax1, sel_max(hi1, t.py[k]));
    }
    const float area = area2(t.px[0], t.py[0], t.px[1], t.py[1], t.px[2], t.py[2]);
    t.x0 = loop_start(lo0 + 0.5f);
    t.y0 = loop_start(lo1 + 0.5f);
    t.nx = trip_count(t.x0, hi0);
    t.ny = trip_count(t.y0, hi1);
    if (!(area != 0.0f)) { t.nx = 0; t.ny = 0; }                 // documented: zero-area triangles are skipped
    t.base_inv = 1 / area;
}
// one candidate pixel of one triangle: barycentric test + perspective depth (renderer.cu:126-140)
__device__ __forceinline__ bool tri_fragment(const float px[3], const float py[3], float base_inv, int x, int y,
                                             float &alpha, float &beta, float &gamma)
{
    const float fx = (float)x, fy = (float)y;
    beta  = area2(px[0], py[0], fx, fy, px[2], py[2]) * base_inv;
    gamma = area2(px[0], py[0], px[1], py[1], fx, fy) * base_inv;
    alpha = 1.0f - beta - gamma;
    return !(alpha < -0.0f || beta < -0.0f || gamma < -0.0f || alpha > 1.0f || beta > 1.0f || gamma > 1.0f);
}
__device__ __forceinline__ int fragment_depth(float alpha, float beta, float gamma, float w0, float w1, float w2)
{
    const float az = alpha / w0, bz = beta / w1, gz = gamma / w2;
    const float frag = (alpha + beta + gamma) / (az + bz + gz);
    return f2i_x86(frag + 0.5f);
}

// Wave-cooperative raster of up to 64 set-up triangles (one per lane, n = candidate pixels of the lane's triangle, 0 for none).
// Thread-per-triangle pixel loops waste most lanes (the average triangle of obj_06 tests 7 pixel centres, the largest 36), so
// the wavefront expands its triangles into one dense list of candidate pixels (exclusive scan of the per-triangle counts) and
// walks that list 64 candidates at a time: every lane finds the owner of its candidate by a 6-step search over the scanned
// offsets and reads the owner's setup back from LDS.  Candidates that pass the inside test are rare (about one in four) --
// they are queued per wavefront as packed (owner, x, y) words and drained 64 at a time, so the four IEEE divisions of the
// perspective depth and the depth update (`sink(x, y, depth)`) always run on full wavefronts.  Same arithmetic per candidate
// as renderer.cu:124-140.  `w` / `queue` are this wavefront's private LDS scratch.
constexpr int kSetupWords = 15;
template <class Sink>
__device__ __forceinline__ void wave_raster(const TriSetup &t, int n, float (*w)[64], uint32_t *queue, Sink sink)
{
    const uint32_t lane = threadIdx.x & 63;
    // exclusive scan of n over the wavefront
    int incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
    const int excl = incl - n;
    const int total = __shfl(incl, 63);

    w[0][lane] = t.px[0]; w[1][lane] = t.py[0]; w[2][lane] = t.px[1]; w[3][lane] = t.py[1]; w[4][lane] = t.px[2]; w[5][lane] = t.py[2];
    w[6][lane] = t.w3[0]; w[7][lane] = t.w3[1]; w[8][lane] = t.w3[2]; w[9][lane] = t.base_inv;
    w[10][lane] = __int_as_float(t.x0); w[11][lane] = __int_as_float(t.y0); w[12][lane] = __int_as_float(t.nx > 0 ? t.nx : 1);
    w[13][lane] = __int_as_float(excl);
    w[14][lane] = __builtin_amdgcn_rcpf((float)(t.nx > 0 ? t.nx : 1));
    // wavefronts only touch their own slice; a wave-level fence is enough for LDS ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    int qn = 0;                                                    // wave-uniform fill level, < 64 between iterations
    auto drain = [&](int first, int count) {
        if ((int)lane < count) {
            const uint32_t e = queue[first + lane];
            const int o = (int)(e >> 26), x = (int)((e >> 13) & 0x1fffu), y = (int)(e & 0x1fffu);
            const float px[3] = { w[0][o], w[2][o], w[4][o] }, py[3] = { w[1][o], w[3][o], w[5][o] };
            float alpha, beta, gamma;
            (void)tri_fragment(px, py, w[9][o], x, y, alpha, beta, gamma);
            sink(x, y, fragment_depth(alpha, beta, gamma, w[6][o], w[7][o], w[8][o]));
        }
    };
    for (int c0 = 0; c0 < total; c0 += 64) {
        const int c = c0 + (int)lane;
        bool pass = false;
        uint32_t entry = 0;
        if (c < total) {
            int o = 0;
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) { if (__float_as_int(w[13][o + s]) <= c) o += s; }
            const int k = c - __float_as_int(w[13][o]);
            const int nx = __float_as_int(w[12][o]);
            // k / nx from the owner's reciprocal with a one-step correction; k, nx, q*nx < 2^24 (a frame has < 2^24 pixels),
            // so the 24-bit multiplier (full rate, v_mul_lo_u32 is quarter rate) is exact
            int q = (int)((float)k * w[14][o]);
            if ((int)__umul24((unsigned)q, (unsigned)nx) > k) --q;
            if ((int)__umul24((unsigned)(q + 1), (unsigned)nx) <= k) ++q;
            const int x = __float_as_int(w[10][o]) + (k - (int)__umul24((unsigned)q, (unsigned)nx));
            const int y = __float_as_int(w[11][o]) + q;
            const float px[3] = { w[0][o], w[2][o], w[4][o] }, py[3] = { w[1][o], w[3][o], w[5][o] };
            float alpha, beta, gamma;
            pass = tri_fragment(px, py, w[9][o], x, y, alpha, beta, gamma);
            entry = ((uint32_t)o << 26) | ((uint32_t)x << 13) | (uint32_t)y;
        }
        const unsigned long long m = __ballot(pass);
        if (pass) queue[qn + (int)__popcll(m & ((1ull << lane) - 1ull))] = entry;
        qn += (int)__popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (qn >= 64) { drain(qn - 64, 64); qn -= 64; __builtin_amdgcn_wave_barrier(); }
    }
    drain(0, qn);
    __builtin_amdgcn_wave_barrier();                             // the scratch may be refilled by the caller's next chunk
}

// one lane = one (triangle, hypothesis); depth resolved with int32 atomicMin in global memory (the reference scheme).
// A workgroup keeps its 256 triangles in registers and walks `pose_run` consecutive hypotheses with them: a mesh that does not fit
// the L2 (the 1 M-triangle mesh of BASELINE configs[4]: 36 MB) is then streamed from the Infinity Cache / HBM once per pose_run
// hypotheses instead of once per hypothesis -- at 128 hypotheses that stream (4.6 GB, 3.2 TB/s) was what bounded the kernel.
__global__ __launch_bounds__(256) void raster_kernel(const pr_triangle *__restrict__ tris, uint32_t n_tris,
                                                     const pr_mat4 *__restrict__ poses, int32_t *__restrict__ depth,
                                                     uint32_t width, uint32_t height, pr_mat4 proj, pr_roi roi,
                                                     uint32_t rw, uint32_t rh, const int4 *__restrict__ boxes, uint32_t n_poses, uint32_t pose_run)
{
    __shared__ float sh[4][kSetupWords][64];
    __shared__ uint32_t shq[4][128];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t ti = blockIdx.x * 256 + threadIdx.x;
    float tv[9];
    if (ti < n_tris) {
        const float *src = reinterpret_cast<const float *>(tris + ti);
#pragma unroll
        for (int k = 0; k < 9; ++k) tv[k] = src[k];
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) tv[k] = 0.0f;
    }
    float rmin0 = 0.0f, rmin1 = 0.0f, rmax0 = (float)(width - 1), rmax1 = (float)(height - 1);
    if (roi.width > 0 && roi.height > 0) {                       // renderer.cu:106-113 (image is flipped in y)
        rmin0 = (float)roi.x;
        rmin1 = (float)((unsigned long long)(height - 1) - (unsigned long long)(long long)(roi.y + roi.height - 1));
        rmax0 = (float)((roi.x + roi.width) - 1);
        rmax1 = (float)((unsigned long long)(height - 1) - (unsigned long long)(long long)roi.y);
    }
    for (uint32_t k = 0; k < pose_run; ++k) {
        const uint32_t by = blockIdx.y * pose_run + k;
        if (by >= n_poses) break;
        const float *M = poses[by].m;                            // wave-uniform -> scalar loads
        int32_t *img = depth + (size_t)by * rw * rh;
        float cmin0 = rmin0, cmin1 = rmin1, cmax0 = rmax0, cmax1 = rmax1;
        if (boxes) {                                             // fused path: the hypothesis' pixel box (already intersected with the caller's ROI,
            const int4 bb = boxes[by];                           // if any); a conservative box clips nothing, an ROI clips like renderer.cu:106-113
            cmin0 = (float)bb.x; cmin1 = (float)bb.y; cmax0 = (float)bb.z; cmax1 = (float)bb.w;
        }
        TriSetup t;
        int n = 0;
        if (ti < n_tris) {
            tri_setup(tv, M, proj, width, height, cmin0, cmin1, cmax0, cmax1, t);
            n = t.nx * t.ny;
        } else { t.nx = t.ny = 0; t.x0 = t.y0 = 0; t.base_inv = 0; for (int j = 0; j < 3; ++j) t.px[j] = t.py[j] = t.w3[j] = 0; }
        wave_raster(t, n, sh[wave], shq[wave], [&](int x, int y, int d) {
            const uint32_t xw = (uint32_t)(x - roi.x);
            const uint32_t yw = (uint32_t)((int)height - 1 - y - roi.y);
            atomicMin(&img[xw + (size_t)yw * rw], d);
        });
    }
}

// ================================================================================================
//  fused-path raster: per-hypothesis screen bounding box + LDS depth bands (no global atomics, no
//  full-frame clear).  The depth values are produced by exactly the arithmetic of raster_kernel;
//  only where the min is taken (LDS ds_min instead of L2/memory atomics) and which pixels are
//  touched (the conservative box of the object instead of the whole frame) differ.
// ================================================================================================

// axis-aligned box of the mesh: {minx,miny,minz,maxx,maxy,maxz}.  Any number of workgroups: each reduces a slice and merges it
// into six 32-bit keys with atomicMin -- a float maps to a key that orders like the float (sign bit flipped for positives, all
// bits for negatives); minima store the key, maxima its complement, so all six start from 0xffffffff (one memset).
__device__ __forceinline__ uint32_t f32_key(float v) { const uint32_t b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key_f32(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__global__ __launch_bounds__(256) void model_aabb_kernel(const pr_triangle *__restrict__ tris, uint32_t n_tris, uint32_t *__restrict__ keys)
{
    __shared__ float red[4][6];
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    const float *v = reinterpret_cast<const float *>(tris);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_tris * 3u; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float c = v[(size_t)i * 3 + a]; lo[a] = fminf(lo[a], c); hi[a] = fmaxf(hi[a], c); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) for (int a = 0; a < 3; ++a) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = red[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) r = (threadIdx.x < 3) ? fminf(r, red[w][threadIdx.x]) : fmaxf(r, red[w][threadIdx.x]);
        atomicMin(&keys[threadIdx.x], (threadIdx.x < 3) ? f32_key(r) : ~f32_key(r));
    }
}
// keys -> floats (aabb_out, optional) and/or a check against the box the host assumed (flag_out, optional: 1 = differs).
// An empty mesh leaves the keys untouched: +FLT_MAX / -FLT_MAX like the single-pass form.
struct AabbExpected { float v[6]; };
__global__ void model_aabb_finish_kernel(const uint32_t *__restrict__ keys, float *__restrict__ aabb_out, AabbExpected expect, uint32_t *__restrict__ flag_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool differs = false;
    for (int a = 0; a < 6; ++a) {
        const uint32_t k = keys[a];
        float v = (a < 3) ? key_f32(k) : key_f32(~k);
        if (k == 0xffffffffu) v = (a < 3) ? FLT_MAX : -FLT_MAX;
        if (aabb_out) aabb_out[a] = v;
        if (!(v == expect.v[a])) differs = true;
    }
    if (flag_out) *flag_out = differs ? 1u : 0u;
}

// Conservative pixel box {x0,y0,x1,y1} (raster coordinates, y not yet flipped) of the mesh under
// every pose: the 8 box corners go through the same model / projection / viewport arithmetic as
// the vertices; the projection of any point of the box lies in the hull of the projected corners
// as long as all of them are in front of the camera, and 2 pixels of padding cover float rounding.
// Any corner at or behind the camera plane -> the whole frame.
__global__ __launch_bounds__(256) void pose_bbox_kernel(const float *__restrict__ aabb, const pr_mat4 *__restrict__ poses, uint32_t n_poses,
                                                        pr_mat4 proj, uint32_t width, uint32_t height, pr_roi roi, int4 *__restrict__ bbox)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_poses) return;
    const float *M = poses[p].m;
    float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
    bool all_front = true;
    for (int c = 0; c < 8; ++c) {
        const float x = aabb[(c & 1) ? 3 : 0], y = aabb[(c & 2) ? 4 : 1], z = aabb[(c & 4) ? 5 : 2];
        const float lx = M[0] * x + M[1] * y + M[2] * z + M[3];
        const float ly = M[4] * x + M[5] * y + M[6] * z + M[7];
        const float lz = M[8] * x + M[9] * y + M[10] * z + M[11];
        if (!(lz > 1e-3f)) all_front = false;
        const float cxp = proj.m[0] * lx + proj.m[1] * ly + proj.m[2] * lz + proj.m[3];
        const float cyp = proj.m[4] * lx + proj.m[5] * ly + proj.m[6] * lz + proj.m[7];
        const float sx = cxp / lz * (float)width / 2.0f + (float)width / 2.0f;
        const float sy = cyp / lz * (float)height / 2.0f + (float)height / 2.0f;
        mnx = fminf(mnx, sx); mxx = fmaxf(mxx, sx); mny = fminf(mny, sy); mxy = fmaxf(mxy, sy);
    }
    int x0 = 0, y0 = 0, x1 = (int)width - 1, y1 = (int)height - 1;
    const bool finite = (mnx > -1e8f) && (mxx < 1e8f) && (mny > -1e8f) && (mxy < 1e8f);
    if (all_front && finite) {
        x0 = max(0, (int)floorf(mnx) - 2);  x1 = min((int)width - 1, (int)ceilf(mxx) + 2);
        y0 = max(0, (int)floorf(mny) - 2);  y1 = min((int)height - 1, (int)ceilf(mxy) + 2);
    }
    if (roi.width > 0 && roi.height > 0) {                       // renderer.cu:106-113: the ROI is given in image rows, the raster runs flipped
        x0 = max(x0, roi.x);  x1 = min(x1, roi.x + roi.width - 1);
        y0 = max(y0, (int)height - 1 - (roi.y + roi.height - 1));  y1 = min(y1, (int)height - 1 - roi.y);
    }
    bbox[p] = make_int4(x0, y0, x1, y1);
}

// INT_MAX-fill and per-row valid counts restricted to each hypothesis' pixel box (image rows are
// the flipped raster rows).  One wavefront per image row; rows outside the box only write count 0.
constexpr uint32_t kBoxRowsPerBlock = 16;                        // 4 wavefronts x 4 rows: few, fatter workgroups (dispatch-bound otherwise)
__global__ __launch_bounds__(256) void fill_box_kernel(int32_t *__restrict__ depth, const int4 *__restrict__ bbox, uint32_t width, uint32_t height)
{
    const uint32_t lane = threadIdx.x & 63;
    const int4 bb = bbox[blockIdx.y];
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t row = blockIdx.x * kBoxRowsPerBlock + (threadIdx.x >> 6) * 4 + r;
        if (row >= height) return;
        const int ry = (int)height - 1 - (int)row;                  // raster row of this image row
        if (ry < bb.y || ry > bb.w) continue;
        int32_t *line = depth + ((size_t)blockIdx.y * height + row) * width;
        for (int x = bb.x + (int)lane; x <= bb.z; x += 64) line[x] = INT_MAX;
    }
}
__global__ __launch_bounds__(256) void count_box_kernel(const int32_t *__restrict__ depth, const int4 *__restrict__ bbox, uint32_t width,
                                                        uint32_t height, uint32_t *__restrict__ row_count)
{
    // latency-bound, not bandwidth-bound: every lane keeps 4 rows x 4 column chunks = 16 loads in flight before the first ballot
    const uint32_t lane = threadIdx.x & 63;
    const int4 bb = bbox[blockIdx.y];
    const uint32_t row0 = blockIdx.x * kBoxRowsPerBlock + (threadIdx.x >> 6) * 4;
    uint32_t cnt[4] = { 0, 0, 0, 0 };
    for (int x0 = bb.x; x0 <= bb.z; x0 += 256) {
        int32_t v[4][4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t row = row0 + r;
            const int ry = (int)height - 1 - (int)row;
            const bool live = row < height && ry >= bb.y && ry <= bb.w;
            const int32_t *line = depth + ((size_t)blockIdx.y * height + (live ? row : 0)) * width;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int x = x0 + 64 * j + (int)lane; v[r][j] = (live && x <= bb.z) ? line[x] : 0; }
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) cnt[r] += (uint32_t)__popcll(__ballot(v[r][j] > 0 && v[r][j] != INT_MAX));
    }
    if (lane == 0) {
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) if (row0 + r < height) row_count[(size_t)blockIdx.y * height + row0 + r] = cnt[r];
    }
}

// Persistent workgroups walk the (hypothesis, band) items; a band is a run of raster rows of the
// hypothesis' box that fits the LDS tile.  Each workgroup rasterises ALL triangles against its band
// with ds_min, then writes the band (INT_MAX = empty) and its per-row valid counts.
__global__ __launch_bounds__(1024) void raster_band_kernel(const pr_triangle *__restrict__ tris, uint32_t n_tris,
                                                           const pr_mat4 *__restrict__ poses, uint32_t n_poses,
                                                           const int4 *__restrict__ bbox, int32_t *__restrict__ depth,
                                                           uint32_t *__restrict__ row_count, uint32_t width, uint32_t height,
                                                           pr_mat4 proj, uint32_t cap_px, uint32_t max_bands)
{
    extern __shared__ __attribute__((aligned(16))) int32_t tile[];
    const uint32_t n_items = n_poses * max_bands;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint32_t pose = item / max_bands, band = item - pose * max_bands;
        const int4 bb = bbox[pose];
        const int bw = bb.z - bb.x + 1, bh = bb.w - bb.y + 1;
        if (bw <= 0 || bh <= 0) continue;
        const int rows_per_band = max(1, (int)(cap_px / (uint32_t)bw));
        const int n_bands = (bh + rows_per_band - 1) / rows_per_band;
        if ((int)band >= n_bands) continue;
        const int by0 = bb.y + (int)band * rows_per_band;
        const int by1 = min(bb.w, by0 + rows_per_band - 1);
        const int n_px = (by1 - by0 + 1) * bw;
        for (int i = threadIdx.x; i < n_px; i += 1024) tile[i] = INT_MAX;
        __syncthreads();

        const float *M = poses[pose].m;
        const float cmin0 = (float)bb.x, cmax0 = (float)bb.z, cmin1 = (float)by0, cmax1 = (float)by1;
        for (uint32_t ti = threadIdx.x; ti < n_tris; ti += 1024) {
            const float *tv = reinterpret_cast<const float *>(tris + ti);
            float px[3], py[3], w3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float x = tv[3 * k], y = tv[3 * k + 1], z = tv[3 * k + 2];
                const float lx = M[0] * x + M[1] * y + M[2] * z + M[3];
                const float ly = M[4] * x + M[5] * y + M[6] * z + M[7];
                const float lz = M[8] * x + M[9] * y + M[10] * z + M[11];
                w3[k] = lz;
                const float cxp = proj.m[0] * lx + proj.m[1] * ly + proj.m[2] * lz + proj.m[3];
                const float cyp = proj.m[4] * lx + proj.m[5] * ly + proj.m[6] * lz + proj.m[7];
                px[k] = cxp / lz * (float)width / 2.0f + (float)width / 2.0f;
                py[k] = cyp / lz * (float)height / 2.0f + (float)height / 2.0f;
            }
            float lo0 = FLT_MAX, lo1 = FLT_MAX, hi0 = -FLT_MAX, hi1 = -FLT_MAX;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo0 = sel_max(cmin0, sel_min(lo0, px[k]));  hi0 = sel_min(cmax0, sel_max(hi0, px[k]));
                lo1 = sel_max(cmin1, sel_min(lo1, py[k]));  hi1 = sel_min(cmax1, sel_max(hi1, py[k]));
            }
            const float area = area2(px[0], py[0], px[1], py[1], px[2], py[2]);
            if (!(area != 0.0f)) continue;
            const float base_inv = 1 / area;
            const int x0 = loop_start(lo0 + 0.5f);
            for (int y = loop_start(lo1 + 0.5f); (float)y <= hi1; ++y) {
                const float fy = (float)y;
                for (int x = x0; (float)x <= hi0; ++x) {
                    const float fx = (float)x;
                    const float beta  = area2(px[0], py[0], fx, fy, px[2], py[2]) * base_inv;
                    const float gamma = area2(px[0], py[0], px[1], py[1], fx, fy) * base_inv;
                    const float alpha = 1.0f - beta - gamma;
                    if (alpha < -0.0f || beta < -0.0f || gamma < -0.0f || alpha > 1.0f || beta > 1.0f || gamma > 1.0f) continue;
                    const float az = alpha / w3[0], bz = beta / w3[1], gz = gamma / w3[2];
                    const float frag = (alpha + beta + gamma) / (az + bz + gz);
                    atomicMin(&tile[(y - by0) * bw + (x - bb.x)], f2i_x86(frag + 0.5f));
                }
            }
        }
        __syncthreads();

        // write the band: raster row y lands on image row height-1-y (renderer.cu:142)
        int32_t *img = depth + (size_t)pose * width * height;
        const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int r = (int)wave; r <= by1 - by0; r += 16) {
            const int yw = (int)height - 1 - (by0 + r);
            uint32_t cnt = 0;
            for (int c0 = 0; c0 < bw; c0 += 64) {
                const int c = c0 + (int)lane;
                int32_t v = INT_MAX;
                if (c < bw) { v = tile[r * bw + c]; img[(size_t)yw * width + bb.x + c] = v; }
                cnt += (uint32_t)__popcll(__ballot(c < bw && v > 0 && v != INT_MAX));
            }
            if (lane == 0) row_count[(size_t)pose * height + yw] = cnt;
        }
        __syncthreads();
    }
}

// depth2cloud emit restricted to the hypothesis' pixel box (the fused path never reads outside it)
__global__ __launch_bounds__(256) void d2c_emit_box_kernel(const int32_t *__restrict__ depth, uint32_t width, uint32_t height,
                                                           const int4 *__restrict__ bbox, float fx, float fy, float cx, float cy,
                                                           const uint32_t *__restrict__ row_count, const uint32_t *__restrict__ row_off,
                                                           pr_vec3 *__restrict__ cloud, size_t cloud_stride)
{
    const uint32_t lane = threadIdx.x & 63;
    const int4 bb = bbox[blockIdx.y];
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t row = blockIdx.x * kBoxRowsPerBlock + (threadIdx.x >> 6) * 4 + r;
        if (row >= height) return;
        if (row_count[(size_t)blockIdx.y * height + row] == 0) continue;
        const int32_t *line = depth + ((size_t)blockIdx.y * height + row) * width;
        pr_vec3 *out = cloud + (size_t)blockIdx.y * cloud_stride + row_off[(size_t)blockIdx.y * height + row];
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
        m.start = i * cloud_stride; m.count = c; m.state = c > 0 ? kRun : kSkip; m.pad = 0;
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

// ================================================================================================
//  scene queries
// ================================================================================================
typedef float float2v __attribute__((ext_vector_type(2)));
// base + 32-bit byte offset: lets the compiler address with a scalar base and one 32-bit VGPR (global_load ... v_off, s[base])
// instead of building a 64-bit address per lane (v_lshl_add_u64 / v_mad_u64_u32 plus the copies that come with 64-bit values)
template <class T> __device__ __forceinline__ T ld_off(const void *base, uint32_t byte_off)
{ return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off); }
template <class T> __device__ __forceinline__ void st_off(void *base, uint32_t byte_off, const T &v)
{ *reinterpret_cast<T *>(static_cast<char *>(base) + byte_off) = v; }
struct Corr { float dx, dy, dz, nx, ny, nz; };   // destination point and its normal

// Correctly rounded a / b for every operand pair whose quotient is in the normal range: the Newton/FMA sequence the
// compiler emits for an IEEE f32 division (LLVM LowerFDIV32) without its div_scale / div_fmas / div_fixup range handling
// (3 of 11 instructions).  Used only where an out-of-range or non-finite quotient is rejected right afterwards anyway.
__device__ __forceinline__ float div_normal_range(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    float q = a * r;
    const float e1 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e2, r, q);
}

// Scene_projective::query depth_scene.h:29-48 + pcd2dep common.h:63-73.  The reference converts
// int(x/z*fx + cx - tl_x + 0.5f) and then tests 0 <= px < width; truncation maps (-1, 0) to pixel 0, and NaN / out-of-range
// values end up as INT_MIN on x86 (rejected).  The same decision is taken here on the float before converting: valid iff
// -1 < v < width.  Whenever the quotient is outside the normal range (z = 0, denormal, inf, NaN) the point is rejected on
// both sides -- by this range test or by the |src.z - dst.z| test -- so the cheaper division above cannot change a result.
__device__ __forceinline__ bool proj_pixel(float sx, float sy, float sz, float fx, float fy, float cx, float cy, float tlx, float tly,
                                           uint32_t width, uint32_t height, uint32_t &idx, int &px, int &py)
{
    // common.h:64-67: int(x/z*fx + cx - tl_x + 0.5f); tl_* are size_t in the reference and enter the float expression converted
    // to float (0.0f for a scene that covers the whole frame -- subtracting it is exact).
    // Both coordinates go through div_normal_range's sequence side by side in packed instructions (v_pk_mul_f32 / v_pk_fma_f32 /
    // v_pk_add_f32: two IEEE operations per issue slot, each element rounded exactly like the scalar form); the reciprocal of z and
    // its first refinement are shared.
    float r = __builtin_amdgcn_rcpf(sz);
    const float e0 = __builtin_fmaf(-sz, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    const float2v a{ sx, sy }, rr{ r, r }, nz{ -sz, -sz };
    float2v q = a * rr;
    const float2v e1 = __builtin_elementwise_fma(nz, q, a);
    q = __builtin_elementwise_fma(e1, rr, q);
    const float2v e2 = __builtin_elementwise_fma(nz, q, a);
    q = __builtin_elementwise_fma(e2, rr, q);
    const float2v v = q * float2v{ fx, fy } + float2v{ cx, cy } - float2v{ tlx, tly } + float2v{ 0.5f, 0.5f };
    const float vx = v.x, vy = v.y;
    // no early exit: the conversions of an out-of-range value saturate harmlessly and the caller only uses px / py / idx when
    // the test passed (straight-line code keeps the packed values out of merge copies)
#ifndef PR_BRANCHLESS_PROJ
#define PR_BRANCHLESS_PROJ 0                                   // measured (tools/ab_flags.sh, same box): the straight-line form is 1.8 % slower
#endif
#if PR_BRANCHLESS_PROJ
    const bool in_img = (vx > -1.0f) & (vx < (float)width) & (vy > -1.0f) & (vy < (float)height);
    px = (int)vx; py = (int)vy;
    idx = (uint32_t)px + (uint32_t)py * width;
    return in_img;
#else
    if (!(vx > -1.0f && vx < (float)width && vy > -1.0f && vy < (float)height)) return false;
    px = (int)vx; py = (int)vy;
    idx = (uint32_t)px + (uint32_t)py * width;
    return true;
#endif
}

__device__ __forceinline__ bool query(const SceneProjAoS &s, float sx, float sy, float sz, Corr &c)
{
    uint32_t idx; int px, py;
    if (!proj_pixel(sx, sy, sz, s.fx, s.fy, s.cx, s.cy, s.tlx, s.tly, s.width, s.height, idx, px, py)) return false;
    const float *d = reinterpret_cast<const float *>(s.pcd + idx);
    const float dz = d[2];
    const float diff = sz - dz;
    const float adiff = (diff > 0) ? diff : -diff;
    if (dz <= 0 || adiff > s.max_dist_diff) return false;
    const float *n = reinterpret_cast<const float *>(s.normal + idx);
    c.dx = d[0]; c.dy = d[1]; c.dz = dz; c.nx = n[0]; c.ny = n[1]; c.nz = n[2];
    return true;
}

__device__ __forceinline__ bool query(const SceneProjPacked &s, float sx, float sy, float sz, Corr &c)
{
    uint32_t idx; int px, py;
    if (!proj_pixel(sx, sy, sz, s.fx, s.fy, s.cx, s.cy, s.tlx, s.tly, s.width, s.height, idx, px, py)) return false;
    const float4 r = s.rec[idx];                                 // one 16-byte gather: {nx, ny, nz, z}
    const float dz = r.w;
    const float diff = sz - dz;
    const float adiff = (diff > 0) ? diff : -diff;
    if (dz <= 0 || adiff > s.max_dist_diff) return false;
    // the scene point is re-derived from its depth exactly as dep2pcd (common.h:47-61) built it:
    // colf[x] = ((float)x - cx)/fx and rowf[y] = ((float)y - cy)/fy are tabulated per scene
    // (same float operations, evaluated once per column/row instead of once per point)
    c.dx = s.colf[px] * dz;
    c.dy = s.rowf[py] * dz;
    c.dz = dz; c.nx = r.x; c.ny = r.y; c.nz = r.z;
    return true;
}

// Split form of the two projective queries above for the hot kernel: gather_issue() computes the
// pixel and starts the loads for it without looking at the data, gather_finish() applies the
// validity rules of depth_scene.h:38-45.  Same arithmetic as query(); only the order of memory
// operations differs.
struct Gathered { float a0, a1, a2, a3, b0, b1, b2; };

__device__ __forceinline__ bool gather_issue(const SceneProjAoS &s, float sx, float sy, float sz, bool live, Gathered &g)
{
    uint32_t idx; int px, py;
    const bool in_img = live && proj_pixel(sx, sy, sz, s.fx, s.fy, s.cx, s.cy, s.tlx, s.tly, s.width, s.height, idx, px, py);
    const uint32_t at = in_img ? idx : 0u;
    const float *d = reinterpret_cast<const float *>(s.pcd + at);
    const float *n = reinterpret_cast<const float *>(s.normal + at);
    g.a0 = d[0]; g.a1 = d[1]; g.a2 = d[2]; g.a3 = 0.0f; g.b0 = n[0]; g.b1 = n[1]; g.b2 = n[2];
    return in_img;
}
__device__ __forceinline__ bool gather_finish(const SceneProjAoS &s, bool in_img, float sz, const Gathered &g, Corr &c)
{
    const float dz = g.a2;
    const float diff = sz - dz;
    const float adiff = (diff > 0) ? diff : -diff;
    if (!in_img || dz <= 0 || adiff > s.max_dist_diff) return false;
    c.dx = g.a0; c.dy = g.a1; c.dz = dz; c.nx = g.b0; c.ny = g.b1; c.nz = g.b2;
    return true;
}
__device__ __forceinline__ bool gather_issue(const SceneProjPacked &s, float sx, float sy, float sz, bool live, Gathered &g)
{
    // pixel 0 stands in for points that do not project into the image (the three values keep their zeros on that path)
    uint32_t idx = 0u; int px = 0, py = 0;
    const bool in_img = live && proj_pixel(sx, sy, sz, s.fx, s.fy, s.cx, s.cy, s.tlx, s.tly, s.width, s.height, idx, px, py);
    const float4 r = ld_off<float4>(s.rec, idx * 16u);                                 // {nx, ny, nz, z}
    g.a0 = r.x; g.a1 = r.y; g.a2 = r.z; g.a3 = r.w;
    g.b0 = ld_off<float>(s.colf, (uint32_t)px * 4u); g.b1 = ld_off<float>(s.rowf, (uint32_t)py * 4u); g.b2 = 0.0f;
    return in_img;
}
__device__ __forceinline__ bool gather_finish(const SceneProjPacked &s, bool in_img, float sz, const Gathered &g, Corr &c)
{
    const float dz = g.a3;
    const float diff = sz - dz;
    const float adiff = (diff > 0) ? diff : -diff;
    if (!in_img || dz <= 0 || adiff > s.max_dist_diff) return false;
    c.dx = g.b0 * dz; c.dy = g.b1 * dz; c.dz = dz; c.nx = g.a0; c.ny = g.a1; c.nz = g.a2;
    return true;
}
// never instantiated for the kd-tree scenes (they have their own loops)
__device__ __forceinline__ bool gather_issue(const SceneNNDev &, float, float, float, bool, Gathered &) { return false; }
__device__ __forceinline__ bool gather_finish(const SceneNNDev &, bool, float, const Gathered &, Corr &) { return false; }
__device__ __forceinline__ bool gather_issue(const SceneNNWinners &, float, float, float, bool, Gathered &) { return false; }
__device__ __forceinline__ bool gather_finish(const SceneNNWinners &, bool, float, const Gathered &, Corr &) { return false; }

// Scene_nn::query pcd_scene.h:60-136 -- same stackless near-first traversal, same strict '<' on
// leaf points and '<=' on the bound, but the bound is the FAR CHILD's own tight box instead of
// the current node's box.  That prunes a superset of what the reference prunes and can only skip
// subtrees whose every point is farther than the current best, so winner and distance are
// identical (DESIGN.md "kd-tree bound").
// dequantisation of a compact-record box coordinate (nn_records32_kernel): one multiply, one add
__device__ __forceinline__ float nn_deq(uint32_t q, float qmin, float qscale) { return qmin + (float)q * qscale; }
__device__ __forceinline__ float box_dist_sq(float sx, float sy, float sz, const float4 lo, const float4 hi)
{
    float lb = 0;
    if (sx < lo.x) lb += (lo.x - sx) * (lo.x - sx); else if (sx > hi.x) lb += (hi.x - sx) * (hi.x - sx);
    if (sy < lo.y) lb += (lo.y - sy) * (lo.y - sy); else if (sy > hi.y) lb += (hi.y - sy) * (hi.y - sy);
    if (sz < lo.z) lb += (lo.z - sz) * (lo.z - sz); else if (sz > hi.z) lb += (hi.z - sz) * (hi.z - sz);
    return lb;
}

template <bool kUseLds>
__device__ __forceinline__ bool query_nn(const SceneNNDev &s, const int4 *lds_topo, float sx, float sy, float sz, Corr &c)
{
    int cur = 0, prev = -1, best_i = 0;
    bool climbing = false;
    float best = FLT_MAX;
    while (cur >= 0) {
        const int4 t = (kUseLds && (uint32_t)cur < s.lds_nodes) ? lds_topo[cur] : s.topo[cur];
        const int parent = (t.w & 0x3fffffff) - 1;
        const bool leaf = t.z < 0;
        if (!climbing && leaf) {
            for (int i = t.x; i < t.y; ++i) {
                const float4 p = s.pts[i];
                const float d2 = (sx - p.x) * (sx - p.x) + (sy - p.y) * (sy - p.y) + (sz - p.z) * (sz - p.z);
                if (d2 < best) { best = d2; best_i = i; }
            }
            climbing = true; prev = cur; cur = parent;
            continue;
        }
        const int dim = (int)((uint32_t)t.w >> 30);
        const float q = (dim == 0) ? sx : ((dim == 1) ? sy : sz);
        const float diff = q - __int_as_float(t.x);
        const int near_c = (diff < 0) ? t.y : t.z;
        const int far_c  = (diff < 0) ? t.z : t.y;
        if (!climbing) { prev = cur; cur = near_c; continue; }
        if (prev == near_c) {
            const float lb = box_dist_sq(sx, sy, sz, s.bmin[far_c], s.bmax[far_c]);
            if (lb <= best) { prev = cur; cur = far_c; climbing = false; continue; }
        }
        prev = cur; cur = parent;
    }
    if (!(best < s.max_dist_diff * s.max_dist_diff)) return false;
    const float *d = reinterpret_cast<const float *>(s.pcd + best_i);
    const float *n = reinterpret_cast<const float *>(s.normal + best_i);
    c.dx = d[0]; c.dy = d[1]; c.dz = d[2]; c.nx = n[0]; c.ny = n[1]; c.nz = n[2];
    return true;
}

// Stack variant of the same search.  The order in which leaves are visited is the reference's
// near-first depth-first order: a far child is parked on a per-lane stack (LDS, [entry][lane] so the
// 64 lanes of a wavefront hit 64 different banks) together with the distance of the query to its
// box, and re-tested against the current best when it is popped -- exactly the test the stackless
// walk makes when it climbs back to the parent (pcd_scene.h:115).  Each internal node is fetched
// once, as one 64-byte record that already contains both children's boxes.
//   record = { split_v | left, child1 | right, child2 | -1, dim,  c1.min.xyz c1.max.xyz  c2.min.xyz c2.max.xyz }
#ifndef PR_LEAF_BATCH
#define PR_LEAF_BATCH 10
#endif
constexpr int kLeafBatch = PR_LEAF_BATCH;
#ifndef PR_NN_BOUNDED
#define PR_NN_BOUNDED 1
#endif
#ifndef PR_NN_LEAF12
#define PR_NN_LEAF12 1
#endif
#ifndef PR_NN_NEAR_TEST
#define PR_NN_NEAR_TEST 1
#endif
#ifndef PR_NN_WIDE_BOUND
#define PR_NN_WIDE_BOUND 4.0e-6f                                // (2 mm)^2: above it a node's whole record is fetched at once
#endif
constexpr uint32_t kNoPrev = 0xffffffffu;
#ifndef PR_NN_COVER_PAD
#define PR_NN_COVER_PAD 5.0e-4f                                 // metres a window search looks beyond its bound for the runner-up (about one pixel at 300 mm)
#endif
#ifndef PR_NN_NODESCENT
#define PR_NN_NODESCENT 2.5e-7f                                // squared step up to which the previous winner's distance is bound enough (no descent through the representatives)
#endif
#ifndef PR_ARRIVE_ACQ_REL
#define PR_ARRIVE_ACQ_REL 0
#endif
#ifndef PR_PYR_LOCAL64
#define PR_PYR_LOCAL64 1
#endif
#ifndef PR_RING_ROWS
#define PR_RING_ROWS 2
#endif
#ifndef PR_NN_SETTLE
#define PR_NN_SETTLE 1                                           // grid_search: widest cover for points that have stopped moving
#endif
#ifndef PR_NN_STILL
#define PR_NN_STILL 2.5e-7f                                     // (0.5 mm)^2: below this step a point's previous winner is taken as a tight seed
#endif

// Bound a search may start from when scene point `seed` is known to exist: its distance, inflated by one part in a million so
// that the point itself -- or an equal one visited earlier -- is still found by the strict '<' of the search.  The bound only
// removes subtrees and points that are strictly farther than an existing point; winner, distance and tie-break are those of the
// unseeded search.
__device__ __forceinline__ void nn_seed_bound(const SceneNNDev &s, float sx, float sy, float sz, uint32_t seed, float &best)
{
    const pr_vec3 p = s.pcd[seed != kNoPrev ? seed : 0u];
    if (seed != kNoPrev) {
        const float d2 = (sx - p.x) * (sx - p.x) + (sy - p.y) * (sy - p.y) + (sz - p.z) * (sz - p.z);
        const float b = d2 * 1.000001f + 1e-30f;
        if (b < best) best = b;
    }
}

// kCode = stack entries per lane (16 / 24), + 0x100 when the scene's compact 32-byte records are used.
// best_init: the bound the search starts from (<= max_dist_diff^2, see PR_NN_BOUNDED; tightened by nn_seed_bound).
// work counters of the search (SURVEY 8d "count its own visits"): per lane, summed into SceneNNDev::counters when that is set
struct NNCount { uint32_t nodes = 0, leaves = 0, leaf_points = 0; };
template <int kCode, bool kWantCorr = true>
__device__ __forceinline__ bool query_nn_stack_from(const SceneNNDev &s, const float4 *lds_rec, int *stk_node, float *stk_lb, float sx, float sy, float sz, Corr &c,
                                                    float best_init, uint32_t &winner, NNCount *cnt = nullptr)
{
    constexpr int kDepth = kCode & 0xff;
    constexpr bool kCompact = (kCode & 0x100) != 0;
    int cur = 0, sp = 0, best_i = -1;                            // -1: no point below the starting bound yet
    float best = best_init;
    winner = kNoPrev;
    for (;;) {
        float4 h, b0, b1, b2;
        if (cnt) cnt->nodes++;
        if constexpr (kCompact) {
            // compact records: half the bytes through the L1 (the kernel is bound by the texture-addresser / L1 rate of its
            // divergent loads, not by latency or HBM); only the far child's box is decoded
            // While the bound is wide (first passes, first descent) the whole 32-byte record is fetched at once.  Once it is
            // tight, 8 bytes (split | child | dim) are enough for most nodes: every point of the far side lies beyond the
            // split plane (left_max <= split <= right_min, pcd_scene.cpp:140-160), so (q - split)^2 is a lower bound of its
            // distance in the same float arithmetic, and only if that bound does not exceed `best` is the far box needed.
            const bool wide = best > PR_NN_WIDE_BOUND;
            uint4 A, B;
            if (wide) { A = s.rec32[(size_t)cur * 2]; B = s.rec32[(size_t)cur * 2 + 1]; }
            else { const uint2 d = s.desc[cur]; A = make_uint4(d.x, d.y, 0u, 0u); B = make_uint4(0u, 0u, 0u, 0u); }
            const uint32_t tag = A.y >> 30;
            if (tag == 3u) { h = make_float4(__uint_as_float(A.x), __uint_as_float(A.y & 0x3fffffffu), __int_as_float(-1), 0.0f); b0 = b1 = b2 = h; }
            else {
                const float q = (tag == 0u) ? sx : ((tag == 1u) ? sy : sz);
                const float diff = q - __uint_as_float(A.x);
                const bool left_near = diff < 0;
                const uint32_t c1 = A.y & 0x3fffffffu;
                const int near_c = (int)(left_near ? c1 : c1 + 1u), far_c = (int)(left_near ? c1 + 1u : c1);
                if (diff * diff <= best && sp < kDepth) {
                    if (!wide) { A = s.rec32[(size_t)cur * 2]; B = s.rec32[(size_t)cur * 2 + 1]; }
                    const uint32_t u0 = left_near ? B.y : A.z, u1 = left_near ? B.z : A.w, u2 = left_near ? B.w : B.x;
                    const float4 lo = make_float4(nn_deq(u0 & 0xffffu, s.qmin[0], s.qscale[0]), nn_deq(u0 >> 16, s.qmin[1], s.qscale[1]),
                                                  nn_deq(u1 & 0xffffu, s.qmin[2], s.qscale[2]), 0.0f);
                    const float4 hi = make_float4(nn_deq(u1 >> 16, s.qmin[0], s.qscale[0]), nn_deq(u2 & 0xffffu, s.qmin[1], s.qscale[1]),
                                                  nn_deq(u2 >> 16, s.qmin[2], s.qscale[2]), 0.0f);
                    const float lb = box_dist_sq(sx, sy, sz, lo, hi);
                    if (lb <= best) { stk_node[sp * kBlockThreads] = far_c; stk_lb[sp * kBlockThreads] = lb; ++sp; }
                }
#if PR_NN_NEAR_TEST
                // While the bound is wide the record is in registers anyway: the near child is entered only if its own box is
                // within the bound (the reference walks into it unconditionally and finds nothing there).
                if (wide) {
                    const uint32_t v0 = left_near ? A.z : B.y, v1 = left_near ? A.w : B.z, v2 = left_near ? B.x : B.w;
                    const float4 nlo = make_float4(nn_deq(v0 & 0xffffu, s.qmin[0], s.qscale[0]), nn_deq(v0 >> 16, s.qmin[1], s.qscale[1]),
                                                   nn_deq(v1 & 0xffffu, s.qmin[2], s.qscale[2]), 0.0f);
                    const float4 nhi = make_float4(nn_deq(v1 >> 16, s.qmin[0], s.qscale[0]), nn_deq(v2 & 0xffffu, s.qmin[1], s.qscale[1]),
                                                   nn_deq(v2 >> 16, s.qmin[2], s.qscale[2]), 0.0f);
                    if (!(box_dist_sq(sx, sy, sz, nlo, nhi) <= best)) {
                        bool found = false;
                        while (sp > 0) {
                            --sp;
                            if (stk_lb[sp * kBlockThreads] <= best) { cur = stk_node[sp * kBlockThreads]; found = true; break; }
                        }
                        if (!found) break;
                        continue;
                    }
                }
#endif
                cur = near_c;
                continue;
            }
        } else
        // 64-byte records; the leading (top-level) ones may be staged in LDS
        if ((uint32_t)cur < s.lds_nodes) {
            const float4 *q = lds_rec + (size_t)cur * 4;
            h = q[0]; b0 = q[1]; b1 = q[2]; b2 = q[3];
        } else {
            const float4 *q = s.rec + (size_t)cur * 4;
            h = q[0]; b0 = q[1]; b1 = q[2]; b2 = q[3];
        }
        const int hz = __float_as_int(h.z);
        if (hz < 0) {                                            // leaf: points [left, right)
            const int lo = __float_as_int(h.x), hi = __float_as_int(h.y);
            if (cnt) { cnt->leaves++; cnt->leaf_points += (uint32_t)(hi - lo); }
            // four point loads are issued before the first compare (one memory round trip per batch); indices past
            // the end are clamped to the last point, whose repeated distance can never pass the strict '<' again,
            // so the in-order compares keep the reference's first-occurrence winner
#pragma unroll 1
            for (int i = lo; i < hi; i += kLeafBatch) {
                float d2[kLeafBatch];
                int idx[kLeafBatch];
#pragma unroll
                for (int k = 0; k < kLeafBatch; ++k) {
                    idx[k] = (i + k < hi) ? (i + k) : (hi - 1);
#if PR_NN_LEAF12
                    const pr_vec3 p = s.pcd[idx[k]];                 // 12-byte points: a quarter fewer bytes through the L1 than the padded copy
#else
                    const float4 p = s.pts[idx[k]];
#endif
                    d2[k] = (sx - p.x) * (sx - p.x) + (sy - p.y) * (sy - p.y) + (sz - p.z) * (sz - p.z);
                }
#pragma unroll
                for (int k = 0; k < kLeafBatch; ++k)
                    if (d2[k] < best) { best = d2[k]; best_i = idx[k]; }
            }
            bool found = false;
            while (sp > 0) {
                --sp;
                if (stk_lb[sp * kBlockThreads] <= best) { cur = stk_node[sp * kBlockThreads]; found = true; break; }
            }
            if (!found) break;
        } else {
            const int dim = __float_as_int(h.w);
            const float q = (dim == 0) ? sx : ((dim == 1) ? sy : sz);
            const float diff = q - h.x;
            const bool left_near = diff < 0;
            const int near_c = left_near ? __float_as_int(h.y) : hz;
            const int far_c  = left_near ? hz : __float_as_int(h.y);
            const float4 lo = left_near ? make_float4(b1.z, b1.w, b2.x, 0.f) : make_float4(b0.x, b0.y, b0.z, 0.f);
            const float4 hi = left_near ? make_float4(b2.y, b2.z, b2.w, 0.f) : make_float4(b0.w, b1.x, b1.y, 0.f);
            const float lb = box_dist_sq(sx, sy, sz, lo, hi);
            if (lb <= best && sp < kDepth) { stk_node[sp * kBlockThreads] = far_c; stk_lb[sp * kBlockThreads] = lb; ++sp; }
            cur = near_c;
        }
    }
    if (!(best < s.max_dist_diff * s.max_dist_diff) || best_i < 0) return false;
    winner = (uint32_t)best_i;
    if constexpr (kWantCorr) {
        const float4 d = s.pts[best_i];                          // {x,y,z,0} copy of pcd[best_i]
        const float *n = reinterpret_cast<const float *>(s.normal + best_i);
        c.dx = d.x; c.dy = d.y; c.dz = d.z; c.nx = n[0]; c.ny = n[1]; c.nz = n[2];
    }
    return true;
}

// the search started from the acceptance bound and, for compact records, from up to two known scene points:
//   seed  = the previous pass' winner of this cloud point (temporal),
//   seed2 = the winner of the cloud point this lane handled just before (its neighbour in the image, 1-2 mm away): on
//           the first passes, when the cloud is still centimetres off the surface, that neighbour's answer is a far
//           tighter bound than anything the descent finds early
template <int kCode>
__device__ __forceinline__ bool query_nn_stack(const SceneNNDev &s, const float4 *lds_rec, int *stk_node, float *stk_lb, float sx, float sy, float sz, Corr &c,
                                               uint32_t seed, uint32_t seed2, uint32_t &winner)
{
#if PR_NN_BOUNDED
    // a winner is only accepted below max_dist_diff^2 (pcd_scene.h query tail), so the search can start from that bound:
    // subtrees and points at or beyond it could only produce a neighbour the final test rejects
    float best = s.max_dist_diff * s.max_dist_diff;
#else
    float best = FLT_MAX;
#endif
    if constexpr ((kCode & 0x100) != 0) { nn_seed_bound(s, sx, sy, sz, seed, best); nn_seed_bound(s, sx, sy, sz, seed2, best); }
    return query_nn_stack_from<kCode>(s, lds_rec, stk_node, stk_lb, sx, sy, sz, c, best, winner);
}

// ---- wide records: eight subtree boxes per 128-byte line, searched as tasks of four lanes (nn_tree_wide_kernel) -----------------
// A per-lane walk pays for every (lane, load instruction) pair: 44 records of 32 bytes per query while a hypothesis is still
// centimetres off the surface, each lane on a line of its own.  (Measured first: the same wide nodes walked one query per lane --
// 15 node visits instead of 44, but eight 16-byte loads per visit and lane: pass 0 went from 5.2 to 6.8 ms.)  The wide form is made
// for cooperative access instead: a WIDE NODE is one 128-byte line of eight 16-byte slots {box as 6 x uint16 in the root-box
// frame, rounded OUTWARDS and verified with nn_deq_fma | reference}, one slot per descendant of a binary node (the frontier the
// builder reaches by repeatedly opening the largest box, about three binary levels).  A group of four adjacent lanes handles one
// (query, node): lane c loads slots 2c and 2c+1 -- the group's loads are ONE coalesced line -- and tests those two boxes.  A leaf
// reference carries (first point, count): the group reads the leaf's points from the padded point array, two 16-byte points per lane.
//   reference: kWideLeaf | count << 27 | first point   (leaf, count 1..15)   |   index of a wide node   |   kWideEmpty
// The search is ORDER-FREE: every admitted child becomes a task of its own.  That finds the minimum squared distance m and every
// point attaining it, because a box is only skipped when its lower bound EXCEEDS the query's bound (>= m).  The reference's answer (pcd_scene.h:60-136) is the first point at distance m in ITS visiting order: if exactly one
// point attains m, that is the answer under any order (the argument the pixel window already rests on); if several do, or a stack
// overflows, the query is repeated by the ordered stackless walk (query_nn_bounded), started from m.  `second`, a lower bound on the
// squared distance of every scene point other than the winner (skipped boxes count with their bound), lets the following passes
// keep this winner without searching (nn_search_kernel) -- the ordered walks cannot report it.
constexpr uint32_t kWideLeaf = 0x80000000u, kWideEmpty = 0xffffffffu;
constexpr uint32_t kWideMaxLeafPoints = 15u, kWideFirstMask = 0x07ffffffu;
__device__ __forceinline__ float nn_deq_fma(uint32_t q, float qmin, float qscale) { return __builtin_fmaf((float)q, qscale, qmin); }
__device__ __forceinline__ float wide_box_lb(float sx, float sy, float sz, uint32_t u0, uint32_t u1, uint32_t u2, const SceneNNDev &s)
{
    const float lox = nn_deq_fma(u0 & 0xffffu, s.qmin[0], s.qscale[0]), loy = nn_deq_fma(u0 >> 16, s.qmin[1], s.qscale[1]);
    const float loz = nn_deq_fma(u1 & 0xffffu, s.qmin[2], s.qscale[2]), hix = nn_deq_fma(u1 >> 16, s.qmin[0], s.qscale[0]);
    const float hiy = nn_deq_fma(u2 & 0xffffu, s.qmin[1], s.qscale[1]), hiz = nn_deq_fma(u2 >> 16, s.qmin[2], s.qscale[2]);
    // = box_dist_sq: per axis (lo - q)^2 below the box, (hi - q)^2 == (q - hi)^2 above it, 0 inside
    const float dx = fmaxf(fmaxf(lox - sx, sx - hix), 0.0f), dy = fmaxf(fmaxf(loy - sy, sy - hiy), 0.0f), dz = fmaxf(fmaxf(loz - sz, sz - hiz), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}
// Scene_nn::query pcd_scene.h:60-136 as query_nn above, started from a bound (an existing point's distance, inflated: nn_seed_bound) and
// reporting the winner's index: the ordered walk ties and overflows of the wide search fall back to.  No LDS.
__device__ __forceinline__ uint32_t query_nn_bounded(const SceneNNDev &s, float sx, float sy, float sz, float best_init)
{
    int cur = 0, prev = -1, best_i = -1;
    bool climbing = false;
    float best = best_init;
    while (cur >= 0) {
        const int4 t = s.topo[cur];
        const int parent = (t.w & 0x3fffffff) - 1;
        if (!climbing && t.z < 0) {
            for (int i = t.x; i < t.y; ++i) {
                const float4 p = s.pts[i];
                const float d2 = (sx - p.x) * (sx - p.x) + (sy - p.y) * (sy - p.y) + (sz - p.z) * (sz - p.z);
                if (d2 < best) { best = d2; best_i = i; }
            }
            climbing = true; prev = cur; cur = parent;
            continue;
        }
        const int dim = (int)((uint32_t)t.w >> 30);
        const float q = (dim == 0) ? sx : ((dim == 1) ? sy : sz);
        const float diff = q - __int_as_float(t.x);
        const int near_c = (diff < 0) ? t.y : t.z;
        const int far_c  = (diff < 0) ? t.z : t.y;
        if (!climbing) { prev = cur; cur = near_c; continue; }
        if (prev == near_c) {
            const float lb = box_dist_sq(sx, sy, sz, s.bmin[far_c], s.bmax[far_c]);
            if (lb <= best) { prev = cur; cur = far_c; climbing = false; continue; }
        }
        prev = cur; cur = parent;
    }
    return (best_i >= 0 && best < s.max_dist_diff * s.max_dist_diff) ? (uint32_t)best_i : kNoPrev;
}

// ---- pixel grid of a kd-tree scene ---------------------------------------------------------------------------------------
// A Scene_nn is made from a depth image (pcd_scene.cpp:10-29), so its points are the pixels of that image: cell (px, py) of the
// grid holds the point that projects into it.  With a valid upper bound B on the squared nearest-neighbour distance (from a
// seed: an existing scene point), every scene point closer than sqrt(B) projects into a small pixel window around the query's
// own pixel, and scanning that window IS the exact search -- except for the tie-break between equidistant points, which the
// reference resolves by traversal order: a tie (or an empty window, or a window too large to pay) hands the query to the
// kd-tree search.  A unique strict minimum is the reference's winner under any visiting order.
//   window: u = x/z*fx + cx.  For |p - q| <= r and z_q - r > 0:  |u_p - u_q| <= fx * r * (z_q + |x_q|) / (z_q * (z_q - r)),
//   and |floor(a) - floor(b)| <= ceil(|a - b|); r carries a 1e-4 relative margin and the window another 1e-3 px for rounding.
__device__ __forceinline__ void grid_project(const SceneNNDev &s, float x, float y, float z, float &u, float &v)
{
    u = x / z * s.gfx + s.gcx + 0.5f;
    v = y / z * s.gfy + s.gcy + 0.5f;
}
#ifndef PR_GRID_MAXW
#define PR_GRID_MAXW 2
#endif
constexpr int kGridMaxW = PR_GRID_MAXW;                                     // windows up to 5 x 5 cells; larger ones go to the tree
// half-widths of the pixel window that holds every scene point closer than sqrt(bound); false when it exceeds kGridMaxW
__device__ __forceinline__ bool grid_window(const SceneNNDev &s, float sx, float sy, float sz, float bound, int &wx, int &wy)
{
    const float r = sqrtf(bound) * 1.0001f;
    if (!(sz - r > 0.25f * sz)) return false;                    // also NaN and points at or behind the camera
    const float k = r / (sz * (sz - r));
    const float du = s.gfx * k * (sz + fabsf(sx)), dv = s.gfy * k * (sz + fabsf(sy));
    if (!(du <= (float)kGridMaxW - 1e-3f && dv <= (float)kGridMaxW - 1e-3f)) return false;
    wx = (int)ceilf(du + 1e-3f); wy = (int)ceilf(dv + 1e-3f);
    return true;
}
// Coarse-to-fine descent through the representative points: the nearest of ALL 64 x 64-block representatives, then the nearest
// 16 x 16-block representative in and around that block (the block's 4 x 4 children plus one ring), then 4 x 4 blocks, then pixels.
// Every point met is an existing scene point, so its distance is a valid bound (nn_seed_bound's argument); the descent itself
// decides nothing.  On the test.cpp scene it lands on the true nearest neighbour for 70-85 % of the queries of a hypothesis that
// starts centimetres off the surface and within a few points of it for the rest (188 distance evaluations at 640 x 480) -- which
// turns the tree search that follows from "find the neighbour" into "confirm it".
__device__ __forceinline__ void grid_ring_min(const float4 *__restrict__ level, int lw, int lh, int x0, int y0, int nb, float sx, float sy, float sz,
                                              float &dmin, int &bx, int &by)
{
    dmin = FLT_MAX; bx = min(max(x0, 0), lw - 1); by = min(max(y0, 0), lh - 1);
    // kRingRows rows (6 cells each) are in flight at a time: the descent is a chain of dependent round trips, three levels of them
    constexpr int kRingRows = PR_RING_ROWS;
    for (int dy0 = 0; dy0 < nb; dy0 += kRingRows) {
        float4 c[kRingRows][6];
        int yy[kRingRows];
#pragma unroll
        for (int r = 0; r < kRingRows; ++r) {
            yy[r] = min(max(y0 + min(dy0 + r, nb - 1), 0), lh - 1);
#pragma unroll
            for (int dx = 0; dx < 6; ++dx) c[r][dx] = level[(size_t)yy[r] * lw + min(max(x0 + dx, 0), lw - 1)];
        }
#pragma unroll
        for (int r = 0; r < kRingRows; ++r)
#pragma unroll
            for (int dx = 0; dx < 6; ++dx) {
                const float d2 = (sx - c[r][dx].x) * (sx - c[r][dx].x) + (sy - c[r][dx].y) * (sy - c[r][dx].y) + (sz - c[r][dx].z) * (sz - c[r][dx].z);
                if (d2 < dmin) { dmin = d2; bx = min(max(x0 + dx, 0), lw - 1); by = yy[r]; }
            }
    }
}
__device__ __forceinline__ void grid_pyramid_bound(const SceneNNDev &s, float sx, float sy, float sz, float &best)
{
    const int w4 = ((int)s.gw + 3) / 4, h4 = ((int)s.gh + 3) / 4, w16 = (w4 + 3) / 4, h16 = (h4 + 3) / 4, w64 = (w16 + 3) / 4, h64 = (h16 + 3) / 4;
    float dmin = FLT_MAX, dall;
    int bx = 0, by = 0;
#if PR_PYR_LOCAL64
    {   // the 3 x 3 blocks of 64 x 64 pixels around the query's own projection first: a hypothesis within a few centimetres / degrees of the
        // scene pose has its neighbour there; only a query that finds nothing there looks at all blocks
        float u, v;
        grid_project(s, sx, sy, sz, u, v);
        if (u > -1e6f && u < 1e6f && v > -1e6f && v < 1e6f) {
            const int cx = min(max((int)floorf(u) >> 6, 0), w64 - 1), cy = min(max((int)floorf(v) >> 6, 0), h64 - 1);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int x = min(max(cx + k % 3 - 1, 0), w64 - 1), y = min(max(cy + k / 3 - 1, 0), h64 - 1);
                const float4 c = s.pyr64[y * w64 + x];
                const float d2 = (sx - c.x) * (sx - c.x) + (sy - c.y) * (sy - c.y) + (sz - c.z) * (sz - c.z);
                if (d2 < dmin) { dmin = d2; bx = x; by = y; }
            }
        }
    }
    if (!(dmin < 1.0e20f))
#endif
    for (int i = 0; i < w64 * h64; ++i) {                          // wave-uniform addresses: every lane reads the same few cache lines
        const float4 c = s.pyr64[i];
        const float d2 = (sx - c.x) * (sx - c.x) + (sy - c.y) * (sy - c.y) + (sz - c.z) * (sz - c.z);
        if (d2 < dmin) { dmin = d2; bx = i % w64; by = i / w64; }
    }
    dall = dmin;
    float d;
    grid_ring_min(s.pyr16, w16, h16, bx * 4 - 1, by * 4 - 1, 6, sx, sy, sz, d, bx, by);   dall = fminf(dall, d);
    grid_ring_min(s.pyr4, w4, h4, bx * 4 - 1, by * 4 - 1, 6, sx, sy, sz, d, bx, by);       dall = fminf(dall, d);
    grid_ring_min(s.grid, (int)s.gw, (int)s.gh, bx * 4 - 1, by * 4 - 1, 6, sx, sy, sz, d, bx, by);   dall = fminf(dall, d);
    const float b = dall * 1.000001f + 1e-30f;                       // empty cells hold huge coordinates: inf
    if (b < best) best = b;
}
__device__ __forceinline__ bool grid_search(const SceneNNDev &s, float sx, float sy, float sz, float bound, uint32_t &winner, uint32_t *cells = nullptr,
                                            float *best_sq = nullptr, float *other_sq = nullptr, bool settle = false)
{
    int wx, wy;
    // The window has to hold every point closer than sqrt(bound) for the search to be exact.  When a slightly larger window still
    // fits it is taken instead: the extra ring costs a few cells and tells how far the RUNNER-UP is (other_sq), which is what lets
    // the following passes keep this winner without searching (nn_search_kernel).  `settle`: the point has (nearly) stopped moving,
    // so whatever margin is found will last for the rest of the loop -- the radius is then not sqrt(bound) + half a millimetre but the
    // largest the 5 x 5 window covers (du(r) = fx (z + |x|) r / (z (z - r)) <= W  <=>  r <= W z^2 / (fx (z + |x|) + W z)): with the
    // fixed pad a point whose neighbour is more than half a millimetre away (a quarter of them: the depth image is in whole
    // millimetres) never got a margin at all and went through the window in every pass.
    float cover = bound;
    if (other_sq) {
        const float rb = sqrtf(bound);
        float rc = rb + PR_NN_COVER_PAD;
        if (settle) {
            const float W = (float)kGridMaxW - 4e-3f;
            const float rx = W * sz * sz / (s.gfx * (sz + fabsf(sx)) + W * sz), ry = W * sz * sz / (s.gfy * (sz + fabsf(sy)) + W * sz);
            rc = fminf(rx, ry) * 0.999f;
        }
        if (rc > rb && grid_window(s, sx, sy, sz, rc * rc, wx, wy)) cover = rc * rc;
        else if (settle) { rc = rb + PR_NN_COVER_PAD; if (grid_window(s, sx, sy, sz, rc * rc, wx, wy)) cover = rc * rc; }
    }
    if (cover == bound && !grid_window(s, sx, sy, sz, bound, wx, wy)) return false;
    float u, v;
    grid_project(s, sx, sy, sz, u, v);
    if (!(u > -1e6f && u < 1e6f && v > -1e6f && v < 1e6f)) return false;
    const int cx0 = (int)floorf(u), cy0 = (int)floorf(v);
    const int x0 = max(cx0 - wx, 0), x1 = min(cx0 + wx, (int)s.gw - 1);
    const int y0 = max(cy0 - wy, 0), y1 = min(cy0 + wy, (int)s.gh - 1);
    if (x0 > x1 || y0 > y1) return false;
    if (cells) *cells += (uint32_t)((x1 - x0 + 1) * (y1 - y0 + 1));
    float best = bound, second = cover;                           // `second`: smallest squared distance of any scene point other than the winner,
    int best_i = -1, ties = 0;                                    // capped at what the window covers (points outside it are at least that far)
    if (wx <= 1 && wy <= 1) {
        // the common case once aligned (a bound below one pixel): the 3 x 3 cells in ONE round trip instead of one per row
        float4 c[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) c[i] = s.grid[(size_t)min(y0 + i / 3, y1) * s.gw + min(x0 + i % 3, x1)];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (x0 + i % 3 > x1 || y0 + i / 3 > y1) continue;    // (clamped repeats must not count as ties)
            const float d2 = (sx - c[i].x) * (sx - c[i].x) + (sy - c[i].y) * (sy - c[i].y) + (sz - c[i].z) * (sz - c[i].z);
            if (d2 < best) { if (best_i >= 0) second = best; best = d2; best_i = __float_as_int(c[i].w); ties = 0; }
            else if (d2 == best && best_i >= 0) ++ties;
            else if (d2 < second) second = d2;
        }
        if (best_i < 0 || ties != 0) return false;
        winner = (uint32_t)best_i;
        if (best_sq) *best_sq = best;
        if (other_sq) *other_sq = second;
        return true;
    }
    for (int y = y0; y <= y1; ++y) {
        const float4 *row = s.grid + (size_t)y * s.gw;
        float4 c[2 * kGridMaxW + 1];
#pragma unroll
        for (int i = 0; i < 2 * kGridMaxW + 1; ++i) c[i] = row[min(x0 + i, x1)];          // one round trip per row
#pragma unroll
        for (int i = 0; i < 2 * kGridMaxW + 1; ++i) {
            if (x0 + i > x1) continue;                           // (clamped repeats must not count as ties)
            const float d2 = (sx - c[i].x) * (sx - c[i].x) + (sy - c[i].y) * (sy - c[i].y) + (sz - c[i].z) * (sz - c[i].z);
            if (d2 < best) { if (best_i >= 0) second = best; best = d2; best_i = __float_as_int(c[i].w); ties = 0; }
            else if (d2 == best && best_i >= 0) ++ties;
            else if (d2 < second) second = d2;
        }
    }
    if (best_i < 0 || ties != 0) return false;
    winner = (uint32_t)best_i;
    if (best_sq) *best_sq = best;
    if (other_sq) *other_sq = second;
    return true;
}
// a first bound for a query nothing is known about yet: the nearest of the scene points in the 3 x 3 cells around its own pixel
__device__ __forceinline__ void grid_seed_bound(const SceneNNDev &s, float sx, float sy, float sz, float &best)
{
    float u, v;
    grid_project(s, sx, sy, sz, u, v);
    if (!(u > -1e6f && u < 1e6f && v > -1e6f && v < 1e6f)) return;
    const int cx0 = (int)floorf(u), cy0 = (int)floorf(v);
    float4 c[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int x = min(max(cx0 + (i % 3) - 1, 0), (int)s.gw - 1), y = min(max(cy0 + (i / 3) - 1, 0), (int)s.gh - 1);
        c[i] = s.grid[(size_t)y * s.gw + x];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float d2 = (sx - c[i].x) * (sx - c[i].x) + (sy - c[i].y) * (sy - c[i].y) + (sz - c[i].z) * (sz - c[i].z);
        const float b = d2 * 1.000001f + 1e-30f;                  // empty cells hold huge coordinates: b = inf
        if (b < best) best = b;
    }
}

// ================================================================================================
//  29-term contribution (icp.h:138-206) accumulated straight into the lane's registers
// ================================================================================================
// The 29 running sums of a lane are kept as 15 register pairs: v_pk_mul_f32 / v_pk_add_f32 do two IEEE single-precision
// operations per issue slot (separate multiply and add, no contraction), so every sum sees exactly the operations of the
// scalar form (pcd2Ab functor, icp.h:86-136) at roughly half the instruction count.
#ifndef PR_PACKED_ACC
#define PR_PACKED_ACC 1
#endif
#if PR_PACKED_ACC
// 16 register pairs.  The four natural pairs of a correspondence -- C01 = (J0, J1) and C2R = (J2, r) computed here, N01 = (J3, J4) =
// (nx, ny) and N2Z = (J5, z) = (nz, dz) as the scene record delivers them -- are multiplied pair by pair: a packed instruction may
// take either half of each operand for each of its two results (op_sel), so every instruction below forms two of the 27 products
// without a single register move (the previous pairing needed 11 v_mov per point to line its operands up).  Three result halves
// are surplus (a repeated J1*J0, J4*J3 and dz*J5) and accumulate in halves that are never exported.
//   pair k holds the sums kAccLo[k], kAccHi[k] (icp.h:138-206 numbering; -1 = surplus)
struct Acc29 { float2v pk[16]; };
__device__ constexpr int kAccLo[16] = { 0, 6, 2, 7, 3, 8, 5, 11, 12, 24, 14, 15, 18, 17, 20, 27 };
__device__ constexpr int kAccHi[16] = { 1, -1, 21, 22, 4, 9, 10, 23, 13, 25, 26, 16, -1, 19, -1, 28 };
__device__ __forceinline__ void acc_clear(Acc29 &a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a.pk[i] = float2v{ 0.0f, 0.0f };
}
__device__ __forceinline__ void acc_export(const Acc29 &a, float (&out)[29]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { if (kAccLo[i] >= 0) out[kAccLo[i]] = a.pk[i].x; if (kAccHi[i] >= 0) out[kAccHi[i]] = a.pk[i].y; }
}
#define PR_SEL(v, i, j) __builtin_shufflevector(v, v, i, j)
__device__ __forceinline__ void accumulate(Acc29 &acc, float sx, float sy, float sz, const Corr &c)
{
    const float ex = c.dx - sx, ey = c.dy - sy, ez = c.dz - sz;
    const float r = ex * c.nx + ey * c.ny + ez * c.nz;
    const float2v C01{ c.nz * sy - c.ny * sz, c.nx * sz - c.nz * sx };
    const float2v C2R{ c.ny * sx - c.nx * sy, r };
    const float2v N01{ c.nx, c.ny };
    const float2v N2Z{ c.nz, c.dz };
    const float e2 = ex * ex + ey * ey + ez * ez;
    // sums 0..20: J[a]*J[b] for a <= b, row-major; 21..26: J[a]*r; 27: squared distance; 28: count.  IEEE multiplication
    // commutes, so r*J[a] and J[a]*r are the same float.
    acc.pk[0]  += PR_SEL(C01, 0, 0) * C01;                        // J0J0, J0J1
    acc.pk[1]  += PR_SEL(C01, 1, 1) * PR_SEL(C01, 1, 0);          // J1J1, (J1J0)
    acc.pk[2]  += PR_SEL(C01, 0, 0) * C2R;                        // J0J2, J0r
    acc.pk[3]  += PR_SEL(C01, 1, 1) * C2R;                        // J1J2, J1r
    acc.pk[4]  += PR_SEL(C01, 0, 0) * N01;                        // J0J3, J0J4
    acc.pk[5]  += PR_SEL(C01, 1, 1) * N01;                        // J1J3, J1J4
    acc.pk[6]  += C01 * PR_SEL(N2Z, 0, 0);                        // J0J5, J1J5
    acc.pk[7]  += PR_SEL(C2R, 0, 0) * C2R;                        // J2J2, J2r
    acc.pk[8]  += PR_SEL(C2R, 0, 0) * N01;                        // J2J3, J2J4
    acc.pk[9]  += PR_SEL(C2R, 1, 1) * N01;                        // J3r,  J4r
    acc.pk[10] += C2R * PR_SEL(N2Z, 0, 0);                        // J2J5, J5r
    acc.pk[11] += PR_SEL(N01, 0, 0) * N01;                        // J3J3, J3J4
    acc.pk[12] += PR_SEL(N01, 1, 1) * PR_SEL(N01, 1, 0);          // J4J4, (J4J3)
    acc.pk[13] += N01 * PR_SEL(N2Z, 0, 0);                        // J3J5, J4J5
    acc.pk[14] += PR_SEL(N2Z, 0, 0) * N2Z;                        // J5J5, (J5 dz)
    acc.pk[15] += float2v{ e2, 1.0f };
}
#undef PR_SEL
#else
struct Acc29 { float v[29]; };
__device__ __forceinline__ void acc_clear(Acc29 &a) {
#pragma unroll
    for (int i = 0; i < 29; ++i) a.v[i] = 0.0f;
}
__device__ __forceinline__ void acc_export(const Acc29 &a, float (&out)[29]) {
#pragma unroll
    for (int i = 0; i < 29; ++i) out[i] = a.v[i];
}
__device__ __forceinline__ void accumulate(Acc29 &a, float sx, float sy, float sz, const Corr &c)
{
    float (&acc)[29] = a.v;
    const float ex = c.dx - sx, ey = c.dy - sy, ez = c.dz - sz;
    const float r = ex * c.nx + ey * c.ny + ez * c.nz;
    float J[6];
    J[0] = c.nz * sy - c.ny * sz;
    J[1] = c.nx * sz - c.nz * sx;
    J[2] = c.ny * sx - c.nx * sy;
    J[3] = c.nx; J[4] = c.ny; J[5] = c.nz;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) { acc[k] += J[a] * J[b]; ++k; }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;
    acc[27] += ex * ex + ey * ey + ez * ez;
    acc[28] += 1.0f;
}
#endif

// the last pass of an ICP (iteration == max_iteration, icp.cu:189) only feeds fitness and rmse: sums 27 (squared distance)
// and 28 (count), each updated by exactly the operation the full form applies to it
__device__ __forceinline__ void accumulate_score(Acc29 &acc, float sx, float sy, float sz, const Corr &c)
{
    const float ex = c.dx - sx, ey = c.dy - sy, ez = c.dz - sz;
    const float e2 = ex * ex + ey * ey + ez * ez;
#if PR_PACKED_ACC
    acc.pk[15] += float2v{ e2, 1.0f };
#else
    acc.v[27] += e2;
    acc.v[28] += 1.0f;
#endif
}

// ------------------------------------------------------------------------------------------------
// wave64 sum with a fixed balanced pairwise tree in lane order, result in lane 63:
//   row_shr:1,2,4,8 inside each 16-lane row, row_bcast15 into rows 1 and 3, row_bcast31 into rows 2,3.
// ------------------------------------------------------------------------------------------------
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_get(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, kRowMask, 0xf, true));
}

__device__ __forceinline__ uint32_t ld_sys_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float ld_sys_f32(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys_f32(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// workgroup sums added sequentially in workgroup order, starting from 0 (system-scope loads: the partials were written by
// other workgroups of the same launch)
__device__ __forceinline__ float sum_partials_sys(const float *partial, uint32_t pose, uint32_t nblk, uint32_t used, uint32_t comp)
{
    float total = 0.0f;
    const float *p = partial + (size_t)pose * nblk * kAccStride + comp;
    for (uint32_t g0 = 0; g0 < used; g0 += 16) {
        float v[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) v[k] = (g0 + k < used) ? ld_sys_f32(p + (size_t)(g0 + k) * kAccStride) : 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) if (g0 + k < used) total += v[k];
    }
    return total;
}

__device__ __forceinline__ bool pose_iteration(const float *Ab, uint32_t n, DevIcpState &s, const pr_criteria &crit, uint32_t iter, float (&E)[16]);

// ---- wave-cooperative form of the same iteration logic ---------------------------------------------------------------
// One lane running prs::solve_666_impl is slow twice over: ~1500 dependent double-precision instructions, and the
// data-dependent pivoting makes the compiler specialise the code per pivot sequence (hundreds of KB of instructions,
// fetched cold).  Here the 6x6 lives one element per lane (lane = 6*row + col, the FULL symmetric matrix, so the
// symmetric pivot swap is a single lane permutation), the column update of a step runs on the lanes of that column,
// the six back-substitution divisions and the three sin/cos evaluations run side by side, and everything else is
// computed redundantly (uniformly) by all lanes.  Every element goes through exactly the same sequence of IEEE operations
// as in prs::ldlt6 / prs::solve_666_impl, so the update is bit-identical to the host solver
// (tests/test_parity_gpu.py::test_host_and_device_solve_agree).  All 64 lanes of the wavefront must be active.
__device__ __forceinline__ double wave_gather_d(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int L> __device__ __forceinline__ double wave_bcast_d(double v)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), L);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), L);
    return __hiloint2double(hi, lo);
}
template <int L> __device__ __forceinline__ float wave_bcast_f(float v)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L));
}

// prs::perm_apply as selects (no branches: a branch per pivot value makes the compiler clone everything downstream)
template <int K> __device__ __forceinline__ void wave_perm_apply(double (&y)[6], int p)
{
#pragma unroll
    for (int P = K + 1; P < 6; ++P) {
        const bool sel = (p == P);
        const double a = y[K], b = y[P];
        y[K] = sel ? b : a;
        y[P] = sel ? a : b;
    }
}

// step K of prs::ldlt6_step on the lane-distributed matrix; returns the pivot row (uniform)
template <int K> __device__ __forceinline__ int wave_ldlt6_step(double &m, int lane, int r, int c)
{
    int p = K;
    double top = prs::dabs(wave_bcast_d<7 * K>(m));
    if constexpr (K + 1 < 6) { const double a = prs::dabs(wave_bcast_d<7 * (K + 1 < 6 ? K + 1 : 5)>(m)); const bool g = a > top; top = g ? a : top; p = g ? K + 1 : p; }
    if constexpr (K + 2 < 6) { const double a = prs::dabs(wave_bcast_d<7 * (K + 2 < 6 ? K + 2 : 5)>(m)); const bool g = a > top; top = g ? a : top; p = g ? K + 2 : p; }
    if constexpr (K + 3 < 6) { const double a = prs::dabs(wave_bcast_d<7 * (K + 3 < 6 ? K + 3 : 5)>(m)); const bool g = a > top; top = g ? a : top; p = g ? K + 3 : p; }
    if constexpr (K + 4 < 6) { const double a = prs::dabs(wave_bcast_d<7 * (K + 4 < 6 ? K + 4 : 5)>(m)); const bool g = a > top; top = g ? a : top; p = g ? K + 4 : p; }
    if constexpr (K + 5 < 6) { const double a = prs::dabs(wave_bcast_d<7 * (K + 5 < 6 ? K + 5 : 5)>(m)); const bool g = a > top; top = g ? a : top; p = g ? K + 5 : p; }
    // symmetric swap K <-> p: rows and columns of the full matrix (prs::sym_swap touches the same lower-triangle entries)
    const int rr = (r == K) ? p : ((r == p) ? K : r);
    const int cc = (c == K) ? p : ((c == p) ? K : c);
    m = wave_gather_d(m, rr * 6 + cc);
    double w[K > 0 ? K : 1];
    double dot = 0.0, acc = 0.0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const double lkj = (j == 0) ? wave_bcast_d<6 * K + 0>(m) : (j == 1) ? wave_bcast_d<6 * K + 1>(m) : (j == 2) ? wave_bcast_d<6 * K + 2>(m)
                         : (j == 3) ? wave_bcast_d<6 * K + 3>(m) : wave_bcast_d<6 * K + 4>(m);
        const double dj  = (j == 0) ? wave_bcast_d<0>(m) : (j == 1) ? wave_bcast_d<7>(m) : (j == 2) ? wave_bcast_d<14>(m)
                         : (j == 3) ? wave_bcast_d<21>(m) : wave_bcast_d<28>(m);
        w[j] = dj * lkj;                                       // w[j] = MM(j,j) * MM(K,j)
        dot += lkj * w[j];                                     // dot += MM(K,j) * w[j]
        acc += wave_gather_d(m, lane - (K - j)) * w[j];        // lane (i,K): acc += MM(i,j) * w[j]
    }
    const double dk = wave_bcast_d<7 * K>(m) - dot;            // MM(K,K) -= dot
    const double v = m - acc;
    const double q = (prs::dabs(dk) > 0.0) ? v / dk : v;
    if (c == K && r > K) m = q;
    if (lane == 7 * K) m = dk;
    return p;
}

// `total`: lanes 0..28 hold the 29 reduced sums (component = lane).  Uniform result; `s` and `E` are uniform copies.
__device__ __forceinline__ bool pose_iteration_wave(float total, uint32_t n, DevIcpState &s, const pr_criteria &crit, uint32_t iter, float (&E)[16])
{
    const int lane = (int)(threadIdx.x & 63u);
    s.passes += 1;
    const float cnt = wave_bcast_f<28>(total), err = wave_bcast_f<27>(total);
    if (cnt == 0) return true;                                               // icp.cu:183
    const float prev_fit = s.fitness, prev_rmse = s.rmse;
    s.fitness = cnt / (float)n;                                              // icp.cu:185
    s.rmse = sqrtf(err / cnt);                                               // icp.cu:186
    if (iter == (uint32_t)crit.max_iteration) return true;                   // icp.cu:189
    const float df = s.fitness - prev_fit, dr = s.rmse - prev_rmse;
    if (((df < 0) ? -df : df) < crit.relative_fitness && ((dr < 0) ? -dr : dr) < crit.relative_rmse) return true;   // icp.cu:191-194

    // m(r,c) = (double)A(c,r) + 0.01 [r==c]; A is filled symmetrically from the 21 upper-triangle sums (row-major, k running)
    const int l36 = lane < 36 ? lane : 35;
    const int r = l36 / 6, c = l36 - 6 * r;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int k = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
    const float a = __int_as_float(__builtin_amdgcn_ds_bpermute(k << 2, __float_as_int(total)));
    double m = (double)a + (r == c ? 0.01 : 0.0);
    double y[6] = { (double)wave_bcast_f<21>(total), (double)wave_bcast_f<22>(total), (double)wave_bcast_f<23>(total),
                    (double)wave_bcast_f<24>(total), (double)wave_bcast_f<25>(total), (double)wave_bcast_f<26>(total) };

    const int s0 = wave_ldlt6_step<0>(m, lane, r, c), s1 = wave_ldlt6_step<1>(m, lane, r, c), s2 = wave_ldlt6_step<2>(m, lane, r, c);
    const int s3 = wave_ldlt6_step<3>(m, lane, r, c), s4 = wave_ldlt6_step<4>(m, lane, r, c);
    (void)wave_ldlt6_step<5>(m, lane, r, c);

    wave_perm_apply<0>(y, s0); wave_perm_apply<1>(y, s1); wave_perm_apply<2>(y, s2); wave_perm_apply<3>(y, s3); wave_perm_apply<4>(y, s4);
    // forward substitution with unit-lower L: y[i] -= MM(i,j) * y[j]
    y[1] -= wave_bcast_d<6>(m) * y[0];
    y[2] -= wave_bcast_d<12>(m) * y[0]; y[2] -= wave_bcast_d<13>(m) * y[1];
    y[3] -= wave_bcast_d<18>(m) * y[0]; y[3] -= wave_bcast_d<19>(m) * y[1]; y[3] -= wave_bcast_d<20>(m) * y[2];
    y[4] -= wave_bcast_d<24>(m) * y[0]; y[4] -= wave_bcast_d<25>(m) * y[1]; y[4] -= wave_bcast_d<26>(m) * y[2]; y[4] -= wave_bcast_d<27>(m) * y[3];
    y[5] -= wave_bcast_d<30>(m) * y[0]; y[5] -= wave_bcast_d<31>(m) * y[1]; y[5] -= wave_bcast_d<32>(m) * y[2]; y[5] -= wave_bcast_d<33>(m) * y[3];
    y[5] -= wave_bcast_d<34>(m) * y[4];
    // pseudo-inverse of D: the six divisions side by side (lane i divides y[i] by MM(i,i))
    {
        const int i6 = lane < 6 ? lane : 5;
        const double yl = (i6 == 0) ? y[0] : (i6 == 1) ? y[1] : (i6 == 2) ? y[2] : (i6 == 3) ? y[3] : (i6 == 4) ? y[4] : y[5];
        const double dl = wave_gather_d(m, 7 * i6);
        const double tiny = 1.0 / 1.7976931348623157e308;
        const double ql = (prs::dabs(dl) > tiny) ? yl / dl : 0.0;
        y[0] = wave_bcast_d<0>(ql); y[1] = wave_bcast_d<1>(ql); y[2] = wave_bcast_d<2>(ql);
        y[3] = wave_bcast_d<3>(ql); y[4] = wave_bcast_d<4>(ql); y[5] = wave_bcast_d<5>(ql);
    }
    // back substitution with L^T: y[i] -= MM(j,i) * y[j], j ascending
    y[4] -= wave_bcast_d<34>(m) * y[5];
    y[3] -= wave_bcast_d<27>(m) * y[4]; y[3] -= wave_bcast_d<33>(m) * y[5];
    y[2] -= wave_bcast_d<20>(m) * y[3]; y[2] -= wave_bcast_d<26>(m) * y[4]; y[2] -= wave_bcast_d<32>(m) * y[5];
    y[1] -= wave_bcast_d<13>(m) * y[2]; y[1] -= wave_bcast_d<19>(m) * y[3]; y[1] -= wave_bcast_d<25>(m) * y[4]; y[1] -= wave_bcast_d<31>(m) * y[5];
    y[0] -= wave_bcast_d<6>(m) * y[1];  y[0] -= wave_bcast_d<12>(m) * y[2]; y[0] -= wave_bcast_d<18>(m) * y[3]; y[0] -= wave_bcast_d<24>(m) * y[4];
    y[0] -= wave_bcast_d<30>(m) * y[5];
    wave_perm_apply<4>(y, s4); wave_perm_apply<3>(y, s3); wave_perm_apply<2>(y, s2); wave_perm_apply<1>(y, s1); wave_perm_apply<0>(y, s0);

    // the three half-angle sin/cos pairs side by side (lane 0: x, 1: y, 2: z)
    const int i3 = lane < 3 ? lane : 2;
    const double ang = 0.5 * ((i3 == 0) ? y[0] : (i3 == 1) ? y[1] : y[2]);
    double sl, cl;
    prs::sincos_d(ang, &sl, &cl);
    prs::compose_update(y, wave_bcast_d<0>(sl), wave_bcast_d<0>(cl), wave_bcast_d<1>(sl), wave_bcast_d<1>(cl),
                        wave_bcast_d<2>(sl), wave_bcast_d<2>(cl), E);
    prs::mat4_mul_impl(E, s.T, s.T);                                         // icp.cu:212
    return false;
}

// ================================================================================================
//  THE hot kernel: pending transform + correspondence + 29-term transform-reduce, all hypotheses
//  of a batch in one launch.  grid = (workgroups per hypothesis, hypotheses), 256 lanes.
// ================================================================================================
#ifndef PR_PASS_WAVES
#define PR_PASS_WAVES 1
#endif

// One virtual workgroup of the canonical tree: accumulate the 29 sums of points [first, first + steps*1024) of one
// cloud into the lane's registers (pending transform applied and written back first when xf).
template <class Scene, bool kNN, int kStack, bool kScoreOnly = false>
__device__ __forceinline__ void vb_accumulate(float (&acc_out)[29], float *cl, uint32_t n, uint32_t first, uint32_t steps, bool xf,
                                              const float (&M)[12], const Scene &scene, const int4 *lds_topo, int *stk_node, float *stk_lb,
                                              uint32_t *nn_prev = nullptr, bool nn_seeded = false)
{
    (void)nn_prev; (void)nn_seeded;
    struct { uint32_t steps; } b{ steps };
    Acc29 acc;                                                   // the caller's sums start at zero
    acc_clear(acc);
    uint32_t lane_winner = kNoPrev;                              // kd-tree scenes: the last neighbour this lane found (spatial seed)
    (void)lane_winner;
    (void)lds_topo; (void)stk_node; (void)stk_lb;
    // One 1024-point step: lane t takes points first + s*1024 + i*256 + t, i = 0..3 -- ADJACENT LANES HOLD ADJACENT CLOUD POINTS,
    // i.e. neighbouring pixels of the rendered hypothesis.  The cloud moves as 12-byte loads / stores that are contiguous across
    // the wavefront (768 B per instruction), and -- what matters -- the 64 scene gathers of an instruction land on neighbouring
    // scene pixels: 8-10 cache lines instead of the 32+ of a stride-4 mapping.  The pass is bound by the texture addresser / L1
    // rate of exactly those gathers (TA busy 64 %, VALU 54 %: profiles/r02/sq_proj_*.md), not by issue slots or HBM.
    auto point_of = [&](uint32_t j0, uint32_t i) { return j0 + i * kBlockThreads; };
    auto load_step = [&](uint32_t s, float (&p)[12], uint32_t &j0, uint32_t &cnt) {
        j0 = first + s * kPointsPerStep + threadIdx.x;
        cnt = (j0 >= n) ? 0u : (((n - j0 + kBlockThreads - 1u) / kBlockThreads < kPointsPerLane) ? (n - j0 + kBlockThreads - 1u) / kBlockThreads : kPointsPerLane);
        // unconditional loads (a lane past the end re-reads the cloud's last point and never uses it: `i < cnt` gates every use)
        const uint32_t last = n - 1u;                                // n >= 1 here: the workgroup has points
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            const uint32_t j = point_of(j0, i);
            const pr_vec3 v = ld_off<pr_vec3>(cl, (j < last ? j : last) * 12u);
            p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z;
        }
    };
    auto process_step = [&](float (&p)[12], uint32_t j0, uint32_t cnt) {
        if (cnt == 0) return;
        if (xf) {                                                // icp.cu:142-153 transform_pcd_cuda, fused
            // rows 0 and 1 of the update side by side in packed instructions (same per-element operations, same order:
            // ((m0*x + m1*y) + m2*z) + m3), row 2 scalar
            const float2v Mx{ M[0], M[4] }, My{ M[1], M[5] }, Mz{ M[2], M[6] }, Mt{ M[3], M[7] };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
                float2v t = Mx * float2v{ x, x };
                t = t + My * float2v{ y, y };
                t = t + Mz * float2v{ z, z };
                t = t + Mt;
                p[3 * i]     = t.x;
                p[3 * i + 1] = t.y;
                p[3 * i + 2] = M[8] * x + M[9] * y + M[10] * z + M[11];
            }
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                if (i < cnt) st_off<pr_vec3>(cl, point_of(j0, i) * 12u, pr_vec3{ p[3 * i], p[3 * i + 1], p[3 * i + 2] });
        }
        if constexpr (kNN && kStack == -1) {
            // winners of the search kernel: four indices, then the four (point, normal) pairs, all in flight before the first use
            uint32_t w[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) w[i] = (i < cnt) ? scene.winner[point_of(j0, i)] : kNoPrev;
            float4 d[4]; float nx[4], ny[4], nz[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t at = (w[i] != kNoPrev) ? w[i] : 0u;
                d[i] = scene.pts[at];
                const float *n = reinterpret_cast<const float *>(scene.normal + at);
                nx[i] = n[0]; ny[i] = n[1]; nz[i] = n[2];
            }
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                if (w[i] != kNoPrev) {
                    Corr c; c.dx = d[i].x; c.dy = d[i].y; c.dz = d[i].z; c.nx = nx[i]; c.ny = ny[i]; c.nz = nz[i];
                    if constexpr (kScoreOnly) accumulate_score(acc, p[3 * i], p[3 * i + 1], p[3 * i + 2], c); else accumulate(acc, p[3 * i], p[3 * i + 1], p[3 * i + 2], c);
                }
            }
        } else if constexpr (kNN) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                if (i < cnt) {
                    Corr c;
                    bool ok;
                    if constexpr (kStack > 0) {
                        const uint32_t seed = (nn_prev && nn_seeded) ? nn_prev[point_of(j0, i)] : kNoPrev;
                        uint32_t winner;
                        ok = query_nn_stack<kStack>(scene, reinterpret_cast<const float4 *>(lds_topo), stk_node, stk_lb, p[3 * i], p[3 * i + 1], p[3 * i + 2], c,
                                                    seed, nn_prev ? lane_winner : kNoPrev, winner);
                        if (nn_prev) nn_prev[point_of(j0, i)] = ok ? winner : kNoPrev;
                        if (ok) lane_winner = winner;
                    } else ok = query_nn<true>(scene, lds_topo, p[3 * i], p[3 * i + 1], p[3 * i + 2], c);
                    if (ok) { if constexpr (kScoreOnly) accumulate_score(acc, p[3 * i], p[3 * i + 1], p[3 * i + 2], c); else accumulate(acc, p[3 * i], p[3 * i + 1], p[3 * i + 2], c); }
                }
            }
        } else {
            // projective: all four gathers of the lane are issued back to back, unconditionally
            // (pixel 0 stands in for out-of-image points), and only then tested -- one memory round
            // trip per step instead of eight dependent ones.
#ifndef PR_GATHER_BATCH
#define PR_GATHER_BATCH 4
#endif
#pragma unroll
            for (uint32_t i0 = 0; i0 < 4; i0 += PR_GATHER_BATCH) {
                Gathered gth[PR_GATHER_BATCH];
                bool in_img[PR_GATHER_BATCH];
#pragma unroll
                for (uint32_t k = 0; k < PR_GATHER_BATCH; ++k) {
                    const uint32_t i = i0 + k;
                    in_img[k] = gather_issue(scene, p[3 * i], p[3 * i + 1], p[3 * i + 2], i < cnt, gth[k]);
                }
#pragma unroll
                for (uint32_t k = 0; k < PR_GATHER_BATCH; ++k) {
                    const uint32_t i = i0 + k;
                    Corr c;
                    if (gather_finish(scene, in_img[k], p[3 * i + 2], gth[k], c)) {
                        if constexpr (kScoreOnly) accumulate_score(acc, p[3 * i], p[3 * i + 1], p[3 * i + 2], c);
                        else accumulate(acc, p[3 * i], p[3 * i + 1], p[3 * i + 2], c);
                    }
                }
            }
        }
    };

    for (uint32_t s = 0; s < b.steps; ++s) {
        float p[12];
        uint32_t j0, cnt;
        load_step(s, p, j0, cnt);
        if (cnt == 0) break;
        process_step(p, j0, cnt);
    }
    acc_export(acc, acc_out);
}

// canonical tree, second half: wave (balanced pairwise over lanes) -> ((w0+w1)+w2)+w3.  Returns, in threads 0..28, the
// workgroup sum of component threadIdx.x.  Contains one __syncthreads().
#ifndef PR_TREE_PACKED
#define PR_TREE_PACKED 1
#endif
template <bool kScoreOnly = false>
__device__ __forceinline__ float vb_reduce(float (&acc)[29], float (*wsum)[kAccStride])
{
    constexpr int kFirst = kScoreOnly ? 27 : 0;                  // score-only passes carry zeros in sums 0..26
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if PR_TREE_PACKED
    if constexpr (!kScoreOnly) {
        // Same balanced pairwise tree, but after every level the live partial sums of two registers are interleaved into one
        // (a level's results sit in the upper half of each 2^k-lane group; the lower half is free to carry another sum's
        // results, fetched from its upper half with a row_shl).  29 -> 15 -> 8 -> 4 -> 2 registers, so levels 2..4 cost
        // 15 + 8 + 4 adds instead of 3 x 29; the two cross-row levels run on the 2 remaining registers with ds_bpermute.
        // Every sum still sees acc(lane) + acc(lane - d) at each level, i.e. bit-identical results.
        // Final layout: sum j = 16 h + t ends in register h at lane 63 - t.
        const bool b1 = (lane & 1u) != 0, b2 = (lane & 2u) != 0;
        float p1[16], p2[8], p3[4], p4[2];
#pragma unroll
        for (int i = 0; i < 29; ++i) acc[i] += dpp_get<0x111, 0xf>(acc[i]);                   // row_shr:1, results in odd lanes
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            const float hi = acc[2 * k], lo = (2 * k + 1 < 29) ? dpp_get<0x101, 0xf>(acc[2 * k + 1 < 29 ? 2 * k + 1 : 28]) : 0.0f;   // row_shl:1
            p1[k] = b1 ? hi : lo;
        }
        p1[15] = 0.0f;
#pragma unroll
        for (int k = 0; k < 15; ++k) p1[k] += dpp_get<0x112, 0xf>(p1[k]);                     // row_shr:2
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float lo = dpp_get<0x102, 0xf>(p1[2 * m + 1]);                               // row_shl:2
            p2[m] = b2 ? p1[2 * m] : lo;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) p2[m] += dpp_get<0x114, 0xf>(p2[m]);                      // row_shr:4
#pragma unroll
        for (int q = 0; q < 4; ++q)                                                            // lanes 0-3, 8-11 <- row_shl:4 of the odd register
            p3[q] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(p2[2 * q]), __float_as_int(p2[2 * q + 1]), 0x104, 0xf, 0x5, false));
#pragma unroll
        for (int q = 0; q < 4; ++q) p3[q] += dpp_get<0x118, 0xf>(p3[q]);                      // row_shr:8
#pragma unroll
        for (int h = 0; h < 2; ++h)                                                            // lanes 0-7 <- row_shl:8 of the odd register
            p4[h] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(p3[2 * h]), __float_as_int(p3[2 * h + 1]), 0x108, 0xf, 0x3, false));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            p4[h] += __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane - 16u) & 63u) << 2, __float_as_int(p4[h])));   // rows 1,3 += rows 0,2
            p4[h] += __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane - 32u) & 63u) << 2, __float_as_int(p4[h])));   // row 3 += row 1
        }
        if (lane >= 48) {
            const uint32_t t = 63u - lane;
            wsum[wave][t] = p4[0];
            if (16u + t < 29u) wsum[wave][16u + t] = p4[1];
        }
    } else
#endif
    {
    // level-major: the 29 sums advance through each tree level together, so consecutive DPP instructions are independent
    // (value-major order makes every instruction depend on the previous one and the compiler pads it with s_nop)
#define PR_TREE_LEVEL(CTRL, MASK) _Pragma("unroll") for (int i = kFirst; i < 29; ++i) acc[i] += dpp_get<CTRL, MASK>(acc[i]);
    PR_TREE_LEVEL(0x111, 0xf)      // row_shr:1
    PR_TREE_LEVEL(0x112, 0xf)      // row_shr:2
    PR_TREE_LEVEL(0x114, 0xf)      // row_shr:4
    PR_TREE_LEVEL(0x118, 0xf)      // row_shr:8
    PR_TREE_LEVEL(0x142, 0xa)      // row_bcast15 -> rows 1,3
    PR_TREE_LEVEL(0x143, 0xc)      // row_bcast31 -> rows 2,3
#undef PR_TREE_LEVEL
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < 29; ++i) wsum[wave][i] = acc[i];
    }
    }
    __syncthreads();
    float t = 0.0f;
    if (threadIdx.x < 29) t = ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
    return t;
}

// ================================================================================================
//  THE hot kernel: pending transform + correspondence + 29-term transform-reduce, all hypotheses
//  of a batch in one launch.  grid = (workgroups per hypothesis, hypotheses), 256 lanes.
// ================================================================================================
template <class Scene, bool kNN, int kStack = 0>
__global__ __launch_bounds__(256, PR_PASS_WAVES) void icp_pass_kernel(IcpBatch b, Scene scene)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float wsum[4][kAccStride];

    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];                           // uniform address: one 64-byte scalar load
    const int32_t st = pm.state;
    if (st == kSkip) return;
    const uint32_t n = pm.count;
    const uint32_t ppb = b.steps * kPointsPerStep;
    if ((uint64_t)blockIdx.x * ppb >= n) return;

    const int4 *lds_topo = nullptr;
    int *stk_node = nullptr; float *stk_lb = nullptr;
    if constexpr (kNN && kStack == 0) {
        int4 *dst = reinterpret_cast<int4 *>(lds_raw);
        for (uint32_t i = threadIdx.x; i < scene.lds_nodes; i += kBlockThreads) dst[i] = scene.topo[i];
        __syncthreads();
        lds_topo = dst;
    }
    if constexpr (kNN && kStack > 0) {                           // per-lane stacks: [entry][lane], then the staged records
        stk_node = reinterpret_cast<int *>(lds_raw) + threadIdx.x;
        stk_lb = reinterpret_cast<float *>(lds_raw) + (size_t)(kStack & 0xff) * kBlockThreads + threadIdx.x;
        float4 *recs = reinterpret_cast<float4 *>(lds_raw + (size_t)(kStack & 0xff) * kBlockThreads * 8);
        if constexpr ((kStack & 0x100) == 0) for (uint32_t i = threadIdx.x; i < scene.lds_nodes * 4; i += kBlockThreads) recs[i] = scene.rec[i];
        __syncthreads();
        lds_topo = reinterpret_cast<const int4 *>(recs);
    }
    if constexpr (kNN && kStack == -1) scene.winner += pm.start;  // winners are indexed like the cloud

    float *cl = reinterpret_cast<float *>(b.cloud + pm.start);
    // a pending update that nn_search_kernel has already applied to the cloud must not be applied again
    const bool xf = (st == kRunWithTransform) && !b.pre_transformed;   // also: not the first pass of this cloud (a seed exists)
    uint32_t *nn_prev = (kNN && kStack >= 0 && b.nn_prev) ? b.nn_prev + pm.start : nullptr;
    float M[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) M[i] = xf ? pm.xform[i] : 0.0f;

    // One virtual workgroup of the canonical tree per trip.  The grid normally holds one workgroup per 2048-point block; the
    // asynchronous fused path sizes its grid from the previous batch's cloud sizes (the current ones are still on the device),
    // so a workgroup may have to take more than one block -- partial sums are per virtual block either way.
    const uint32_t used = (n + ppb - 1) / ppb;
    for (uint32_t vb = blockIdx.x; vb < used; vb += gridDim.x) {
        if (vb != blockIdx.x) __syncthreads();                   // wsum of the previous trip has been read
        float acc[29];
#pragma unroll
        for (int i = 0; i < 29; ++i) acc[i] = 0.0f;
        float t;
        if (b.score_only) {                                      // uniform: the final pass needs sums 27 and 28 only
            vb_accumulate<Scene, kNN, kStack, true>(acc, cl, n, vb * ppb, b.steps, xf, M, scene, lds_topo, stk_node, stk_lb, nn_prev, xf);
            t = vb_reduce<true>(acc, wsum);
        } else {
            vb_accumulate<Scene, kNN, kStack>(acc, cl, n, vb * ppb, b.steps, xf, M, scene, lds_topo, stk_node, stk_lb, nn_prev, xf);
            t = vb_reduce(acc, wsum);
        }
        float *slot = b.partial + ((size_t)pose * b.nblk + vb) * kAccStride;
        if (!b.fused) {
            if (threadIdx.x < 29) slot[threadIdx.x] = t;
            continue;
        }
        // Fused finalize + solve: partial sums cross workgroups (and XCDs) through memory with system-scope accesses on both
        // sides; wave 0 drains its stores before the arrival atomic that publishes them.  The workgroup that delivers the last
        // partial sum of the pose (it cannot have another block left) adds the partials in block order (same sequence as
        // icp_finalize_solve_kernel) and runs the iteration logic.  PoseMeta / DevIcpState are only read again by the next
        // launch, so plain accesses suffice for them.
        if (threadIdx.x >= 64) continue;
        if (threadIdx.x < 29) st_sys_f32(slot + threadIdx.x, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t ticket = 0;
#if PR_ARRIVE_ACQ_REL
        // the memory-model form: release / acquire on the arrival counter itself (buffer_wbl2 sc1 + buffer_inv sc1 around the atomic).
        // Measured against the default below: the default is the guide's "sc1 payload -> vmcnt(0) -> flag, sc1
        // loads on the consumer" hand-off, which needs neither cache operation because the payload never rests in L1 / L2.
        if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(&b.arrive[pose], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
        if (threadIdx.x == 0) ticket = atomicAdd(&b.arrive[pose], 1u);
#endif
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket + 1u != used) continue;
        if (threadIdx.x == 0) st_sys_u32(&b.arrive[pose], 0u);
        if (b.fused == 2u) {                                     // solve on the host: the sums of the hypothesis, straight into host memory
            if (threadIdx.x < 29) st_sys_f32(b.sums_out + (size_t)pose * kAccStride + threadIdx.x, sum_partials_sys(b.partial, pose, b.nblk, used, threadIdx.x));
            return;
        }
        DevIcpState s = b.st[pose];                              // uniform; in flight together with the partial sums
        float total = 0.0f;
        if (threadIdx.x < 29) total = sum_partials_sys(b.partial, pose, b.nblk, used, threadIdx.x);
        float E[16];
        const bool finished = pose_iteration_wave(total, n, s, b.crit, b.iter, E);
        if (threadIdx.x != 0) return;
        PoseMeta *wm = const_cast<PoseMeta *>(b.meta) + pose;
        if (finished) { s.done = 1; wm->state = kSkip; }
        else {
#pragma unroll
            for (int i = 0; i < 12; ++i) wm->xform[i] = E[i];
            wm->state = kRunWithTransform;
        }
        b.st[pose] = s;
        return;
    }
}

// ================================================================================================
//  kd-tree scenes, split form: the SEARCH on its own.
//
//  The fused pass ties the search to the canonical reduction tree: lane t owns points 4t..4t+3 of a 1024-point step, so only
//  three of four queries can start from their neighbour's answer, and the unseeded ones -- a hundred node visits each while the
//  hypothesis is still centimetres off the surface -- set the pace of the first passes.  The winner of a query does not depend
//  on how the search is organised, so the search runs here in whatever order suits it and leaves the winners in `nn_prev`;
//  the pass that follows (icp_pass_kernel<SceneNNWinners>) gathers them in canonical order -- bit-identical sums.
//    * a lane walks a RUN of consecutive cloud points (image neighbours): every query but the first of a run starts from the
//      previous one's winner, whose distance is within microns of the answer (the step between neighbours is lateral);
//    * the previous pass' winner of the same point bounds the search as before;
//    * the first query of a run takes a bound from the scene points around its own pixel (grid_seed_bound);
//    * with a bound that tight the candidates are the handful of scene pixels around the query's pixel: grid_search settles the
//      query without touching the tree unless there is a tie (the tree's visiting order then decides, as in the reference).
//  The pending rigid update is applied (and written back) here, so the pass reads the cloud as it is.
// ================================================================================================
__global__ __launch_bounds__(256) void nn_search_kernel(IcpBatch b, SceneNNDev scene, uint32_t run)
{
    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];
    const int32_t st = pm.state;
    if (st == kSkip) return;
    const uint32_t n = pm.count;
    const uint32_t first = blockIdx.x * kBlockThreads * run;
    if (first >= n) return;

    float *cl = reinterpret_cast<float *>(b.cloud + pm.start);
    uint32_t *win = b.nn_prev + pm.start;
    const bool xf = (st == kRunWithTransform);                   // also: the previous pass left winners for this cloud
    float M[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) M[i] = xf ? pm.xform[i] : 0.0f;
    const float accept = scene.max_dist_diff * scene.max_dist_diff;

    // Two phases.  (1) Every lane takes one point per chunk (lane t of chunk k: point first + 256 k + t -- adjacent lanes hold
    // adjacent points), applies the pending update and tries the cheap way: the bound from the previous pass' winner and, if that
    // bound is a pixel or two wide, the exact window scan.  (2) What is left -- a few per cent of the points once aligned, most of
    // them on the first passes -- is packed into dense lanes (in point order, so neighbours stay neighbours) and only then pays for
    // the descent through the representative points and the tree search.  Without the packing four of five wavefronts ran a whole
    // tree search for two or three of their lanes.
    // the hypothesis' queue lives next to its winners (same indexing, so it can never overflow): (point, bound) per entry, filled
    // through this pass' counter of the hypothesis (two counters alternate between passes; nn_tree_kernel re-arms the idle one)
    uint2 *queue = b.nn_queue + pm.start;
    uint32_t *q_count = b.nn_qcount + kQCountStride * pose + (b.iter & 1u);
    __shared__ uint32_t wg_count, wg_base;
    __shared__ uint32_t wave_base[4];
    if (threadIdx.x == 0) wg_count = 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t j0 = first + threadIdx.x;
    NNCount cnt;
    uint32_t n_query = 0, n_window = 0, n_cells = 0, n_kept = 0;
    float *slk = b.nn_slack + pm.start;
    for (uint32_t k = 0; k < run; ++k) {
        const uint32_t j = j0 + k * kBlockThreads;
        const bool live = j < n;
        bool pending = false;
        float best = accept;
        if (live) {
            pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);
            float x = q.x, y = q.y, z = q.z;
            bool still = false;                                  // has this point (nearly) kept the position its winner is from?
            float step_sq = 0.0f;
            if (xf) {                                            // icp.cu:142-153 transform_pcd_cuda
                const float tx = M[0] * x + M[1] * y + M[2]  * z + M[3];
                const float ty = M[4] * x + M[5] * y + M[6]  * z + M[7];
                const float tz = M[8] * x + M[9] * y + M[10] * z + M[11];
                step_sq = (tx - x) * (tx - x) + (ty - y) * (ty - y) + (tz - z) * (tz - z);
                still = step_sq <= PR_NN_STILL;
                x = tx; y = ty; z = tz;
                st_off<pr_vec3>(cl, j * 12u, pr_vec3{ x, y, z });
            }
            ++n_query;
            const uint32_t prev = xf ? win[j] : kNoPrev;
            bool kept = false;
            if (prev != kNoPrev) {
                // KEEP THE WINNER WITHOUT SEARCHING.  `slack` is a lower bound on the distance from this point to every scene point
                // other than its winner, already reduced by every step the point has taken since the bound was established (a step
                // of length s changes any distance by at most s).  If the winner's distance -- computed here with the search's own
                // expression -- is below that bound by more than the float error of a squared distance (3e-7 relative; 1e-5 is
                // demanded), the winner is still the unique strict minimum, which is what the reference's search returns under any
                // visiting order.
                const pr_vec3 pw = ld_off<pr_vec3>(scene.pcd, prev * 12u);
                const float d2 = (x - pw.x) * (x - pw.x) + (y - pw.y) * (y - pw.y) + (z - pw.z) * (z - pw.z);
                const float slack = slk[j] - sqrtf(step_sq) * 1.000001f;
                if (d2 < accept && sqrtf(d2) * 1.00001f < slack) { kept = true; slk[j] = slack; ++n_kept; }
                else { const float bnd = d2 * 1.000001f + 1e-30f; if (bnd < best) best = bnd; }              // = nn_seed_bound
            }
            if (!kept) {
                uint32_t w = kNoPrev;
                float bsq = 0.0f, osq = 0.0f;
                const bool settled = scene.grid && best < accept && grid_search(scene, x, y, z, best, w, &n_cells, &bsq, &osq, still && PR_NN_SETTLE);
                if (settled) { ++n_window; win[j] = w; slk[j] = sqrtf(osq) * 0.99999f; }
                else { pending = true; if (prev != kNoPrev && step_sq <= PR_NN_NODESCENT) best = -best; }   // the sign carries "no descent needed" to the next kernel (best > 0 always)
            }
        }
        // one slot range per workgroup and chunk: waves in order, lanes in order -- the queue keeps the points' order
        const unsigned long long m = __ballot(pending);
        if (lane == 0) wave_base[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t total = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
            wg_base = total ? atomicAdd(q_count, total) : 0u;
        }
        __syncthreads();
        if (pending) {
            uint32_t slot = wg_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            for (uint32_t w2 = 0; w2 < wave; ++w2) slot += wave_base[w2];
            queue[slot] = make_uint2(j, __float_as_uint(best));
        }
        __syncthreads();
    }
    if (scene.counters) {                                        // instrumented runs only (option "nn_count")
        const uint32_t v[8] = { n_query, n_window, 0u, 0u, 0u, 0u, 0u, n_cells };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if ((threadIdx.x & 63u) == 0u && t) atomicAdd(&scene.counters[(size_t)b.iter * 8 + i], (unsigned long long)t);
        }
    }
}

// Second half of the search: the queued queries of every hypothesis in DENSE lanes (queue order = point order, so neighbouring lanes
// still hold neighbouring points).  A fixed number of workgroups per hypothesis walks the queue in chunks of 256; per-lane stacks in
// LDS as before.  Kept apart from the first half so that a handful of tree searches no longer pins a whole workgroup, its 32 KB of
// stacks and its place on the CU for fifteen dependent round trips while the other 250 lanes have long finished.
template <int kCode>
__global__ __launch_bounds__(256) void nn_tree_kernel(IcpBatch b, SceneNNDev scene)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];
    if (pm.state == kSkip) return;
    uint32_t *counts = b.nn_qcount + kQCountStride * pose;
    const uint32_t queued = counts[b.iter & 1u];
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[(b.iter + 1u) & 1u] = 0u;         // re-arm the counter the NEXT pass will fill
    if (blockIdx.x * kBlockThreads >= queued) return;
    int *stk_node = reinterpret_cast<int *>(lds_raw) + threadIdx.x;
    float *stk_lb = reinterpret_cast<float *>(lds_raw) + (size_t)(kCode & 0xff) * kBlockThreads + threadIdx.x;
    const float *cl = reinterpret_cast<const float *>(b.cloud + pm.start);
    uint32_t *win = b.nn_prev + pm.start;
    float *slk = b.nn_slack + pm.start;
    const uint2 *queue = b.nn_queue + pm.start;
    const float accept = scene.max_dist_diff * scene.max_dist_diff;
    NNCount cnt;
    uint32_t n_window = 0, n_tree = 0, n_pyramid = 0, n_cells = 0;
    for (uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x; i < queued; i += gridDim.x * kBlockThreads) {
        const uint2 e = queue[i];
        const uint32_t j = e.x;
        float best = __uint_as_float(e.y);
        const bool still = best < 0.0f;
        best = still ? -best : best;
        const pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);          // as nn_search_kernel stored it
        const float x = q.x, y = q.y, z = q.z;
        // A point that has hardly moved still has (nearly) its true neighbour as temporal seed: d_old - step <= d_new <= d_old + step.
        // Only queries without such a seed -- the first passes, while the updates are still millimetres -- pay for the descent
        // through the representative points.
        uint32_t w = kNoPrev;
        bool settled = false;
        float other = 0.0f;
        if (scene.grid && !still) {
            float bsq = 0.0f, osq = 0.0f;
            grid_pyramid_bound(scene, x, y, z, best); ++n_pyramid;
            settled = best < accept && grid_search(scene, x, y, z, best, w, &n_cells, &bsq, &osq);
            if (settled) other = sqrtf(osq) * 0.99999f;
        }
        if (settled) ++n_window;
        else {
            Corr c;
            ++n_tree;
            if (!query_nn_stack_from<kCode, false>(scene, nullptr, stk_node, stk_lb, x, y, z, c, best, w, &cnt)) w = kNoPrev;
        }
        win[j] = w;
        slk[j] = other;                                           // the ordered binary walk does not report its runner-up: no shortcut next pass
    }
    if (scene.counters) {                                        // instrumented runs only (option "nn_count")
        const uint32_t v[8] = { 0u, n_window, n_tree, n_pyramid, cnt.nodes, cnt.leaves, cnt.leaf_points, n_cells };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if ((threadIdx.x & 63u) == 0u) atomicAdd(&scene.counters[(size_t)b.iter * 8 + i], (unsigned long long)t);
        }
    }
}

// The queued queries of every hypothesis, searched over the wide records as TASKS (see "wide records" above).  Two walks over the same
// records came first and both lost to the binary per-lane walk's 5.1 ms in pass 0 (256 hypotheses, 5.4 M tree searches): one query per
// lane (eight 16-byte loads per visit and lane: 6.8 ms, bound by the rate of divergent loads) and one query per group of eight / four
// lanes with a stack per group (6.6 / 6.0 ms: every lane of a wavefront pays for the node branch AND the leaf branch of every iteration
// and for the longest walk among its groups -- VALU-bound at a quarter of the lanes doing useful work).  So the walk is cut into
// uniform pieces: a NODE TASK = (query, wide node): four lanes test the node's eight boxes, two each; a LEAF TASK = (query, leaf): four
// lanes test its points, two each per round.  A wavefront owns 64 queries at a time and two LIFO task queues in LDS; a step pops sixteen
// tasks of ONE kind (one per group of four lanes), so all lanes run the same code, and pushes what the boxes admit.  What a query has
// found so far lives in LDS and is updated with LDS atomics: `bound` (float bits, atomicMin), `best` = (distance bits << 32 | point index)
// (64-bit atomicMin: smallest distance, lowest index among equals), `tied` (smallest distance two different points were seen at),
// `second` (smallest distance / box bound of everything that is not the best).  When both queues are empty all 64 queries are finished:
// exactly one point at the minimum -> that is the reference's answer under any visiting order; a tie (tied == minimum) or a queue
// overflow -> the ordered stackless walk repeats the query from the minimum, as the reference would resolve it.
// The bound (descent through the representative points) and the exact pixel window run in a kernel of their own, nn_bound_kernel, which
// hands what it cannot settle to nn_tree_wide_kernel through a second queue: the descent needs twice the registers of the task walk, and
// in one kernel (as first built: 3.6 ms for pass 0) it held the walk to four wavefronts per SIMD and three workgroup barriers per chunk.
#ifndef PR_WIDE_WAVES
#define PR_WIDE_WAVES 5                                        // wavefronts per SIMD the task walk is compiled for
#endif
#ifndef PR_WIDE_LANES
#define PR_WIDE_LANES 2                                        // lanes per task
#endif
#ifndef PR_WIDE_QCAP
#define PR_WIDE_QCAP 384
#endif
#ifndef PR_WIDE_LCAP
#define PR_WIDE_LCAP 288
#endif
constexpr uint32_t kTaskQCap = PR_WIDE_QCAP, kTaskLCap = PR_WIDE_LCAP;   // entries of the node / leaf task queue of a wavefront (a leaf queue is drained from 16 entries on)
constexpr uint32_t kNoIdx = 0xffffffffu;
// Queue 1 (nn_search_kernel's leftovers) -> bound + pixel window, one query per lane -> winners, or queue 2 (point, bound).
__global__ __launch_bounds__(256) void nn_bound_kernel(IcpBatch b, SceneNNDev scene)
{
    __shared__ uint32_t wave_n[4], wg_base;
    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];
    if (pm.state == kSkip) return;
    uint32_t *counts = b.nn_qcount + kQCountStride * pose;
    const uint32_t queued = counts[b.iter & 1u];
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[(b.iter + 1u) & 1u] = 0u;         // re-arm the counter the NEXT pass' search kernel will fill
    if (blockIdx.x * kBlockThreads >= queued) return;
    const float *cl = reinterpret_cast<const float *>(b.cloud + pm.start);
    uint32_t *win = b.nn_prev + pm.start;
    float *slk = b.nn_slack + pm.start;
    const uint2 *queue = b.nn_queue + pm.start;
    uint2 *queue2 = b.nn_queue2 + pm.start;
    uint32_t *q2_count = counts + 2u + (b.iter & 1u);
    const float accept = scene.max_dist_diff * scene.max_dist_diff;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t n_window = 0, n_pyramid = 0, n_cells = 0;
    for (uint32_t i0 = blockIdx.x * kBlockThreads; i0 < queued; i0 += gridDim.x * kBlockThreads) {
        const uint32_t i = i0 + threadIdx.x;
        bool pending = false;
        uint32_t j = 0; float bst = 0.0f;
        if (i < queued) {
            const uint2 e = queue[i];
            j = e.x;
            bst = __uint_as_float(e.y);
            const bool still = bst < 0.0f;                       // the sign carries "no descent needed" (nn_search_kernel)
            bst = still ? -bst : bst;
            pending = true;
            if (!scene.grid && !still) {
                // No pixel grid (a bare ICP call has no camera) and the query is new or has moved (its previous winner, centimetres away
                // now, is a loose bound): the task walk is order-free, so with a loose bound it would open a good part of the tree before
                // a leaf near the query tightens it.  The classic descent does that first: follow the split planes to the query's own leaf (8 bytes
                // per level) and take the nearest of its points as the seed -- an existing point's distance, inflated (nn_seed_bound).
                const pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);
                uint32_t cur = 0u;
                for (int guard = 0; guard < 64; ++guard) {
                    const uint2 d = scene.desc[cur];
                    const uint32_t tag = d.y >> 30;
                    if (tag == 3u) {
                        float dmin = FLT_MAX;
                        for (uint32_t i = d.x; i < (d.y & 0x3fffffffu); ++i) {
                            const float4 p = scene.pts[i];
                            const float d2 = (q.x - p.x) * (q.x - p.x) + (q.y - p.y) * (q.y - p.y) + (q.z - p.z) * (q.z - p.z);
                            dmin = d2 < dmin ? d2 : dmin;
                        }
                        const float bb = dmin * 1.000001f + 1e-30f;
                        if (bb < bst) bst = bb;
                        break;
                    }
                    const float c = (tag == 0u) ? q.x : ((tag == 1u) ? q.y : q.z);
                    const uint32_t c1 = d.y & 0x3fffffffu;
                    cur = (c - __uint_as_float(d.x) < 0.0f) ? c1 : c1 + 1u;
                }
            }
            if (scene.grid && !still) {
                const pr_vec3 q = ld_off<pr_vec3>(cl, j * 12u);
                uint32_t w = kNoPrev; float bsq = 0.0f, osq = 0.0f;
                grid_pyramid_bound(scene, q.x, q.y, q.z, bst);
                ++n_pyramid;
                if (bst < accept && grid_search(scene, q.x, q.y, q.z, bst, w, &n_cells, &bsq, &osq)) { pending = false; ++n_window; win[j] = w; slk[j] = sqrtf(osq) * 0.99999f; }
            }
        }
        const unsigned long long m = __ballot(pending);
        if (lane == 0) wave_n[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) { const uint32_t t = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3]; wg_base = t ? atomicAdd(q2_count, t) : 0u; }
        __syncthreads();
        if (pending) {
            uint32_t slot = wg_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            for (uint32_t w2 = 0; w2 < wave; ++w2) slot += wave_n[w2];
            queue2[slot] = make_uint2(j, __float_as_uint(bst));
        }
        __syncthreads();
    }
    if (scene.counters) {                                        // instrumented runs only (option "nn_count")
        const uint32_t v[8] = { 0u, n_window, 0u, n_pyramid, 0u, 0u, 0u, n_cells };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if ((threadIdx.x & 63u) == 0u && t) atomicAdd(&scene.counters[(size_t)b.iter * 8 + i], (unsigned long long)t);
        }
    }
}
// minimum of two floats that are known not to be NaN: one v_min_f32 (fminf's IEEE minNum semantics cost a canonicalising v_max per operand)
__device__ __forceinline__ float min_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// exclusive prefix sum over the 64 lanes of a wavefront (and the total): in-row Hillis-Steele with DPP row shifts, row totals by readlane
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);      // row_shr:1 (lanes shifted in from outside the row read 0)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);      // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);      // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);      // row_shr:8
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 31),
                   t2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 47), t3 = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    x += (row > 0u ? t0 : 0u) + (row > 1u ? t1 : 0u) + (row > 2u ? t2 : 0u);
    total = t0 + t1 + t2 + t3;
    return x - v;
}
// Queue 2 -> the task walk.  Wavefronts are independent (no workgroup barrier): each takes 64 queries at a time.  kLanes lanes share a task:
// 8 / kLanes boxes of a node, or 8 / kLanes points of a leaf per round, per lane; a step pops 64 / kLanes tasks of one kind.
template <int kLanes>
__global__ __launch_bounds__(256, PR_WIDE_WAVES) void nn_tree_wide_kernel(IcpBatch b, SceneNNDev scene, uint32_t qbatch)
{
    constexpr uint32_t kPer = 8u / kLanes, kTasks = 64u / kLanes;
    __shared__ uint2 s_nodeq[4][kTaskQCap], s_leafq[4][kTaskLCap];               // {reference, bound bits (low 6 bits cleared: rounded DOWN) | query slot}
    __shared__ float4 s_q[4][64];                                                // query point | bound (float bits, lowered with atomicMin)
    __shared__ uint32_t s_second[4][64], s_tied[4][64], s_ovf[4][64], s_root[4][64];
    __shared__ unsigned long long s_best[4][64];
    const uint32_t pose = blockIdx.y;
    const PoseMeta &pm = b.meta[pose];
    if (pm.state == kSkip) return;
    uint32_t *counts = b.nn_qcount + kQCountStride * pose + 2u;
    const uint32_t queued = counts[b.iter & 1u];
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[(b.iter + 1u) & 1u] = 0u;         // re-arm the counter the NEXT pass' bound kernel will fill
    if (blockIdx.x * 4u * qbatch >= queued) return;               // qbatch = queries a wavefront takes at a time (64, or 16 when the whole launch has few)
    const float *cl = reinterpret_cast<const float *>(b.cloud + pm.start);
    uint32_t *win = b.nn_prev + pm.start;
    float *slk = b.nn_slack + pm.start;
    const uint2 *todo = b.nn_queue2 + pm.start;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, c = lane % kLanes, grp = lane / kLanes;
    uint2 *nodeq = s_nodeq[wave], *leafq = s_leafq[wave];
    float4 *qs = s_q[wave];
    uint32_t *second = s_second[wave], *tied = s_tied[wave], *ovf = s_ovf[wave], *root = s_root[wave];
    unsigned long long *best = s_best[wave];
    uint32_t n_tree = 0, n_nodes = 0, n_leaves = 0, n_leaf_points = 0, n_redo_q = 0, n_steps = 0;
    for (uint32_t base = (blockIdx.x * 4u + wave) * qbatch; base < queued; base += gridDim.x * 4u * qbatch) {
        const bool have_q = lane < qbatch && base + lane < queued;
        uint32_t nN = 0, nL = 0;                                    // fill levels of the two queues (wave-uniform)
        const uint2 mine = have_q ? todo[base + lane] : make_uint2(0u, 0u);    // this lane's query: (point, bound bits)
        uint32_t bound_now = mine.y;                                // the bound this lane's query was (last) started from
        {
            const pr_vec3 q = have_q ? ld_off<pr_vec3>(cl, mine.x * 12u) : pr_vec3{ 0.0f, 0.0f, 0.0f };
            qs[lane] = make_float4(q.x, q.y, q.z, __uint_as_float(mine.y));
            best[lane] = ((unsigned long long)mine.y << 32) | kNoIdx;
            second[lane] = 0x7f7fffffu; tied[lane] = 0xffffffffu; ovf[lane] = 0u; root[lane] = lane;
            if (have_q) ++n_tree;
        }
        // The walks of the 64 queries are started kTasks at a time, whenever the node queue runs low: all 64 root tasks at once would
        // spread three levels of every walk over the queues before the first leaf is reached (depth-first order keeps them short).
        // A query that lost a task to a full queue is walked again in a second round, alone with the other such queries and from the
        // minimum it did find (a handful of queries cannot fill the queues); only if that fails too does it go to the ordered walk.
        uint32_t n_q = (queued - base < qbatch) ? (queued - base) : qbatch;
        for (int round = 0; round < 2 && n_q; ++round) {
        uint32_t started = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while (nN | nL | (n_q - started)) {
            if (nN < kTasks && started < n_q) {                     // root tasks: wide node 0, bound 0.0f
                const uint32_t add = (n_q - started < kTasks) ? (n_q - started) : kTasks;
                if (lane < add) nodeq[nN + lane] = make_uint2(0u, root[started + lane]);
                nN += add; started += add;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const bool leaf_step = (nL >= kTasks) || (nN == 0u);
            if (lane == 0u) ++n_steps;
            const uint32_t avail = leaf_step ? nL : nN, k = avail < kTasks ? avail : kTasks;
            const bool active = grp < k;
            const uint2 e = active ? (leaf_step ? leafq : nodeq)[avail - 1u - grp] : make_uint2(0u, 0u);
            if (leaf_step) nL -= k; else nN -= k;
            const uint32_t q = e.y & 63u, ref = e.x;
            const float lb_in = __uint_as_float(e.y & ~63u);
            const float4 qp = qs[q];
            const float sx = qp.x, sy = qp.y, sz = qp.z, bnd = qp.w;
            uint32_t *bound_q = reinterpret_cast<uint32_t *>(&qs[q]) + 3;
            const bool alive = active && lb_in <= bnd;
            float sec_l = (active && !alive) ? lb_in : FLT_MAX;      // what this lane rules out (lower bounds of other points' distances)
            if (!leaf_step) {
                // ---------------- node tasks: lane c of a group tests slots kPer*c .. kPer*c + kPer-1
                uint4 r[kPer];
#pragma unroll
                for (uint32_t i = 0; i < kPer; ++i) r[i] = make_uint4(0u, 0u, 0u, kWideEmpty);
                if (alive) {
                    const uint4 *rec = scene.wide + (size_t)ref * 8u + kPer * c;
#pragma unroll
                    for (uint32_t i = 0; i < kPer; ++i) r[i] = rec[i];
                    if (c == 0u) ++n_nodes;
                }
                uint32_t cntI = 0, cntL = 0;
                float lb[kPer]; bool keep[kPer], leaf[kPer];
#pragma unroll
                for (uint32_t i = 0; i < kPer; ++i) {
                    const bool v = r[i].w != kWideEmpty;
                    lb[i] = wide_box_lb(sx, sy, sz, r[i].x, r[i].y, r[i].z, scene);
#ifdef PR_BOX_TWICE                                                 // experiment: what the box test's share of the step is (same result, computed twice)
                    { float sx2 = sx; asm volatile("" : "+v"(sx2)); lb[i] = min_f32(lb[i], wide_box_lb(sx2, sy, sz, r[i].x, r[i].y, r[i].z, scene)); }
#endif
                    keep[i] = v && lb[i] <= bnd;
                    leaf[i] = (r[i].w & kWideLeaf) != 0u;
                    if (v && !keep[i]) sec_l = min_f32(sec_l, lb[i]);
                    cntI += (keep[i] && !leaf[i]) ? 1u : 0u; cntL += (keep[i] && leaf[i]) ? 1u : 0u;
                }
                uint32_t tot = 0;
                const uint32_t ex = wave_excl_scan(cntI | (cntL << 16), tot);
                uint32_t pI = nN + (ex & 0xffffu), pL = nL + (ex >> 16);
#pragma unroll
                for (uint32_t i = 0; i < kPer; ++i) {                  // straight-line: destination by selects, one predicated store per slot
                    const uint32_t pos = leaf[i] ? pL : pI, cap = leaf[i] ? kTaskLCap : kTaskQCap;
                    uint2 *dst = (leaf[i] ? leafq : nodeq) + pos;
                    const bool fits = pos < cap;
                    if (keep[i] && fits) *dst = make_uint2(r[i].w, (__float_as_uint(lb[i]) & ~63u) | q);
                    if (keep[i] && !fits) ovf[q] = 1u;
                    pL += (keep[i] && leaf[i]) ? 1u : 0u; pI += (keep[i] && !leaf[i]) ? 1u : 0u;
                }
                nN += tot & 0xffffu; if (nN > kTaskQCap) nN = kTaskQCap;
                nL += tot >> 16; if (nL > kTaskLCap) nL = kTaskLCap;
            } else {
                // ---------------- leaf tasks: kPer points per lane and round, all loaded before any is looked at
                const uint32_t first = ref & kWideFirstMask, cnt = alive ? ((ref >> 27) & 15u) : 0u;
                if (alive && c == 0u) { ++n_leaves; n_leaf_points += cnt; }
                for (uint32_t kb = 0; kb < cnt; kb += 8u) {
                    const uint32_t ka = kb + kPer * c;
                    float4 pt[kPer];
#pragma unroll
                    for (uint32_t h = 0; h < kPer; ++h) pt[h] = scene.pts[first + (ka + h < cnt ? ka + h : 0u)];
#pragma unroll
                    for (uint32_t h = 0; h < kPer; ++h) {
                        if (ka + h >= cnt) continue;
                        const float d2 = (sx - pt[h].x) * (sx - pt[h].x) + (sy - pt[h].y) * (sy - pt[h].y) + (sz - pt[h].z) * (sz - pt[h].z);   // pcd_scene.h:88-91
                        if (d2 <= bnd) {                               // rare: a point that may be the minimum
                            const uint32_t idx = first + ka + h, db = __float_as_uint(d2);
                            const unsigned long long key = ((unsigned long long)db << 32) | idx;
                            const unsigned long long old = atomicMin(&best[q], key);
                            const uint32_t old_d = (uint32_t)(old >> 32), old_i = (uint32_t)old;
                            if (old_d == db && old_i != idx && old_i != kNoIdx) atomicMin(&tied[q], db);
                            if (key < old) { if (old_i != kNoIdx) sec_l = min_f32(sec_l, __uint_as_float(old_d)); if (d2 < bnd) atomicMin(bound_q, db); }
                            else sec_l = min_f32(sec_l, d2);
                        } else sec_l = min_f32(sec_l, d2);
                    }
                }
            }
            nN = (uint32_t)__builtin_amdgcn_readfirstlane((int)nN); nL = (uint32_t)__builtin_amdgcn_readfirstlane((int)nL);     // wave-uniform by construction
            float sec_g = sec_l;
            if (kLanes >= 2) sec_g = min_f32(sec_l, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sec_l), 0xB1, 0xf, 0xf, true)));          // quad_perm [1,0,3,2]
            if (kLanes == 4) sec_g = min_f32(sec_g, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sec_g), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
            if (c == 0u && sec_g < FLT_MAX) atomicMin(&second[q], __float_as_uint(sec_g));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
            n_q = 0u;
            if (round == 0) {                                       // queries that lost a task: reset, tighten, list for the second round
                const bool lost = have_q && ovf[lane] != 0u;
                const unsigned long long ml = __ballot(lost);
                if (ml) {
                    if (lost) {
                        const unsigned long long key = best[lane];
                        const uint32_t m_bits = (uint32_t)(key >> 32), idx = (uint32_t)key;
                        uint32_t bb = mine.y;                         // (positive float bits order like the floats)
                        if (idx != kNoIdx && m_bits < bb) { const uint32_t t = __float_as_uint(__uint_as_float(m_bits) * 1.000001f + 1e-30f); if (t < bb) bb = t; }   // = nn_seed_bound
                        reinterpret_cast<uint32_t *>(&qs[lane])[3] = bb;
                        best[lane] = ((unsigned long long)bb << 32) | kNoIdx;
                        second[lane] = 0x7f7fffffu; tied[lane] = 0xffffffffu; ovf[lane] = 0u;
                        root[__popcll(ml & ((1ull << lane) - 1ull))] = lane;
                        bound_now = bb;
                    }
                    n_q = (uint32_t)__builtin_amdgcn_readfirstlane((int)__popcll(ml));
                }
            }
        }
        // all 64 queries of this wavefront are finished: one lane per query delivers
        if (have_q) {
            const unsigned long long key = best[lane];
            const uint32_t m_bits = (uint32_t)(key >> 32), idx = (uint32_t)key, b0 = bound_now, j = mine.x;
            const bool found = idx != kNoIdx && m_bits < b0;
            if (ovf[lane] != 0u || (found && tied[lane] == m_bits)) {
                // a tie or a dropped task: the ordered walk, from the minimum found (= nn_seed_bound: an existing point's distance)
                ++n_redo_q;
                const float bnd = found ? fminf(__uint_as_float(b0), __uint_as_float(m_bits) * 1.000001f + 1e-30f) : __uint_as_float(b0);
                const float4 qp = qs[lane];
                win[j] = query_nn_bounded(scene, qp.x, qp.y, qp.z, bnd);
                slk[j] = 0.0f;                                      // an ordered walk does not report its runner-up: no shortcut next pass
            } else if (found) { win[j] = idx; slk[j] = sqrtf(__uint_as_float(second[lane])) * 0.99999f; }
            else { win[j] = kNoPrev; slk[j] = 0.0f; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (scene.counters) {                                        // instrumented runs only (option "nn_count")
        uint32_t n_pyr = 0u;
#ifdef PR_COUNT_STEPS
        n_pyr = n_steps;                                            // experiment builds: the 'pyramid' column counts the steps of the task walk
#endif
#ifdef PR_COUNT_REDO
        n_pyr = n_redo_q;                                           // experiment builds: the 'pyramid' column also counts the queries handed to the ordered walk
#endif
        (void)n_redo_q; (void)n_steps;
        const uint32_t v[8] = { 0u, 0u, n_tree, n_pyr, n_nodes, n_leaves, n_leaf_points, 0u };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if ((threadIdx.x & 63u) == 0u && t) atomicAdd(&scene.counters[(size_t)b.iter * 8 + i], (unsigned long long)t);
        }
    }
}

// scene points -> grid cells.  cell_idx starts at -1; a second claim on a cell, or a point outside the image, clears info[0].
__global__ __launch_bounds__(256) void nn_grid_claim_kernel(const pr_vec3 *__restrict__ pcd, uint32_t n_points, SceneNNDev g, int32_t *__restrict__ cell_idx,
                                                            uint32_t *__restrict__ info)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_points) return;
    const pr_vec3 p = pcd[i];
    float u, v;
    grid_project(g, p.x, p.y, p.z, u, v);
    if (!(p.z > 0 && u >= 0.0f && u < (float)g.gw && v >= 0.0f && v < (float)g.gh)) { info[0] = 0u; return; }
    const int cell = (int)floorf(v) * (int)g.gw + (int)floorf(u);
    if (atomicCAS(&cell_idx[cell], -1, (int)i) != -1) info[0] = 0u;
}
// one coarser level: cell (bx, by) takes the occupied child cell nearest its centre (children = 4 x 4 cells of the finer level)
__global__ __launch_bounds__(256) void nn_grid_coarsen_kernel(const float4 *__restrict__ fine, int fw, int fh, float4 *__restrict__ coarse, int cw, int ch)
{
    const int c = (int)(blockIdx.x * 256 + threadIdx.x);
    if (c >= cw * ch) return;
    const int bx = c % cw, by = c / cw;
    float4 pick = make_float4(1e30f, 1e30f, 1e30f, __int_as_float(-1));
    int pick_d = 1 << 30;
    for (int dy = 0; dy < 4; ++dy)
        for (int dx = 0; dx < 4; ++dx) {
            const int x = bx * 4 + dx, y = by * 4 + dy;
            if (x >= fw || y >= fh) continue;
            const float4 v = fine[(size_t)y * fw + x];
            const int dist = (2 * dx - 3) * (2 * dx - 3) + (2 * dy - 3) * (2 * dy - 3);      // squared distance to the block centre, in half cells
            if (__float_as_int(v.w) >= 0 && dist < pick_d) { pick = v; pick_d = dist; }
        }
    coarse[c] = pick;
}
__global__ __launch_bounds__(256) void nn_grid_fill_kernel(const pr_vec3 *__restrict__ pcd, const int32_t *__restrict__ cell_idx, uint32_t n_cells,
                                                           float4 *__restrict__ grid)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    const int i = cell_idx[c];
    grid[c] = (i >= 0) ? make_float4(pcd[i].x, pcd[i].y, pcd[i].z, __int_as_float(i)) : make_float4(1e30f, 1e30f, 1e30f, __int_as_float(-1));
}

// second stage: workgroup sums added sequentially in workgroup order, starting from 0
__device__ __forceinline__ float sum_partials(const float *partial, uint32_t pose, uint32_t nblk, uint32_t used, uint32_t comp)
{
    // the loads of a chunk are independent and issued together; only the additions are ordered
    float total = 0.0f;
    const float *p = partial + (size_t)pose * nblk * kAccStride + comp;
    for (uint32_t g0 = 0; g0 < used; g0 += 16) {
        float v[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) v[k] = (g0 + k < used) ? p[(size_t)(g0 + k) * kAccStride] : 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) if (g0 + k < used) total += v[k];
    }
    return total;
}

__global__ __launch_bounds__(64) void icp_finalize_kernel(const float *__restrict__ partial, const PoseMeta *__restrict__ meta,
                                                          uint32_t nblk, uint32_t ppb, float *__restrict__ sums)
{
    const uint32_t pose = blockIdx.x;
    if (meta[pose].state == kSkip) return;
    const uint32_t n = meta[pose].count;
    const uint32_t used = (n + ppb - 1) / ppb;
    if (threadIdx.x < kAccStride)
        sums[(size_t)pose * kAccStride + threadIdx.x] = (threadIdx.x < 29) ? sum_partials(partial, pose, nblk, used, threadIdx.x) : 0.0f;
}

// The per-iteration host logic of icp.cu:178-212 for one hypothesis, on the 29 reduced sums: scores, convergence test,
// 6x6 solve, accumulation of the transform.  Returns true when the hypothesis is finished; otherwise E holds the update.
__device__ __forceinline__ bool pose_iteration(const float *Ab, uint32_t n, DevIcpState &s, const pr_criteria &crit, uint32_t iter, float (&E)[16])
{
    s.passes += 1;
    const float cnt = Ab[28], err = Ab[27];
    if (cnt == 0) return true;                                               // icp.cu:183
    const float prev_fit = s.fitness, prev_rmse = s.rmse;
    s.fitness = cnt / (float)n;                                              // icp.cu:185
    s.rmse = sqrtf(err / cnt);                                               // icp.cu:186
    if (iter == (uint32_t)crit.max_iteration) return true;                   // icp.cu:189
    const float df = s.fitness - prev_fit, dr = s.rmse - prev_rmse;
    if (((df < 0) ? -df : df) < crit.relative_fitness && ((dr < 0) ? -dr : dr) < crit.relative_rmse) return true;   // icp.cu:191-194
    float A[36], bb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) bb[i] = Ab[21 + i];
    {
        int k = 0;
#pragma unroll
        for (int y = 0; y < 6; ++y) {
#pragma unroll
            for (int x = y; x < 6; ++x) { A[x + y * 6] = Ab[k]; A[y + x * 6] = Ab[k]; ++k; }
        }
    }
    prs::solve_666_impl(A, bb, E);
    prs::mat4_mul_impl(E, s.T, s.T);                                         // icp.cu:212
    return false;
}

// PR_SOLVE_DEVICE: the per-iteration host logic of icp.cu:178-212 for one hypothesis per wavefront
__global__ __launch_bounds__(64) void icp_finalize_solve_kernel(const float *__restrict__ partial, PoseMeta *__restrict__ meta,
                                                                uint32_t nblk, uint32_t ppb, DevIcpState *__restrict__ st,
                                                                pr_criteria crit, uint32_t iter)
{
    const uint32_t pose = blockIdx.x;
    if (meta[pose].state == kSkip) return;
    const uint32_t n = meta[pose].count;
    const uint32_t used = (n + ppb - 1) / ppb;
    DevIcpState s = st[pose];
    float total = 0.0f;
    if (threadIdx.x < 29) total = sum_partials(partial, pose, nblk, used, threadIdx.x);
    float E[16];
    const bool finished = pose_iteration_wave(total, n, s, crit, iter, E);
    if (threadIdx.x != 0) return;
    if (finished) { s.done = 1; meta[pose].state = kSkip; }
    else {
#pragma unroll
        for (int i = 0; i < 12; ++i) meta[pose].xform[i] = E[i];
        meta[pose].state = kRunWithTransform;
    }
    st[pose] = s;
}

__global__ void pack_results_kernel(const DevIcpState *__restrict__ st, pr_result *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pr_result r;
    for (int k = 0; k < 16; ++k) r.T[k] = st[i].T[k];
    r.inlier_rmse = st[i].rmse; r.fitness = st[i].fitness;
    out[i] = r;
}
// results to the device buffer and, in the same launch, results / cloud sizes to pinned host staging (stores over the host link)
__global__ void pack_export_kernel(const DevIcpState *__restrict__ st, pr_result *__restrict__ out, const uint32_t *__restrict__ counts,
                                   uint32_t *__restrict__ host_counts, pr_result *__restrict__ host_results, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pr_result r;
    for (int k = 0; k < 16; ++k) r.T[k] = st[i].T[k];
    r.inlier_rmse = st[i].rmse; r.fitness = st[i].fitness;
    out[i] = r;
    host_counts[i] = counts[i];
    if (host_results) host_results[i] = r;
}
// small host-staged inputs (poses, pixel boxes) pulled into device memory by a kernel: an SDMA copy of 20 KB costs ~20 us of latency
__global__ void stage_words_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

// ================================================================================================
//  Dataflow ICP: ALL iterations of ALL hypotheses in one persistent launch.
//
//  The multi-launch loop serialises [pass over every pose] -> [solve of every pose] 21 times although pose i only ever
//  waits for pose i: at 256 poses about a third of each pass and all of each solve launch is dependency latency.  Here
//  every resident workgroup owns a fixed, strided set of virtual workgroups (pose, g) of the canonical tree and walks
//  them iteration by iteration; the last workgroup to deliver a partial sum for a pose runs that pose's finalize + 6x6
//  solve and publishes the update, everyone else picks it up when it next touches the pose.  Sums, solve and results are
//  bit-identical to the multi-launch path (same tree, same code).
//
//  Cross-workgroup data (partials, PoseMeta, DevIcpState, counters) moves with system-scope (sc0 sc1) accesses, i.e.
//  through memory, never through a possibly stale L1/L2 line; a producer drains its stores (s_waitcnt vmcnt(0)) before
//  the arrival atomic / ready flag that publishes them (cdna_hip_programming.md G16, valid form "sc0 sc1 stores and loads
//  both sides").  A cloud block is only ever touched by its owning workgroup, so plain accesses are fine there.
//  Every spin is bounded; a timeout raises the abort flag and the whole grid drains.
// ================================================================================================

template <class Scene, bool kNN, int kStack>
__global__ __launch_bounds__(256, 5) void icp_flow_kernel(FlowArgs a, Scene scene)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float wsum[4][kAccStride];
    __shared__ float Ab[kAccStride];
    __shared__ uint32_t sm_meta[16];
    __shared__ int sm_go;

    const int4 *lds_topo = nullptr;
    int *stk_node = nullptr; float *stk_lb = nullptr;
    if constexpr (kNN && kStack == 0) {
        int4 *dst = reinterpret_cast<int4 *>(lds_raw);
        for (uint32_t i = threadIdx.x; i < scene.lds_nodes; i += kBlockThreads) dst[i] = scene.topo[i];
        __syncthreads();
        lds_topo = dst;
    }
    if constexpr (kNN && kStack > 0) {
        stk_node = reinterpret_cast<int *>(lds_raw) + threadIdx.x;
        stk_lb = reinterpret_cast<float *>(lds_raw) + (size_t)(kStack & 0xff) * kBlockThreads + threadIdx.x;
        float4 *recs = reinterpret_cast<float4 *>(lds_raw + (size_t)(kStack & 0xff) * kBlockThreads * 8);
        if constexpr ((kStack & 0x100) == 0) for (uint32_t i = threadIdx.x; i < scene.lds_nodes * 4; i += kBlockThreads) recs[i] = scene.rec[i];
        __syncthreads();
        lds_topo = reinterpret_cast<const int4 *>(recs);
    }

    const uint32_t ppb = a.steps * kPointsPerStep;
    constexpr uint32_t kSpinLimit = 1u << 22;

    for (uint32_t it = 0; it <= (uint32_t)a.crit.max_iteration; ++it) {
        for (uint32_t v = blockIdx.x; v < a.n_vbs; v += gridDim.x) {
            const uint2 d = a.vb_desc[v];                          // {pose, g}: uniform
            const uint32_t pose = d.x, g = d.y;
            if (threadIdx.x == 0) {
                int go = 1;
                uint32_t spins = 0;
                while (ld_sys_u32(&a.ready[pose]) < it) {          // the solve that closes iteration it-1 of this pose
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > kSpinLimit || ((spins & 255u) == 0 && ld_sys_u32(a.abort_flag) != 0)) { st_sys_u32(a.abort_flag, 1u); go = -1; break; }
                }
                sm_go = go;
            }
            __syncthreads();
            if (sm_go < 0) return;
            if (threadIdx.x < 16) sm_meta[threadIdx.x] = ld_sys_u32(reinterpret_cast<const uint32_t *>(a.meta + pose) + threadIdx.x);
            __syncthreads();
            const uint32_t start = sm_meta[0], n = sm_meta[1];
            const int32_t st = (int32_t)sm_meta[2];
            const uint32_t first = g * ppb;
            if (st != kSkip && first < n) {
                const bool xf = (st == kRunWithTransform);
                float M[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) M[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(sm_meta[4 + i]));
                float acc[29];
#pragma unroll
                for (int i = 0; i < 29; ++i) acc[i] = 0.0f;
                float *cl = reinterpret_cast<float *>(a.cloud + start);
                vb_accumulate<Scene, kNN, kStack>(acc, cl, n, first, a.steps, xf, M, scene, lds_topo, stk_node, stk_lb);
                const float t = vb_reduce(acc, wsum);
                const uint32_t used = (n + ppb - 1) / ppb;
                if (threadIdx.x < 29) st_sys_f32(&a.partial[((size_t)pose * a.nblk + g) * kAccStride + threadIdx.x], t);
                if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // wave 0 issued the stores
                if (threadIdx.x == 0) {
                    const uint32_t ticket = atomicAdd(&a.arrive[pose], 1u);
                    sm_go = (ticket + 1u == used * (it + 1u)) ? 2 : 1;
                }
                __syncthreads();
                if (sm_go == 2) {                                  // last arrival of this iteration: finalize + solve
                    if (threadIdx.x < 29) {
                        float total = 0.0f;
                        const float *pp = a.partial + (size_t)pose * a.nblk * kAccStride + threadIdx.x;
                        for (uint32_t k = 0; k < used; ++k) total += ld_sys_f32(pp + (size_t)k * kAccStride);
                        Ab[threadIdx.x] = total;
                    }
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        DevIcpState s;
                        uint32_t *sw = reinterpret_cast<uint32_t *>(&s);
                        uint32_t *gw = reinterpret_cast<uint32_t *>(a.st + pose);
#pragma unroll
                        for (uint32_t i = 0; i < sizeof(DevIcpState) / 4; ++i) sw[i] = ld_sys_u32(gw + i);
                        float E[16];
                        const bool finished = pose_iteration(Ab, n, s, a.crit, it, E);
                        uint32_t *mw = reinterpret_cast<uint32_t *>(a.meta + pose);
                        if (finished) { s.done = 1; st_sys_u32(mw + 2, (uint32_t)kSkip); }
                        else {
#pragma unroll
                            for (int i = 0; i < 12; ++i) st_sys_f32(reinterpret_cast<float *>(mw + 4) + i, E[i]);
                            st_sys_u32(mw + 2, (uint32_t)kRunWithTransform);
                        }
#pragma unroll
                        for (uint32_t i = 0; i < sizeof(DevIcpState) / 4; ++i) st_sys_u32(gw + i, sw[i]);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        st_sys_u32(&a.ready[pose], finished ? 0xffffffffu : it + 1u);
                    }
                }
            }
            __syncthreads();
        }
    }
}

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

// ================================================================================================
//  kd-tree build on the device -- the reference's level-order build (pcd_scene.cpp:45-184) level by level:
//  one small kernel hands out child slots to the nodes of the level that split (children are appended pairwise in node order),
//  one workgroup per node then computes the box, picks the widest axis, and performs the stable two-ended partition with the
//  alternating tie rule through block scans (left part keeps its order, right part is filled from the end, the k-th tie goes
//  left iff k is even).  Same float operations, so nodes and permutation are bit-identical to the CPU build.
// ================================================================================================
__global__ __launch_bounds__(256) void kd_init_kernel(pr_kdnode *__restrict__ nodes, uint32_t cap, int *__restrict__ idx, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) idx[i] = (int)i;
    if (i < cap) {
        pr_kdnode b;
        b.parent = b.child1 = b.child2 = -1; b.split_v = 0.0f; b.split_dim = 0; b.left = 0; b.right = 0;
        for (int k = 0; k < 6; ++k) b.bbox[k] = 0.0f;
        if (i == 0) b.right = (int)n;
        nodes[i] = b;
    }
}

// ctrl = {level_lo, level_hi, count}.  Nodes [lo,hi) that hold more than max_leaf points get child slots count + 2*rank.
__global__ __launch_bounds__(256) void kd_level_plan_kernel(const pr_kdnode *__restrict__ nodes, uint32_t *__restrict__ ctrl, int max_leaf,
                                                            int *__restrict__ child_of, uint32_t cap)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t running;
    const uint32_t lo = ctrl[0], hi = ctrl[1];
    if (threadIdx.x == 0) running = ctrl[2];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = lo; base < hi; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const bool split = (i < hi) && (nodes[i].right - nodes[i].left > max_leaf);
        const unsigned long long m = __ballot(split);
        const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = 0, total = 0;
        for (uint32_t w = 0; w < 4; ++w) { if (w < wave) off += wsum[w]; total += wsum[w]; }
        if (i < hi) {
            const uint32_t c = running + 2 * (off + before);
            child_of[i - lo] = (split && c + 2 <= cap) ? (int)c : -1;
        }
        __syncthreads();
        if (threadIdx.x == 0) running += 2 * total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { ctrl[3] = running; }                   // count after this level (may exceed cap: host checks)
}

__device__ __forceinline__ float axis_coord(const pr_vec3 &p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }

// The CPU build keeps an extreme with "if (v > best) best = v" while walking the points in order, i.e. the FIRST occurrence
// among equal values survives -- observable only through the sign of a zero, but the node records are compared bit for bit.
// (value, sequence position) pairs reproduce that under any reduction order.
struct Ext { float v; int k; };
__device__ __forceinline__ Ext ext_max(Ext a, Ext b) { return (b.v > a.v || (b.v == a.v && b.k < a.k)) ? b : a; }
__device__ __forceinline__ Ext ext_min(Ext a, Ext b) { return (b.v < a.v || (b.v == a.v && b.k < a.k)) ? b : a; }
__device__ __forceinline__ Ext ext_shfl(Ext a, int off) { Ext r; r.v = __shfl_xor(a.v, off); r.k = __shfl_xor(a.k, off); return r; }

__global__ __launch_bounds__(256) void kd_level_split_kernel(pr_kdnode *__restrict__ nodes, const uint32_t *__restrict__ ctrl,
                                                             const int *__restrict__ child_of, const pr_vec3 *__restrict__ pcd,
                                                             int *__restrict__ idx, int *__restrict__ scratch)
{
    __shared__ float red[4][8];
    __shared__ int redk[4][8];
    __shared__ uint32_t wcnt[4][3];
    __shared__ uint32_t run[3];
    __shared__ float s_split;
    __shared__ int s_axis;
    const uint32_t node = ctrl[0] + blockIdx.x;
    const int child = child_of[blockIdx.x];
    if (child < 0) return;
    const int L = nodes[node].left, R = nodes[node].right;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // 1. box of the node's points (first occurrence wins among equal extremes, like the sequential CPU loop)
    const int kNone = 0x7fffffff;
    Ext mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a].v = FLT_MAX; mn[a].k = kNone; mx[a].v = -FLT_MAX; mx[a].k = kNone; }
    for (int k = L + (int)threadIdx.x; k < R; k += 256) {
        const pr_vec3 p = pcd[idx[k]];
        const float c3[3] = { p.x, p.y, p.z };
#pragma unroll
        for (int a = 0; a < 3; ++a) { Ext e; e.v = c3[a]; e.k = k; if (e.v > mx[a].v) mx[a] = e; if (e.v < mn[a].v) mn[a] = e; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) { mn[a] = ext_min(mn[a], ext_shfl(mn[a], off)); mx[a] = ext_max(mx[a], ext_shfl(mx[a], off)); }
    if (lane == 0) for (int a = 0; a < 3; ++a) { red[wave][a] = mn[a].v; redk[wave][a] = mn[a].k; red[wave][3 + a] = mx[a].v; redk[wave][3 + a] = mx[a].k; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bmin[3], bmax[3];
        for (int a = 0; a < 3; ++a) {
            Ext lo{ red[0][a], redk[0][a] }, hi{ red[0][3 + a], redk[0][3 + a] };
            for (int w = 1; w < 4; ++w) { lo = ext_min(lo, Ext{ red[w][a], redk[w][a] }); hi = ext_max(hi, Ext{ red[w][3 + a], redk[w][3 + a] }); }
            bmin[a] = lo.v; bmax[a] = hi.v;
        }
        int axis = 0; float cut = 0.0f, widest = -FLT_MAX;
        for (int a = 0; a < 3; ++a) {                             // first strictly widest axis, box midpoint (pcd_scene.cpp:96-110)
            const float extent = bmax[a] - bmin[a];
            if (extent > widest) { widest = extent; axis = a; cut = (bmin[a] + bmax[a]) / 2; }
        }
        s_axis = axis; s_split = cut;
        pr_kdnode &nd = nodes[node];
        for (int a = 0; a < 3; ++a) { nd.bbox[2 * a] = bmin[a]; nd.bbox[2 * a + 1] = bmax[a]; }
        nd.split_dim = axis; nd.child1 = child; nd.child2 = child + 1;
        run[0] = run[1] = run[2] = 0;
    }
    __syncthreads();
    const int axis = s_axis; const float cut = s_split;

    // 2. stable two-ended partition, 256 points at a time
    Ext left_max{ -FLT_MAX, kNone }, right_min{ FLT_MAX, kNone };
    for (int base = L; base < R; base += 256) {
        const int k = base + (int)threadIdx.x;
        const bool live = k < R;
        int id = 0; float v = 0.0f;
        if (live) { id = idx[k]; v = axis_coord(pcd[id], axis); }
        const bool tie = live && (v == cut);
        const unsigned long long mt = __ballot(tie);
        if (lane == 0) wcnt[wave][0] = (uint32_t)__popcll(mt);
        __syncthreads();
        uint32_t tie_before = run[0] + (uint32_t)__popcll(mt & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) tie_before += wcnt[w][0];
        // the toggle starts true and flips at every tie before the test: the k-th tie (k = tie_before + 1) goes left iff k is even
        const bool goes_left = live && (v < cut || (tie && ((tie_before + 1u) % 2u == 0u)));
        const bool goes_right = live && !goes_left;
        const unsigned long long ml = __ballot(goes_left), mr = __ballot(goes_right);
        if (lane == 0) { wcnt[wave][1] = (uint32_t)__popcll(ml); wcnt[wave][2] = (uint32_t)__popcll(mr); }
        __syncthreads();
        uint32_t lb = run[1] + (uint32_t)__popcll(ml & ((1ull << lane) - 1ull));
        uint32_t rb = run[2] + (uint32_t)__popcll(mr & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) { lb += wcnt[w][1]; rb += wcnt[w][2]; }
        if (goes_left) { scratch[L + (int)lb] = id; if (v > left_max.v) { left_max.v = v; left_max.k = k; } }
        if (goes_right) { scratch[R - 1 - (int)rb] = id; if (v < right_min.v) { right_min.v = v; right_min.k = k; } }
        __syncthreads();
        if (threadIdx.x == 0) for (uint32_t w = 0; w < 4; ++w) { run[0] += wcnt[w][0]; run[1] += wcnt[w][1]; run[2] += wcnt[w][2]; }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) { left_max = ext_max(left_max, ext_shfl(left_max, off)); right_min = ext_min(right_min, ext_shfl(right_min, off)); }
    if (lane == 0) { red[wave][6] = left_max.v; redk[wave][6] = left_max.k; red[wave][7] = right_min.v; redk[wave][7] = right_min.k; }
    __syncthreads();
    for (int k = L + (int)threadIdx.x; k < R; k += 256) idx[k] = scratch[k];
    if (threadIdx.x == 0) {
        Ext lmx{ red[0][6], redk[0][6] }, rmn{ red[0][7], redk[0][7] };
        for (int w = 1; w < 4; ++w) { lmx = ext_max(lmx, Ext{ red[w][6], redk[w][6] }); rmn = ext_min(rmn, Ext{ red[w][7], redk[w][7] }); }
        nodes[node].split_v = (lmx.v + rmn.v) / 2;                // pcd_scene.cpp:135
        const int head = L + (int)run[1];
        pr_kdnode &c1 = nodes[child], &c2 = nodes[child + 1];
        c1.parent = (int)node; c1.left = L; c1.right = head;
        c2.parent = (int)node; c2.left = head; c2.right = R;
    }
}

__global__ __launch_bounds__(256) void kd_advance_kernel(uint32_t *ctrl) { if (threadIdx.x == 0 && blockIdx.x == 0) { ctrl[0] = ctrl[1]; ctrl[1] = ctrl[3]; ctrl[2] = ctrl[3]; } }

__global__ __launch_bounds__(256) void kd_permute_kernel(const pr_vec3 *__restrict__ pcd, const pr_vec3 *__restrict__ nrm, const int *__restrict__ idx,
                                                         uint32_t n, pr_vec3 *__restrict__ pcd_out, pr_vec3 *__restrict__ nrm_out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    pcd_out[i] = pcd[idx[i]]; nrm_out[i] = nrm[idx[i]];
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

__global__ __launch_bounds__(256) void nn_accel_kernel(const pr_kdnode *__restrict__ nodes, uint32_t n_nodes,
                                                       const pr_vec3 *__restrict__ pcd, uint32_t n_points,
                                                       int4 *__restrict__ topo, float4 *__restrict__ bmin,
                                                       float4 *__restrict__ bmax, float4 *__restrict__ pts)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_points) pts[i] = make_float4(pcd[i].x, pcd[i].y, pcd[i].z, 0.0f);
    if (i >= n_nodes) return;
    const pr_kdnode nd = nodes[i];
    const bool leaf = (nd.child1 < 0 || nd.child2 < 0);              // pcd_scene.h:21-24
    const int pw = ((nd.parent + 1) & 0x3fffffff) | (int)((uint32_t)(nd.split_dim & 3) << 30);
    if (leaf) {
        float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for (int k = nd.left; k < nd.right; ++k) {
            const float c3[3] = { pcd[k].x, pcd[k].y, pcd[k].z };
            for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], c3[d]); hi[d] = fmaxf(hi[d], c3[d]); }
        }
        topo[i] = make_int4(nd.left, nd.right, -1, pw);
        bmin[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        bmax[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    } else {
        topo[i] = make_int4(__float_as_int(nd.split_v), nd.child1, nd.child2, pw);
        bmin[i] = make_float4(nd.bbox[0], nd.bbox[2], nd.bbox[4], 0.0f);
        bmax[i] = make_float4(nd.bbox[1], nd.bbox[3], nd.bbox[5], 0.0f);
    }
}

// 64-byte traversal records for the stack variant + the depth of the tree (longest root-to-node path)
__global__ __launch_bounds__(256) void nn_records_kernel(const int4 *__restrict__ topo, const float4 *__restrict__ bmin,
                                                         const float4 *__restrict__ bmax, uint32_t n_nodes,
                                                         float4 *__restrict__ rec, uint32_t *__restrict__ max_depth)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    const int4 t = topo[i];
    float4 r0, r1 = make_float4(0, 0, 0, 0), r2 = r1, r3 = r1;
    if (t.z < 0) r0 = make_float4(__int_as_float(t.x), __int_as_float(t.y), __int_as_float(-1), 0.0f);
    else {
        const int dim = (int)((uint32_t)t.w >> 30);
        r0 = make_float4(__int_as_float(t.x), __int_as_float(t.y), __int_as_float(t.z), __int_as_float(dim));
        const int y = ((uint32_t)t.y < n_nodes) ? t.y : 0, z = ((uint32_t)t.z < n_nodes) ? t.z : 0;   // malformed links are reported below
        const float4 a0 = bmin[y], a1 = bmax[y], c0 = bmin[z], c1 = bmax[z];
        r1 = make_float4(a0.x, a0.y, a0.z, a1.x);
        r2 = make_float4(a1.y, a1.z, c0.x, c0.y);
        r3 = make_float4(c0.z, c1.x, c1.y, c1.z);
    }
    rec[(size_t)i * 4] = r0; rec[(size_t)i * 4 + 1] = r1; rec[(size_t)i * 4 + 2] = r2; rec[(size_t)i * 4 + 3] = r3;
    uint32_t depth = 0;
    for (int p = (t.w & 0x3fffffff) - 1; p >= 0 && depth < 4096; p = (topo[p].w & 0x3fffffff) - 1) ++depth;
    // The stack size is derived from this depth, which follows the PARENT links, while the search follows the CHILD links: the
    // nodes are the caller's, so the two are cross-checked.  Any disagreement reports an impossible depth and the scene is
    // searched with the reference's stackless walk instead (which uses both kinds of link exactly like pcd_scene.h:60-136).
    if (t.z >= 0) {
        const bool in_range = (uint32_t)t.y < n_nodes && (uint32_t)t.z < n_nodes && t.y > (int)i && t.z > (int)i;
        if (!in_range || (topo[t.y].w & 0x3fffffff) - 1 != (int)i || (topo[t.z].w & 0x3fffffff) - 1 != (int)i) depth = 0x7fffffffu;
    }
    atomicMax(max_depth, depth);
}

// Compact traversal records: 32 bytes per node.
//   word 0: split value (internal) | first point (leaf)
//   word 1: child1 | dim << 30 (internal; child2 = child1 + 1, as KDTree_cpu::build_tree appends children pairwise) | end point | 3 << 30 (leaf)
//   words 2..7: the boxes of child1 and child2 as 12 uint16 {lo.x lo.y lo.z hi.x hi.y hi.z} x 2, quantised in the frame of
//   the root box and rounded OUTWARDS (checked with the very dequantisation arithmetic the query uses).  A looser box can
//   only make the search visit a subtree it could have skipped -- a subtree whose every point is farther than the current
//   best -- so winners, distances and tie-breaks are those of the exact boxes.
__global__ void nn_frame_kernel(const float4 *__restrict__ bmin, const float4 *__restrict__ bmax, uint32_t *__restrict__ info)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float lo[3] = { bmin[0].x, bmin[0].y, bmin[0].z }, hi[3] = { bmax[0].x, bmax[0].y, bmax[0].z };
    for (int a = 0; a < 3; ++a) {
        float sc = (hi[a] - lo[a]) / 65535.0f * 1.000001f;
        if (!(sc > 1e-30f)) sc = 1e-30f;
        info[2 + a] = __float_as_uint(lo[a]);
        info[5 + a] = __float_as_uint(sc);
    }
    info[1] = 1u;
}
__global__ __launch_bounds__(256) void nn_records32_kernel(const int4 *__restrict__ topo, const float4 *__restrict__ bmin,
                                                           const float4 *__restrict__ bmax, uint32_t n_nodes, uint4 *__restrict__ rec32,
                                                           uint2 *__restrict__ desc, uint32_t *__restrict__ info)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    const int4 t = topo[i];
    uint32_t w[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    bool ok = true;
    if (t.z < 0) { w[0] = (uint32_t)t.x; w[1] = (uint32_t)t.y | (3u << 30); ok = ((uint32_t)t.y < (1u << 30)); }
    else {
        const uint32_t dim = (uint32_t)t.w >> 30;
        w[0] = (uint32_t)t.x; w[1] = (uint32_t)t.y | (dim << 30);
        ok = (t.z == t.y + 1) && ((uint32_t)t.y < (1u << 30)) && dim < 3 && (uint32_t)t.z < n_nodes;
        // the 8-byte descent skips a far side on (q - split)^2 alone, which needs left_max <= split <= right_min
        // (pcd_scene.cpp:135 builds trees that way; other trees take the exact 64-byte records)
        if (ok) {
            const float lmax = (dim == 0) ? bmax[t.y].x : ((dim == 1) ? bmax[t.y].y : bmax[t.y].z);
            const float rmin = (dim == 0) ? bmin[t.z].x : ((dim == 1) ? bmin[t.z].y : bmin[t.z].z);
            const float split = __int_as_float(t.x);
            if (!(lmax <= split && split <= rmin)) ok = false;
        }
        uint32_t q[12];
        for (int c = 0; c < 2; ++c) {
            const int ch = ok ? (c ? t.z : t.y) : 0;
            const float lo[3] = { bmin[ch].x, bmin[ch].y, bmin[ch].z }, hi[3] = { bmax[ch].x, bmax[ch].y, bmax[ch].z };
            for (int a = 0; a < 3; ++a) {
                const float qmin = __uint_as_float(info[2 + a]), qs = __uint_as_float(info[5 + a]);
                float fl = floorf((lo[a] - qmin) / qs) - 1.0f;
                uint32_t ql = fl > 0.0f ? (fl < 65535.0f ? (uint32_t)fl : 65535u) : 0u;
                while (ql > 0 && !(nn_deq(ql, qmin, qs) <= lo[a])) --ql;
                if (!(nn_deq(ql, qmin, qs) <= lo[a])) ok = false;
                float fh = ceilf((hi[a] - qmin) / qs) + 1.0f;
                uint32_t qh = fh > 0.0f ? (fh < 65535.0f ? (uint32_t)fh : 65535u) : 0u;
                while (qh < 65535u && !(nn_deq(qh, qmin, qs) >= hi[a])) ++qh;
                if (!(nn_deq(qh, qmin, qs) >= hi[a])) ok = false;
                q[c * 6 + a] = ql; q[c * 6 + 3 + a] = qh;
            }
        }
        for (int k = 0; k < 6; ++k) w[2 + k] = q[2 * k] | (q[2 * k + 1] << 16);
    }
    rec32[(size_t)i * 2] = make_uint4(w[0], w[1], w[2], w[3]);
    rec32[(size_t)i * 2 + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    desc[i] = make_uint2(w[0], w[1]);
    if (!ok) info[1] = 0u;                                       // any node that does not fit: the 64-byte records are used instead
}

// ---- wide records (query_nn_wide) ---------------------------------------------------------------------------------------------
// Box of a binary node as 6 uint16 in the root-box frame, rounded outwards and checked with the dequantisation the query uses.
__device__ __forceinline__ bool wide_quant_box(const float4 lo4, const float4 hi4, const uint32_t *__restrict__ info, uint32_t (&u)[3])
{
    const float lo[3] = { lo4.x, lo4.y, lo4.z }, hi[3] = { hi4.x, hi4.y, hi4.z };
    uint32_t q[6];
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
        const float qmin = __uint_as_float(info[2 + a]), qs = __uint_as_float(info[5 + a]);
        const float fl = floorf((lo[a] - qmin) / qs) - 1.0f;
        uint32_t ql = fl > 0.0f ? (fl < 65535.0f ? (uint32_t)fl : 65535u) : 0u;
        while (ql > 0 && !(nn_deq_fma(ql, qmin, qs) <= lo[a])) --ql;
        if (!(nn_deq_fma(ql, qmin, qs) <= lo[a])) ok = false;
        const float fh = ceilf((hi[a] - qmin) / qs) + 1.0f;
        uint32_t qh = fh > 0.0f ? (fh < 65535.0f ? (uint32_t)fh : 65535u) : 0u;
        while (qh < 65535u && !(nn_deq_fma(qh, qmin, qs) >= hi[a])) ++qh;
        if (!(nn_deq_fma(qh, qmin, qs) >= hi[a])) ok = false;
        q[a] = ql; q[3 + a] = qh;
    }
    u[0] = q[0] | (q[1] << 16); u[1] = q[2] | (q[3] << 16); u[2] = q[4] | (q[5] << 16);
    return ok;
}
// One workgroup builds the wide nodes level by level (a scene is prepared once; 6 287 binary nodes give 5 levels).  Per level:
// (A) every wide node opens its binary root into a frontier of up to eight descendants -- repeatedly the internal frontier node
// with the longest box diagonal -- and counts the internal ones, (B) an exclusive scan of those counts numbers the next level's
// wide nodes (deterministic: a wide node's index does not depend on timing), (C) the records are written.  `wq[k]` = binary root
// of wide node k, `cnt[k]` = scan scratch.  A tree whose links are out of order (child index <= parent index) clears info[8].
constexpr uint32_t kWideBuildThreads = 1024;
__global__ __launch_bounds__(kWideBuildThreads) void nn_wide_build_kernel(const int4 *__restrict__ topo, const float4 *__restrict__ bmin,
                                                                           const float4 *__restrict__ bmax, uint32_t n_nodes, uint4 *__restrict__ wide,
                                                                           uint32_t cap_wide, uint32_t *__restrict__ wq, uint32_t *__restrict__ cnt,
                                                                           uint32_t n_points, uint32_t *__restrict__ info)
{
    __shared__ uint32_t part[kWideBuildThreads];
    __shared__ uint32_t s_bad;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { wq[0] = 0u; s_bad = (info[1] == 1u) ? 0u : 1u; }
    __syncthreads();
    uint32_t begin = 0, end = 1;
    for (int level = 0; level < 64 && begin < end; ++level) {
        // s_bad is read into a register BETWEEN two barriers (the one that ended the previous level and this one): stage (A) below sets it,
        // and a thread still evaluating the loop condition while a faster one is already in (A) would leave the loop alone -- a divergent barrier
        const bool bad_so_far = s_bad != 0u;
        __syncthreads();
        if (bad_so_far) break;
        // (A) frontiers; the words of the record hold binary node ids for now
        for (uint32_t k = begin + tid; k < end; k += kWideBuildThreads) {
            uint32_t fr[8]; float fsz[8]; int nf = 1;
            fr[0] = wq[k];
            auto size_of = [&](uint32_t n) -> float {              // -1 for a leaf (never opened)
                if (topo[n].z < 0) return -1.0f;
                const float4 a = bmin[n], b = bmax[n];
                return (b.x - a.x) * (b.x - a.x) + (b.y - a.y) * (b.y - a.y) + (b.z - a.z) * (b.z - a.z);
            };
            fsz[0] = (topo[fr[0]].z < 0) ? -1.0f : FLT_MAX;        // the root of a wide node is always opened (unless the whole tree is one leaf)
            while (nf < 8) {
                int pick = -1;
                for (int i = 0; i < nf; ++i) if (fsz[i] >= 0.0f && (pick < 0 || fsz[i] > fsz[pick])) pick = i;
                if (pick < 0) break;
                const uint32_t n = fr[pick];
                const int4 t = topo[n];
                if (!((uint32_t)t.y < n_nodes && (uint32_t)t.z < n_nodes && (uint32_t)t.y > n && (uint32_t)t.z > n)) { s_bad = 1u; fsz[pick] = -1.0f; continue; }
                fr[pick] = (uint32_t)t.y; fsz[pick] = size_of((uint32_t)t.y);
                fr[nf] = (uint32_t)t.z;   fsz[nf] = size_of((uint32_t)t.z);
                ++nf;
            }
            uint32_t internal = 0;
            for (int c = 0; c < 8; ++c) {                            // slot c = {box, reference}; internal children carry their BINARY id until (C)
                uint32_t u[3] = { 0u, 0u, 0u }, ref = kWideEmpty;
                if (c < nf) {
                    const uint32_t n = fr[c];
                    const int4 t = topo[n];
                    if (!wide_quant_box(bmin[n], bmax[n], info, u)) s_bad = 1u;
                    if (t.z < 0) {
                        const int lo = t.x, hi = t.y;
                        if (lo >= 0 && hi > lo && (uint32_t)(hi - lo) <= kWideMaxLeafPoints && (uint32_t)lo <= kWideFirstMask && (uint32_t)hi <= n_points)
                            ref = kWideLeaf | ((uint32_t)(hi - lo) << 27) | (uint32_t)lo;
                        else if (hi != lo) s_bad = 1u;               // (an empty leaf stays an empty slot)
                    } else { ref = n; ++internal; if (n >= 0x7fffffffu) s_bad = 1u; }
                }
                wide[(size_t)k * 8 + c] = make_uint4(u[0], u[1], u[2], ref);
            }
            cnt[k] = internal;
        }
        __syncthreads();
        // (B) exclusive scan of cnt[begin, end): contiguous chunk per thread, Hillis-Steele over the chunk sums
        const uint32_t span = end - begin, chunk = (span + kWideBuildThreads - 1) / kWideBuildThreads;
        const uint32_t c0 = begin + tid * chunk, c1 = (c0 + chunk < end) ? c0 + chunk : end;
        uint32_t sum = 0;
        for (uint32_t k = c0; k < c1 && k >= begin; ++k) sum += cnt[k];
        part[tid] = sum;
        __syncthreads();
        for (uint32_t off = 1; off < kWideBuildThreads; off <<= 1) {
            const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const uint32_t total = part[kWideBuildThreads - 1];
        uint32_t run = part[tid] - sum;                              // exclusive prefix of this thread's chunk
        for (uint32_t k = c0; k < c1 && k >= begin; ++k) { const uint32_t v = cnt[k]; cnt[k] = run; run += v; }
        if (tid == 0 && (size_t)end + total > (size_t)cap_wide) s_bad = 1u;
        __syncthreads();
        if (s_bad) break;
        // (C) number the internal children: wide node k's go to end + cnt[k] ...
        for (uint32_t k = begin + tid; k < end; k += kWideBuildThreads) {
            uint32_t next = end + cnt[k];
            for (int c = 0; c < 8; ++c) {
                uint4 r = wide[(size_t)k * 8 + c];
                if (r.w != kWideEmpty && !(r.w & kWideLeaf)) { wq[next] = r.w; r.w = next; ++next; wide[(size_t)k * 8 + c] = r; }
            }
        }
        __syncthreads();
        begin = end; end = end + total;
    }
    __syncthreads();
    if (tid == 0) { info[8] = (s_bad || begin < end) ? 0u : 1u; info[9] = end; }
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
    uint32_t h = 0;
    const uint32_t *arr[3] = { a, b, c };
    const unsigned long long len[3] = { na, nb, nc };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!arr[k] || len[k] == 0) continue;
        // sample s of 4096 sits in stripe s of the array, at a hashed offset inside it (32-bit arithmetic: __umulhi maps a hash onto a range)
        const uint32_t n = len[k] > 0xffffffffull ? 0xffffffffu : (uint32_t)len[k];
        const uint32_t stripe = n / 4096u;                            // 0 for short arrays: every word is visited, wrapping around
        uint32_t w[16];
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) {
            const uint32_t s = threadIdx.x * 16u + i;
            const uint32_t pos = stripe ? s * stripe + __umulhi(s * 2654435761u + 0x9e3779b9u, stripe) : (n >= 4096u ? s : s % n);
            w[i] = arr[k][pos] ^ pos;
        }
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) h += w[i] * 2654435761u + (uint32_t)k;      // a sum: the order of the lanes does not matter
    }
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

// ================================================================================================
//  launchers
// ================================================================================================
// Dynamic LDS beyond 64 KiB is an opt-in PER DEVICE (hipFuncSetAttribute applies to the current device): one flag per kernel slot and
// device, so that a process driving several GPUs (one host thread each) opts in on every one of them.
static bool lds_opt_in(const void *fn, int slot, uint32_t bytes)
{
    static std::atomic<uint64_t> done[4];                         // bit = device ordinal (<= 64 devices), one word per kernel slot
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (done[slot].load(std::memory_order_acquire) & (1ull << dev)) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    done[slot].fetch_or(1ull << dev, std::memory_order_release);
    return true;
}
static inline uint32_t cap_grid(size_t want) { return (uint32_t)(want < 1 ? 1 : (want > 8192 ? 8192 : want)); }

hipError_t launch_fill_i32(int32_t *dst, size_t n, int32_t v, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(cap_grid((n + 1023) / 1024)), dim3(256), 0, s, dst, n, v);
    return hipGetLastError();
}

hipError_t launch_max2zero(int32_t *depth, size_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(max2zero_kernel, dim3(cap_grid((n + 1023) / 1024)), dim3(256), 0, s, depth, n);
    return hipGetLastError();
}

// hypotheses a raster workgroup walks with its 256 triangles: 1 while the mesh is L2-resident and the grid is what fills the chip, more once
// the mesh has to be streamed (PR_RASTER_RUN overrides: experiments)
static uint32_t raster_pose_run(uint32_t n_tris, uint32_t n_poses)
{
    static const int env_run = getenv("PR_RASTER_RUN") ? atoi(getenv("PR_RASTER_RUN")) : 0;
    uint32_t run = 1;
    if (env_run > 0) run = (uint32_t)env_run;
    else if ((size_t)n_tris * sizeof(pr_triangle) > ((size_t)3 << 20)) run = 8;      // beyond what one XCD's 4 MiB L2 keeps next to the depth images
    if (run > n_poses) run = n_poses ? n_poses : 1u;
    return run;
}
hipError_t launch_raster(const pr_triangle *tris, uint32_t n_tris, const pr_mat4 *poses_dev, uint32_t n_poses,
                         int32_t *depth, uint32_t width, uint32_t height, const pr_mat4 &proj, pr_roi roi,
                         uint32_t rw, uint32_t rh, hipStream_t s)
{
    if (n_tris == 0 || n_poses == 0) return hipSuccess;
    // grid.y is limited to 65535: split very large batches
    for (uint32_t p0 = 0; p0 < n_poses; p0 += 32768) {
        const uint32_t np = (n_poses - p0 < 32768) ? (n_poses - p0) : 32768;
        const uint32_t run = raster_pose_run(n_tris, np);
        hipLaunchKernelGGL(raster_kernel, dim3((n_tris + 255) / 256, (np + run - 1) / run), dim3(256), 0, s, tris, n_tris, poses_dev + p0,
                           depth + (size_t)p0 * rw * rh, width, height, proj, roi, rw, rh, (const int4 *)nullptr, np, run);
    }
    return hipGetLastError();
}

// keys: 6 x uint32 scratch; aabb_out (device, 6 floats) and/or flag_out (device-visible word: 1 when the box differs from `expect`)
hipError_t launch_model_aabb(const pr_triangle *tris, uint32_t n_tris, uint32_t *keys, float *aabb_out, const float *expect, uint32_t *flag_out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(keys, 0xff, 6 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (n_tris > 0) {
        const uint32_t blocks = (n_tris * 3u + 2047u) / 2048u;           // >= 8 vertices per lane
        hipLaunchKernelGGL(model_aabb_kernel, dim3(blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks)), dim3(256), 0, s, tris, n_tris, keys);
    }
    AabbExpected ex{};
    if (expect) for (int a = 0; a < 6; ++a) ex.v[a] = expect[a];
    hipLaunchKernelGGL(model_aabb_finish_kernel, dim3(1), dim3(64), 0, s, keys, aabb_out, ex, flag_out);
    return hipGetLastError();
}

// fused-path render: boxes, LDS-band raster (depth + row counts), row scan.  Leaves depth valid only
// inside each hypothesis' box; row_count/row_off/counts as launch_depth2cloud(emit=false) would.
hipError_t launch_render_bands(const pr_triangle *tris, uint32_t n_tris, const pr_mat4 *poses_dev, uint32_t n_poses, const float *aabb,
                               int4 *bbox, int32_t *depth, uint32_t *row_count, uint32_t *row_off, uint32_t *counts,
                               uint32_t width, uint32_t height, const pr_mat4 &proj, pr_roi roi, uint32_t n_cus, hipStream_t s)
{
    if (n_poses == 0) return hipSuccess;
    const uint32_t cap_px = 36864;                                  // 144 KiB of the 160 KiB LDS
    if (!lds_opt_in(reinterpret_cast<const void *>(raster_band_kernel), 0, (uint32_t)(cap_px * sizeof(int32_t)))) return hipErrorInvalidValue;   // per device
    hipLaunchKernelGGL(pose_bbox_kernel, dim3((n_poses + 255) / 256), dim3(256), 0, s, aabb, poses_dev, n_poses, proj, width, height, roi, bbox);
    hipError_t e = hipMemsetAsync(row_count, 0, sizeof(uint32_t) * (size_t)n_poses * height, s);
    if (e != hipSuccess) return e;
    const uint32_t rows_min = cap_px / width > 0 ? cap_px / width : 1;
    const uint32_t max_bands = (height + rows_min - 1) / rows_min;
    const uint32_t items = n_poses * max_bands;
    const uint32_t grid = items < n_cus ? items : n_cus;
    hipLaunchKernelGGL(raster_band_kernel, dim3(grid), dim3(1024), cap_px * sizeof(int32_t), s, tris, n_tris, poses_dev, n_poses, bbox, depth,
                       row_count, width, height, proj, cap_px, max_bands);
    hipLaunchKernelGGL(d2c_scan_kernel, dim3(n_poses), dim3(256), 0, s, row_count, height, row_off, counts);
    return hipGetLastError();
}

// fused-path render, reference scheme (global int32 atomicMin) but only inside each hypothesis' box
hipError_t launch_render_boxes(const pr_triangle *tris, uint32_t n_tris, const pr_mat4 *poses_dev, uint32_t n_poses, const float *aabb,
                               int4 *bbox, int32_t *depth, uint32_t *row_count, uint32_t *row_off, uint32_t *counts,
                               uint32_t width, uint32_t height, const pr_mat4 &proj, pr_roi roi, hipStream_t s, bool compute_boxes,
                               PoseMeta *meta, DevIcpState *st, uint32_t *arrive, uint32_t cloud_stride)
{
    if (n_poses == 0) return hipSuccess;
    if (compute_boxes)
        hipLaunchKernelGGL(pose_bbox_kernel, dim3((n_poses + 255) / 256), dim3(256), 0, s, aabb, poses_dev, n_poses, proj, width, height, roi, bbox);
    const pr_roi none{ 0, 0, 0, 0 };
    for (uint32_t p0 = 0; p0 < n_poses; p0 += 32768) {
        const uint32_t np = (n_poses - p0 < 32768) ? (n_poses - p0) : 32768;
        const size_t off = (size_t)p0 * width * height;
        hipLaunchKernelGGL(fill_box_kernel, dim3((height + kBoxRowsPerBlock - 1) / kBoxRowsPerBlock, np), dim3(256), 0, s, depth + off, bbox + p0, width, height);
        if (n_tris > 0)                                          // an empty mesh renders nothing: every cloud is empty
        { const uint32_t run = raster_pose_run(n_tris, np);
        hipLaunchKernelGGL(raster_kernel, dim3((n_tris + 255) / 256, (np + run - 1) / run), dim3(256), 0, s, tris, n_tris, poses_dev + p0, depth + off,
                           width, height, proj, none, width, height, (const int4 *)(bbox + p0), np, run); }
        hipLaunchKernelGGL(count_box_kernel, dim3((height + kBoxRowsPerBlock - 1) / kBoxRowsPerBlock, np), dim3(256), 0, s, depth + off, bbox + p0, width, height,
                           row_count + (size_t)p0 * height);
    }
    if (meta) hipLaunchKernelGGL(d2c_scan_init_kernel, dim3(n_poses), dim3(256), 0, s, row_count, height, row_off, counts, meta, st, arrive, cloud_stride);
    else hipLaunchKernelGGL(d2c_scan_kernel, dim3(n_poses), dim3(256), 0, s, row_count, height, row_off, counts);
    return hipGetLastError();
}

hipError_t launch_emit_box(const int32_t *depth, uint32_t n_poses, uint32_t width, uint32_t height, const int4 *bbox, float fx, float fy,
                           float cx, float cy, const uint32_t *row_count, const uint32_t *row_off, pr_vec3 *cloud, size_t cloud_stride,
                           hipStream_t s)
{
    if (n_poses == 0) return hipSuccess;
    for (uint32_t i0 = 0; i0 < n_poses; i0 += 32768) {
        const uint32_t ni = (n_poses - i0 < 32768) ? (n_poses - i0) : 32768;
        hipLaunchKernelGGL(d2c_emit_box_kernel, dim3((height + kBoxRowsPerBlock - 1) / kBoxRowsPerBlock, ni), dim3(256), 0, s, depth + (size_t)i0 * width * height, width, height,
                           bbox + i0, fx, fy, cx, cy, row_count + (size_t)i0 * height, row_off + (size_t)i0 * height,
                           cloud + (size_t)i0 * cloud_stride, cloud_stride);
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

template <class Scene, bool kNN, int kStack = 0>
static hipError_t launch_pass(const IcpBatch &b, const Scene &sc, uint32_t n_poses, size_t lds_bytes, hipStream_t s)
{
    if (n_poses == 0 || b.nblk == 0) return hipSuccess;
    for (uint32_t p0 = 0; p0 < n_poses; p0 += 32768) {
        const uint32_t np = (n_poses - p0 < 32768) ? (n_poses - p0) : 32768;
        IcpBatch bb = b;
        bb.meta += p0;
        bb.partial += (size_t)p0 * b.nblk * kAccStride;
        if (bb.st) bb.st += p0;                                  // everything indexed by the hypothesis moves with the piece
        if (bb.arrive) bb.arrive += p0;
        if (bb.sums_out) bb.sums_out += (size_t)p0 * kAccStride;
        if (bb.nn_qcount) bb.nn_qcount += kQCountStride * (size_t)p0;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(icp_pass_kernel<Scene, kNN, kStack>), dim3(b.grid_x ? b.grid_x : b.nblk, np), dim3(kBlockThreads), lds_bytes, s, bb, sc);
    }
    return hipGetLastError();
}
hipError_t launch_icp_pass_proj_aos(const IcpBatch &b, const SceneProjAoS &sc, uint32_t n_poses, hipStream_t s)
{ return launch_pass<SceneProjAoS, false>(b, sc, n_poses, 0, s); }
hipError_t launch_icp_pass_proj_packed(const IcpBatch &b, const SceneProjPacked &sc, uint32_t n_poses, hipStream_t s)
{ return launch_pass<SceneProjPacked, false>(b, sc, n_poses, 0, s); }
hipError_t launch_icp_pass_nn(const IcpBatch &b, const SceneNNDev &sc, uint32_t n_poses, hipStream_t s)
{
    if (sc.stack_depth == 16 && sc.rec32) return launch_pass<SceneNNDev, true, 16 + 0x100>(b, sc, n_poses, (size_t)16 * kBlockThreads * 8, s);
    if (sc.stack_depth == 24 && sc.rec32) return launch_pass<SceneNNDev, true, 24 + 0x100>(b, sc, n_poses, (size_t)24 * kBlockThreads * 8, s);
    if (sc.stack_depth == 16) return launch_pass<SceneNNDev, true, 16>(b, sc, n_poses, (size_t)16 * kBlockThreads * 8 + (size_t)sc.lds_nodes * 64, s);
    if (sc.stack_depth == 24) return launch_pass<SceneNNDev, true, 24>(b, sc, n_poses, (size_t)24 * kBlockThreads * 8 + (size_t)sc.lds_nodes * 64, s);
    return launch_pass<SceneNNDev, true, 0>(b, sc, n_poses, (size_t)sc.lds_nodes * sizeof(int4), s);
}

hipError_t launch_icp_pass_nn_winners(const IcpBatch &b, const SceneNNWinners &sc, uint32_t n_poses, hipStream_t s)
{ return launch_pass<SceneNNWinners, true, -1>(b, sc, n_poses, 0, s); }

hipError_t launch_nn_search(const IcpBatch &b, const SceneNNDev &sc, uint32_t n_poses, uint32_t max_points, uint32_t run, hipStream_t s, hipEvent_t *marks)
{
    if (n_poses == 0 || max_points == 0) return hipSuccess;
    if (!sc.rec32 || (sc.stack_depth != 16 && sc.stack_depth != 24) || !b.nn_prev || !b.nn_slack || !b.nn_queue || !b.nn_queue2 || !b.nn_qcount) return hipErrorInvalidValue;
    if (run == 0) run = 1;
    if (run > 8) run = 8;
    const uint32_t per_block = kBlockThreads * run;
    const uint32_t gx = (max_points + per_block - 1) / per_block;
#ifndef PR_TREE_GX
#define PR_TREE_GX 8
#endif
    // workgroups per hypothesis walking its queue: 8 when there are hundreds of hypotheses, more for a handful (a single cloud -- the
    // reference's own ICP() call -- would otherwise run its tree searches on 8 CUs of 256)
    uint32_t want = (uint32_t)PR_TREE_GX;
    if (n_poses * want < 1024u) want = (1024u + n_poses - 1u) / n_poses;
    const uint32_t tree_gx = gx < want ? gx : want;
    for (uint32_t p0 = 0; p0 < n_poses; p0 += 32768) {
        const uint32_t np = (n_poses - p0 < 32768) ? (n_poses - p0) : 32768;
        IcpBatch bb = b;
        bb.meta += p0; bb.nn_qcount += kQCountStride * p0;
        hipLaunchKernelGGL(nn_search_kernel, dim3(gx, np), dim3(kBlockThreads), 0, s, bb, sc, run);
        if (marks) (void)hipEventRecord(marks[0], s);
        if (!sc.wide && marks) (void)hipEventRecord(marks[1], s);
        if (sc.wide) {
            hipLaunchKernelGGL(nn_bound_kernel, dim3(tree_gx, np), dim3(kBlockThreads), 0, s, bb, sc);
            if (marks) (void)hipEventRecord(marks[1], s);
            // a wavefront of the task walk takes 64 queries at a time -- unless the whole launch has too few to fill the chip that way (a single
            // cloud: 26 k queries = 103 workgroups of 4 x 64): then 16 at a time in four times as many workgroups
            uint32_t qbatch = 64u, walk_gx = tree_gx;
            if ((size_t)np * max_points < (size_t)64 * 4 * 1024) { qbatch = 16u; walk_gx = (max_points + 4u * qbatch - 1u) / (4u * qbatch); if (walk_gx * np > 4096u) walk_gx = (4096u + np - 1u) / np; if (walk_gx < tree_gx) walk_gx = tree_gx; }
            hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_tree_wide_kernel<PR_WIDE_LANES>), dim3(walk_gx, np), dim3(kBlockThreads), 0, s, bb, sc, qbatch);
        }
        else if (sc.stack_depth == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_tree_kernel<16 + 0x100>), dim3(tree_gx, np), dim3(kBlockThreads), (size_t)16 * kBlockThreads * 8, s, bb, sc);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_tree_kernel<24 + 0x100>), dim3(tree_gx, np), dim3(kBlockThreads), (size_t)24 * kBlockThreads * 8, s, bb, sc);
        if (marks) (void)hipEventRecord(marks[2], s);
    }
    return hipGetLastError();
}

hipError_t launch_build_nn_grid(const pr_vec3 *pcd, uint32_t n_points, uint32_t gw, uint32_t gh, float fx, float fy, float cx, float cy,
                                int32_t *cell_idx, float4 *grid, uint32_t *info, hipStream_t s)
{
    const uint32_t cells = gw * gh;
    if (cells == 0 || n_points == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(cell_idx, 0xff, sizeof(int32_t) * cells, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(info, 1, sizeof(uint32_t), s);                 // non-zero: usable so far
    if (e != hipSuccess) return e;
    SceneNNDev g{};
    g.gw = gw; g.gh = gh; g.gfx = fx; g.gfy = fy; g.gcx = cx; g.gcy = cy;
    hipLaunchKernelGGL(nn_grid_claim_kernel, dim3((n_points + 255) / 256), dim3(256), 0, s, pcd, n_points, g, cell_idx, info);
    hipLaunchKernelGGL(nn_grid_fill_kernel, dim3((cells + 255) / 256), dim3(256), 0, s, pcd, cell_idx, cells, grid);
    int fw = (int)gw, fh = (int)gh;
    float4 *fine = grid;
    for (int level = 0; level < 3; ++level) {
        const int cw = (fw + 3) / 4, ch = (fh + 3) / 4;
        float4 *coarse = fine + (size_t)fw * fh;
        hipLaunchKernelGGL(nn_grid_coarsen_kernel, dim3((uint32_t)((cw * ch + 255) / 256)), dim3(256), 0, s, fine, fw, fh, coarse, cw, ch);
        fine = coarse; fw = cw; fh = ch;
    }
    return hipGetLastError();
}
size_t nn_grid_cells(uint32_t gw, uint32_t gh)
{
    size_t total = (size_t)gw * gh, w = gw, h = gh;
    for (int level = 0; level < 3; ++level) { w = (w + 3) / 4; h = (h + 3) / 4; total += w * h; }
    return total;
}

template <class Scene, bool kNN, int kStack>
static hipError_t launch_flow_t(const FlowArgs &a, const Scene &sc, size_t lds_bytes, uint32_t n_cus, hipStream_t s, uint32_t *grid_out)
{
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(icp_flow_kernel<Scene, kNN, kStack>),
                                                                (int)kBlockThreads, lds_bytes);
    if (e != hipSuccess) return e;
    if (per_cu < 1) return hipErrorInvalidValue;
    if (per_cu > 8) per_cu = 8;
    // every workgroup must be resident (they wait on each other): grid <= what one wave of dispatch can hold
    uint32_t grid = n_cus * (uint32_t)per_cu;
    if (grid > a.n_vbs) grid = a.n_vbs;
    if (grid_out) *grid_out = grid;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(icp_flow_kernel<Scene, kNN, kStack>), dim3(grid), dim3(kBlockThreads), lds_bytes, s, a, sc);
    return hipGetLastError();
}
hipError_t launch_icp_flow_proj_aos(const FlowArgs &a, const SceneProjAoS &sc, uint32_t n_cus, hipStream_t s, uint32_t *grid_out)
{ return launch_flow_t<SceneProjAoS, false, 0>(a, sc, 0, n_cus, s, grid_out); }
hipError_t launch_icp_flow_proj_packed(const FlowArgs &a, const SceneProjPacked &sc, uint32_t n_cus, hipStream_t s, uint32_t *grid_out)
{ return launch_flow_t<SceneProjPacked, false, 0>(a, sc, 0, n_cus, s, grid_out); }
hipError_t launch_icp_flow_nn(const FlowArgs &a, const SceneNNDev &sc, uint32_t n_cus, hipStream_t s, uint32_t *grid_out)
{
    if (sc.stack_depth == 16) return launch_flow_t<SceneNNDev, true, 16>(a, sc, (size_t)16 * kBlockThreads * 8 + (size_t)sc.lds_nodes * 64, n_cus, s, grid_out);
    if (sc.stack_depth == 24) return launch_flow_t<SceneNNDev, true, 24>(a, sc, (size_t)24 * kBlockThreads * 8 + (size_t)sc.lds_nodes * 64, n_cus, s, grid_out);
    return launch_flow_t<SceneNNDev, true, 0>(a, sc, (size_t)sc.lds_nodes * sizeof(int4), n_cus, s, grid_out);
}

hipError_t launch_icp_finalize(const float *partial, const PoseMeta *meta, uint32_t nblk,
                               uint32_t steps, float *sums, uint32_t n_poses, hipStream_t s)
{
    if (n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(icp_finalize_kernel, dim3(n_poses), dim3(64), 0, s, partial, meta, nblk, steps * kPointsPerStep, sums);
    return hipGetLastError();
}
hipError_t launch_icp_finalize_solve(const float *partial, PoseMeta *meta, uint32_t nblk,
                                     uint32_t steps, DevIcpState *st, pr_criteria crit, uint32_t iter,
                                     uint32_t n_poses, hipStream_t s)
{
    if (n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(icp_finalize_solve_kernel, dim3(n_poses), dim3(64), 0, s, partial, meta, nblk,
                       steps * kPointsPerStep, st, crit, iter);
    return hipGetLastError();
}
hipError_t launch_pack_export(const DevIcpState *st, pr_result *out, const uint32_t *counts, uint32_t *host_counts, pr_result *host_results,
                              uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_export_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st, out, counts, host_counts, host_results, n);
    return hipGetLastError();
}
hipError_t launch_stage_words(const void *src_host_mapped, void *dst, size_t bytes, hipStream_t s)
{
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    if (n16 == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_words_kernel, dim3((n16 + 255) / 256), dim3(256), 0, s, static_cast<const uint4 *>(src_host_mapped), static_cast<uint4 *>(dst), n16);
    return hipGetLastError();
}
hipError_t launch_pack_results(const DevIcpState *st, pr_result *out, uint32_t n_poses, hipStream_t s)
{
    if (n_poses == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_results_kernel, dim3((n_poses + 255) / 256), dim3(256), 0, s, st, out, n_poses);
    return hipGetLastError();
}

hipError_t launch_kd_init(pr_kdnode *nodes, uint32_t cap, int *idx, uint32_t n, uint32_t *ctrl, hipStream_t s)
{
    const uint32_t m = cap > n ? cap : n;
    hipLaunchKernelGGL(kd_init_kernel, dim3((m + 255) / 256), dim3(256), 0, s, nodes, cap, idx, n);
    const uint32_t init[4] = { 0, 1, 1, 1 };
    return hipMemcpyAsync(ctrl, init, sizeof init, hipMemcpyHostToDevice, s);
}
hipError_t launch_kd_level(pr_kdnode *nodes, uint32_t *ctrl, int max_leaf, int *child_of, uint32_t cap, uint32_t level_nodes,
                           const pr_vec3 *pcd, int *idx, int *scratch, bool plan_only, hipStream_t s)
{
    if (plan_only) { hipLaunchKernelGGL(kd_level_plan_kernel, dim3(1), dim3(256), 0, s, nodes, ctrl, max_leaf, child_of, cap); return hipGetLastError(); }
    if (level_nodes) hipLaunchKernelGGL(kd_level_split_kernel, dim3(level_nodes), dim3(256), 0, s, nodes, ctrl, child_of, pcd, idx, scratch);
    hipLaunchKernelGGL(kd_advance_kernel, dim3(1), dim3(64), 0, s, ctrl);
    return hipGetLastError();
}
hipError_t launch_kd_permute(const pr_vec3 *pcd, const pr_vec3 *nrm, const int *idx, uint32_t n, pr_vec3 *pcd_out, pr_vec3 *nrm_out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kd_permute_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pcd, nrm, idx, n, pcd_out, nrm_out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_nn_gather(const T *depth, uint32_t W, uint32_t H, float fx, float fy, float cx, float cy, const pr_vec3 *normal_full,
                            uint32_t *row_count, uint32_t *row_off, uint32_t *count, pr_vec3 *pcd, pr_vec3 *nrm, bool emit, hipStream_t s)
{
    if (!emit) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(nn_gather_count_kernel<T>), dim3((H + 3) / 4), dim3(256), 0, s, depth, W, H, row_count);
        hipLaunchKernelGGL(d2c_scan_kernel, dim3(1), dim3(256), 0, s, row_count, H, row_off, count);
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
hipError_t launch_build_nn_accel(const pr_kdnode *nodes, uint32_t n_nodes, const pr_vec3 *pcd, uint32_t n_points,
                                 int4 *topo, float4 *bmin, float4 *bmax, float4 *pts, float4 *rec, uint4 *rec32, uint2 *desc, uint32_t *info, hipStream_t s,
                                 uint4 *wide, uint32_t *wide_scratch)
{
    const uint32_t m = (n_nodes > n_points) ? n_nodes : n_points;
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(nn_accel_kernel, dim3((m + 255) / 256), dim3(256), 0, s, nodes, n_nodes, pcd, n_points, topo, bmin, bmax, pts);
    hipError_t e = hipMemsetAsync(info, 0, 16 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nn_records_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, topo, bmin, bmax, n_nodes, rec, info);
    hipLaunchKernelGGL(nn_frame_kernel, dim3(1), dim3(64), 0, s, bmin, bmax, info);
    hipLaunchKernelGGL(nn_records32_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, topo, bmin, bmax, n_nodes, rec32, desc, info);
    if (wide && wide_scratch) {
        const uint32_t cap = (uint32_t)nn_wide_capacity(n_nodes);
        hipLaunchKernelGGL(nn_wide_build_kernel, dim3(1), dim3(kWideBuildThreads), 0, s, topo, bmin, bmax, n_nodes, wide, cap, wide_scratch, wide_scratch + cap, n_points, info);
    }
    return hipGetLastError();
}

}  // namespace prk
