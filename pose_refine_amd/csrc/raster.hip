// raster.hip -- render_triangle + rasterization (cuda_renderer/renderer.cu:83-187) for a batch of hypotheses: clear, per-hypothesis pixel boxes, triangle raster with int32 atomicMin, row counts
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#include "pr_launch.h"

namespace prk {

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
// the same test with the triangle's two edge vectors from vertex 0 formed once per triangle (e1 = p1 - p0, e2 = p2 - p0: the very subtractions area2 makes per
// pixel, so every product and difference below is the same float as in tri_fragment)
__device__ __forceinline__ bool tri_fragment_edges(float px0, float py0, float e1x, float e1y, float e2x, float e2y, float base_inv, int x, int y,
                                                   float &alpha, float &beta, float &gamma)
{
    const float dx = (float)x - px0, dy = (float)y - py0;
    beta  = 0.5f * (e2x * dy - dx * e2y) * base_inv;               // area2(p0, f, p2) * base_inv
    gamma = 0.5f * (dx * e1y - e1x * dy) * base_inv;               // area2(p0, p1, f) * base_inv
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

    w[0][lane] = t.px[0]; w[1][lane] = t.py[0]; w[2][lane] = t.px[1] - t.px[0]; w[3][lane] = t.py[1] - t.py[0]; w[4][lane] = t.px[2] - t.px[0]; w[5][lane] = t.py[2] - t.py[0];
    w[6][lane] = t.w3[0]; w[7][lane] = t.w3[1]; w[8][lane] = t.w3[2]; w[9][lane] = t.base_inv;
    w[10][lane] = __int_as_float(t.x0); w[11][lane] = __int_as_float(t.y0); w[12][lane] = __int_as_float(t.nx > 0 ? t.nx : 1);
    w[13][lane] = __int_as_float(excl);
    w[14][lane] = __builtin_amdgcn_rcpf((float)(t.nx > 0 ? t.nx : 1));
    // wavefronts only touch their own slice; a wave-level fence is enough for LDS ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // The owner of candidate c = the last triangle whose run starts at or before c.  Round 6: instead of a six-step search over the scanned offsets (six
    // dependent LDS reads per candidate), the triangles whose runs start inside the current window of 64 candidates mark their first slot with lane + 1
    // and an inclusive maximum over the lanes (four row shifts + two row broadcasts, no LDS) carries each mark to the end of its run; a run that began in
    // an earlier window arrives through `carry`, the owner of that window's last candidate.
    uint32_t *mark = queue + 128;
    int carry = 0;
    int qn = 0;                                                    // wave-uniform fill level, < 64 between iterations
    auto drain = [&](int first, int count) {
        if ((int)lane < count) {
            const uint32_t e = queue[first + lane];
            const int o = (int)(e >> 26), x = (int)((e >> 13) & 0x1fffu), y = (int)(e & 0x1fffu);
            float alpha, beta, gamma;
            (void)tri_fragment_edges(w[0][o], w[1][o], w[2][o], w[3][o], w[4][o], w[5][o], w[9][o], x, y, alpha, beta, gamma);
            sink(x, y, fragment_depth(alpha, beta, gamma, w[6][o], w[7][o], w[8][o]));
        }
    };
    for (int c0 = 0; c0 < total; c0 += 64) {
        const int c = c0 + (int)lane;
        bool pass = false;
        uint32_t entry = 0;
        mark[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (n > 0 && excl >= c0 && excl < c0 + 64) mark[excl - c0] = lane + 1u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int own = (int)mark[lane];
        own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x111, 0xf, 0xf, true));      // row_shr:1
        own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x112, 0xf, 0xf, true));      // row_shr:2
        own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x114, 0xf, 0xf, true));      // row_shr:4
        own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x118, 0xf, 0xf, true));      // row_shr:8
        own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x142, 0xa, 0xf, true));      // row_bcast15 -> rows 1, 3
        own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x143, 0xc, 0xf, true));      // row_bcast31 -> rows 2, 3
        own = max(own, carry);
        carry = __builtin_amdgcn_readlane(own, 63);                 // (only used when another window follows: lane 63 then held a real candidate)
        if (c < total) {
            const int o = own - 1;
            const int k = c - __float_as_int(w[13][o]);
            const int nx = __float_as_int(w[12][o]);
            // k / nx from the owner's reciprocal with a one-step correction; k, nx, q*nx < 2^24 (a frame has < 2^24 pixels),
            // so the 24-bit multiplier (full rate, v_mul_lo_u32 is quarter rate) is exact
            int q = (int)((float)k * w[14][o]);
            if ((int)__umul24((unsigned)q, (unsigned)nx) > k) --q;
            if ((int)__umul24((unsigned)(q + 1), (unsigned)nx) <= k) ++q;
            const int x = __float_as_int(w[10][o]) + (k - (int)__umul24((unsigned)q, (unsigned)nx));
            const int y = __float_as_int(w[11][o]) + q;
            float alpha, beta, gamma;
            pass = tri_fragment_edges(w[0][o], w[1][o], w[2][o], w[3][o], w[4][o], w[5][o], w[9][o], x, y, alpha, beta, gamma);
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
__device__ __forceinline__ uint32_t f32_key(float v) { const uint32_t b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key_f32(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
// The per-batch safety nets, carried by the raster (BatchCheck, pr_internal.h): called by the workgroups of the launch's first hypothesis with
// their 256 triangles in registers.  `red` / `part`: LDS scratch (6 x 4 floats, 4 + 1 words).
__device__ __forceinline__ void raster_batch_check(const float (&tv)[9], bool have, const BatchCheck &chk, float (*red)[6], uint32_t *part)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && chk.fp_expected) {
        uint32_t h = fingerprint_lane(chk.fa, chk.na, chk.fb, chk.nb, chk.fc, chk.nc);
        for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off);
        if (lane == 0) part[wave] = h;
        __syncthreads();
        if (threadIdx.x == 0 && *chk.fp_expected != part[0] + part[1] + part[2] + part[3]) *chk.flag = 1u;
        __syncthreads();
    }
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = have ? fminf(fminf(tv[a], tv[3 + a]), tv[6 + a]) : FLT_MAX;
        hi[a] = have ? fmaxf(fmaxf(tv[a], tv[3 + a]), tv[6 + a]) : -FLT_MAX;
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
    }
    if (lane == 0) for (int a = 0; a < 3; ++a) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = red[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) r = (threadIdx.x < 3) ? fminf(r, red[w][threadIdx.x]) : fmaxf(r, red[w][threadIdx.x]);
        atomicMin(&chk.keys[threadIdx.x], (threadIdx.x < 3) ? f32_key(r) : ~f32_key(r));
        __threadfence();                                             // this workgroup's minima are in place before its ticket is drawn
    }
    __syncthreads();
    if (threadIdx.x == 0) part[4] = (atomicAdd(&chk.keys[6], 1u) == gridDim.x - 1u) ? 1u : 0u;
    __syncthreads();
    if (part[4] != 0u && threadIdx.x == 0) {                         // the last workgroup to arrive: compare, and re-arm the words for the next batch
        __threadfence();
        bool differs = false;
        for (int a = 0; a < 6; ++a) {
            const uint32_t k = atomicExch(&chk.keys[a], 0xffffffffu);
            float val = (a < 3) ? key_f32(k) : key_f32(~k);
            if (k == 0xffffffffu) val = (a < 3) ? FLT_MAX : -FLT_MAX;
            if (!(val == chk.expect.v[a])) differs = true;
        }
        atomicExch(&chk.keys[6], 0u);
        if (differs) *chk.flag = 1u;
    }
    __syncthreads();
}

// A workgroup keeps its 256 triangles in registers and walks `pose_run` consecutive hypotheses with them: a mesh that does not fit
// the L2 (the 1 M-triangle mesh of BASELINE configs[4]: 36 MB) is then streamed from the Infinity Cache / HBM once per pose_run
// hypotheses instead of once per hypothesis -- at 128 hypotheses that stream (4.6 GB, 3.2 TB/s) was what bounded the kernel.
__global__ __launch_bounds__(256) void raster_kernel(const pr_triangle *__restrict__ tris, uint32_t n_tris,
                                                     const pr_mat4 *__restrict__ poses, int32_t *__restrict__ depth,
                                                     uint32_t width, uint32_t height, pr_mat4 proj, pr_roi roi,
                                                     uint32_t rw, uint32_t rh, const int4 *__restrict__ boxes, uint32_t n_poses, uint32_t pose_run,
                                                     const uint32_t *__restrict__ box_off, BatchCheck chk)
{
    __shared__ float sh[4][kSetupWords][64];
    __shared__ uint32_t shq[4][192];                              // per wavefront: 128 words of fragment queue, 64 of run marks (wave_raster)
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
    if (chk.keys && blockIdx.y == 0)                             // (wave-uniform) the asynchronous path's per-batch checks ride on the first hypothesis' workgroups
        raster_batch_check(tv, ti < n_tris, chk, reinterpret_cast<float (*)[6]>(&sh[0][0][0]), &shq[0][0]);
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
        uint32_t pitch = rw;                                     // image row pitch; packed boxes (box_off): the box's own width, origin at its top-left pixel
        if (boxes) {                                             // fused path: the hypothesis' pixel box (already intersected with the caller's ROI,
            const int4 bb = boxes[by];                           // if any); a conservative box clips nothing, an ROI clips like renderer.cu:106-113
            cmin0 = (float)bb.x; cmin1 = (float)bb.y; cmax0 = (float)bb.z; cmax1 = (float)bb.w;
            if (box_off) {
                pitch = (uint32_t)max(bb.z - bb.x + 1, 0);
                // pixel (x, image row r) of the box lives at off + (r - r0) * pitch + (x - bb.x), r0 = height - 1 - bb.w: fold the origin into the base
                img = depth + box_off[by] - ((ptrdiff_t)((int)height - 1 - bb.w) * (ptrdiff_t)pitch + (ptrdiff_t)bb.x);
            }
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
            atomicMin(&img[xw + (size_t)yw * pitch], d);
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
// box_off (fused asynchronous path): the boxes of a sub-batch PACKED one behind the other, each as its own little image (pitch = its width) at
// box_off[hypothesis] ints -- like the clouds (d2c_pack_starts_kernel), and for the same reason: 290 KB boxes 1.2 MB apart make the render depend
// on the frame's size (a frame 17 or 32 rows taller: -3 %).  nullptr: full frames [hypothesis][height][width].
__device__ __forceinline__ int32_t *box_line(int32_t *depth, const uint32_t *box_off, const int4 bb, uint32_t pose, uint32_t row, uint32_t width, uint32_t height)
{
    if (!box_off) return depth + ((size_t)pose * height + row) * width;
    const ptrdiff_t pitch = max(bb.z - bb.x + 1, 0);
    return depth + box_off[pose] + ((ptrdiff_t)row - (ptrdiff_t)((int)height - 1 - bb.w)) * pitch - (ptrdiff_t)bb.x;      // line[x] for bb.x <= x <= bb.z
}
__global__ __launch_bounds__(256) void fill_box_kernel(int32_t *__restrict__ depth, const int4 *__restrict__ bbox, uint32_t width, uint32_t height,
                                                       const uint32_t *__restrict__ box_off)
{
    const uint32_t lane = threadIdx.x & 63;
    const int4 bb = bbox[blockIdx.y];
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t row = blockIdx.x * kBoxRowsPerBlock + (threadIdx.x >> 6) * 4 + r;
        if (row >= height) return;
        const int ry = (int)height - 1 - (int)row;                  // raster row of this image row
        if (ry < bb.y || ry > bb.w) continue;
        int32_t *line = box_line(depth, box_off, bb, blockIdx.y, row, width, height);
        for (int x = bb.x + (int)lane; x <= bb.z; x += 64) line[x] = INT_MAX;
    }
}
__global__ __launch_bounds__(256) void count_box_kernel(const int32_t *__restrict__ depth, const int4 *__restrict__ bbox, uint32_t width,
                                                        uint32_t height, uint32_t *__restrict__ row_count, const uint32_t *__restrict__ box_off)
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
            const int32_t *line = live ? box_line(const_cast<int32_t *>(depth), box_off, bb, blockIdx.y, row, width, height) : depth;
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
                           depth + (size_t)p0 * rw * rh, width, height, proj, roi, rw, rh, (const int4 *)nullptr, np, run, (const uint32_t *)nullptr, BatchCheck{});
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
    { hipError_t e = launch_d2c_scan(row_count, height, row_off, counts, n_poses, s); if (e != hipSuccess) return e; }
    return hipGetLastError();
}

// offsets of the packed boxes from boxes computed on the device (synchronous path; the asynchronous path's host computes both): exclusive scan of
// the areas rounded up to kBoxPack pixels, one workgroup
__global__ __launch_bounds__(256) void box_pack_offsets_kernel(const int4 *__restrict__ bbox, uint32_t n, uint32_t *__restrict__ off)
{
    __shared__ uint32_t buf[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + threadIdx.x;
        uint32_t v = 0;
        if (i < n) { const int4 b = bbox[i]; const uint32_t area = (uint32_t)max(b.z - b.x + 1, 0) * (uint32_t)max(b.w - b.y + 1, 0); v = (area + (kBoxPack - 1u)) / (kBoxPack ? kBoxPack : 1u) * kBoxPack; }
        buf[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 256; o <<= 1) {
            const uint32_t t = (threadIdx.x >= o) ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) off[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += buf[255];
        __syncthreads();
    }
}
// fused-path render, reference scheme (global int32 atomicMin) but only inside each hypothesis' box
hipError_t launch_render_boxes(const pr_triangle *tris, uint32_t n_tris, const pr_mat4 *poses_dev, uint32_t n_poses, const float *aabb,
                               int4 *bbox, int32_t *depth, uint32_t *row_count, uint32_t *row_off, uint32_t *counts,
                               uint32_t width, uint32_t height, const pr_mat4 &proj, pr_roi roi, hipStream_t s, bool compute_boxes,
                               PoseMeta *meta, DevIcpState *st, uint32_t *arrive, uint32_t cloud_stride, const uint32_t *box_off, const BatchCheck *check)
{
    if (n_poses == 0) return hipSuccess;
    if (compute_boxes) {
        hipLaunchKernelGGL(pose_bbox_kernel, dim3((n_poses + 255) / 256), dim3(256), 0, s, aabb, poses_dev, n_poses, proj, width, height, roi, bbox);
        if (box_off && kBoxPack) hipLaunchKernelGGL(box_pack_offsets_kernel, dim3(1), dim3(256), 0, s, (const int4 *)bbox, n_poses, const_cast<uint32_t *>(box_off));   // (the caller's scratch: filled here)
    }
    const pr_roi none{ 0, 0, 0, 0 };
    for (uint32_t p0 = 0; p0 < n_poses; p0 += 32768) {
        const uint32_t np = (n_poses - p0 < 32768) ? (n_poses - p0) : 32768;
        const size_t off = box_off ? 0 : (size_t)p0 * width * height;
        const uint32_t *bo = box_off ? box_off + p0 : nullptr;
        hipLaunchKernelGGL(fill_box_kernel, dim3((height + kBoxRowsPerBlock - 1) / kBoxRowsPerBlock, np), dim3(256), 0, s, depth + off, bbox + p0, width, height, bo);
        if (n_tris > 0)                                          // an empty mesh renders nothing: every cloud is empty
        { const uint32_t run = raster_pose_run(n_tris, np);
          // PR_RASTER_CHUNKS > 1 (experiment of round 5): the hypotheses of a launch in that many launches one behind the other
          const uint32_t chunks = (uint32_t)PR_RASTER_CHUNKS < np ? (uint32_t)PR_RASTER_CHUNKS : 1u;
          for (uint32_t c = 0; c < chunks; ++c) {
              const uint32_t c0 = (uint32_t)(((uint64_t)np * c) / chunks), c1 = (uint32_t)(((uint64_t)np * (c + 1)) / chunks), nc = c1 - c0;
              if (nc == 0) continue;
              hipLaunchKernelGGL(raster_kernel, dim3((n_tris + 255) / 256, (nc + run - 1) / run), dim3(256), 0, s, tris, n_tris, poses_dev + p0 + c0,
                                 depth + off + (bo ? 0 : (size_t)c0 * width * height), width, height, proj, none, width, height, (const int4 *)(bbox + p0 + c0), nc, run,
                                 bo ? bo + c0 : nullptr, (check && p0 == 0 && c == 0) ? *check : BatchCheck{});
          } }
        hipLaunchKernelGGL(count_box_kernel, dim3((height + kBoxRowsPerBlock - 1) / kBoxRowsPerBlock, np), dim3(256), 0, s, depth + off, bbox + p0, width, height,
                           row_count + (size_t)p0 * height, bo);
    }
    if (meta) { hipError_t e = launch_d2c_scan_init(row_count, height, row_off, counts, n_poses, meta, st, arrive, cloud_stride, s); if (e != hipSuccess) return e; }
    else { hipError_t e = launch_d2c_scan(row_count, height, row_off, counts, n_poses, s); if (e != hipSuccess) return e; }
    return hipGetLastError();
}

}  // namespace prk
