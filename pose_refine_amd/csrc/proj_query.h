// proj_query.h -- Scene_projective::query (depth_scene.h:29-48) + pcd2dep (common.h:63-73), for the caller's two arrays and for the packed 16-byte record
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#pragma once
#include "pr_device.h"

namespace prk {

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
    if (!(vx > -1.0f && vx < (float)width && vy > -1.0f && vy < (float)height)) return false;
    px = (int)vx; py = (int)vy;
    idx = (uint32_t)px + (uint32_t)py * width;
    return true;
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

}  // namespace prk
