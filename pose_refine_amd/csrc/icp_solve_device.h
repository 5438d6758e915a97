// icp_solve_device.h -- the per-hypothesis iteration logic of icp.cu:178-212 on the device: system-scope hand-off of the partial sums, convergence test, wave-cooperative 6x6 LDL^T (pr_solver.inl arithmetic)
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#pragma once
#include "pr_device.h"
#define PR_HD __host__ __device__
#include "pr_solver.inl"

namespace prk {

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

// The tail of a correspondence pass (icp_pass_kernel, nn_late_pass_kernel): thread `c` < 29 of the workgroup holds sum `c` of virtual block `vb`.
// Not fused: the partial sums go to memory and a second launch adds them.  Fused finalize + solve: partial sums cross workgroups (and XCDs)
// through memory with system-scope accesses on both sides; wave 0 drains its stores before the arrival atomic that publishes them.  The
// workgroup that delivers the last partial sum of the hypothesis (it cannot have another block left) adds the partials in block order (same
// sequence as icp_finalize_solve_kernel) and runs the iteration logic.  PoseMeta / DevIcpState are only read again by the next launch, so
// plain accesses suffice for them.  Returns true in the lanes that are done with the kernel.
__device__ __forceinline__ bool pass_deliver(const IcpBatch &b, uint32_t pose, uint32_t vb, uint32_t used, uint32_t n, float t)
{
    float *slot = b.partial + ((size_t)pose * b.nblk + vb) * kAccStride;
    if (!b.fused) {
        if (threadIdx.x < 29) slot[threadIdx.x] = t;
        return false;
    }
    if (threadIdx.x >= 64) return false;
    if (threadIdx.x < 29) st_sys_f32(slot + threadIdx.x, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = 0;
    if (threadIdx.x == 0) ticket = atomicAdd(&b.arrive[pose], 1u);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket + 1u != used) return false;
    if (threadIdx.x == 0) st_sys_u32(&b.arrive[pose], 0u);
    if (b.fused == 2u) {                                         // solve on the host: the sums of the hypothesis, straight into host memory
        if (threadIdx.x < 29) st_sys_f32(b.sums_out + (size_t)pose * kAccStride + threadIdx.x, sum_partials_sys(b.partial, pose, b.nblk, used, threadIdx.x));
        // ... and, when the host polls (round 6, option host_poll), the group's count: the wavefront waits until its 29 sum stores have been acknowledged,
        // then counts its hypothesis in; the one that completes the pose group re-arms the counter and stores the iteration's tag into pinned host
        // memory -- behind every hypothesis' sums (each was acknowledged before its count).  The host then solves the group while the launch is still
        // winding down (the end-of-kernel write-back of the clouds it moved), instead of waiting for the stream.  Plain write-through stores: a release
        // at system scope would write the whole L2 back per hypothesis (measured: 82 k instead of 234 k poses/s).
        // Each row also carries the tag (word 31), stored by the wavefront that stored the row's sums, after them: the host checks it when it has seen the
        // group's flag -- a flag that overtook a row on the way to host memory (writes of different compute units, in principle re-orderable by the
        // fabric) is noticed there and costs a stream wait, never a solve on stale sums.
        if (b.grp_flag) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if PR_HOST_ROW_TAG
            if (threadIdx.x == 0) st_sys_u32(reinterpret_cast<uint32_t *>(b.sums_out + (size_t)pose * kAccStride) + 31, b.iter + 1u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            if (threadIdx.x == 0 && atomicAdd(b.grp_count, 1u) + 1u == b.grp_expected) { st_sys_u32(b.grp_count, 0u); st_sys_u32(b.grp_flag, b.iter + 1u); }
        }
        return true;
    }
    DevIcpState s = b.st[pose];                                  // uniform; in flight together with the partial sums
    float total = 0.0f;
    if (threadIdx.x < 29) total = sum_partials_sys(b.partial, pose, b.nblk, used, threadIdx.x);
    float E[16];
    const bool finished = pose_iteration_wave(total, n, s, b.crit, b.iter, E);
    if (threadIdx.x != 0) return true;
    PoseMeta *wm = const_cast<PoseMeta *>(b.meta) + pose;
    if (finished) { s.done = 1; wm->state = kSkip; }
    else {
#pragma unroll
        for (int i = 0; i < 12; ++i) wm->xform[i] = E[i];
        wm->state = kRunWithTransform;
    }
    b.st[pose] = s;
    return true;
}

}  // namespace prk
