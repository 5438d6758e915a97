// icp_accumulate.h -- thrust__pcd2Ab (icp.h:128-209): the 29-term contribution of a point, one virtual workgroup of the canonical tree (transform_pcd fused in), and the fixed reduction tree
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#pragma once
#include "nn_query.h"

namespace prk {

// ================================================================================================
//  29-term contribution (icp.h:138-206) accumulated straight into the lane's registers
// ================================================================================================
// The 29 running sums of a lane are kept as 15 register pairs: v_pk_mul_f32 / v_pk_add_f32 do two IEEE single-precision
// operations per issue slot (separate multiply and add, no contraction), so every sum sees exactly the operations of the
// scalar form (pcd2Ab functor, icp.h:86-136) at roughly half the instruction count.
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

// the last pass of an ICP (iteration == max_iteration, icp.cu:189) only feeds fitness and rmse: sums 27 (squared distance)
// and 28 (count), each updated by exactly the operation the full form applies to it
__device__ __forceinline__ void accumulate_score(Acc29 &acc, float sx, float sy, float sz, const Corr &c)
{
    const float ex = c.dx - sx, ey = c.dy - sy, ez = c.dz - sz;
    const float e2 = ex * ex + ey * ey + ez * ez;
    acc.pk[15] += float2v{ e2, 1.0f };
}

// ================================================================================================
//  THE hot kernel: pending transform + correspondence + 29-term transform-reduce, all hypotheses
//  of a batch in one launch.  grid = (workgroups per hypothesis, hypotheses), 256 lanes.
// ================================================================================================

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
        if ((!kNN || kStack != -1) && xf) {                      // icp.cu:142-153 transform_pcd_cuda, fused (never in the winners pass: nn_search_kernel has moved the cloud -- said at compile time, so that no conditional store stands between the point loads and the winner loads)
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
#if !PR_PASS_LATE_STORE
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                if (i < cnt) st_off<pr_vec3>(cl, point_of(j0, i) * 12u, pr_vec3{ p[3 * i], p[3 * i + 1], p[3 * i + 2] });
#endif
        }
        auto store_back = [&]() {
#if PR_PASS_LATE_STORE
            if ((!kNN || kStack != -1) && xf) {
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i)
                    if (i < cnt) st_off<pr_vec3>(cl, point_of(j0, i) * 12u, pr_vec3{ p[3 * i], p[3 * i + 1], p[3 * i + 2] });
            }
#endif
        };
        if constexpr (kNN) store_back();
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
#pragma unroll
            for (uint32_t i0 = 0; i0 < 4; i0 += PR_GATHER_BATCH) {
                Gathered gth[PR_GATHER_BATCH];
                bool in_img[PR_GATHER_BATCH];
#pragma unroll
                for (uint32_t k = 0; k < PR_GATHER_BATCH; ++k) {
                    const uint32_t i = i0 + k;
                    in_img[k] = gather_issue(scene, p[3 * i], p[3 * i + 1], p[3 * i + 2], i < cnt, gth[k]);
                }
                if (i0 == 0) store_back();                           // (PR_PASS_LATE_STORE: the moved points go out BEHIND the step's first gathers instead of ahead of them)
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

#if PR_PASS_PREFETCH
    // the NEXT step's cloud points are requested before this step's gathers are looked at: one dependent round trip per step instead of two
    {
        float p[12];
        uint32_t j0, cnt;
        load_step(0, p, j0, cnt);
        for (uint32_t s = 0; s < b.steps && cnt != 0; ++s) {
            float pn[12];
            uint32_t j0n = 0, cntn = 0;
            if (s + 1 < b.steps) load_step(s + 1, pn, j0n, cntn);
            process_step(p, j0, cnt);
#pragma unroll
            for (int i = 0; i < 12; ++i) p[i] = pn[i];
            j0 = j0n; cnt = cntn;
        }
    }
#else
    for (uint32_t s = 0; s < b.steps; ++s) {
        float p[12];
        uint32_t j0, cnt;
        load_step(s, p, j0, cnt);
        if (cnt == 0) break;
        process_step(p, j0, cnt);
    }
#endif
    acc_export(acc, acc_out);
}

// canonical tree, second half: wave (balanced pairwise over lanes) -> ((w0+w1)+w2)+w3.  Returns, in threads 0..28, the
// workgroup sum of component threadIdx.x.  Contains one __syncthreads().
template <bool kScoreOnly = false>
__device__ __forceinline__ float vb_reduce(float (&acc)[29], float (*wsum)[kAccStride])
{
    constexpr int kFirst = kScoreOnly ? 27 : 0;                  // score-only passes carry zeros in sums 0..26
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if PR_PASS_TAIL_PRIO
    __builtin_amdgcn_s_setprio(PR_PASS_TAIL_PRIO);               // a wavefront in its reduction tail goes first: its workgroup retires sooner and frees its slots
#endif
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

}  // namespace prk
