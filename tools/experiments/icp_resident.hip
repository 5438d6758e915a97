// icp_resident.hip -- NOT built into the library: round 6's RESIDENT form of the projective ICP loop, the record of an experiment that lost.
// (It was compiled inside icp_pass.hip, behind option "resident" in refine_submit_async: one launch per sub-batch instead of 21 x 2.)
//
// Idea: workgroup (hypothesis, block) keeps its 3072 cloud points in registers over all 21 passes -- the cloud is read once and never written (24 B per
// point and pass less through the fabric), no launch chain -- and the workgroups of a hypothesis meet once per pass through memory.
// Result (MI355X, 256 hypotheses of the bench, same box, bit-identical records with fixed-20 and the default criteria):
//   * arrival counters of neighbouring hypotheses in one array, polled by every waiting workgroup: 2.35 ms per pipelined step (109 k poses/s) -- 900 polling
//     loads and the atomics on four cache lines = one memory channel;
//   * one meeting point per hypothesis 4352 bytes apart, the polled line separate from the atomic's word: 1.07 ms (240 k poses/s) against 0.92 ms (277 k)
//     for the multi-launch loop; one batch at a time 1.39 against 1.11 ms.  The loop alone takes ~1.0 ms = 24 us per pass and round (128 hypotheses are
//     resident at 128 VGPRs, two rounds), of which ~13 us are the six dependent memory round trips of the meeting (partials + acknowledgement, ticket,
//     partial loads, update + acknowledgement, flag, poll, update loads) and ~10 us the three dependent gather rounds of a lane's 12 points;
//   * starting every other group of hypotheses 8 us late (so that the four workgroups of a compute unit are out of phase): no change (239-241 k) -- the chain
//     is the limit, not issue contention; a longer sleep between polls: no change.
// Why it cannot win as built: the multi-launch loop already overlaps four chains (two pose groups x two slots) and the other slot's raster on the same
// compute units at ~50 % VALU issue; the resident workgroups own all 512 VGPRs of a SIMD (4 x 128), so nothing else runs beside them, and their per-pass
// chain is longer than a launch's fixed cost.  A tag-in-line hand-off without acknowledgements (three round trips) would still leave ~17 us per pass and
// round = 0.71 ms per batch before any render.

// ================================================================================================
//  Resident form of the loop (option "resident", asynchronous fused path, projective scenes, device solve): ONE launch runs all passes of a
//  sub-batch.  Workgroup (hypothesis, g) of the canonical tree keeps its 3072 cloud points IN REGISTERS (12 per lane) from the first pass to
//  the last -- the cloud is read once and never written -- and the workgroups of a hypothesis meet once per pass: partial sums through memory
//  (system-scope accesses, as in pass_deliver), one arrival counter per hypothesis that only ever grows (a pass is complete at
//  (used + 1) * it + used arrivals; the workgroup that completes it adds the partials in workgroup order, runs the iteration logic, stores the
//  update and counts itself in once more: (used + 1) * (it + 1) = "the update of pass `it` is published"), everyone else polls that word.
//  Same per-point arithmetic, same tree, same solve as the multi-launch loop: bit-identical results.
//  Workgroups wait for each other, so the blocks of a hypothesis must be able to be resident together: the grid is one-dimensional and
//  hypothesis-major (the dispatcher hands out workgroups in index order: every hypothesis before the last one that has a workgroup on the chip
//  is complete on the chip and makes progress, so slots keep being freed); the launcher refuses clouds of more than kResidentMaxBlocks blocks.
// ================================================================================================
constexpr uint32_t kResidentSteps = 3;
constexpr uint32_t kResidentMaxBlocks = 128;
constexpr uint32_t kResidentSyncWords = 1088;                     // 4352 bytes between the meeting points of two hypotheses
template <class Scene, bool kScoreOnly>
__device__ __forceinline__ void resident_accumulate(float (&acc_out)[29], const float (&p)[kResidentSteps][12], const uint32_t (&cnt)[kResidentSteps], const Scene &scene)
{
    Acc29 acc;
    acc_clear(acc);
#pragma unroll
    for (uint32_t s = 0; s < kResidentSteps; ++s) {
        if (cnt[s] == 0) break;
        Gathered gth[4];
        bool in_img[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) in_img[i] = gather_issue(scene, p[s][3 * i], p[s][3 * i + 1], p[s][3 * i + 2], i < cnt[s], gth[i]);
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            Corr c;
            if (gather_finish(scene, in_img[i], p[s][3 * i + 2], gth[i], c)) {
                if constexpr (kScoreOnly) accumulate_score(acc, p[s][3 * i], p[s][3 * i + 1], p[s][3 * i + 2], c);
                else accumulate(acc, p[s][3 * i], p[s][3 * i + 1], p[s][3 * i + 2], c);
            }
        }
    }
    acc_export(acc, acc_out);
}
template <class Scene>
__global__ __launch_bounds__(256, PR_RESIDENT_WAVES) void icp_resident_kernel(IcpBatch b, Scene scene, uint32_t n_poses, uint32_t *rsync)
{
    __shared__ float wsum[4][kAccStride];
    __shared__ uint32_t sm_meta[16];
    const uint32_t pose = blockIdx.x / b.nblk, vb = blockIdx.x - pose * b.nblk;
    if (pose >= n_poses) return;
    const PoseMeta &pm = b.meta[pose];
    if (pm.state == kSkip) return;
    const uint32_t n = pm.count;
    constexpr uint32_t ppb = kResidentSteps * kPointsPerStep;
    if ((uint64_t)vb * ppb >= n) return;
    const uint32_t used = (n + ppb - 1) / ppb;
    const float *cl = reinterpret_cast<const float *>(b.cloud + pm.start);
    float p[kResidentSteps][12];
    uint32_t cnt[kResidentSteps];
    {
        const uint32_t last = n - 1u;
#pragma unroll
        for (uint32_t s = 0; s < kResidentSteps; ++s) {
            const uint32_t j0 = vb * ppb + s * kPointsPerStep + threadIdx.x;
            cnt[s] = (j0 >= n) ? 0u : (((n - j0 + kBlockThreads - 1u) / kBlockThreads < kPointsPerLane) ? (n - j0 + kBlockThreads - 1u) / kBlockThreads : kPointsPerLane);
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t j = j0 + i * kBlockThreads;
                const pr_vec3 v = ld_off<pr_vec3>(cl, (j < last ? j : last) * 12u);
                p[s][3 * i] = v.x; p[s][3 * i + 1] = v.y; p[s][3 * i + 2] = v.z;
            }
        }
    }
    // the hypothesis' meeting point: a record of its own, kResidentSyncWords apart from the next (another memory channel): word 0 = arrivals (atomics only),
    // words 32.. = a 64-byte line {published pass, state, -, -, update[12]} that the waiting workgroups poll -- never the word the atomics hit
    uint32_t *arrive = rsync + (size_t)pose * kResidentSyncWords;
    uint32_t *wm = arrive + 32;
    uint32_t *ws = reinterpret_cast<uint32_t *>(b.st + pose);
    float *slot = b.partial + ((size_t)pose * b.nblk + vb) * kAccStride;
    const uint32_t last_it = (uint32_t)b.crit.max_iteration;
#ifdef PR_RESIDENT_STAGGER
    // experiment: hypotheses start out of phase (the four workgroups of a compute unit belong to hypotheses 32 apart)
    if (((pose >> PR_RESIDENT_STAGGER_SHIFT) & 1u) != 0u) { const uint64_t t0 = __builtin_readcyclecounter(); while (__builtin_readcyclecounter() - t0 < (uint64_t)PR_RESIDENT_STAGGER) __builtin_amdgcn_s_sleep(8); }
#endif
    for (uint32_t it = 0;; ++it) {
        float acc[29], t;
        if (it == last_it) { resident_accumulate<Scene, true>(acc, p, cnt, scene); t = vb_reduce<true>(acc, wsum); }
        else { resident_accumulate<Scene, false>(acc, p, cnt, scene); t = vb_reduce(acc, wsum); }
        if (threadIdx.x < 64) {
            if (threadIdx.x < 29) st_sys_f32(slot + threadIdx.x, t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t ticket = 0;
            if (threadIdx.x == 0) ticket = atomicAdd(arrive, 1u);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket + 1u == used * (it + 1u)) {            // the last partial sum of this pass: add up, iterate, publish
                DevIcpState s;
                uint32_t *sw = reinterpret_cast<uint32_t *>(&s);
#pragma unroll
                for (uint32_t i = 0; i < sizeof(DevIcpState) / 4; ++i) sw[i] = ld_sys_u32(ws + i);
                float total = 0.0f;
                if (threadIdx.x < 29) total = sum_partials_sys(b.partial, pose, b.nblk, used, threadIdx.x);
                float E[16];
                const bool finished = pose_iteration_wave(total, n, s, b.crit, it, E);
                if (threadIdx.x == 0) {
                    if (finished) { s.done = 1; st_sys_u32(wm + 2, (uint32_t)kSkip); }
                    else {
#pragma unroll
                        for (int i = 0; i < 12; ++i) st_sys_f32(reinterpret_cast<float *>(wm + 4) + i, E[i]);
                        st_sys_u32(wm + 2, (uint32_t)kRunWithTransform);
                    }
#pragma unroll
                    for (uint32_t i = 0; i < sizeof(DevIcpState) / 4; ++i) st_sys_u32(ws + i, sw[i]);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_sys_u32(wm, it + 1u);
                }
            }
        }
        if (it == last_it) return;
        if (threadIdx.x == 0) {
            while (ld_sys_u32(wm) < it + 1u) __builtin_amdgcn_s_sleep(PR_RESIDENT_SLEEP);
        }
        __syncthreads();
        if (threadIdx.x < 16) sm_meta[threadIdx.x] = ld_sys_u32(wm + threadIdx.x);
        __syncthreads();
        if ((int32_t)sm_meta[2] == kSkip) return;
        float M[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) M[i] = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)sm_meta[4 + i]));
        // icp.cu:142-153 transform_pcd_cuda on the registers (same operations as vb_accumulate's fused form)
        const float2v Mx{ M[0], M[4] }, My{ M[1], M[5] }, Mz{ M[2], M[6] }, Mt{ M[3], M[7] };
#pragma unroll
        for (uint32_t s = 0; s < kResidentSteps; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = p[s][3 * i], y = p[s][3 * i + 1], z = p[s][3 * i + 2];
                float2v tt = Mx * float2v{ x, x };
                tt = tt + My * float2v{ y, y };
                tt = tt + Mz * float2v{ z, z };
                tt = tt + Mt;
                p[s][3 * i]     = tt.x;
                p[s][3 * i + 1] = tt.y;
                p[s][3 * i + 2] = M[8] * x + M[9] * y + M[10] * z + M[11];
            }
    }
}


// the resident form: false when the batch cannot run that way (the caller then enqueues the multi-launch loop)
bool resident_fits(const IcpBatch &b) { return b.steps == kResidentSteps && b.nblk >= 1 && b.nblk <= kResidentMaxBlocks && b.fused == 1; }
size_t resident_sync_bytes(uint32_t n_poses) { return (size_t)n_poses * kResidentSyncWords * 4u; }
hipError_t launch_icp_resident_proj_packed(const IcpBatch &b, const SceneProjPacked &sc, uint32_t n_poses, uint32_t *rsync, hipStream_t s)
{
    if (n_poses == 0 || b.nblk == 0) return hipSuccess;
    if (!resident_fits(b)) return hipErrorInvalidValue;
    for (uint32_t p0 = 0; p0 < n_poses; p0 += 32768) {
        const uint32_t np = (n_poses - p0 < 32768) ? (n_poses - p0) : 32768;
        IcpBatch bb = b;
        bb.meta += p0; bb.partial += (size_t)p0 * b.nblk * kAccStride; bb.st += p0;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(icp_resident_kernel<SceneProjPacked>), dim3(np * b.nblk), dim3(kBlockThreads), 0, s, bb, sc, np, rsync + (size_t)p0 * kResidentSyncWords);
    }
    return hipGetLastError();
}
