// pr_refine.cpp -- render, depth -> cloud, and the fused batch path render -> cloud -> ICP: synchronous form, two asynchronous slots, helper threads
#include "pr_runtime.h"

namespace prr {

// the raster packs pixel coordinates into 13 bits each and enumerates a triangle's candidate pixels with 24-bit arithmetic
bool frame_size_ok(size_t W, size_t H)
{
    if (W > 8192 || H > 8192 || W * H > ((size_t)1 << 24)) { set_error("frames larger than 8192 on a side or 2^24 pixels are not supported (got %zux%zu)", W, H); return false; }
    return true;
}

int render_impl(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, size_t P, size_t W, size_t H,
                const pr_mat4 *proj, pr_roi roi, int32_t *depth_dev, bool zero_empty)
{
    if ((!tris_dev && n_tris > 0) || (P && (!poses_host || !depth_dev)) || !proj || W == 0 || H == 0) { set_error("pr_render: bad arguments"); return PR_ERR_INVALID; }   // (an empty model has no array, no hypotheses need none)
    if (!frame_size_ok(W, H)) return PR_ERR_INVALID;
    size_t rw = W, rh = H;
    if (roi.width > 0 && roi.height > 0) {
        if (roi.x < 0 || roi.y < 0 || (size_t)(roi.x + roi.width) > W || (size_t)(roi.y + roi.height) > H) {
            set_error("pr_render: roi out of image");        // renderer.cu:202-203 asserts
            return PR_ERR_INVALID;
        }
        rw = (size_t)roi.width; rh = (size_t)roi.height;
    }
    if (P == 0) return PR_OK;
    PR_TRY(g->poses.ensure(sizeof(pr_mat4) * P));
    SpanGuard sp(kSpanRender);
    {   // poses through the pinned staging array, pulled by a kernel (no copy command from the caller's pageable array: see refine_impl)
        PR_TRY(g->h_poses.ensure(sizeof(pr_mat4) * P));
        std::memcpy(g->h_poses.p, poses_host, sizeof(pr_mat4) * P);
        void *hp = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&hp, g->h_poses.p, 0));
        HIP_TRY(prk::launch_stage_words(hp, g->poses.p, sizeof(pr_mat4) * P, g->stream));
    }
    HIP_TRY(prk::launch_fill_i32(depth_dev, P * rw * rh, INT32_MAX, g->stream));
    HIP_TRY(prk::launch_raster(tris_dev, (uint32_t)n_tris, g->poses.as<pr_mat4>(), (uint32_t)P, depth_dev, (uint32_t)W, (uint32_t)H,
                               *proj, roi, (uint32_t)rw, (uint32_t)rh, g->stream));
    if (zero_empty) HIP_TRY(prk::launch_max2zero(depth_dev, P * rw * rh, g->stream));
    return PR_OK;
}

template <typename T>
int depth2cloud_impl(const T *depth_dev, uint32_t W, uint32_t H, const float K[9], uint32_t stride, uint32_t tl_x, uint32_t tl_y,
                     pr_vec3 **cloud_out, uint32_t *n_out)
{
    if (!depth_dev || !K || !cloud_out || !n_out || W == 0 || H == 0 || stride == 0) { set_error("pr_depth2cloud: bad arguments"); return PR_ERR_INVALID; }
    const uint32_t gh = H / stride;
    PR_TRY(g->row_count.ensure(sizeof(uint32_t) * std::max(1u, gh)));
    PR_TRY(g->row_off.ensure(sizeof(uint32_t) * std::max(1u, gh)));
    PR_TRY(g->counts.ensure(sizeof(uint32_t)));
    HIP_TRY(prk::launch_depth2cloud<T>(depth_dev, 1, 0, W, H, stride, tl_x, tl_y, K[0], K[4], K[2], K[5], false, g->row_count.as<uint32_t>(),
                                       g->row_off.as<uint32_t>(), g->counts.as<uint32_t>(), nullptr, 0, false, g->stream));
    uint32_t n = 0;
    PR_TRY(read_back_words(g->counts.p, &n, 1, g->stream));
    pr_vec3 *cloud = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&cloud), sizeof(pr_vec3) * std::max(1u, n)));
    if (n > 0) {
        hipError_t e = prk::launch_depth2cloud<T>(depth_dev, 1, 0, W, H, stride, tl_x, tl_y, K[0], K[4], K[2], K[5], false, g->row_count.as<uint32_t>(),
                                                  g->row_off.as<uint32_t>(), g->counts.as<uint32_t>(), cloud, 0, true, g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) { hipFree(cloud); set_error("depth2cloud emit failed: %s", hipGetErrorString(e)); return PR_ERR_HIP; }
    }
    *cloud_out = cloud; *n_out = n;
    return PR_OK;
}

bool roi_ok(pr_roi roi, uint32_t W, uint32_t H)
{
    if (roi.width <= 0 || roi.height <= 0) return true;            // no ROI
    if (roi.x < 0 || roi.y < 0 || (size_t)roi.x + (size_t)roi.width > W || (size_t)roi.y + (size_t)roi.height > H) {
        set_error("roi out of image");                               // renderer.cu:202-203 asserts
        return false;
    }
    return true;
}

int refine_impl(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t P, uint32_t W, uint32_t H,
                const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                pr_result *results_host, pr_result *results_dev, uint32_t *sizes_host)
{
    if (!K || W == 0 || H == 0) { set_error("pr_refine_batch: bad arguments"); return PR_ERR_INVALID; }
    if (!frame_size_ok(W, H) || !roi_ok(roi, W, H)) return PR_ERR_INVALID;
    if (P == 0) return PR_OK;
    SceneSel sc;
    std::memset(static_cast<void *>(&sc), 0, sizeof sc);           // also zeroes padding: sc is part of the graph-cache key
    const Camera cam{ W, H, K[0], K[4], K[2], K[5] };
    trace_mark("refine_impl: enter");
    PR_TRY(make_scene(scene_kind, scene, /*want_packed=*/true, sc, nullptr, nullptr, &cam));
    trace_mark("refine_impl: scene ready");
    // bound the depth workspace to ~4 GiB per chunk (288 GB of HBM would allow far more; this keeps
    // first-touch cost and the 2^32 element index space comfortable)
    const size_t img = (size_t)W * H;
    uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(P, ((size_t)4 << 30) / (img * sizeof(int32_t))));
    std::vector<uint32_t> start(chunk), count(chunk);
    // model box: recomputed from the triangle buffer on every call (one pass over the mesh; nothing is cached by address here)
    PR_TRY(g->aabb.ensure(6 * sizeof(float)));
    PR_TRY(g->aabb_keys.ensure(6 * sizeof(uint32_t)));
    HIP_TRY(prk::launch_model_aabb(tris_dev, (uint32_t)n_tris, g->aabb_keys.as<uint32_t>(), g->aabb.as<float>(), nullptr, nullptr, g->stream));
    for (uint32_t p0 = 0; p0 < P; p0 += chunk) {
        const uint32_t np = std::min(chunk, P - p0);
        PR_TRY(g->depth.ensure(sizeof(int32_t) * (img + prk::kBoxPack) * np));
        PR_TRY(g->row_count.ensure(sizeof(uint32_t) * (size_t)H * np));
        PR_TRY(g->row_off.ensure(sizeof(uint32_t) * (size_t)H * np));
        PR_TRY(g->counts.ensure(sizeof(uint32_t) * np));
        PR_TRY(g->h_counts.ensure(sizeof(uint32_t) * np));
        uint32_t *h_counts = g->h_counts.as<uint32_t>();
        // per-pose pixel boxes; raster + row counts + row scan
        PR_TRY(g->bbox.ensure(sizeof(int4) * np + sizeof(uint32_t) * np));   // boxes, then the offsets of the packed boxes (box_pack_offsets_kernel)
        PR_TRY(g->poses.ensure(sizeof(pr_mat4) * np));
        uint32_t *box_off = (prk::kBoxPack && opt.raster_mode != 1) ? reinterpret_cast<uint32_t *>(g->bbox.as<int4>() + np) : nullptr;
        {
            SpanGuard sp(kSpanRender);
            {   // the hypotheses' poses: through the context's pinned staging array, pulled by a kernel (as the asynchronous path stages its inputs) --
                // a copy command from the caller's pageable array goes through the runtime's bounce buffers and its copy-engine path (see icp_drive's
                // result block: that path is where the host-solve pipeline's one-off multi-millisecond stalls came from)
                PR_TRY(g->h_poses.ensure(sizeof(pr_mat4) * np));
                std::memcpy(g->h_poses.p, poses_host + p0, sizeof(pr_mat4) * np);
                void *hp = nullptr;
                HIP_TRY(hipHostGetDevicePointer(&hp, g->h_poses.p, 0));
                HIP_TRY(prk::launch_stage_words(hp, g->poses.p, sizeof(pr_mat4) * np, g->stream));
            }
            if (opt.raster_mode == 1)
                HIP_TRY(prk::launch_render_bands(tris_dev, (uint32_t)n_tris, g->poses.as<pr_mat4>(), np, g->aabb.as<float>(), g->bbox.as<int4>(),
                                                 g->depth.as<int32_t>(), g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(),
                                                 g->counts.as<uint32_t>(), W, H, *proj, roi, (uint32_t)g->n_cus, g->stream));
            else
                HIP_TRY(prk::launch_render_boxes(tris_dev, (uint32_t)n_tris, g->poses.as<pr_mat4>(), np, g->aabb.as<float>(), g->bbox.as<int4>(),
                                                 g->depth.as<int32_t>(), g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(),
                                                 g->counts.as<uint32_t>(), W, H, *proj, roi, g->stream, /*compute_boxes=*/true, nullptr, nullptr, nullptr, 0, box_off));
        }
        {   // the cloud sizes come back through a kernel's stores into the pinned array, not through a copy command (see icp_drive's result block)
            void *hc = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&hc, h_counts, 0));
            HIP_TRY(prk::launch_copy_words32(g->counts.p, hc, np, g->stream));
        }
        HIP_TRY(hipStreamSynchronize(g->stream));
        trace_mark("refine_impl: render done, counts on host");
        uint32_t max_n = 0;
        for (uint32_t i = 0; i < np; ++i) max_n = std::max(max_n, h_counts[i]);
        // the clouds packed one behind the other, as in the asynchronous path (d2c_pack_starts_kernel; the host adds the same rounded sizes)
        size_t total = 0;
        for (uint32_t i = 0; i < np; ++i) {
            start[i] = (uint32_t)total; count[i] = h_counts[i];
            total += prk::kCloudAlign ? ((size_t)h_counts[i] + prk::kCloudAlign - 1) / prk::kCloudAlign * prk::kCloudAlign : (((size_t)max_n + 3) & ~(size_t)3);
        }
        if (total > 0xffffffffull) { set_error("pr_refine_batch: more than 2^32 cloud points in one chunk"); return PR_ERR_INVALID; }
        const size_t cstride = ((size_t)max_n + 3) & ~(size_t)3;           // (fixed-stride layout, kCloudAlign == 0: keeps every cloud 16-byte aligned)
        PR_TRY(g->cloud.ensure(sizeof(pr_vec3) * std::max<size_t>(4, total)));
        if (max_n > 0) {
            SpanGuard sp(kSpanCloud);
            if (prk::kCloudAlign) {
                PR_TRY(g->meta.ensure(sizeof(prk::PoseMeta) * np));
                HIP_TRY(prk::launch_d2c_pack_starts(g->counts.as<uint32_t>(), np, g->meta.as<prk::PoseMeta>(), g->stream));
            }
            HIP_TRY(prk::launch_emit_box(g->depth.as<int32_t>(), np, W, H, g->bbox.as<int4>(), K[0], K[4], K[2], K[5],
                                         g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(), g->cloud.as<pr_vec3>(), cstride, g->stream,
                                         prk::kCloudAlign ? g->meta.as<prk::PoseMeta>() : nullptr, box_off));
        }
        if (sizes_host) std::memcpy(sizes_host + p0, count.data(), sizeof(uint32_t) * np);
        trace_mark("refine_impl: clouds emitted, icp_drive next");
        PR_TRY(icp_drive(g->cloud.as<pr_vec3>(), start.data(), count.data(), np, sc, crit,
                         results_host ? results_host + p0 : nullptr, results_dev ? results_dev + p0 : nullptr));
    }
    return PR_OK;
}

// ---- asynchronous fused refinement: two slots, everything of a batch enqueued without a host round trip -------------------
// The synchronous path above reads the cloud sizes back in the middle of a step (they size the cloud stride and the grid),
// uploads the start state and returns only after the results are in: ≈150 µs of a 1.3 ms step during which the GPU idles.
// Here the host computes the per-pose pixel boxes itself (same arithmetic as pose_bbox_kernel, 8 corners per pose), which
// bounds every cloud by its box area, the start state is written by a kernel from the device-side counts, and a batch is
// only waited for when its results are wanted -- so the next batch can be enqueued while this one runs.
// The streams of BOTH slots are created together, in one run: main and first side stream of slot 0, then of slot 1.  Which
// hardware queue -- and with it which of the command processor's four pipes -- a stream lands on follows from the order in which
// the process created its queues (queue k -> pipe k mod 4), and two of these four streams on one pipe cost up to half the
// throughput (tools/queue_fairness.hip: two chains of dependent launches on one pipe take 2.1x a lone chain each, on two pipes
// 1.25x; bench.py with PR_STREAM_PADS-style gaps: 250 k -> 118-213 k poses/s).  Created lazily, slot by slot, anything the host does
// in between (its first synchronous copy creates the null stream's queue) shifts slot 1 onto slot 0's pipes; four queues
// created back to back always sit on four different pipes.  Further side streams (3-4 pose groups) come lazily, ensure_stream.
int slot_streams(Slot &sl)
{
    if (sl.stream) return PR_OK;
    for (Slot &o : g->slots) {
        if (o.stream || (!opt.eager_streams && &o != &sl)) continue;
        HIP_TRY(hipStreamCreateWithFlags(&o.stream, hipStreamNonBlocking));
        if (opt.eager_streams) PR_TRY(ensure_stream(o.side[0], &o.join[0]));
    }
    for (Slot &o : g->slots) {
        if (o.done) continue;
        HIP_TRY(hipEventCreateWithFlags(&o.scene_ready, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&o.fork, hipEventDisableTiming));
        // `done` is the one event a host thread waits on (pr_refine_wait): option "blocking_wait" makes that wait sleep instead of spin --
        // eight ranks spinning on a node whose cgroup grants sixteen CPUs leave nothing for RCCL's proxy threads
        HIP_TRY(hipEventCreateWithFlags(&o.done, hipEventDisableTiming | (opt.blocking_wait ? hipEventBlockingSync : 0u)));
        HIP_TRY(hipEventCreateWithFlags(&o.progress, hipEventDisableTiming));
    }
    return PR_OK;
}
// wait for everything a slot has in flight (its stream and side streams)
void slot_worker_wait(Slot &sl)      // the helper thread's batch, if one is running, has finished when this returns
{
    if (!sl.worker || !sl.worker_job) return;
    std::unique_lock<std::mutex> lk(sl.worker->mu);
    sl.worker->cv.wait(lk, [&] { return sl.worker->done || !sl.worker->alive; });
}
void slot_worker_stop(Slot &sl)
{
    if (!sl.worker) return;
    { std::lock_guard<std::mutex> lk(sl.worker->mu); sl.worker->quit = true; }
    sl.worker->cv.notify_all();
    if (sl.worker->th.joinable()) sl.worker->th.join();
    sl.worker.reset();
    sl.worker_job = false;
}
void slot_drain(Slot &sl)
{
    slot_worker_wait(sl);
    for (int i = 0; i < 3; ++i) if (sl.side[i]) (void)hipStreamSynchronize(sl.side[i]);
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
}
void drain_all_slots()
{
    for (Slot &o : g->slots) if (o.pending && !o.delivered) slot_drain(o);
}
void drain_slots_reading(const void *p, size_t bytes)
{
    if (!p || bytes == 0) return;
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
    auto hit = [&](const void *a, size_t ab) { const uintptr_t s = reinterpret_cast<uintptr_t>(a); return a && ab && s < hi && lo < s + ab; };
    for (Slot &sl : g->slots) {
        if (!sl.pending || sl.delivered) continue;
        const Resubmit &r = (sl.worker_job && sl.worker) ? sl.worker->in : sl.again;     // (a helper thread's job is posted and read under g->mu by this thread only)
        bool reads = hit(r.tris, r.n_tris * sizeof(pr_triangle)) || hit(r.results_dev, (size_t)sl.P * sizeof(pr_result));
        if (r.scene_kind == PR_SCENE_NN)
            reads = reads || hit(r.sn.pcd, (size_t)r.sn.n_points * sizeof(pr_vec3)) || hit(r.sn.normal, (size_t)r.sn.n_points * sizeof(pr_vec3)) ||
                    hit(r.sn.nodes, (size_t)r.sn.n_nodes * sizeof(pr_kdnode));
        else {
            const size_t n = (size_t)r.sp.view.width * r.sp.view.height;
            reads = reads || hit(r.sp.view.pcd, n * sizeof(pr_vec3)) || hit(r.sp.view.normal, n * sizeof(pr_vec3));
        }
        if (reads) slot_drain(sl);
    }
}
void slot_release(Slot &sl)
{
    slot_drain(sl);
    slot_worker_stop(sl);
    for (DevBuf *b : { &sl.poses_bbox, &sl.depth, &sl.row_count, &sl.row_off, &sl.counts, &sl.cloud, &sl.meta, &sl.partial, &sl.dstate,
                       &sl.dresults, &sl.arrive, &sl.aabb_keys, &sl.nn_prev, &sl.packed.rec }) b->release();
    sl.packed = PackedCache();
    sl.h_in.release(); sl.h_out.release();
    for (int i = 0; i < 3; ++i) {
        if (sl.side[i]) hipStreamDestroy(sl.side[i]);
        if (sl.join[i]) hipEventDestroy(sl.join[i]);
        sl.side[i] = nullptr; sl.join[i] = nullptr;
    }
    if (sl.fork) hipEventDestroy(sl.fork);
    if (sl.done) hipEventDestroy(sl.done);
    if (sl.scene_ready) hipEventDestroy(sl.scene_ready);
    if (sl.progress) hipEventDestroy(sl.progress);
    sl.progress = nullptr; sl.progress_valid = false;
    if (sl.stream) hipStreamDestroy(sl.stream);
    sl.fork = sl.done = sl.scene_ready = nullptr; sl.stream = nullptr; sl.pending = false;
    for (hipEvent_t e : sl.t_events) (void)hipEventDestroy(e);
    sl.t_events.clear(); sl.t_used = 0; sl.t_spans.clear(); sl.timed = false;
}

// Host-side copy of a triangle buffer's box, once per (pointer, size): it feeds the per-pose pixel boxes computed on the host,
// which size a batch before anything of it has run.  The copy is an ASSUMPTION about memory the caller owns: every batch that
// uses it re-derives the box on the device (one pass over the mesh, on a stream that is idle at that point) and refine_wait
// re-runs the batch through the synchronous path if the two differ -- a rewritten mesh costs one repeated batch, never a wrong one.
// (An indexed form of the soup with a per-pose vertex stage was built and measured as well: three divergent 16-byte gathers
// per triangle cost the texture addresser more than the 170 saved VALU instructions give back -- 1.27 -> 1.32 ms per step.)
int ensure_model_box(const pr_triangle *tris_dev, size_t n_tris)
{
    if (g->mesh_key == tris_dev && g->mesh_n == n_tris && g->aabb_host_valid) return PR_OK;
    std::vector<pr_triangle> h(n_tris);
    if (n_tris) HIP_TRY(hipMemcpy(h.data(), tris_dev, sizeof(pr_triangle) * n_tris, hipMemcpyDeviceToHost));
    const float *f = reinterpret_cast<const float *>(h.data());
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (size_t v = 0; v < n_tris * 3; ++v)
        for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], f[3 * v + d]); hi[d] = fmaxf(hi[d], f[3 * v + d]); }
    for (int d = 0; d < 3; ++d) { g->aabb_host[d] = lo[d]; g->aabb_host[3 + d] = hi[d]; }
    g->mesh_key = tris_dev; g->mesh_n = n_tris; g->aabb_host_valid = true;
    return PR_OK;
}

// pose_bbox_kernel on the host (same operations in the same order; any conservative box gives the same images and clouds)
void pose_bbox_host(const float *aabb, const pr_mat4 &pose, const pr_mat4 &proj, uint32_t width, uint32_t height, pr_roi roi, int32_t out[4])
{
    const float *M = pose.m;
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
        x0 = std::max(0, (int)floorf(mnx) - 2);  x1 = std::min((int)width - 1, (int)ceilf(mxx) + 2);
        y0 = std::max(0, (int)floorf(mny) - 2);  y1 = std::min((int)height - 1, (int)ceilf(mxy) + 2);
    }
    if (roi.width > 0 && roi.height > 0) {
        x0 = std::max(x0, roi.x);  x1 = std::min(x1, roi.x + roi.width - 1);
        y0 = std::max(y0, (int)height - 1 - (roi.y + roi.height - 1));  y1 = std::min(y1, (int)height - 1 - roi.y);
    }
    out[0] = x0; out[1] = y0; out[2] = x1; out[3] = y1;
}

// the synchronous entry points run on whichever slot holds no unfinished batch of the caller's (-1: none)
int free_slot() { for (int i = 0; i < kSlots; ++i) if (!g->slots[i].pending) return i; return -1; }

int refine_wait(int slot)
{
    if (slot < 0 || slot >= kSlots) { set_error("slot must be 0..%d", kSlots - 1); return PR_ERR_INVALID; }
    Slot &sl = g->slots[slot];
    if (!sl.pending) { set_error("pr_refine_wait: nothing was submitted on slot %d", slot); return PR_ERR_INVALID; }
    if (sl.delivered) { sl.pending = false; return PR_OK; }
    if (sl.worker_job) {                                          // the batch ran on the slot's helper thread: collect its verdict
        slot_worker_wait(sl);
        sl.worker_job = false; sl.pending = false;
        std::lock_guard<std::mutex> lk(sl.worker->mu);
        if (!sl.worker->done) { set_error("pr_refine_wait: the slot's helper thread ended before its batch did"); return PR_ERR_HIP; }
        if (sl.worker->rc != PR_OK) set_error("%s", sl.worker->err.c_str());
        return sl.worker->rc;
    }
    {
        // the slot stays occupied until its batch has really finished: if the wait itself fails, everything the slot has in flight is
        // drained before the error goes back (a later submit must never reuse the slot's buffers under a running batch)
        const hipError_t we = hipEventSynchronize(sl.done);
        if (we != hipSuccess) { slot_drain(sl); sl.pending = false; set_error("pr_refine_wait: hipEventSynchronize failed: %s", hipGetErrorString(we)); return PR_ERR_HIP; }
    }
    sl.pending = false;
    const unsigned char *h_out = sl.h_out.as<unsigned char>();
    if (*reinterpret_cast<const volatile uint32_t *>(h_out + sl.flag_off) != 0u) {
        // the triangle buffer no longer has the box this batch was sized with: forget the host copy and run the batch again,
        // synchronously (that path derives every box on the device); outputs are overwritten in full
        g->aabb_host_valid = false; g->mesh_key = nullptr;
        g->stat_repeated++;                                         // (pr_stats: the safety net is not free -- a caller that sees this count grow should call pr_invalidate)
        // (or a scene array no longer has the content its cached form was derived from: same cure)
        drain_all_slots();
        for (Slot &o : g->slots) o.packed.valid = false;
        g->packed.valid = false;
        for (NNDerived &d : g->nn_sets) { d.valid = false; d.grid_valid = false; }
        const Resubmit &r = sl.again;
        const void *scene = (r.scene_kind == PR_SCENE_NN) ? static_cast<const void *>(&r.sn) : static_cast<const void *>(&r.sp);
        return refine_impl(r.tris, r.n_tris, sl.h_in.as<pr_mat4>(), sl.P, r.W, r.H, &r.proj, r.K, r.scene_kind, scene, r.crit, r.roi,
                           sl.user_results_host, r.results_dev, sl.user_sizes);
    }
    const uint32_t *h_counts = sl.h_out.as<uint32_t>();
    uint32_t largest = 1;
    for (uint32_t i = 0; i < sl.P; ++i) largest = std::max(largest, h_counts[i]);
    g->cloud_hint = largest;
    if (sl.timed) {                                              // the batch carried timing events: the same accounts as the synchronous timed path keeps
        for (const Slot::TSpan &t : sl.t_spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, sl.t_events[t.e0], sl.t_events[t.e1]) != hipSuccess) continue;
            if (t.kind == kSpanRender) g->render_ms += ms;
            else if (t.kind == kSpanCloud) g->cloud_ms += ms;
            else {
                uint64_t pts = 0;
                for (uint32_t i = t.q0; i < t.q0 + t.nq; ++i) pts += h_counts[i];
                g->icp_ms += ms; g->icp_launches++; note_launch_us(ms);
                g->icp_points += pts; g->icp_bytes += pts * (t.edge ? 36u : 48u);
                if (t.marks) {                                      // kd-tree pass: search | bound | task walk | winners pass
                    const hipEvent_t ev[5] = { sl.t_events[t.e0], sl.t_events[t.m[0]], sl.t_events[t.m[1]], sl.t_events[t.m[2]], sl.t_events[t.e1] };
                    float part[4]; bool ok = true;
                    for (int k = 0; k < 4; ++k) ok = ok && hipEventElapsedTime(&part[k], ev[k], ev[k + 1]) == hipSuccess;
                    if (ok) { for (int k = 0; k < 4; ++k) g->nn_part_ms[k] += part[k]; g->nn_part_n++; }
                }
            }
        }
        sl.t_spans.clear(); sl.t_used = 0; sl.timed = false;
    }
    if (sl.user_sizes) std::memcpy(sl.user_sizes, h_counts, sizeof(uint32_t) * sl.P);
    if (sl.user_results_host) std::memcpy(sl.user_results_host, h_out + (((size_t)sl.P * 4 + 63) & ~(size_t)63), sizeof(pr_result) * sl.P);
    return PR_OK;
}

int refine_impl(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t P, uint32_t W, uint32_t H,
                const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                pr_result *results_host, pr_result *results_dev, uint32_t *sizes_host);
// the helper thread of a slot: a private context on the slot's device, then one synchronous batch per job
void slot_worker_main(SlotWorker *w)
{
    bool ready = bind_shared(w->device) == PR_OK && pr_thread_context(1) == PR_OK;
    // the streams this thread's batches will run on -- its context's own and the first side stream (two pose groups) -- are created NOW, while the thread
    // that starts the helpers waits: the helpers of a context's slots are started one after the other (slot_worker_post), so their four streams get four
    // consecutive hardware queues = four different pipes of the command processor.  Created lazily by whichever helper came first, two of them could share a
    // pipe: the host-solve pipeline then ran at 208 k instead of 240-254 k poses/s, in about one process out of three (round 6, same box).
    if (ready) {
        std::lock_guard<std::mutex> ck(g->mu);
        ready = require_ctx() == PR_OK && ensure_stream(g->side[0], &g->ev_join[0]) == PR_OK;
    }
    { std::lock_guard<std::mutex> lk(w->mu); w->started = true; }
    w->cv.notify_all();
    for (;;) {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) break;
        lk.unlock();
        trace_mark("worker: job taken");
        int rc = PR_ERR_HIP;
        if (!ready) set_error("the slot's helper thread could not create its context: %s", std::string(prh::g_err).c_str());
        else {
            std::lock_guard<std::mutex> ck(g->mu);
            rc = require_ctx();
            if (rc == PR_OK) {
                const Resubmit &r = w->in;
                const void *scene = (r.scene_kind == PR_SCENE_NN) ? static_cast<const void *>(&r.sn)
                                    : (r.scene_kind == PR_SCENE_PROJ_CROP ? static_cast<const void *>(&r.sp) : static_cast<const void *>(&r.sp.view));
                rc = refine_impl(r.tris, r.n_tris, w->poses.data(), (uint32_t)w->poses.size(), r.W, r.H, &r.proj, r.K, r.scene_kind, scene, r.crit, r.roi,
                                 w->results_host, r.results_dev, w->sizes_host);
            }
        }
        trace_mark("worker: job done");
        lk.lock();
        w->rc = rc; w->err = (rc == PR_OK) ? std::string() : prh::g_err;
        w->has_job = false; w->done = true;
        lk.unlock();
        w->cv.notify_all();
    }
    if (ready) (void)pr_thread_context(0);
    { std::lock_guard<std::mutex> lk(w->mu); w->alive = false; }
    w->cv.notify_all();
}
int slot_worker_post(Slot &sl, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t P, uint32_t W, uint32_t H,
                     const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                     pr_result *results_host, pr_result *results_dev, uint32_t *sizes_host)
{
    if (scene_kind != PR_SCENE_NN && scene_kind != PR_SCENE_PROJ && scene_kind != PR_SCENE_PROJ_CROP) { set_error("unknown scene kind %d", scene_kind); return PR_ERR_INVALID; }
    if (!scene) { set_error("pr_refine_submit: null scene"); return PR_ERR_INVALID; }
    if (!sl.worker) {
        // the helpers of ALL slots of this context, one after the other, each with its streams made before the next one starts (see slot_worker_main)
        for (Slot &o : g->slots) {
            if (o.worker) continue;
            o.worker.reset(new SlotWorker());
            o.worker->device = g->device; o.worker->alive = true;
            try { o.worker->th = std::thread(slot_worker_main, o.worker.get()); }
            catch (...) { o.worker.reset(); if (&o == &sl) { set_error("pr_refine_submit: cannot start the slot's helper thread"); return PR_ERR_NOMEM; } continue; }
            std::unique_lock<std::mutex> lk(o.worker->mu);
            o.worker->cv.wait(lk, [&] { return o.worker->started; });
        }
    }
    SlotWorker &w = *sl.worker;
    {
        std::lock_guard<std::mutex> lk(w.mu);
        Resubmit &r = w.in;
        r.tris = tris_dev; r.n_tris = n_tris; r.W = W; r.H = H; r.proj = *proj; std::memcpy(r.K, K, sizeof r.K);
        r.scene_kind = scene_kind; r.crit = crit; r.roi = roi; r.results_dev = results_dev;
        if (scene_kind == PR_SCENE_NN) r.sn = *static_cast<const pr_scene_nn *>(scene);
        else if (scene_kind == PR_SCENE_PROJ_CROP) r.sp = *static_cast<const pr_scene_proj_crop *>(scene);
        else { r.sp.view = *static_cast<const pr_scene_proj *>(scene); r.sp.tl_x = r.sp.tl_y = 0; }
        w.poses.assign(poses_host, poses_host + P);
        w.results_host = results_host; w.sizes_host = sizes_host;
        w.done = false; w.has_job = true;
    }
    w.cv.notify_all();
    return PR_OK;
}

int refine_submit_async(Slot &sl, const pr_triangle *tris_dev, size_t n_tris, uint32_t P, uint32_t W, uint32_t H,
                        const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                        pr_result *results_host, pr_result *results_dev);

int refine_submit(int slot, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t P, uint32_t W, uint32_t H,
                  const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                  pr_result *results_host, pr_result *results_dev, uint32_t *sizes_host)
{
    if (slot < 0 || slot >= kSlots) { set_error("slot must be 0..%d", kSlots - 1); return PR_ERR_INVALID; }
    Slot &sl = g->slots[slot];
    if (sl.pending) { set_error("pr_refine_submit: slot %d still holds an unfinished batch (call pr_refine_wait)", slot); return PR_ERR_INVALID; }
    if ((!tris_dev && n_tris > 0) || !poses_host || !proj || !K || W == 0 || H == 0 || (!results_host && !results_dev)) { set_error("pr_refine_submit: bad arguments"); return PR_ERR_INVALID; }
    if (crit.max_iteration < 0) { set_error("max_iteration must be >= 0"); return PR_ERR_INVALID; }
    if (!frame_size_ok(W, H) || !roi_ok(roi, W, H)) return PR_ERR_INVALID;
    sl.P = P; sl.user_results_host = results_host; sl.user_sizes = sizes_host; sl.delivered = false;
    const size_t img = (size_t)W * H;
    const uint64_t period = (uint64_t)std::max(1, opt.sample_period);
    const bool sample_call = (opt.profile == 2) && (g->sample_clock % period == 0);
    const bool proj_scene = (scene_kind == PR_SCENE_PROJ || scene_kind == PR_SCENE_PROJ_CROP);
    const bool nn_scene = (scene_kind == PR_SCENE_NN) && !opt.nn_count;     // (an instrumented kd-tree run stays synchronous)
    (void)img;                                                   // (large frames: the asynchronous path sizes its sub-batches to its workspace bound)
    const bool async_ok = P > 0 && opt.solve_mode == PR_SOLVE_DEVICE && opt.raster_mode == 0 && (proj_scene || nn_scene)
                          && (opt.profile == 0 || opt.profile == 3 || (opt.profile == 2 && !sample_call));
    if (!async_ok && P > 0 && opt.solve_mode == PR_SOLVE_HOST && opt.host_worker && opt.profile == 0 && !opt.nn_count) {
        // host solve, nothing to time: the batch goes to the slot's helper thread (see SlotWorker) and this call returns
        PR_TRY(slot_worker_post(sl, tris_dev, n_tris, poses_host, P, W, H, proj, K, scene_kind, scene, crit, roi, results_host, results_dev, sizes_host));
        sl.pending = true; sl.delivered = false; sl.worker_job = true;
        return PR_OK;
    }
    if (!async_ok) {
        // the synchronous path (host solve, timed calls, oversized batches): let the other slot drain first so
        // that a timed launch has the chip to itself, then run to completion; pr_refine_wait has nothing left to do
        for (Slot &o : g->slots) if (o.pending && !o.delivered && o.done) HIP_TRY(hipEventSynchronize(o.done));
        PR_TRY(refine_impl(tris_dev, n_tris, poses_host, P, W, H, proj, K, scene_kind, scene, crit, roi, results_host, results_dev, sizes_host));
        sl.pending = true; sl.delivered = true;
        return PR_OK;
    }
    g->sample_clock++;
    // the host copies of this batch's inputs (the poses are staged from here; the rest is what a re-run needs)
    const size_t in_bytes = (sizeof(pr_mat4) + sizeof(int4) + sizeof(uint32_t)) * (size_t)P;      // [poses][pixel boxes][box offsets]
    PR_TRY(sl.h_in.ensure(in_bytes + 16));
    std::memcpy(sl.h_in.p, poses_host, sizeof(pr_mat4) * P);
    Resubmit &r = sl.again;
    r.tris = tris_dev; r.n_tris = n_tris; r.W = W; r.H = H; r.proj = *proj; std::memcpy(r.K, K, sizeof r.K);
    r.scene_kind = scene_kind; r.crit = crit; r.roi = roi; r.results_dev = results_dev;
    if (scene_kind == PR_SCENE_NN) r.sn = *static_cast<const pr_scene_nn *>(scene);
    else if (scene_kind == PR_SCENE_PROJ_CROP) r.sp = *static_cast<const pr_scene_proj_crop *>(scene);
    else { r.sp.view = *static_cast<const pr_scene_proj *>(scene); r.sp.tl_x = r.sp.tl_y = 0; }
    const int rc = refine_submit_async(sl, tris_dev, n_tris, P, W, H, proj, K, scene_kind, scene, crit, roi, results_host, results_dev);
    if (rc != PR_OK) {                                            // part of the batch may already be queued: do not leave it running
        slot_drain(sl);                                           // behind the caller's back (its buffers may go away next)
        sl.pending = false;
        return rc;
    }
    sl.pending = true;
    return PR_OK;
}

int refine_submit_async(Slot &sl, const pr_triangle *tris_dev, size_t n_tris, uint32_t P, uint32_t W, uint32_t H,
                        const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                        pr_result *results_host, pr_result *results_dev)
{
    const size_t img = (size_t)W * H;
    PR_TRY(slot_streams(sl));
    SceneSel sc;
    std::memset(static_cast<void *>(&sc), 0, sizeof sc);
    // the packed copy of the scene does not depend on the render: when it has to be (re)built that happens on the first side
    // stream (idle until the loop forks), the loop waits for it; batches without pose groups build it in line
    const uint32_t groups_hint = std::max(1u, std::min({ pose_groups_for(scene_kind), 4u, std::min<uint32_t>(P, (uint32_t)std::max(32, opt.sub_batch)) / 32u }));
    for (uint32_t k = 1; k < groups_hint; ++k) PR_TRY(ensure_stream(sl.side[k - 1], &sl.join[k - 1]));
    hipStream_t scene_stream = groups_hint > 1 ? sl.side[0] : sl.stream;
    const Camera cam{ W, H, K[0], K[4], K[2], K[5] };            // kd-tree scenes: the pixel grid of the scene points under this camera
    // nothing of this batch may run beside the loop of a TIMED batch on the other slot -- not even the small checks on the scene stream
    if (scene_stream != sl.stream)
        for (Slot &o : g->slots) if (&o != &sl && o.pending && !o.delivered && o.timed && o.progress_valid) HIP_TRY(hipStreamWaitEvent(scene_stream, o.progress, 0));
    PR_TRY(make_scene(scene_kind, scene, /*want_packed=*/true, sc, &sl.packed, scene_stream, scene_kind == PR_SCENE_NN ? &cam : nullptr, /*verify_now=*/false));
    sl.nn_set = (scene_kind == PR_SCENE_NN) ? sc.nn_set : -1;        // (what make_scene must not rebuild under this batch while it is in flight)

    PR_TRY(ensure_model_box(tris_dev, n_tris));                 // once per triangle buffer ...
    const size_t res_off = ((size_t)P * 4 + 63) & ~(size_t)63;
    sl.flag_off = (res_off + sizeof(pr_result) * P + 63) & ~(size_t)63;
    PR_TRY(sl.h_out.ensure(sl.flag_off + 64));
    void *h_in_dev = nullptr, *h_out_dev = nullptr;              // the pinned staging buffers as the device sees them
    HIP_TRY(hipHostGetDevicePointer(&h_in_dev, sl.h_in.p, 0));
    HIP_TRY(hipHostGetDevicePointer(&h_out_dev, sl.h_out.p, 0));
    HIP_TRY(hipEventRecord(sl.scene_ready, scene_stream));      // (complete as it is recorded unless a cache missed and the scene is being rebuilt on this stream)
    // ... and checked against the buffer's present content by every batch (refine_wait acts on the flag), together with the scene caches: a
    // sampled fingerprint of the caller's arrays against the one taken when the cache was built.  Round 5: both checks ride on the raster launch
    // (BatchCheck, pr_internal.h) -- as launches of their own on the scene stream (a memset, two box kernels, the fingerprint kernel) they cost
    // 1.3-1.5 % of the headline whatever their number (skipped: 277.6 against 273.6 k poses/s on one box; fused into one launch: 274.1 k).
    const bool keys_new = sl.aabb_keys.p == nullptr;
    PR_TRY(sl.aabb_keys.ensure(8 * sizeof(uint32_t)));
    if (keys_new) {                                               // armed once; the last workgroup of every check re-arms the words itself
        HIP_TRY(hipMemsetAsync(sl.aabb_keys.p, 0xff, 6 * sizeof(uint32_t), sl.stream));
        HIP_TRY(hipMemsetAsync(sl.aabb_keys.as<uint32_t>() + 6, 0, 2 * sizeof(uint32_t), sl.stream));
    }
    *reinterpret_cast<volatile uint32_t *>(sl.h_out.as<unsigned char>() + sl.flag_off) = 0u;
    prk::BatchCheck chk{};
    {
        chk.keys = sl.aabb_keys.as<uint32_t>();
        for (int a = 0; a < 6; ++a) chk.expect.v[a] = g->aabb_host[a];
        chk.flag = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(h_out_dev) + sl.flag_off);
        if (opt.scene_cache) {
            if (scene_kind == PR_SCENE_NN) {
                const pr_scene_nn *sn = static_cast<const pr_scene_nn *>(scene);
                chk.fa = reinterpret_cast<const uint32_t *>(sn->pcd); chk.na = (size_t)sn->n_points * sizeof(pr_vec3) / 4;
                chk.fb = reinterpret_cast<const uint32_t *>(sn->nodes); chk.nb = (size_t)sn->n_nodes * sizeof(pr_kdnode) / 4;
                chk.fc = reinterpret_cast<const uint32_t *>(sn->normal); chk.nc = (size_t)sn->n_points * sizeof(pr_vec3) / 4;
                chk.fp_expected = g->nn_sets[sc.nn_set].nndepth.as<uint32_t>() + 12;
            } else if (sl.packed.valid) {
                const pr_scene_proj *sp = static_cast<const pr_scene_proj *>(scene);
                const size_t n = (size_t)sp->width * sp->height;
                const size_t tables = ((sp->width + sp->height) * sizeof(float) + 15) & ~(size_t)15;
                const uint32_t *exact_dev = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(sl.packed.rec.as<float4>() + n) + tables);
                chk.fa = reinterpret_cast<const uint32_t *>(sp->pcd); chk.na = n * sizeof(pr_vec3) / 4;
                chk.fb = reinterpret_cast<const uint32_t *>(sp->normal); chk.nb = n * sizeof(pr_vec3) / 4;
                chk.fp_expected = exact_dev + 1;
            }
        }
    }

    // staging: [poses][boxes][box offsets]; the cloud stride and the grid come from the largest box
    const size_t in_bytes = (sizeof(pr_mat4) + sizeof(int4) + sizeof(uint32_t)) * (size_t)P;
    PR_TRY(sl.poses_bbox.ensure(in_bytes + 16));
    pr_mat4 *h_poses = sl.h_in.as<pr_mat4>();
    int32_t *h_box = reinterpret_cast<int32_t *>(h_poses + P);
    size_t max_area = 1;
    for (uint32_t i = 0; i < P; ++i) {
        pose_bbox_host(g->aabb_host, h_poses[i], *proj, W, H, roi, h_box + 4 * (size_t)i);
        const int32_t *b = h_box + 4 * (size_t)i;
        max_area = std::max(max_area, (size_t)std::max(0, b[2] - b[0] + 1) * (size_t)std::max(0, b[3] - b[1] + 1));   // an off-screen pose has an empty box
    }
    // capacity per hypothesis (the clouds themselves are packed: PoseMeta::start; a multiple of the packing's alignment, so that no sum of rounded sizes exceeds P x cstride)
    const size_t cstride = prk::kCloudAlign ? ((max_area + prk::kCloudAlign - 1) / prk::kCloudAlign) * prk::kCloudAlign : ((max_area + 3) & ~(size_t)3);
    const uint32_t steps = (uint32_t)std::max(1, opt.steps);
    const uint32_t ppb = steps * prk::kPointsPerStep;
    const uint32_t nblk = (uint32_t)((max_area + ppb - 1) / ppb);       // bound from the pixel boxes: capacity of the partial sums
    // grid: one workgroup per block of the largest cloud of the previous batch (workgroups loop if this batch's clouds are
    // larger, surplus workgroups exit at once); the box bound itself would launch ~60 % empty workgroups (-2.5 % poses/s)
    const uint32_t grid_x = g->cloud_hint ? std::min(nblk, (g->cloud_hint + ppb - 1) / ppb) : nblk;

    // Large batches run as consecutive sub-batches that reuse the same depth / cloud / partial-sum memory: the clouds of
    // <= 512 hypotheses (~140 MB touched) stay in the 256 MiB Infinity Cache over their 21 passes (1024 poses as one batch:
    // 199 k poses/s, as 2 x 512: see DESIGN.md)
    // ... and, for large frames, small enough that the depth workspace of a sub-batch stays within ~4 GiB (as the synchronous path
    // bounds its chunks) and the clouds within 2^30 points (offsets are 32-bit): 64 hypotheses per sub-batch at 4096 x 4096, all 512 up to 2 M pixels
    uint32_t sub_cap = (uint32_t)std::max(32, opt.sub_batch);
    sub_cap = (uint32_t)std::max<size_t>(1, std::min<size_t>({ (size_t)sub_cap, ((size_t)4 << 30) / (img * sizeof(int32_t)), ((size_t)1 << 30) / cstride }));     // (2^30 cloud points per sub-batch: 12 GiB of clouds, 24 GiB of kd-tree search state at most)
    const uint32_t n_sub = (P + sub_cap - 1) / sub_cap;
    const uint32_t sub = (P + n_sub - 1) / n_sub;
    PR_TRY(sl.depth.ensure(sizeof(int32_t) * (img + prk::kBoxPack) * sub));
    // the pixel boxes of a sub-batch packed into the depth workspace: box i at h_off[i] ints, its own width as pitch (fill_box_kernel)
    uint32_t *h_off = reinterpret_cast<uint32_t *>(h_box + 4 * (size_t)P);
    if (prk::kBoxPack) {
        size_t acc = 0;
        for (uint32_t i = 0; i < P; ++i) {
            if (i % sub == 0) acc = 0;
            const int32_t *b = h_box + 4 * (size_t)i;
            const size_t area = (size_t)std::max(0, b[2] - b[0] + 1) * (size_t)std::max(0, b[3] - b[1] + 1);
            h_off[i] = (uint32_t)acc;
            acc += (area + prk::kBoxPack - 1) / prk::kBoxPack * prk::kBoxPack;
        }
    }
    PR_TRY(sl.row_count.ensure(sizeof(uint32_t) * (size_t)H * sub));
    PR_TRY(sl.row_off.ensure(sizeof(uint32_t) * (size_t)H * sub));
    PR_TRY(sl.counts.ensure(sizeof(uint32_t) * P));
    PR_TRY(sl.cloud.ensure(sizeof(pr_vec3) * cstride * sub));
    PR_TRY(sl.meta.ensure(sizeof(prk::PoseMeta) * P));
    PR_TRY(sl.partial.ensure(sizeof(float) * prk::kAccStride * (size_t)nblk * sub));
    PR_TRY(sl.dstate.ensure(sizeof(prk::DevIcpState) * P));
    PR_TRY(sl.arrive.ensure(sizeof(uint32_t) * P));
    pr_result *dres = results_dev;
    if (!dres) { PR_TRY(sl.dresults.ensure(sizeof(pr_result) * P)); dres = sl.dresults.as<pr_result>(); }
    // kd-tree scene: winners | slack | queue | two queue counters per hypothesis, indexed like the clouds of one sub-batch (icp_drive
    // has the same layout); the search kernel's grid comes from the box bound, surplus workgroups exit at once
    uint32_t *nn_prev = nullptr;
    const size_t nn_span = cstride * (size_t)sub;
    if (sc.kind == PR_SCENE_NN) {
        sc.nn_split = (sc.nn.rec32 && opt.nn_split) ? 1u : 0u;
        sc.nn_max_points = (uint32_t)std::min<size_t>(max_area, 0xffffffffu);
        if (sc.nn.rec32 && (opt.nn_seed || sc.nn_split)) {
            PR_TRY(sl.nn_prev.ensure(sizeof(uint32_t) * (nn_span * prk::kNNWordsPerPoint + prk::kQCountStride * (size_t)sub) + 64));
            nn_prev = sl.nn_prev.as<uint32_t>();
        }
    }

    hipStream_t st = sl.stream;
    // A TIMED batch (profile 3) stays asynchronous: its render runs under the other slot's loop like any other, but its own loop waits
    // until the other slot's batch is complete, runs as ONE pose group, lets the other slot's next render start only when it has
    // finished, and carries HIP events around every launch -- the timed launches have the chip to themselves as in the synchronous timed
    // path (profile 1), without that path's host round trips and its idle render.
    const bool timed = (opt.profile == 3);
    sl.timed = timed; sl.t_used = 0; sl.t_spans.clear();
    // A batch submitted while the other slot is idle starts a pipeline.  Option start_overlap says when the NEXT batch's render may start
    // in that case: -1 (default) = by the rule below, like any other batch; a pass number releases it earlier (see the option's note).
    bool pipeline_start = true;
    for (Slot &o : g->slots) if (&o != &sl && o.pending && !o.delivered) pipeline_start = false;
    // a failed event creation or record drops the TIMING of this batch (never the batch): t_fail, checked when everything is enqueued
    bool t_fail = false;
    auto t_event = [&]() -> size_t {
        if (sl.t_used == sl.t_events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess || !e) { (void)hipGetLastError(); t_fail = true; return kNoEvent; }
            sl.t_events.push_back(e);
        }
        return sl.t_used++;
    };
    auto t_record = [&](size_t e) { if (e == kNoEvent) { t_fail = true; return; } if (hipEventRecord(sl.t_events[e], st) != hipSuccess) { (void)hipGetLastError(); t_fail = true; } };
    auto t_begin = [&]() -> size_t { const size_t e = t_event(); t_record(e); return e; };
    auto t_end = [&](size_t e0, int kind, uint32_t q0, uint32_t nq, bool edge) { const size_t e1 = t_event(); t_record(e1); if (!t_fail) sl.t_spans.push_back({ e0, e1, kind, q0, nq, edge, false, { 0, 0, 0 } }); };
    pr_mat4 *d_poses = sl.poses_bbox.as<pr_mat4>();
    int4 *d_box = reinterpret_cast<int4 *>(d_poses + P);
    const uint32_t *d_off = prk::kBoxPack ? reinterpret_cast<const uint32_t *>(d_box + P) : nullptr;
    // Both phases are bound by the same units, so a batch that renders while the other slot is in the middle of its ICP loop
    // slows that loop by more than it gains; its render is therefore held back until the other slot has issued pass
    // `overlap_pass` of its (last sub-batch's) loop -- late enough to disturb little, early enough that the GPU never idles.
    for (Slot &o : g->slots)
        if (&o != &sl && o.pending && !o.delivered && o.progress_valid) HIP_TRY(hipStreamWaitEvent(st, o.progress, 0));
    sl.progress_valid = false;
    HIP_TRY(prk::launch_stage_words(h_in_dev, d_poses, in_bytes, st));
    const bool fused = opt.fused_solve != 0;
    const pr_roi none{ 0, 0, 0, 0 };                              // the ROI is already part of the host-computed boxes
    for (uint32_t q0 = 0; q0 < P; q0 += sub) {
        const uint32_t nq = std::min(sub, P - q0);
        prk::PoseMeta *meta = sl.meta.as<prk::PoseMeta>() + q0;
        prk::DevIcpState *dstate = sl.dstate.as<prk::DevIcpState>() + q0;
        uint32_t *arrive = sl.arrive.as<uint32_t>() + q0;
        size_t te = timed ? t_begin() : 0;
        // (the first render carries the batch's checks: the fingerprint's expected value belongs to the scene build, so the scene comes first --
        // an event that is complete when recorded unless a cache missed)
        // (ADVICE r05 asked whether this wait costs a rebuilt scene its overlap with the render: it does not -- make_scene reads the rebuilt form's
        // `exact` flag / tree depth back before it returns, so a rebuild has finished on the host's clock before the render is even enqueued; on a
        // cache hit the event is complete when recorded)
        if (q0 == 0) HIP_TRY(hipStreamWaitEvent(st, sl.scene_ready, 0));
        HIP_TRY(prk::launch_render_boxes(tris_dev, (uint32_t)n_tris, d_poses + q0, nq, nullptr, d_box + q0, sl.depth.as<int32_t>(),
                                         sl.row_count.as<uint32_t>(), sl.row_off.as<uint32_t>(), sl.counts.as<uint32_t>() + q0, W, H, *proj, none, st,
                                         /*compute_boxes=*/false, meta, dstate, arrive, (uint32_t)cstride, d_off ? d_off + q0 : nullptr, q0 == 0 ? &chk : nullptr));
        if (timed) { t_end(te, kSpanRender, q0, nq, false); te = t_begin(); }
        HIP_TRY(prk::launch_emit_box(sl.depth.as<int32_t>(), nq, W, H, d_box + q0, K[0], K[4], K[2], K[5], sl.row_count.as<uint32_t>(),
                                     sl.row_off.as<uint32_t>(), sl.cloud.as<pr_vec3>(), cstride, st, prk::kCloudAlign ? meta : nullptr, d_off ? d_off + q0 : nullptr));
        if (timed) t_end(te, kSpanCloud, q0, nq, false);
        if (timed && q0 == 0)                                     // the timed loop starts when the other slot's batch is complete
            for (Slot &o : g->slots) if (&o != &sl && o.pending && !o.delivered && o.done) HIP_TRY(hipStreamWaitEvent(st, o.done, 0));

        // the iteration loop: (max_iteration+1) x [pass (+ fused finalize/solve)], pose groups on the slot's side streams
        const uint32_t n_groups = timed ? 1u : std::max(1u, std::min({ pose_groups_for(scene_kind), 4u, nq / 32u }));
        auto group_begin = [&](uint32_t grp) { return (uint32_t)(((uint64_t)nq * grp) / n_groups); };
        if (nn_prev) HIP_TRY(prk::launch_fill_i32(reinterpret_cast<int32_t *>(nn_prev + 6 * nn_span), (size_t)prk::kQCountStride * nq, 0, st));
        if (n_groups > 1) {
            for (uint32_t k = 1; k < n_groups; ++k) PR_TRY(ensure_stream(sl.side[k - 1], &sl.join[k - 1]));
            HIP_TRY(hipEventRecord(sl.fork, st));
            for (uint32_t k = 1; k < n_groups; ++k) HIP_TRY(hipStreamWaitEvent(sl.side[k - 1], sl.fork, 0));
        }
        // When may the OTHER slot start rendering?  Measured optimum (21 passes of obj_06, overlap_pass swept per batch size):
        // pass 4 or earlier at 128 hypotheses, 9-10 at 256, 14-16 at 384, 16 at 512 -- i.e. when about 2800 hypothesis-passes of
        // this loop are left (never fewer than 5 passes): that much loop work is what a render hides behind without stretching the
        // passes it runs beside.  The render's own weight scales that figure: triangles per hypothesis against cloud points per
        // hypothesis (31 468 and 27 400 there; cloud_hint = the largest cloud of the previous batch).
        uint32_t auto_overlap = 0;
        {
            const double weight = ((double)std::max<size_t>(n_tris, 1) / 31468.0) * (27400.0 / (double)std::max(g->cloud_hint, 1000u));
            double left = std::max(5.0, 2800.0 * weight / (double)std::max(nq, 1u));   // passes of this sub-batch's loop still to run
            // Since the clouds are packed (end of round 4) a render disturbs the loop beside it less, as long as the clouds of BOTH slots stay in
            // the 256 MiB Infinity Cache: then the render is released with 13 passes to go (256 hypotheses: passes 5-9 give 272 k poses/s, the rule
            // above -- pass 10 -- 268 k; 384: pass 8 277 k against 272 k at 14; 512, whose two slots' clouds do not fit: 275 k at pass 16, 263 k at 10).
            if (2.0 * (double)nq * (double)std::max(g->cloud_hint, 1000u) * sizeof(pr_vec3) <= 240.0e6) left = std::max(left, 13.0);
            const double passes = (double)crit.max_iteration + 1.0;
            auto_overlap = left >= passes ? 0u : (uint32_t)(passes - left + 0.5);
            // kd-tree scenes: the loop is seven times a render and its first passes are the heavy ones -- the other slot's render (and with it
            // that slot's own first passes) should start at once: 37.3 / 37.5 k against 36.0 k poses/s for the rule above (pass 1 / 2 / 3 / 6: 37.2 /
            // 37.2 / 37.1 / 37.0 k; 10: 35.7 k; 16: 34.0 k)
            if (scene_kind == PR_SCENE_NN) auto_overlap = 0u;
        }
        prk::IcpBatch b{};
        b.cloud = sl.cloud.as<pr_vec3>(); b.nblk = nblk; b.grid_x = grid_x; b.steps = steps;
        if (nn_prev) {
            b.nn_prev = nn_prev; b.nn_slack = reinterpret_cast<float *>(nn_prev + nn_span);
            b.nn_queue = reinterpret_cast<uint2 *>(nn_prev + 2 * nn_span); b.nn_queue2 = reinterpret_cast<uint2 *>(nn_prev + 4 * nn_span); b.nn_qcount = nn_prev + 6 * nn_span;
        }
        for (uint32_t it = 0; it <= (uint32_t)crit.max_iteration; ++it) {
            for (uint32_t grp = 0; grp < n_groups; ++grp) {
                const uint32_t p0 = group_begin(grp), np = group_begin(grp + 1) - p0;
                hipStream_t gs = grp ? sl.side[grp - 1] : st;
                prk::IcpBatch bb = b;
                bb.meta = meta + p0; bb.partial = sl.partial.as<float>() + (size_t)p0 * nblk * prk::kAccStride;
                if (bb.nn_qcount) bb.nn_qcount += prk::kQCountStride * (size_t)p0;
                bb.iter = it;
                if (fused) { bb.fused = 1; bb.crit = crit; bb.st = dstate + p0; bb.arrive = arrive + p0; }
                bb.score_only = (it == (uint32_t)crit.max_iteration) ? 1u : 0u;
                if (timed) {                                         // (one group: gs == st)
                    const size_t e0 = t_begin();
                    // a kd-tree pass is four kernels: three more events between them give each kernel's own time (pr_profile_nn)
                    const bool marks = sc.kind == PR_SCENE_NN && sc.nn_split && bb.nn_prev && np <= 32768u;
                    size_t mi[3] = { 0, 0, 0 }; hipEvent_t me[3] = { nullptr, nullptr, nullptr };
                    bool marks_ok = marks;
                    if (marks) for (int k = 0; k < 3; ++k) { mi[k] = t_event(); if (mi[k] == kNoEvent) marks_ok = false; else me[k] = sl.t_events[mi[k]]; }
                    HIP_TRY(launch_pass(bb, sc, np, gs, marks_ok ? me : nullptr));
                    t_end(e0, kSpanIcp, q0 + p0, np, it == 0 || it == (uint32_t)crit.max_iteration);
                    if (marks_ok && !t_fail) { Slot::TSpan &ts = sl.t_spans.back(); ts.marks = true; for (int k = 0; k < 3; ++k) ts.m[k] = mi[k]; }
                } else HIP_TRY(launch_pass(bb, sc, np, gs));
                if (!fused) HIP_TRY(prk::launch_icp_finalize_solve(bb.partial, meta + p0, nblk, steps, dstate + p0, crit, it, np, gs));
            }
            if (q0 + sub >= P && it == (timed ? (uint32_t)crit.max_iteration : std::min<uint32_t>((uint32_t)crit.max_iteration, (pipeline_start && opt.start_overlap >= 0) ? (uint32_t)opt.start_overlap : (opt.overlap_pass >= 0 ? (uint32_t)opt.overlap_pass : auto_overlap)))) {
                HIP_TRY(hipEventRecord(sl.progress, st));
                sl.progress_valid = true;
            }
        }
        for (uint32_t k = 1; k < n_groups; ++k) { HIP_TRY(hipEventRecord(sl.join[k - 1], sl.side[k - 1])); HIP_TRY(hipStreamWaitEvent(st, sl.join[k - 1], 0)); }
    }
    HIP_TRY(prk::launch_pack_export(sl.dstate.as<prk::DevIcpState>(), dres, sl.counts.as<uint32_t>(), static_cast<uint32_t *>(h_out_dev),
                                    results_host ? reinterpret_cast<pr_result *>(static_cast<unsigned char *>(h_out_dev) + res_off) : nullptr, P, st));
    HIP_TRY(hipEventRecord(sl.done, st));
    if (t_fail) { sl.timed = false; sl.t_spans.clear(); sl.t_used = 0; g->stat_timing_dropped++; }   // the batch runs; its timing is dropped, and counted
    return PR_OK;
}

}  // namespace prr

using namespace prr;

extern "C" {

int pr_render(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, size_t n_poses, size_t width, size_t height,
              const pr_mat4 *proj, pr_roi roi, int32_t *depth_dev_out)
{
    PR_ENTER();
    if (depth_dev_out) note_write(depth_dev_out, sizeof(int32_t) * n_poses * ((roi.width > 0 && roi.height > 0) ? (size_t)roi.width * roi.height : width * height));
    PR_TRY(render_impl(tris_dev, n_tris, poses_host, n_poses, width, height, proj, roi, depth_dev_out, true));
    HIP_TRY(hipStreamSynchronize(g->stream));            // renderer.cu:295 cudaDeviceSynchronize
    drain_spans();
    return PR_OK;
}

int pr_render_to_host(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, size_t n_poses, size_t width, size_t height,
                      const pr_mat4 *proj, pr_roi roi, int32_t *depth_host_out)
{
    PR_ENTER();
    size_t rw = width, rh = height;
    if (roi.width > 0 && roi.height > 0) { rw = (size_t)roi.width; rh = (size_t)roi.height; }
    PR_TRY(g->depth.ensure(sizeof(int32_t) * std::max<size_t>(1, n_poses * rw * rh)));
    PR_TRY(render_impl(tris_dev, n_tris, poses_host, n_poses, width, height, proj, roi, g->depth.as<int32_t>(), true));
    HIP_TRY(hipMemcpyAsync(depth_host_out, g->depth.p, sizeof(int32_t) * n_poses * rw * rh, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    drain_spans();
    return PR_OK;
}

int pr_depth2cloud_i32(const int32_t *depth_dev, uint32_t width, uint32_t height, const float K[9], uint32_t stride, uint32_t tl_x,
                       uint32_t tl_y, pr_vec3 **cloud_dev_out, uint32_t *n_points)
{
    PR_ENTER();
    return depth2cloud_impl<int32_t>(depth_dev, width, height, K, stride, tl_x, tl_y, cloud_dev_out, n_points);
}

int pr_depth2cloud_u16(const uint16_t *depth_dev, uint32_t width, uint32_t height, const float K[9], uint32_t stride, uint32_t tl_x,
                       uint32_t tl_y, pr_vec3 **cloud_dev_out, uint32_t *n_points)
{
    PR_ENTER();
    return depth2cloud_impl<uint16_t>(depth_dev, width, height, K, stride, tl_x, tl_y, cloud_dev_out, n_points);
}

int pr_refine_batch_roi(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                        const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                        pr_result *results_host, uint32_t *cloud_sizes_host)
{
    PR_ENTER();
    if (!results_host) { set_error("pr_refine_batch: results_host is null"); return PR_ERR_INVALID; }
    if (n_poses == 0) return PR_OK;
    const int slot = free_slot();
    if (slot < 0) { set_error("pr_refine_batch: both asynchronous slots hold unfinished batches (pr_refine_wait one of them first)"); return PR_ERR_INVALID; }
    PR_TRY(refine_submit(slot, tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, roi, results_host, nullptr, cloud_sizes_host));
    return refine_wait(slot);
}

int pr_refine_batch(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                    const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit,
                    pr_result *results_host, uint32_t *cloud_sizes_host)
{
    return pr_refine_batch_roi(tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, pr_roi{ 0, 0, 0, 0 }, results_host, cloud_sizes_host);
}

int pr_refine_batch_dev(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                        const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit,
                        pr_result *results_dev, uint32_t *cloud_sizes_host)
{
    PR_ENTER();
    if (!results_dev) { set_error("pr_refine_batch_dev: results_dev is null"); return PR_ERR_INVALID; }
    if (n_poses == 0) return PR_OK;
    const int slot = free_slot();
    if (slot < 0) { set_error("pr_refine_batch_dev: both asynchronous slots hold unfinished batches (pr_refine_wait one of them first)"); return PR_ERR_INVALID; }
    PR_TRY(refine_submit(slot, tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, pr_roi{ 0, 0, 0, 0 }, nullptr, results_dev, cloud_sizes_host));
    return refine_wait(slot);
}

int pr_refine_submit_roi(int slot, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                         const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                         pr_result *results_host, pr_result *results_dev, uint32_t *cloud_sizes_host)
{
    trace_mark("pr_refine_submit: enter");
    int rc;
    { PR_ENTER(); rc = refine_submit(slot, tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, roi, results_host, results_dev, cloud_sizes_host); }
    trace_mark("pr_refine_submit: exit");
    return rc;
}

int pr_refine_submit(int slot, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                     const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit,
                     pr_result *results_host, pr_result *results_dev, uint32_t *cloud_sizes_host)
{
    return pr_refine_submit_roi(slot, tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, pr_roi{ 0, 0, 0, 0 }, results_host, results_dev, cloud_sizes_host);
}

int pr_refine_wait(int slot)
{
    trace_mark("pr_refine_wait: enter");
    int rc;
    { PR_ENTER(); rc = refine_wait(slot); }
    trace_mark("pr_refine_wait: exit");
    return rc;
}

}  // extern "C"
