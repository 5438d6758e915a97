// pr_scene.cpp -- scene caches (packed projective record, kd-tree traversal records, pixel grid) and device-side scene preparation
#include "pr_runtime.h"

namespace prr {

// Packed copy of a projective scene in `pc`: reused while the caller's arrays are unchanged as far as the library can tell
// (option scene_cache, WriteLog above), rebuilt otherwise.  A rebuild also learns (one 4-byte read-back) whether the pcd array
// is exactly what dep2pcd produces -- only then may the packed form stand in for it.
// Synchronous callers (pr_icp_*, the synchronous fused path -- everything the C++ adapters of the reference's API reach) check a cache hit on the
// spot, and since round 6 against a fingerprint of EVERY word of the source arrays (`slot[2]`, taken when the cache was built; `slot[3]` is this
// call's): the reference reads the caller's arrays at every call (depth_scene.h:29-48), so an in-place edit through a pointer the adapters handed
// out must be seen even when nobody announced it.  ~20 us per call; the asynchronous path compares the 4096-word SAMPLE
// (`slot[0]`, verdict in `slot[1]`) inside its raster launch and documents pr_invalidate for edits the sample cannot see.
int fingerprint_differs(const void *a, size_t ab, const void *b, size_t bb, const void *c, size_t cb, uint32_t *slot, hipStream_t st, bool &differs)
{
    HIP_TRY(prk::launch_scene_fingerprint_full(a, ab, b, bb, c, cb, slot + 3, st));
    // both words come back through a kernel's stores into pinned memory (no copy command on the per-batch path: see icp_drive's result block)
    uint32_t v[2] = { 0u, 1u };
    PR_TRY(read_back_words(slot + 2, v, 2, st));
    differs = v[0] != v[1];
    return PR_OK;
}
int ensure_packed(PackedCache &pc, const pr_scene_proj &s, uint32_t tl_x, uint32_t tl_y, hipStream_t st, bool verify_now)
{
    const size_t n = (size_t)s.width * s.height;
    const float k[4] = { s.K[0], s.K[4], s.K[2], s.K[5] };
    const bool same = pc.valid && pc.pcd == s.pcd && pc.normal == s.normal && pc.w == s.width && pc.h == s.height &&
                      std::memcmp(pc.k, k, sizeof k) == 0 && pc.tl[0] == tl_x && pc.tl[1] == tl_y;
    if (same && opt.scene_cache && !g_writes.written_since(pc.gen, s.pcd, n * sizeof(pr_vec3)) &&
        !g_writes.written_since(pc.gen, s.normal, n * sizeof(pr_vec3))) {
        bool stale = false;
        if (verify_now) {
            const size_t tb = ((s.width + s.height) * sizeof(float) + 15) & ~(size_t)15;
            uint32_t *ex = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(pc.rec.as<float4>() + n) + tb);
            PR_TRY(fingerprint_differs(s.pcd, n * sizeof(pr_vec3), s.normal, n * sizeof(pr_vec3), nullptr, 0, ex + 1, st, stale));
        }
        if (!stale) { pc.gen = g_writes.now(); return PR_OK; }
    }
    pc.valid = false;
    const uint64_t gen = g_writes.now();
    const size_t tables = ((s.width + s.height) * sizeof(float) + 15) & ~(size_t)15;
    PR_TRY(pc.rec.ensure(n * sizeof(float4) + tables + 32));      // behind the tables: [0] exact flag [1] sampled fingerprint [2] its verdict (asynchronous check) [3] full fingerprint [4] a call's full fingerprint
    float *colf = reinterpret_cast<float *>(pc.rec.as<float4>() + n);
    float *rowf = colf + s.width;
    uint32_t *exact_dev = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(colf) + tables);
    HIP_TRY(prk::launch_pack_proj_scene(s.pcd, s.normal, pc.rec.as<float4>(), n, colf, rowf, (uint32_t)s.width, (uint32_t)s.height,
                                        k[0], k[1], k[2], k[3], tl_x, tl_y, exact_dev, st));
    HIP_TRY(prk::launch_scene_fingerprint(s.pcd, n * sizeof(pr_vec3), s.normal, n * sizeof(pr_vec3), nullptr, 0, exact_dev + 1, nullptr, false, st));
    HIP_TRY(prk::launch_scene_fingerprint_full(s.pcd, n * sizeof(pr_vec3), s.normal, n * sizeof(pr_vec3), nullptr, 0, exact_dev + 3, st));
    uint32_t exact = 0;
    PR_TRY(read_back_words(exact_dev, &exact, 1, st));
    pc.pcd = s.pcd; pc.normal = s.normal; pc.w = s.width; pc.h = s.height; std::memcpy(pc.k, k, sizeof k); pc.tl[0] = tl_x; pc.tl[1] = tl_y;
    pc.gen = gen; pc.exact = exact != 0; pc.valid = true;
    return PR_OK;
}

void drain_all_slots();
int make_scene(int kind, const void *scene, bool want_packed, SceneSel &out, PackedCache *pc_in, hipStream_t st, const Camera *cam,
               bool verify_now)
{
    PackedCache &pc = pc_in ? *pc_in : g->packed;
    if (!st) st = g->stream;
    out.kind = kind;
    if (kind == PR_SCENE_PROJ || kind == PR_SCENE_PROJ_CROP) {
        const pr_scene_proj *s = static_cast<const pr_scene_proj *>(scene);       // pr_scene_proj_crop starts with the plain view
        if (!s || !s->pcd || !s->normal || s->width == 0 || s->height == 0) { set_error("invalid pr_scene_proj"); return PR_ERR_INVALID; }
        if (s->width > 0x7fffffffull || s->height > 0x7fffffffull) { set_error("pr_scene_proj: frame too large"); return PR_ERR_INVALID; }
        uint32_t tl_x = 0, tl_y = 0;
        if (kind == PR_SCENE_PROJ_CROP) { const pr_scene_proj_crop *c = static_cast<const pr_scene_proj_crop *>(scene); tl_x = c->tl_x; tl_y = c->tl_y; }
        out.kind = PR_SCENE_PROJ;
        out.aos = prk::SceneProjAoS{ (uint32_t)s->width, (uint32_t)s->height, s->max_dist_diff, s->K[0], s->K[4], s->K[2], s->K[5],
                                     (float)tl_x, (float)tl_y, s->pcd, s->normal };
        out.packed = false;
        if (want_packed) {
            PR_TRY(ensure_packed(pc, *s, tl_x, tl_y, st, verify_now));
            if (pc.exact) {                                         // else: the caller's pcd is not dep2pcd's -- use the arrays as they are
                const size_t n = (size_t)s->width * s->height;
                float *colf = reinterpret_cast<float *>(pc.rec.as<float4>() + n);
                out.packed = true;
                out.pk = prk::SceneProjPacked{ (uint32_t)s->width, (uint32_t)s->height, s->max_dist_diff, s->K[0], s->K[4], s->K[2], s->K[5],
                                               (float)tl_x, (float)tl_y, pc.rec.as<float4>(), colf, colf + s->width };
            }
        }
        return PR_OK;
    }
    if (kind == PR_SCENE_NN) {
        const pr_scene_nn *s = static_cast<const pr_scene_nn *>(scene);
        if (!s || !s->pcd || !s->normal || !s->nodes || s->n_nodes == 0 || s->n_points == 0) { set_error("invalid pr_scene_nn"); return PR_ERR_INVALID; }
        // which of the sets of derived records: one that holds this scene and is still good, else a victim -- a stale set of the same arrays, an
        // empty one, the least recently used one that no batch in flight reads; only when every set is being read does a slot have to drain first
        auto matches = [&](const NNDerived &d) {
            return d.valid && d.pcd == s->pcd && d.normal == s->normal && d.nodes == s->nodes && d.n_points == s->n_points && d.n_nodes == s->n_nodes &&
                   s->max_dist_diff * 1.01f <= d.frame_margin;       // (a smaller radius keeps its pruning inside the larger frame; a larger one would lose it outside the old)
        };
        auto in_flight = [&](int set) { for (Slot &o : g->slots) if (o.pending && !o.delivered && o.nn_set == set) return true; return false; };
        auto drain_readers = [&](int set) { for (Slot &o : g->slots) if (o.pending && !o.delivered && o.nn_set == set) slot_drain(o); };
        int pick = -1;
        bool hit = false;
        for (int i = 0; i < kNNSets && pick < 0; ++i) {
            const NNDerived &d = g->nn_sets[i];
            if (matches(d) && opt.scene_cache && !g_writes.written_since(d.gen, s->pcd, (size_t)s->n_points * sizeof(pr_vec3)) &&
                !g_writes.written_since(d.gen, s->nodes, (size_t)s->n_nodes * sizeof(pr_kdnode))) { pick = i; hit = true; }
        }
        if (hit && verify_now) {
            bool stale = false;
            PR_TRY(fingerprint_differs(s->pcd, (size_t)s->n_points * sizeof(pr_vec3), s->nodes, (size_t)s->n_nodes * sizeof(pr_kdnode), s->normal,
                                       (size_t)s->n_points * sizeof(pr_vec3), g->nn_sets[pick].nndepth.as<uint32_t>() + 12, g->stream, stale));
            hit = !stale;
        }
        if (!hit) {
            if (pick < 0) for (int i = 0; i < kNNSets && pick < 0; ++i) { const NNDerived &d = g->nn_sets[i]; if (d.valid && d.pcd == s->pcd && d.nodes == s->nodes) pick = i; }   // this scene's arrays, rewritten
            if (pick < 0) for (int i = 0; i < kNNSets && pick < 0; ++i) if (!g->nn_sets[i].valid) pick = i;
            if (pick < 0) { for (int i = 0; i < kNNSets; ++i) if (!in_flight(i) && (pick < 0 || g->nn_sets[i].used < g->nn_sets[pick].used)) pick = i; }
            if (pick < 0) { pick = 0; for (int i = 1; i < kNNSets; ++i) if (g->nn_sets[i].used < g->nn_sets[pick].used) pick = i; }
        }
        NNDerived &nc = g->nn_sets[pick];
        nc.used = ++g->nn_clock;
        out.nn_set = pick;
        if (hit) {
            nc.gen = g_writes.now();                                // (the normals are read through the caller's pointer, never copied)
        } else {
            drain_readers(pick);                                    // (a batch in flight that reads this very set: it finishes first)
            nc.valid = false; nc.grid_valid = false;
            const uint64_t gen = g_writes.now();
            PR_TRY(nc.topo.ensure((size_t)s->n_nodes * sizeof(int4)));
            PR_TRY(nc.bmin.ensure((size_t)s->n_nodes * sizeof(float4)));
            PR_TRY(nc.bmax.ensure((size_t)s->n_nodes * sizeof(float4)));
            PR_TRY(nc.pts.ensure(((size_t)s->n_points + 16u) * sizeof(float4)));   // + 16: a leaf task of the walk reads its ten slots whatever the leaf holds (nn_tree_wide_kernel)
            PR_TRY(nc.nnrec.ensure((size_t)s->n_nodes * 4 * sizeof(float4)));
            PR_TRY(nc.nndepth.ensure(24 * sizeof(uint32_t)));      // [0] depth [1] rec32 valid [2..7] rec32 frame [8] wide valid [9] wide nodes [12] sampled fingerprint [13] its verdict [14] full fingerprint [15] a call's full fingerprint [16..19] wide frame
            PR_TRY(nc.nnrec32.ensure((size_t)s->n_nodes * 2 * sizeof(uint4)));
            PR_TRY(nc.nndesc.ensure((size_t)s->n_nodes * sizeof(uint2)));
            // wide records: one 128-byte line per wide node
            PR_TRY(nc.nnwide.ensure(prk::nn_wide_capacity(s->n_nodes) * 128));
            PR_TRY(nc.nnwq.ensure(prk::nn_wide_scratch_words(s->n_nodes) * sizeof(uint32_t)));
            HIP_TRY(prk::launch_build_nn_accel(s->nodes, s->n_nodes, s->pcd, s->n_points, nc.topo.as<int4>(), nc.bmin.as<float4>(),
                                               nc.bmax.as<float4>(), nc.pts.as<float4>(), nc.nnrec.as<float4>(), nc.nnrec32.as<uint4>(),
                                               nc.nndesc.as<uint2>(), nc.nndepth.as<uint32_t>(), g->stream, s->max_dist_diff * 1.01f));
            HIP_TRY(prk::launch_scene_fingerprint(s->pcd, (size_t)s->n_points * sizeof(pr_vec3), s->nodes, (size_t)s->n_nodes * sizeof(pr_kdnode), s->normal,
                                                  (size_t)s->n_points * sizeof(pr_vec3), nc.nndepth.as<uint32_t>() + 12, nullptr, false, g->stream));
            HIP_TRY(prk::launch_scene_fingerprint_full(s->pcd, (size_t)s->n_points * sizeof(pr_vec3), s->nodes, (size_t)s->n_nodes * sizeof(pr_kdnode), s->normal,
                                                       (size_t)s->n_points * sizeof(pr_vec3), nc.nndepth.as<uint32_t>() + 14, g->stream));
            // the wide records, eight levels per synchronisation (8^8 wide nodes: deeper only for a lopsided tree)
            for (uint32_t level = 0; ; level += 8) {
                HIP_TRY(prk::launch_nn_wide_levels(nc.topo.as<int4>(), nc.bmin.as<float4>(), nc.bmax.as<float4>(), s->n_nodes, s->n_points, nc.nnwide.as<uint4>(),
                                                   nc.nnwq.as<uint32_t>(), nc.nndepth.as<uint32_t>(), level, 8, g->stream));
                PR_TRY(read_back_words(nc.nndepth.p, nc.info, (uint32_t)(sizeof nc.info / sizeof(uint32_t)), g->stream));
                if (nc.info[8] != 2u) break;
                if (level + 8 >= 64) { nc.info[8] = 0u; break; }    // (as before: more than 64 wide levels -> the binary records)
            }
            nc.pcd = s->pcd; nc.normal = s->normal; nc.nodes = s->nodes; nc.n_points = s->n_points; nc.n_nodes = s->n_nodes; nc.gen = gen; nc.valid = true;
            nc.frame_margin = s->max_dist_diff * 1.01f;
        }
        const uint32_t *info = nc.info;
        const uint32_t depth = info[0];
        if (depth >= 0x7fffffffu) { nc.valid = false; set_error("pr_scene_nn: the nodes do not form a consistent tree (child / parent links disagree or point outside the array)"); return PR_ERR_INVALID; }
        uint32_t lds = (uint32_t)std::max(0, opt.nn_lds_nodes);
        lds = std::min(lds, s->n_nodes);
        lds = std::min<uint32_t>(lds, 8192);                      // <= 128 KiB of LDS
        // pending far children on the stack never exceed the tree depth; deeper trees use the stackless walk
        uint32_t stack = 0;
        if (opt.nn_stack) stack = (depth <= 16) ? 16u : ((depth <= 24) ? 24u : 0u);
        if (stack) lds = std::min<uint32_t>({ (uint32_t)std::max(0, opt.nn_lds_records), s->n_nodes, (65536u - stack * 2048u) / 64u });   // stacks + records <= 64 KiB
        out.nn = prk::SceneNNDev{ s->max_dist_diff, nc.topo.as<int4>(), nc.bmin.as<float4>(), nc.bmax.as<float4>(), nc.pts.as<float4>(),
                                  s->pcd, s->normal, s->n_nodes, lds, nc.nnrec.as<float4>(), stack, nullptr, { 0, 0, 0 }, { 0, 0, 0 }, nc.nndesc.as<uint2>() };
        if (stack && opt.nn_compact && info[1] == 1u) {
            out.nn.rec32 = nc.nnrec32.as<uint4>();
            for (int a = 0; a < 3; ++a) { std::memcpy(&out.nn.qmin[a], &info[2 + a], 4); std::memcpy(&out.nn.qscale[a], &info[5 + a], 4); }
            if (opt.nn_wide && info[8] == 1u && info[9] > 0u) {
                out.nn.wide = nc.nnwide.as<uint4>(); out.nn.n_wide = info[9];
                for (int a = 0; a < 3; ++a) std::memcpy(&out.nn.wmin[a], &info[16 + a], 4);
                std::memcpy(&out.nn.wscale, &info[19], 4);
            }
        }
        // a bare ICP call has no camera of its own: the scene's hint, if it carries one (pose_refine.h)
        Camera hinted;
        if (!cam && s->cam_magic == PR_SCENE_NN_CAM_MAGIC && s->cam_w && s->cam_h && s->cam_fx > 0.0f && s->cam_fy > 0.0f) { hinted = Camera{ s->cam_w, s->cam_h, s->cam_fx, s->cam_fy, s->cam_cx, s->cam_cy }; cam = &hinted; }
        // pixel grid of the scene points under the hypotheses' camera (fused paths), or under the camera the scene says it was made with.  Usable
        // when every scene point owns a cell -- a Scene_nn made from a depth image with these intrinsics -- else the tree alone.
        if (cam && opt.nn_grid && out.nn.rec32 && (size_t)cam->w * cam->h <= ((size_t)1 << 24)) {
            const float gk[4] = { cam->fx, cam->fy, cam->cx, cam->cy };
            if (!(nc.grid_valid && nc.gw == cam->w && nc.gh == cam->h && std::memcmp(nc.gk, gk, sizeof gk) == 0)) {
                drain_readers(pick);
                const size_t cells = (size_t)cam->w * cam->h;
                PR_TRY(nc.nn_cells.ensure(cells * sizeof(int32_t) + 16));
                PR_TRY(nc.nn_grid.ensure(prk::nn_grid_cells(cam->w, cam->h) * sizeof(float4)));
                uint32_t *flag = reinterpret_cast<uint32_t *>(nc.nn_cells.as<int32_t>() + cells);
                HIP_TRY(prk::launch_build_nn_grid(s->pcd, s->n_points, cam->w, cam->h, gk[0], gk[1], gk[2], gk[3], nc.nn_cells.as<int32_t>(),
                                                  nc.nn_grid.as<float4>(), flag, g->stream));
                uint32_t usable = 0;
                PR_TRY(read_back_words(flag, &usable, 1, g->stream));
                nc.grid_valid = true; nc.grid_usable = usable != 0; nc.gw = cam->w; nc.gh = cam->h; std::memcpy(nc.gk, gk, sizeof gk);
            }
            if (nc.grid_usable) {
                out.nn.grid = nc.nn_grid.as<float4>(); out.nn.gw = cam->w; out.nn.gh = cam->h; out.nn.gfx = gk[0]; out.nn.gfy = gk[1]; out.nn.gcx = gk[2]; out.nn.gcy = gk[3];
                const size_t w4 = (cam->w + 3) / 4, h4 = (cam->h + 3) / 4, w16 = (w4 + 3) / 4, h16 = (h4 + 3) / 4;
                out.nn.pyr4 = out.nn.grid + (size_t)cam->w * cam->h; out.nn.pyr16 = out.nn.pyr4 + w4 * h4; out.nn.pyr64 = out.nn.pyr16 + w16 * h16;
            }
        }
        return PR_OK;
    }
    set_error("unknown scene kind %d", kind);
    return PR_ERR_INVALID;
}

// KDTree_cpu::build_tree on the device (kd_build.hip): three launches per level over all points; the host reads the 32-byte control record back
// after twelve levels and then every four (launches past the last level return at once)
int kd_build_dev(pr_vec3 *pcd, pr_vec3 *nrm, uint32_t n, int max_leaf, pr_kdnode *nodes, size_t cap, uint32_t *n_nodes)
{
    if (n == 0 || cap == 0) { set_error("kd-tree build: no points"); return PR_ERR_INVALID; }
    if (n >= 0x7fffffffu) { set_error("kd-tree build: %u points are more than the node records' int ranges hold", n); return PR_ERR_INVALID; }
    const uint32_t cap32 = (uint32_t)std::min<size_t>(cap, 0x7fffffff);
    prk::KdWork w;
    PR_TRY(g->kd_scratch.ensure(prk::kd_work_bytes(n, cap32, max_leaf, nullptr)));
    w.base = g->kd_scratch.p;
    prk::kd_work_bytes(n, cap32, max_leaf, &w);
    PR_TRY(g->kd_tmp.ensure(sizeof(pr_vec3) * 2 * (size_t)n));
    HIP_TRY(prk::launch_kd_init(w, nodes, cap32, pcd, n, max_leaf, g->stream));
    prk::KdCtrl ctrl{};
    uint32_t level = 0;
    for (uint32_t until = 12; ; until += 4) {
        for (; level < until; ++level) HIP_TRY(prk::launch_kd_level(w, nodes, cap32, pcd, n, max_leaf, level, g->stream));
        PR_TRY(read_back_words(w.ctrl[level & 1u], &ctrl, (uint32_t)(sizeof ctrl / sizeof(uint32_t)), g->stream));
        if (ctrl.error == 2u) { set_error("kd-tree build: a level holds more nodes than the work arrays were sized for (%u points, max_leaf %d) -- an internal bound, not the caller's capacity", n, max_leaf); return PR_ERR_INVALID; }
        if (ctrl.error) { set_error("kd-tree build: node capacity %u too small", cap32); return PR_ERR_NOMEM; }
        if (ctrl.done) break;
        if (level >= 4096) { set_error("kd-tree build: more than 4096 levels (max_leaf %d)", max_leaf); return PR_ERR_INVALID; }
    }
    pr_vec3 *tp = g->kd_tmp.as<pr_vec3>(), *tn = tp + n;
    HIP_TRY(prk::launch_kd_permute(w, level, pcd, nrm, n, tp, tn, g->stream));
    HIP_TRY(hipMemcpyAsync(pcd, tp, sizeof(pr_vec3) * n, hipMemcpyDeviceToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(nrm, tn, sizeof(pr_vec3) * n, hipMemcpyDeviceToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (n_nodes) *n_nodes = ctrl.next;
    return PR_OK;
}

template <typename T>
int scene_nn_prepare_dev_t(const T *depth, const float K[9], uint32_t W, uint32_t H, int max_leaf, pr_vec3 *pcd, pr_vec3 *nrm,
                           pr_kdnode *nodes, size_t cap, uint32_t *n_points, uint32_t *n_nodes)
{
    const size_t px = (size_t)W * H;
    PR_TRY(g->nn_full.ensure(sizeof(pr_vec3) * 2 * px));
    PR_TRY(g->row_count.ensure(sizeof(uint32_t) * H));
    PR_TRY(g->row_off.ensure(sizeof(uint32_t) * H));
    PR_TRY(g->counts.ensure(sizeof(uint32_t)));
    pr_vec3 *full_pcd = g->nn_full.as<pr_vec3>(), *full_nrm = full_pcd + px;
    // normals of every pixel (get_normal sees the uint16 image), then the valid pixels in row-major order
    HIP_TRY(prk::launch_scene_proj_prepare<T>(depth, W, H, K[0], K[4], K[2], K[5], full_pcd, full_nrm, g->stream));
    HIP_TRY(prk::launch_nn_gather<T>(depth, W, H, K[0], K[4], K[2], K[5], full_nrm, g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(),
                                     g->counts.as<uint32_t>(), nullptr, nullptr, false, g->stream));
    uint32_t n = 0;
    PR_TRY(read_back_words(g->counts.p, &n, 1, g->stream));
    if (n_points) *n_points = n;
    if (n == 0) { if (n_nodes) *n_nodes = 0; return PR_OK; }
    HIP_TRY(prk::launch_nn_gather<T>(depth, W, H, K[0], K[4], K[2], K[5], full_nrm, g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(),
                                     g->counts.as<uint32_t>(), pcd, nrm, true, g->stream));
    return kd_build_dev(pcd, nrm, n, max_leaf, nodes, cap, n_nodes);
}

}  // namespace prr

using namespace prr;

extern "C" {

int pr_scene_proj_prepare_dev(const void *depth_dev, int depth_is_i32, const float K[9], size_t width, size_t height,
                              pr_vec3 *pcd_dev_out, pr_vec3 *normal_dev_out)
{
    PR_ENTER();
    if (!depth_dev || !K || !pcd_dev_out || !normal_dev_out || width == 0 || height == 0) { set_error("pr_scene_proj_prepare_dev: bad arguments"); return PR_ERR_INVALID; }
    note_write(pcd_dev_out, width * height * sizeof(pr_vec3)); note_write(normal_dev_out, width * height * sizeof(pr_vec3));
    if (depth_is_i32) HIP_TRY(prk::launch_scene_proj_prepare<int32_t>(static_cast<const int32_t *>(depth_dev), (uint32_t)width, (uint32_t)height, K[0], K[4], K[2], K[5], pcd_dev_out, normal_dev_out, g->stream));
    else HIP_TRY(prk::launch_scene_proj_prepare<uint16_t>(static_cast<const uint16_t *>(depth_dev), (uint32_t)width, (uint32_t)height, K[0], K[4], K[2], K[5], pcd_dev_out, normal_dev_out, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PR_OK;
}

int pr_kdtree_build_dev(pr_vec3 *pcd_dev, pr_vec3 *normal_dev, size_t n_points, int max_leaf, pr_kdnode *nodes_dev_out, size_t cap_nodes, uint32_t *n_nodes)
{
    PR_ENTER();
    if (!pcd_dev || !normal_dev || !nodes_dev_out) { set_error("pr_kdtree_build_dev: bad arguments"); return PR_ERR_INVALID; }
    note_write(pcd_dev, n_points * sizeof(pr_vec3)); note_write(normal_dev, n_points * sizeof(pr_vec3)); note_write(nodes_dev_out, cap_nodes * sizeof(pr_kdnode));
    return kd_build_dev(pcd_dev, normal_dev, (uint32_t)n_points, max_leaf, nodes_dev_out, cap_nodes, n_nodes);
}

int pr_scene_nn_prepare_dev(const void *depth_dev, int depth_is_i32, const float K[9], int width, int height, int max_leaf,
                            pr_vec3 *pcd_dev_out, pr_vec3 *normal_dev_out, pr_kdnode *nodes_dev_out, size_t cap_nodes,
                            uint32_t *n_points, uint32_t *n_nodes)
{
    PR_ENTER();
    if (!depth_dev || !K || !pcd_dev_out || !normal_dev_out || !nodes_dev_out || width <= 0 || height <= 0) { set_error("pr_scene_nn_prepare_dev: bad arguments"); return PR_ERR_INVALID; }
    note_write(pcd_dev_out, (size_t)width * height * sizeof(pr_vec3)); note_write(normal_dev_out, (size_t)width * height * sizeof(pr_vec3)); note_write(nodes_dev_out, cap_nodes * sizeof(pr_kdnode));
    if (depth_is_i32) return scene_nn_prepare_dev_t<int32_t>(static_cast<const int32_t *>(depth_dev), K, (uint32_t)width, (uint32_t)height, max_leaf,
                                                             pcd_dev_out, normal_dev_out, nodes_dev_out, cap_nodes, n_points, n_nodes);
    return scene_nn_prepare_dev_t<uint16_t>(static_cast<const uint16_t *>(depth_dev), K, (uint32_t)width, (uint32_t)height, max_leaf,
                                            pcd_dev_out, normal_dev_out, nodes_dev_out, cap_nodes, n_points, n_nodes);
}

int pr_raw2depth_mask(const int32_t *raw_dev, size_t count, uint16_t *depth_host_out, uint8_t *mask_host_out)
{
    PR_ENTER();
    if (count == 0) return PR_OK;                                  // (an empty render stack has no array)
    if (!raw_dev || (!depth_host_out && !mask_host_out)) { set_error("pr_raw2depth_mask: bad arguments"); return PR_ERR_INVALID; }
    if (depth_host_out) PR_TRY(g->conv16.ensure(count * sizeof(uint16_t) + 16));
    if (mask_host_out) PR_TRY(g->conv8.ensure(count + 16));
    HIP_TRY(prk::launch_raw2depth_mask(raw_dev, count, depth_host_out ? g->conv16.as<uint16_t>() : nullptr, mask_host_out ? g->conv8.as<uint8_t>() : nullptr, g->stream));
    // ONE copy per output for the whole stack (the reference issues one thrust::copy per pose, renderer.cu:370-373)
    if (depth_host_out) HIP_TRY(hipMemcpyAsync(depth_host_out, g->conv16.p, count * sizeof(uint16_t), hipMemcpyDeviceToHost, g->stream));
    if (mask_host_out) HIP_TRY(hipMemcpyAsync(mask_host_out, g->conv8.p, count, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PR_OK;
}

// Scene_projective restricted to a window of the frame: copies the window's rows of the full-frame arrays
int pr_scene_proj_crop_dev(const pr_vec3 *pcd_full_dev, const pr_vec3 *normal_full_dev, size_t width, size_t height, pr_roi window,
                           pr_vec3 *pcd_out_dev, pr_vec3 *normal_out_dev)
{
    PR_ENTER();
    if (!pcd_full_dev || !normal_full_dev || !pcd_out_dev || !normal_out_dev || window.width <= 0 || window.height <= 0 || window.x < 0 || window.y < 0 ||
        (size_t)window.x + (size_t)window.width > width || (size_t)window.y + (size_t)window.height > height) { set_error("pr_scene_proj_crop_dev: bad arguments"); return PR_ERR_INVALID; }
    const size_t cw = (size_t)window.width, ch = (size_t)window.height;
    note_write(pcd_out_dev, cw * ch * sizeof(pr_vec3)); note_write(normal_out_dev, cw * ch * sizeof(pr_vec3));
    const size_t off = (size_t)window.y * width + (size_t)window.x;
    HIP_TRY(hipMemcpy2DAsync(pcd_out_dev, cw * sizeof(pr_vec3), pcd_full_dev + off, width * sizeof(pr_vec3), cw * sizeof(pr_vec3), ch, hipMemcpyDeviceToDevice, g->stream));
    HIP_TRY(hipMemcpy2DAsync(normal_out_dev, cw * sizeof(pr_vec3), normal_full_dev + off, width * sizeof(pr_vec3), cw * sizeof(pr_vec3), ch, hipMemcpyDeviceToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PR_OK;
}

}  // extern "C"
