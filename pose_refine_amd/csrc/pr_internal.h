// pr_internal.h -- declarations shared by the HIP kernels TU and the C-ABI/host TUs.
// Not part of the public boundary (that is include/pose_refine.h).
#pragma once

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#include "pose_refine.h"
#include "pr_tuning.h"

namespace prk {

// ---- canonical reduction geometry (DESIGN.md "canonical tree"; oracle: sum29_canonical) -----------
constexpr uint32_t kBlockThreads   = 256;                       // 4 wavefronts
constexpr uint32_t kPointsPerLane  = 4;                         // consecutive points per lane per step
constexpr uint32_t kPointsPerStep  = kBlockThreads * kPointsPerLane;   // 1024
constexpr uint32_t kAccStride      = 32;                        // 29 sums padded to 32 floats
constexpr uint32_t kCloudAlign     = PR_CLOUD_PACK;                 // fused path: clouds of a sub-batch packed one behind the other, each rounded up to this many points (0: one per fixed stride)
constexpr uint32_t kBoxPack        = PR_BOX_PACK;                   // fused asynchronous path: depth boxes of a sub-batch packed (each rounded up to this many ints; 0: full frames)
constexpr uint32_t kQCountStride   = 4;                         // queue counters per hypothesis (IcpBatch::nn_qcount)
constexpr uint32_t kNNWordsPerPoint = 6;                        // per cloud point: winner | slack | queue 1 (2 words) | queue 2 (2 words)

// per-hypothesis run state consumed by the correspondence kernel
enum : int32_t { kSkip = 0, kRun = 1, kRunWithTransform = 2 };

// everything a workgroup needs to know about its hypothesis, in ONE 64-byte record so that the
// wavefront fetches it with a single scalar load instead of a chain of dependent ones
struct alignas(64) PoseMeta {
    uint32_t start;             // first point of the cloud (in points, relative to IcpBatch::cloud)
    uint32_t count;             // points in the cloud
    int32_t  state;             // kSkip / kRun / kRunWithTransform
    uint32_t pad;
    float    xform[12];         // pending rigid update, rows 0..2 of the 4x4
};
static_assert(sizeof(PoseMeta) == 64, "PoseMeta must stay one 64-byte record");

struct DevIcpState;
// batch of model clouds handed to one correspondence pass
struct IcpBatch {
    pr_vec3        *cloud;      // base of all clouds (dev)
    const PoseMeta *meta;       // [P] (dev)
    float          *partial;    // [P][nblk][kAccStride] workgroup sums (dev)
    uint32_t        nblk;       // 2048-point blocks a hypothesis may have = stride of `partial` (and grid.x unless grid_x is set)
    uint32_t        grid_x;     // 0, or the number of workgroups per hypothesis when it differs from nblk (workgroups then loop over blocks)
    uint32_t        steps;      // steps of 1024 points per workgroup
    // PR_SOLVE_DEVICE with the solve fused into the pass (option "fused_solve"): the workgroup that delivers the last
    // partial sum of a hypothesis also adds the partials up and runs that hypothesis' iteration logic
    uint32_t        score_only; // 1 on the pass of iteration == max_iteration (PR_SOLVE_DEVICE): only sums 27 (error) and 28 (count) are formed
    uint32_t        fused;      // 0 = separate icp_finalize[_solve] launch; 1 = finalize + solve in the pass; 2 = finalize in the pass, the 29 sums of a
                                // hypothesis go to sums_out (PR_SOLVE_HOST: pinned host memory, read by the host after the stream has drained)
    float          *sums_out;   // fused == 2: [P][kAccStride], same pose indexing as `meta`
    uint32_t       *grp_count;  // fused == 2, optional: device counter of the hypotheses of this launch's pose GROUP that have delivered their sums; the one that makes it
    uint32_t        grp_expected;   // grp_expected re-zeroes it and stores iter + 1 into *grp_flag (pinned host memory): the host polls that one word instead of waiting for the stream
    uint32_t       *grp_flag;
    uint32_t        iter;       // iteration index of this pass (icp.cu:178 loop variable)
    DevIcpState    *st;         // [P]
    uint32_t       *arrive;     // [P] zero before the first pass; the solving workgroup re-zeroes its entry
    uint32_t       *nn_prev;    // kd-tree scenes: per cloud point (same indexing as `cloud`) the scene index of the previous pass' winner, or null
    float          *nn_slack;   // kd-tree scenes, search kernel: per cloud point, a lower bound on the distance of every scene point OTHER than the
                                // winner, minus the distance the point has travelled since that bound was established (<= 0: none)
    uint2          *nn_queue;   // search kernel -> tree kernel: per hypothesis (same indexing as the cloud) the (point, bound) of every query the first
                                // half could not settle;  nn_qcount: kQCountStride counters per hypothesis (same pose indexing as `meta`): fill levels
                                // of queue 1 and of queue 2, two each, alternating per pass
    uint32_t       *nn_qcount;
    uint2          *nn_queue2;  // bound kernel -> task walk: what the pixel window could not settle either, (point, bound)
    uint32_t        pre_transformed;   // 1: the pending update has already been applied to the cloud (nn_search_kernel): the pass must not apply it again
    pr_criteria     crit;
};

// projective scene exactly as the reference API hands it over (two Vec3f arrays)
struct SceneProjAoS {
    uint32_t width, height;
    float max_dist_diff, fx, fy, cx, cy;
    float tlx, tly;             // (float)tl_x, (float)tl_y of pcd2dep (common.h:64-67): 0 unless the scene buffers cover a crop of the frame
    const pr_vec3 *pcd, *normal;
};
// projective scene repacked by the fused pipeline: one 16-byte record {nx,ny,nz,z} per pixel
struct SceneProjPacked {
    uint32_t width, height;
    float max_dist_diff, fx, fy, cx, cy;
    float tlx, tly;
    const float4 *rec;
    const float *colf, *rowf;   // colf[x] = ((float)(x + tl_x) - cx)/fx, rowf[y] = ((float)(y + tl_y) - cy)/fy
};
// kd-tree scene: reference arrays + the traversal structure derived from them (build_nn_accel)
struct SceneNNDev {
    float max_dist_diff;
    const int4   *topo;         // per node: {split_v|left, child1|right, child2|-1, (parent+1)|dim<<30}
    const float4 *bmin, *bmax;  // per node tight bbox (leaves included)
    const float4 *pts;          // per point {x,y,z,0}
    const pr_vec3 *pcd, *normal;
    uint32_t n_nodes;
    uint32_t lds_nodes;         // how many leading (top-level) nodes the kernel stages in LDS (stackless: 16-byte topo entries; stack: 64-byte records)
    const float4 *rec;          // stack variant: 64-byte record per node (4 x float4), see nn_records_kernel
    uint32_t stack_depth;       // 0 = stackless traversal, else per-lane LDS stack entries (>= tree depth)
    // stack variant, compact form: 32-byte record per node (see nn_records32_kernel) with the two child boxes quantised to
    // 16 bits in the frame of the root box, rounded outwards; null when the tree cannot be expressed that way
    const uint4 *rec32;
    float qmin[3], qscale[3];   // dequantisation: qmin[a] + (float)q * qscale[a]
    const uint2 *desc;          // first 8 bytes of every compact record on their own (split | child | dim): all a descent needs when the
                                // split plane alone already rules the far side out
    // Pixel grid of the scene points (build_nn_grid): cell (px, py) of a gw x gh image holds the scene point that projects into
    // it under (gfx, gfy, gcx, gcy) as {x, y, z, index}, or {huge, huge, huge, -1}.  Null unless every scene point owns a cell of
    // its own (always the case for a Scene_nn made from a depth image with the same intrinsics, pcd_scene.cpp:10-29).
    const float4 *grid;
    uint32_t gw, gh;
    float gfx, gfy, gcx, gcy;
    // three coarser levels of the same grid: one representative scene point per 4 x 4, 16 x 16 and 64 x 64 pixel block
    // (the occupied sub-block nearest the block's centre, recursively), same {x, y, z, index} cells
    const float4 *pyr4, *pyr16, *pyr64;
    // wide records (nn_wide_build_kernel): one 128-byte line per wide node = eight slots {subtree box, reference}; null when the tree
    // cannot be expressed that way (the binary walk of nn_tree_kernel is used)
    const uint4 *wide;
    uint32_t n_wide;
    float wmin[3], wscale;      // frame of the wide records: coordinate = wmin[a] + (float)q * wscale (one scale for the three axes)
    // instrumented runs (option "nn_count"): per ICP pass (IcpBatch::iter) eight 64-bit counters -- queries, settled by the pixel
    // window, handed to the tree, pyramid descents, tree nodes visited, leaves scanned, leaf points tested, window cells read; else null
    unsigned long long *counters;
};
// kd-tree scene after the search kernel has run: the correspondence pass only gathers the winners (no search, no transform)
struct SceneNNWinners {
    float max_dist_diff;
    const float4 *pts;          // {x,y,z,0} per scene point
    const pr_vec3 *normal;
    const uint32_t *winner;     // per cloud point (same indexing as IcpBatch::cloud): scene index or 0xffffffff
};

// device-side solver state for PR_SOLVE_DEVICE (one record per hypothesis)
struct DevIcpState {
    float T[16];
    float fitness, rmse;
    int32_t done;               // 0 running, 1 finished
    int32_t passes;
};

// ---- launchers (all asynchronous on `s`) ----------------------------------------------------------
hipError_t launch_fill_i32(int32_t *dst, size_t n, int32_t v, hipStream_t s);
hipError_t launch_raster(const pr_triangle *tris, uint32_t n_tris, const pr_mat4 *poses_dev, uint32_t n_poses,
                         int32_t *depth, uint32_t width, uint32_t height, const pr_mat4 &proj, pr_roi roi,
                         uint32_t rw, uint32_t rh, hipStream_t s);
hipError_t launch_max2zero(int32_t *depth, size_t n, hipStream_t s);

// depth -> cloud: rows are counted, scanned per image, then emitted in row-major order.
// images: n_img images of width*height pixels each `img_stride` elements apart; cloud i is written
// at cloud + i*cloud_stride; counts[i] receives its size.  empty_intmax: INT_MAX also means "no depth".
template <typename T>
hipError_t launch_depth2cloud(const T *depth, uint32_t n_img, size_t img_stride, uint32_t width, uint32_t height,
                              uint32_t stride, uint32_t tl_x, uint32_t tl_y, float fx, float fy, float cx, float cy,
                              bool empty_intmax, uint32_t *row_count, uint32_t *row_off, uint32_t *counts,
                              pr_vec3 *cloud, size_t cloud_stride, bool emit, hipStream_t s);

hipError_t launch_model_aabb(const pr_triangle *tris, uint32_t n_tris, uint32_t *keys, float *aabb_out, const float *expect, uint32_t *flag_out, hipStream_t s);
hipError_t launch_render_bands(const pr_triangle *tris, uint32_t n_tris, const pr_mat4 *poses_dev, uint32_t n_poses, const float *aabb,
                               int4 *bbox, int32_t *depth, uint32_t *row_count, uint32_t *row_off, uint32_t *counts,
                               uint32_t width, uint32_t height, const pr_mat4 &proj, pr_roi roi, uint32_t n_cus, hipStream_t s);
// The two per-batch safety nets of the asynchronous path, carried by the raster launch itself (round 5): the workgroups of the launch's FIRST
// hypothesis fold their triangles' vertices into keys[0..5] (running minima; keys[6] = arrival ticket; armed once by the caller with six
// words of ones and a zero, re-armed by the last workgroup), the last one compares with the box the host assumed; workgroup (0, 0) takes
// the sampled fingerprint of the caller's scene arrays (fp_expected null: none).  Both only ever RAISE *flag.  keys null: no check.
struct AabbExpected { float v[6]; };
struct BatchCheck {
    uint32_t *keys; AabbExpected expect;
    const uint32_t *fa, *fb, *fc; unsigned long long na, nb, nc; const uint32_t *fp_expected;
    uint32_t *flag;
};
hipError_t launch_render_boxes(const pr_triangle *tris, uint32_t n_tris, const pr_mat4 *poses_dev, uint32_t n_poses, const float *aabb,
                               int4 *bbox, int32_t *depth, uint32_t *row_count, uint32_t *row_off, uint32_t *counts,
                               uint32_t width, uint32_t height, const pr_mat4 &proj, pr_roi roi, hipStream_t s, bool compute_boxes = true,
                               PoseMeta *meta = nullptr, DevIcpState *st = nullptr, uint32_t *arrive = nullptr, uint32_t cloud_stride = 0,
                               const uint32_t *box_off = nullptr,    // box_off: the boxes packed into `depth` at these offsets (ints), each with its own pitch (fill_box_kernel)
                               const BatchCheck *check = nullptr);
hipError_t launch_pack_export(const DevIcpState *st, pr_result *out, const uint32_t *counts, uint32_t *host_counts, pr_result *host_results,
                              uint32_t n, hipStream_t s);
hipError_t launch_stage_words(const void *src_host_mapped, void *dst, size_t bytes, hipStream_t s);
hipError_t launch_stage_words64(const void *src_host_mapped, void *dst, size_t bytes, hipStream_t s);
hipError_t launch_copy_words32(const void *src, void *dst, uint32_t n_words, hipStream_t s);   // either direction (pinned host memory through its device pointer)   // 8-byte words: exact for 72-byte records
// meta[i].start = exclusive scan of counts[0..i) rounded up to kCloudAlign points each (the packed layout of a batch's clouds)
hipError_t launch_d2c_pack_starts(const uint32_t *counts, uint32_t n, PoseMeta *meta, hipStream_t s);
hipError_t launch_emit_box(const int32_t *depth, uint32_t n_poses, uint32_t width, uint32_t height, const int4 *bbox, float fx, float fy,
                           float cx, float cy, const uint32_t *row_count, const uint32_t *row_off, pr_vec3 *cloud, size_t cloud_stride,
                           hipStream_t s, const PoseMeta *meta = nullptr, const uint32_t *box_off = nullptr);     // meta: cloud i starts at meta[i].start (packed, launch_render_boxes wrote it) instead of i * cloud_stride

hipError_t launch_icp_pass_proj_aos(const IcpBatch &b, const SceneProjAoS &sc, uint32_t n_poses, hipStream_t s);
hipError_t launch_icp_pass_proj_packed(const IcpBatch &b, const SceneProjPacked &sc, uint32_t n_poses, hipStream_t s);
hipError_t launch_icp_pass_nn(const IcpBatch &b, const SceneNNDev &sc, uint32_t n_poses, hipStream_t s);
// split form of the kd-tree pass: search (pending transform applied, winners written to b.nn_prev) + gather/accumulate pass
// marks (or null): three events recorded after the search, the bound and the walk kernel (timed batches; n_poses <= 32768)
hipError_t launch_nn_search(const IcpBatch &b, const SceneNNDev &sc, uint32_t n_poses, uint32_t max_points, uint32_t run, hipStream_t s, hipEvent_t *marks = nullptr);
hipError_t launch_icp_pass_nn_winners(const IcpBatch &b, const SceneNNWinners &sc, uint32_t n_poses, hipStream_t s);
// audit entry (icp_debug.hip): the 29 terms of every point of one cloud, out[n][29]; update12 (rows 0..2 of a 4x4) or null is applied first
hipError_t launch_contrib29_proj_aos(pr_vec3 *cloud, uint32_t n, const float *update12, const SceneProjAoS &sc, float *out, hipStream_t s);
hipError_t launch_contrib29_proj_packed(pr_vec3 *cloud, uint32_t n, const float *update12, const SceneProjPacked &sc, float *out, hipStream_t s);
hipError_t launch_contrib29_nn(pr_vec3 *cloud, uint32_t n, const float *update12, const SceneNNDev &sc, float *out, hipStream_t s);
// info[0] = 1 when every scene point owns a grid cell (the grid may be used), 0 otherwise
// grid: gw*gh cells followed by the three pyramid levels (nn_grid_cells(gw, gh) cells in all)
size_t nn_grid_cells(uint32_t gw, uint32_t gh);
hipError_t launch_build_nn_grid(const pr_vec3 *pcd, uint32_t n_points, uint32_t gw, uint32_t gh, float fx, float fy, float cx, float cy,
                                int32_t *cell_idx, float4 *grid, uint32_t *info, hipStream_t s);
hipError_t launch_icp_finalize(const float *partial, const PoseMeta *meta, uint32_t nblk,
                               uint32_t steps, float *sums, uint32_t n_poses, hipStream_t s);
// PR_SOLVE_DEVICE: finalize + convergence test + 6x6 solve + state update in one kernel
hipError_t launch_icp_finalize_solve(const float *partial, PoseMeta *meta, uint32_t nblk,
                                     uint32_t steps, DevIcpState *st, pr_criteria crit, uint32_t iter,
                                     uint32_t n_poses, hipStream_t s);
hipError_t launch_pack_results(const DevIcpState *st, pr_result *out, uint32_t n_poses, hipStream_t s);

// kd_build.hip: the level-order build's workspace (one allocation, carved by kd_work_bytes) and its launches
struct KdLevelNode { int left, right, child, axis; float cut; int pad0, pad1, pad2; };   // a node of the level being split: position range, first child's id (-1: a leaf), axis, cut
struct KdCtrl { uint32_t lo, hi, next, done, levels, error, pad0, pad1; };               // nodes [lo, hi) = the level, their children [hi, next); levels = scatter passes done
struct KdWork {
    void *base = nullptr;
    KdCtrl *ctrl[2] = { nullptr, nullptr };                                            // by level parity: a level's launches read [level & 1], its plan writes the other
    int *idx[2] = { nullptr, nullptr }, *owner[2] = { nullptr, nullptr };                // permutation and each position's node (index within its level), double-buffered
    KdLevelNode *lv[2] = { nullptr, nullptr };
    unsigned long long *bbkeys = nullptr, *lrkeys = nullptr;
    uint32_t *left_total = nullptr, *tile_agg = nullptr, *chunk_cnt = nullptr;
    unsigned long long *root_part = nullptr;
    uint32_t max_level = 0;
};
size_t kd_work_bytes(uint32_t n, uint32_t cap, int max_leaf, KdWork *w);           // bytes needed; with w->base set, also fills in the pointers
hipError_t launch_kd_init(const KdWork &w, pr_kdnode *nodes, uint32_t cap, const pr_vec3 *pcd, uint32_t n, int max_leaf, hipStream_t s);
hipError_t launch_kd_level(const KdWork &w, pr_kdnode *nodes, uint32_t cap, const pr_vec3 *pcd, uint32_t n, int max_leaf, uint32_t level, hipStream_t s);
hipError_t launch_kd_permute(const KdWork &w, uint32_t levels_launched, const pr_vec3 *pcd, const pr_vec3 *nrm, uint32_t n, pr_vec3 *pcd_out, pr_vec3 *nrm_out, hipStream_t s);
template <typename T>
hipError_t launch_nn_gather(const T *depth, uint32_t W, uint32_t H, float fx, float fy, float cx, float cy, const pr_vec3 *normal_full,
                            uint32_t *row_count, uint32_t *row_off, uint32_t *count, pr_vec3 *pcd, pr_vec3 *nrm, bool emit, hipStream_t s);
template <typename T>
hipError_t launch_scene_proj_prepare(const T *depth, uint32_t W, uint32_t H, float fx, float fy, float cx, float cy, pr_vec3 *pcd, pr_vec3 *normal, hipStream_t s);
hipError_t launch_raw2depth_mask(const int32_t *raw, size_t n, uint16_t *depth16, uint8_t *mask8, hipStream_t s);
// sampled hash of up to three arrays (null = absent): check = false stores it in *expected, check = true sets *flag = 1 if it differs
hipError_t launch_scene_fingerprint(const void *a, size_t a_bytes, const void *b, size_t b_bytes, const void *c, size_t c_bytes,
                                    uint32_t *expected, uint32_t *flag, bool check, hipStream_t s);
hipError_t launch_scene_fingerprint_full(const void *a, size_t a_bytes, const void *b, size_t b_bytes, const void *c, size_t c_bytes, uint32_t *sum, hipStream_t s);
hipError_t launch_pack_proj_scene(const pr_vec3 *pcd, const pr_vec3 *normal, float4 *rec, size_t n, float *colf, float *rowf,
                                  uint32_t width, uint32_t height, float fx, float fy, float cx, float cy, uint32_t tl_x, uint32_t tl_y,
                                  uint32_t *exact, hipStream_t s);
// info[0] = tree depth, info[1] = 1 when the 32-byte records are valid, info[2..7] = qmin[3], qscale[3] (float bits)
// wide records (nn_wide_build: launch_nn_wide_levels): info[8] = 1 when they are valid, 2 when the tree has more wide levels than were launched,
// info[9] = number of wide nodes; wide: nn_wide_capacity(n_nodes) lines of 128 bytes, scratch: nn_wide_scratch_words(n_nodes) words
inline size_t nn_wide_capacity(uint32_t n_nodes) { return (size_t)n_nodes / 2 + 2; }
inline size_t nn_wide_scratch_words(uint32_t n_nodes) { const size_t cap = nn_wide_capacity(n_nodes); return 2 * cap + 4 + 2 * (cap / 1024 + 4) + 4 + 8; }   // wq, cnt, chunk sums x 2, flag, control records x 2
hipError_t launch_build_nn_accel(const pr_kdnode *nodes, uint32_t n_nodes, const pr_vec3 *pcd, uint32_t n_points,
                                 int4 *topo, float4 *bmin, float4 *bmax, float4 *pts, float4 *rec, uint4 *rec32, uint2 *desc, uint32_t *info, hipStream_t s,
                                 float wide_margin = 0.0f);   // wide_margin: how far beyond the root box the wide records' frame reaches (the acceptance radius)
hipError_t launch_nn_wide_levels(const int4 *topo, const float4 *bmin, const float4 *bmax, uint32_t n_nodes, uint32_t n_points, uint4 *wide, uint32_t *scratch,
                                 uint32_t *info, uint32_t first, uint32_t count, hipStream_t s);

}  // namespace prk

// ---- host-side math shared by the C ABI (pr_host.cpp) ------------------------------------------
namespace prh {
void solve_666(const float A[36], const float b[6], float T[16]);
void mat4_mul(const float A[16], const float B[16], float C[16]);
}
