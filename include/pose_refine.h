/*
 * pose_refine.h -- C ABI of the MI355X-native render -> cloud -> point-to-plane ICP hot path.
 *
 * This is the drop-in boundary.  The reference (meiqua/pose_refine) has no FFI layer: its boundary
 * is the C++ header API (cuda_renderer/renderer.h, cuda_icp/icp.h, cuda_icp/scene/...).  The C++
 * adapter headers in include/cuda_renderer and include/cuda_icp keep that API source-compatible
 * and forward every device-side entry point to the functions below.  Each function cites the
 * reference interface it replaces (paths relative to the reference root).
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  "dev" pointers are device (HBM)
 * pointers obtained from pr_malloc (or any hipMalloc'd memory of the current device); everything
 * else is host memory.  All functions return PR_OK (0) or a negative error code; the message of
 * the last failure on the calling thread is available from pr_last_error().  There is NO CPU
 * fallback: every device entry point fails with PR_ERR_NO_DEVICE when no gfx950 device is usable.
 * Work is issued on one library-owned HIP stream per context; calls are synchronous with respect to the host unless
 * stated otherwise.
 *
 * Contexts and threads.  There is one shared context per device, created by pr_init(device) (or on first use: device 0).
 * A host thread works on the context it last selected with pr_init / pr_set_device; threads that never selected one use the
 * device of the process' first pr_init.  Calls on one context are serialised (a mutex per context); contexts of different
 * devices run side by side, which is how a C++ host drives N GPUs: one thread per device, pr_set_device(d) first.  Threads
 * that each refine their own hypotheses on the SAME device (the reference's usage, README.md:15 / icp.cu:170 per-thread
 * streams) call pr_thread_context(1) once to get a private context (own stream and workspaces) instead of queueing on the
 * shared one (once per thread, for the thread's life: the runtime hands out hardware queues in the order streams are created, and a
 * process that keeps creating and releasing contexts ends up with streams that share a queue -- two host-solve threads 235 k poses/s on
 * fresh contexts, 154 k on the twelfth pair).  Options (pr_set_option) are process-wide.
 *
 * Caches and caller-owned memory.  Derived data is cached by the address of the buffers it was derived from: the packed copy
 * of a projective scene (pcd / normal arrays), the traversal records of a kd-tree scene (pcd / nodes arrays) and the model box
 * of a triangle buffer.  The model box is re-verified on the device by every batch that uses it, so a rewritten mesh is never
 * rendered with a stale box.  The scene caches are dropped by every write that goes through this library (pr_memcpy_*,
 * pr_fill_i32, pr_free, pr_render, the *_prepare_dev / *_build_dev / *_crop_dev functions); a caller that rewrites a scene
 * array by other means (its own kernels, raw hipMemcpy) MUST announce it with pr_invalidate(ptr, bytes) before the next ICP /
 * refine call, or switch the caches off with pr_set_option("scene_cache", 0).
 * Behind that (round 6): every SYNCHRONOUS call that finds a cached form (pr_icp_*, pr_refine_batch*: what the C++ adapters of the reference's API
 * reach) compares a fingerprint of EVERY word of the scene arrays with the one taken when the cache was built (~20 us) and rebuilds on a mismatch --
 * an in-place edit through any pointer is seen without pr_invalidate, as with the reference, which reads the arrays at every call
 * (depth_scene.h:29-48).  Asynchronous batches (pr_refine_submit) compare a SAMPLE of 4096 words per array inside their raster launch (a mismatch
 * repeats the batch with fresh caches, pr_stats): a frame replaced as a whole is noticed, an edit confined to words the sample does not look at
 * needs pr_invalidate.  A write THROUGH this library into a range that a batch still in flight on the calling context reads (its scene arrays, its
 * mesh, its result block) waits for that batch first: a scene object may be re-initialised while its previous frame's batch is running.
 */
#ifndef POSE_REFINE_H
#define POSE_REFINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PR_OK                 0
#define PR_ERR_NO_DEVICE     -1
#define PR_ERR_HIP           -2
#define PR_ERR_INVALID       -3
#define PR_ERR_IO            -4
#define PR_ERR_NOMEM         -5
#define PR_ERR_COMM          -6     /* RCCL: library not loadable, communicator missing, or a collective failed */

/* ---- POD mirrors of the reference types (layouts verified by static_assert in the adapters) -- */
typedef struct { float x, y, z; } pr_vec3;                         /* ::Vec3f geometry.h:83-103; Model::float3 renderer.h:50-57 (12 B) */
typedef struct { pr_vec3 v0, v1, v2; } pr_triangle;                /* Model::Triangle renderer.h:58-68 (36 B)                          */
typedef struct { float m[16]; } pr_mat4;                           /* Model::mat4x4 renderer.h:69-141 = ::Mat4x4f, row-major (64 B)    */
typedef struct { int x, y, width, height; } pr_roi;                /* Model::ROI renderer.h:43-48                                      */
typedef struct {                                                   /* ::Node_kdtree pcd_scene.h:5-25 (52 B)                            */
    int parent, child1, child2;
    float split_v;
    float bbox[6];
    int split_dim;
    int left, right;
} pr_kdnode;
typedef struct { float T[16]; float inlier_rmse; float fitness; } pr_result;        /* cuda_icp::RegistrationResult icp.h:26-36 (72 B) */
typedef struct { float relative_fitness, relative_rmse; int max_iteration; } pr_criteria; /* cuda_icp::ICPConvergenceCriteria icp.h:38-50 */

/* Scene_projective (depth_scene.h:7-48): non-owning view, passed by value like the reference does. */
typedef struct {
    uint64_t width, height;
    float max_dist_diff;
    float K[9];
    const pr_vec3 *pcd;        /* dev, width*height, (0,0,0) where depth==0 */
    const pr_vec3 *normal;     /* dev, width*height */
} pr_scene_proj;

/* Scene_nn (pcd_scene.h:48-137): non-owning view of a KDTree_cuda (pcd_scene.h:38-45). */
typedef struct {
    float max_dist_diff;
    const pr_vec3 *pcd;        /* dev, n_points, in tree order */
    const pr_vec3 *normal;     /* dev, n_points */
    const pr_kdnode *nodes;    /* dev, n_nodes, level order, nodes[0] = root */
    uint32_t n_points, n_nodes;
    /* Optional (not in the reference's struct, which drops K after init_Scene_nn_*): the camera of the depth image the scene was made
     * from -- cam_w = 0: unknown.  With it a bare ICP call (no render, hence no camera of its own) can index the scene points by pixel as
     * the fused path does; it is a HINT: the pixel grid is only used if every scene point really projects into a cell of its own. */
    float cam_fx, cam_fy, cam_cx, cam_cy;
    uint32_t cam_w, cam_h;
    /* The hint is read only when cam_magic == PR_SCENE_NN_CAM_MAGIC: a caller that fills the reference's six members field by field and
     * leaves the rest of the struct uninitialised (or zero) never triggers a grid build from garbage.  The struct has grown since round 2
     * (it ends here; sizeof == 72): callers must be compiled against this header -- PR_ABI_VERSION / pr_abi_version() tell them apart. */
    uint32_t cam_magic;
} pr_scene_nn;
#define PR_SCENE_NN_CAM_MAGIC 0x4d414350u   /* 'PCAM' */
#define PR_ABI_VERSION 4                    /* bumped whenever a struct of this header changes size or layout */

/* A Scene_projective whose arrays cover only a window of the frame: pcd2dep(src, K, tl_x, tl_y) / dep2pcd(x, y, d, K, tl_x, tl_y)
 * (common.h:47-73) with the offsets the reference declares but never passes.  view.width / view.height are
 * the window's size, pixel (x, y) of the arrays is frame pixel (x + tl_x, y + tl_y). */
typedef struct {
    pr_scene_proj view;
    uint32_t tl_x, tl_y;
} pr_scene_proj_crop;

#define PR_SCENE_PROJ      0
#define PR_SCENE_NN        1
#define PR_SCENE_PROJ_CROP 2      /* scene argument is a pr_scene_proj_crop */

/* where the 6x6 solve of every ICP iteration runs (icp.cu:207 does it on the host) */
#define PR_SOLVE_HOST   0
#define PR_SOLVE_DEVICE 1

/* ---- library / device ------------------------------------------------------------------------ */
const char *pr_last_error(void);
const char *pr_version(void);
int pr_abi_version(void);                        /* PR_ABI_VERSION the library was built with: compare with the header a caller was compiled against */
int pr_device_count(void);                       /* number of visible HIP devices (0 if none)              */
int pr_init(int device);                         /* bind the calling thread to the device's shared context, create its stream (test.cpp:12-20 warm-up) */
int pr_set_device(int device);                   /* same (cudaSetDevice, which test.cpp:14 leaves commented out)                 */
int pr_thread_context(int enable);               /* 1: private context for the calling thread on its current device; 0: release it */
int pr_shutdown(void);                           /* release the calling thread's context: workspaces, streams, communicator    */
int pr_sync(void);                               /* wait for the context's stream                                               */

/* device_vector_holder<T> storage: common.cu:3-40, renderer.cu:15-50 */
int pr_malloc(void **dev_ptr, size_t bytes);
int pr_free(void *dev_ptr);                     /* waits for what every context of the device has in flight (the caller's and, one at a time, the others') before the memory goes */
int pr_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes);
int pr_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes);
int pr_memcpy_d2d(void *dev_dst, const void *dev_src, size_t bytes);
int pr_fill_i32(int32_t *dev_dst, size_t count, int32_t value);   /* holder(size, init) fill ctor */
/* [dev_ptr, dev_ptr + bytes) was written behind the library's back (bytes = 0: the whole allocation) -- see "Caches" above */
int pr_invalidate(const void *dev_ptr, size_t bytes);

/* ---- host-side model / scene preparation (CPU in the reference too) --------------------------- */
/* Model::Model(fileName) / LoadModel renderer.cpp:11-104 + get_bounding_box :106-150.  The reference imports through assimp; here:
 * PLY (ASCII, binary little / big endian, typed properties) and Wavefront OBJ (single mesh, identity node transform), and glTF 2.0
 * (.gltf with external or base64 buffers, .glb) with its node hierarchy walked like recursive_render (renderer.cpp:69-104): node
 * transforms multiplied down the tree, one mesh per primitive, triangles transformed, vertices_out untransformed and faces_out
 * per-mesh indices exactly as the reference keeps them, the box over the TRANSFORMED vertices (get_bounding_box_for_node).
 * Faces with < 3 indices are dropped (renderer.cpp:78), polygons / strips / fans become triangles (assimp's aiProcess_Triangulate).
 * Any output pointer may be NULL; faces_out receives 3 vertex indices per triangle (Model::faces), vertices_out the file's vertex
 * list (Model::vertices), bbox_min / bbox_max the component-wise extremes of the vertices (Model::bbox_min / bbox_max). */
int pr_mesh_count(const char *path, size_t *n_triangles, size_t *n_vertices);
int pr_mesh_load(const char *path, pr_triangle *tris_out, size_t cap_triangles, size_t *n_triangles,
                 pr_vec3 *vertices_out, size_t cap_vertices, size_t *n_vertices, int32_t *faces_out,
                 float bbox_min[3], float bbox_max[3]);
/* triangles only (the hot path consumes nothing else) */
int pr_ply_count(const char *path, size_t *n_triangles, size_t *n_vertices);
int pr_ply_load(const char *path, pr_triangle *tris_out, size_t cap_triangles, size_t *n_triangles);
/* compute_proj renderer.cpp:161-185 */
void pr_compute_proj(const float K[9], int width, int height, float near_, float far_, pr_mat4 *proj_out);
/* get_normal common.cpp:17-107 (depth already uint16, mm) */
int pr_get_normal(const uint16_t *depth16, int width, int height, const float K[9], pr_vec3 *normals_out);
/* init_Scene_projective_cpu depth_scene.cpp:3-35: fills width*height pcd + normal host buffers */
int pr_scene_proj_prepare(const void *depth, int depth_is_i32, const float K[9], size_t width, size_t height,
                          pr_vec3 *pcd_out, pr_vec3 *normal_out);
/* SURVEY 8f rank 1 -- the same preparation entirely on the device (the reference's init_Scene_projective_cuda still runs it
 * on the CPU, depth_scene.cu:8, and its README names that as the remaining bottleneck): depth_dev -> pcd_dev, normal_dev.
 * Bit-identical to pr_scene_proj_prepare. */
int pr_scene_proj_prepare_dev(const void *depth_dev, int depth_is_i32, const float K[9], size_t width, size_t height,
                              pr_vec3 *pcd_dev_out, pr_vec3 *normal_dev_out);
/* A window of a prepared full-frame scene as arrays of its own (rows of `window`, window.width * window.height entries each) */
int pr_scene_proj_crop_dev(const pr_vec3 *pcd_full_dev, const pr_vec3 *normal_full_dev, size_t width, size_t height, pr_roi window,
                           pr_vec3 *pcd_out_dev, pr_vec3 *normal_out_dev);
/* init_Scene_nn_cpu pcd_scene.cpp:4-37 + KDTree_cpu::build_tree pcd_scene.cpp:45-184.
 * pcd_out/normal_out need width*height entries, nodes_out 2*width*height+1 entries (worst case). */
int pr_scene_nn_prepare(const void *depth, int depth_is_i32, const float K[9], int width, int height,
                        int max_leaf, pr_vec3 *pcd_out, pr_vec3 *normal_out, pr_kdnode *nodes_out,
                        size_t cap_nodes, uint32_t *n_points, uint32_t *n_nodes);
/* KDTree_cpu::build_tree on caller-provided points (reorders pcd/normal in place) */
int pr_kdtree_build(pr_vec3 *pcd, pr_vec3 *normal, size_t n_points, int max_leaf,
                    pr_kdnode *nodes_out, size_t cap_nodes, uint32_t *n_nodes);
/* SURVEY 8f rank 1 -- the same two functions on the device (bit-identical nodes, point order and normals): the level-order
 * build runs one workgroup per node and level (block scans reproduce the stable two-ended partition and its tie rule).
 * pcd/normal/nodes are device buffers sized like their host counterparts above. */
int pr_kdtree_build_dev(pr_vec3 *pcd_dev, pr_vec3 *normal_dev, size_t n_points, int max_leaf,
                        pr_kdnode *nodes_dev_out, size_t cap_nodes, uint32_t *n_nodes);
int pr_scene_nn_prepare_dev(const void *depth_dev, int depth_is_i32, const float K[9], int width, int height, int max_leaf,
                            pr_vec3 *pcd_dev_out, pr_vec3 *normal_dev_out, pr_kdnode *nodes_dev_out, size_t cap_nodes,
                            uint32_t *n_points, uint32_t *n_nodes);
/* eigen_slover_666 icp.cpp:29-45 (public in icp.h:54) */
void pr_solve_666(const float A[36], const float b[6], pr_mat4 *T_out);
/* Mat4x4f * Mat4x4f geometry.h:292-298 (entries summed over k = 3,2,1,0): what `result.transformation_ = extrinsic * result.transformation_`
 * (icp.cu:212) evaluates; C_out may alias A or B */
void pr_mat4_mul(const pr_mat4 *A, const pr_mat4 *B, pr_mat4 *C_out);

/* ---- renderer (cuda_renderer/renderer.cu) ------------------------------------------------------ */
/* render_cuda_keep_in_gpu renderer.cu:269-336: depth_dev_out[n_poses*rw*rh] int32 mm, 0 = empty.
 * tris_dev: device triangles (the device_vector_holder<Triangle> overloads); poses on the host. */
int pr_render(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, size_t n_poses,
              size_t width, size_t height, const pr_mat4 *proj, pr_roi roi, int32_t *depth_dev_out);
/* render_cuda renderer.cu:189-267: same, result copied to the host */
int pr_render_to_host(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, size_t n_poses,
                      size_t width, size_t height, const pr_mat4 *proj, pr_roi roi, int32_t *depth_host_out);

/* raw2depth_uint16_cuda / raw2mask_uint8_cuda / raw2depth_mask_cuda renderer.cu:338-439: the int32 depth stack of a render ->
 * uint16 depth (uint16_t(x)) and/or uint8 mask (x>0 ? 255 : 0) on the host; either output may be NULL. */
int pr_raw2depth_mask(const int32_t *raw_dev, size_t count, uint16_t *depth_host_out, uint8_t *mask_host_out);

/* ---- depth -> cloud (cuda_icp/icp.cu:228-291 depth2cloud_cuda<T>) ------------------------------ */
/* Allocates *cloud_dev_out (release with pr_free); points in row-major pixel order, metres. */
int pr_depth2cloud_i32(const int32_t *depth_dev, uint32_t width, uint32_t height, const float K[9],
                       uint32_t stride, uint32_t tl_x, uint32_t tl_y, pr_vec3 **cloud_dev_out, uint32_t *n_points);
int pr_depth2cloud_u16(const uint16_t *depth_dev, uint32_t width, uint32_t height, const float K[9],
                       uint32_t stride, uint32_t tl_x, uint32_t tl_y, pr_vec3 **cloud_dev_out, uint32_t *n_points);

/* ---- ICP (cuda_icp/icp.cu:156-223 ICP_Point2Plane_cuda<Scene>) --------------------------------- */
/* One cloud: mutates cloud_dev in place exactly like the reference. */
int pr_icp_proj(pr_vec3 *cloud_dev, uint32_t n_points, const pr_scene_proj *scene, pr_criteria crit, pr_result *result_out);
int pr_icp_nn(pr_vec3 *cloud_dev, uint32_t n_points, const pr_scene_nn *scene, pr_criteria crit, pr_result *result_out);
/* Many clouds against one scene in one launch per iteration (what the reference needs P host
 * threads + cudaStreamPerThread for, README.md:15): cloud i = clouds_dev[offsets[i] .. offsets[i+1]). */
int pr_icp_batch(pr_vec3 *clouds_dev, const uint32_t *offsets_host, uint32_t n_clouds, int scene_kind,
                 const void *scene, pr_criteria crit, pr_result *results_host);

/* ---- fused hypothesis refinement: render -> cloud -> ICP for n_poses hypotheses --------------- */
/* test.cpp:143-172 for a whole batch, everything resident on the device; results on the host.
 * cloud_sizes_host (optional) receives the model-cloud size of every hypothesis.  The synchronous calls (pr_refine_batch, _dev, _roi) run on
 * whichever asynchronous slot of the context holds no unfinished batch; with both slots pending they return PR_ERR_INVALID. */
int pr_refine_batch(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses,
                    uint32_t width, uint32_t height, const pr_mat4 *proj, const float K[9],
                    int scene_kind, const void *scene, pr_criteria crit,
                    pr_result *results_host, uint32_t *cloud_sizes_host);
/* same, results left on the device (for the RCCL gather of the sharded job): results_dev[n_poses] */
int pr_refine_batch_dev(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses,
                        uint32_t width, uint32_t height, const pr_mat4 *proj, const float K[9],
                        int scene_kind, const void *scene, pr_criteria crit,
                        pr_result *results_dev, uint32_t *cloud_sizes_host);

/* The same with the hypotheses rendered only inside `roi` (renderer.h:199 ROI {x, y, width, height} in image rows, renderer.cu:106-113)
 * and the clouds extracted with tl_x = roi.x, tl_y = roi.y (icp.h:57-60): clouds hold the rendered pixels inside the window, with
 * the coordinates they have in the full frame.  roi.width / roi.height <= 0 means no ROI. */
int pr_refine_batch_roi(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses,
                        uint32_t width, uint32_t height, const pr_mat4 *proj, const float K[9],
                        int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                        pr_result *results_host, uint32_t *cloud_sizes_host);

/* Asynchronous form of the two calls above (no counterpart in the reference, whose ICP() blocks): pr_refine_submit
 * enqueues one batch on `slot` (0 or 1) and returns; pr_refine_wait(slot) blocks until that batch is finished and
 * fills results_host / cloud_sizes_host (results_dev is complete at that point as well).  Exactly one of results_host /
 * results_dev may be NULL.  poses_host is copied before the call returns; tris_dev, the scene arrays and every output
 * pointer must stay valid until the wait.  Two slots let a host enqueue batch k+1 while batch k runs.  Batches the
 * asynchronous path does not cover (host solve, instrumented and timed calls) run to completion inside pr_refine_submit. */
int pr_refine_submit(int slot, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses,
                     uint32_t width, uint32_t height, const pr_mat4 *proj, const float K[9],
                     int scene_kind, const void *scene, pr_criteria crit,
                     pr_result *results_host, pr_result *results_dev, uint32_t *cloud_sizes_host);
int pr_refine_submit_roi(int slot, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses,
                         uint32_t width, uint32_t height, const pr_mat4 *proj, const float K[9],
                         int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                         pr_result *results_host, pr_result *results_dev, uint32_t *cloud_sizes_host);
int pr_refine_wait(int slot);

/* ---- sharding of a hypothesis batch over ranks (contiguous blocks) ---------------- */
void pr_shard_range(uint32_t n_items, uint32_t rank, uint32_t world, uint32_t *first, uint32_t *count);

/* ---- the job's one collective: gather of the solved transforms over RCCL / xGMI -- */
/* The reference is single-GPU (test.cpp:14).  Hypotheses are independent, so ranks never exchange data on the path itself;
 * the P/G x 72-byte RegistrationResult records of every rank are gathered once.  librccl is opened on first use.
 *   one process, one host thread per GPU:  pr_comm_init_all(G) once, then thread d: pr_set_device(d) ... pr_gather_results(...)
 *   one process per GPU:  rank 0 calls pr_comm_id and hands the 128 bytes to the others (any channel), all call pr_comm_init_rank */
#define PR_COMM_ID_BYTES 128
int pr_comm_id(unsigned char id_out[PR_COMM_ID_BYTES]);                                   /* ncclGetUniqueId */
int pr_comm_init_rank(const unsigned char id[PR_COMM_ID_BYTES], int rank, int world);     /* the calling thread's context becomes rank `rank` */
int pr_comm_init_all(int n_devices);                                                      /* shared contexts of devices 0..n-1, rank = device */
int pr_comm_rank(int *rank, int *world);
int pr_comm_destroy(void);
/* rank r sends its pr_shard_range(n_total, r, world) block (n_local records, device memory); recv_dev (root only, n_total
 * records) receives all blocks in global hypothesis order.  Enqueued on the context's stream: pr_sync / pr_memcpy_d2h order after it. */
int pr_gather_results(const pr_result *send_dev, uint32_t n_local, uint32_t n_total, int root, pr_result *recv_dev);

/* ---- options / instrumentation ----------------------------------------------------------------- */
/* pr_set_option names (all int; defaults in brackets).  None of them changes a result bit, except "points_per_block", which
 * selects the reduction tree and therefore the last bits of the sums (DESIGN.md "canonical tree").
 *   "solve"            [PR_SOLVE_HOST] PR_SOLVE_HOST = reference-style host solve per iteration, PR_SOLVE_DEVICE = loop on the device
 *   "points_per_block" [3072]  points per workgroup of the correspondence pass (multiple of 1024)
 *   "fused_solve"      [1]     the workgroup that delivers a hypothesis' last partial sum finalizes in the tail of the pass kernel instead of
 *                              a second launch: device solve = finalize + 6x6 solve there; host solve = the 29 totals stored into pinned host memory
 *   "pose_groups"      [0]     streams the batch is split over (1..4; 0 = 2): device solve = one group's solve tail under another's pass;
 *                              host solve = software pipeline (the host solves one group while another group's pass runs)
 *   "graph"            [1]     device solve, one pose group, synchronous path: replay the loop as a hipGraph
 *   "sub_batch"        [512]   asynchronous path: hypotheses per sub-batch (cache residency of the clouds)
 *   "overlap_pass"     [-1]    asynchronous path: the other slot's render may start after this pass of a slot's loop (-1 = chosen per batch)
 *   "raster_mode"      [0]     fused render: 0 = global atomicMin inside the pose's pixel box, 1 = LDS depth bands (synchronous path)
 *   "nn_stack"         [1]     kd-tree query: per-lane LDS stack (1) or the reference's stackless walk (0)
 *   "nn_compact"       [1]     stack query on 32-byte node records with 16-bit outward-rounded boxes (0: exact 64-byte records)
 *   "host_worker"      [1]     host solve: a batch given to pr_refine_submit runs on a library-owned helper thread of its slot (private context: its own
 *                              workspaces and scene caches -- about as much device memory again per slot in use); 0: inside the call, on the caller's thread
 *   "blocking_wait"    [0]     pr_refine_wait sleeps on the slot's event (hipEventBlockingSync) instead of spinning; takes effect for slots created afterwards
 *   "nn_wide"          [1]     queued tree searches walk 128-byte lines (wide nodes of eight subtree boxes, one line per leaf) in nearest-first
 *                              order; a query whose minimum is attained by more than one point is repeated by the ordered binary walk (0: binary walk only)
 *   "nn_seed"          [1]     compact records: start every search from the previous pass' / previous point's winner distance
 *   "nn_split"         [1]     compact records: the search runs in a kernel of its own (adjacent lanes = adjacent cloud points, a workgroup
 *                              takes "nn_run" [8] chunks of 256 points) and the pass gathers its winners in canonical order
 *   "nn_count"         [0]     instrumented runs: the search kernel counts its work (pr_nn_counters)
 *   "nn_grid"          [1]     fused refinement with a kd-tree scene made from a depth image: scene points are also indexed by pixel
 *                              (first bounds, and an exact window scan once the bound is a few pixels wide)
 *   "nn_lds_nodes"     [1024]  stackless query: leading nodes staged in LDS;  "nn_lds_records" [0]: the same for 64-byte records
 *   "profile"          [0]     see below;  "sample_period" [32]: profile 2 times one call in this many
 *   "scene_cache"      [1]     keep derived scene data between calls (see "Caches" at the top) */
int  pr_set_option(const char *name, int value);
int  pr_get_option(const char *name, int *value);
/* HIP-event timing of the correspondence kernel on the library stream: option "profile" = 1 times every launch (all calls run
 * synchronously as one pose group), 2 does so for one call in `sample_period` (the other calls are unaffected), 3 times every launch
 * of pr_refine_submit batches WITHOUT making them synchronous: a timed batch renders under the other slot's loop as usual, its own loop
 * waits until the other slot's batch is complete, runs as one pose group and holds the other slot's next render back until it has
 * finished -- the timed launches have the chip to themselves, the figures appear with that batch's pr_refine_wait.
 * Accumulated since the last reset: launches timed, model points they processed, and their algorithmic bytes (36 B/point on
 * the first pass of a cloud and on the score-only last pass, 48 B/point in between). */
/* Work counters of the kd-tree search kernel, collected while option "nn_count" is 1 (instrumented runs; SURVEY 8d "count its own
 * visits"): out[pass * 8 + k] for ICP passes 0..passes-1 (<= 64), k = 0 queries, 1 settled by the pixel window, 2 handed to the tree,
 * 3 pyramid descents, 4 tree nodes visited, 5 leaves scanned, 6 leaf points tested, 7 window cells read.  Reading resets the counters. */
int  pr_nn_counters(uint64_t *out, uint32_t passes);
int  pr_profile_reset(void);
int  pr_profile_read(double *kernel_ms, uint64_t *launches, uint64_t *points, uint64_t *algorithmic_bytes,
                     double *render_ms, double *cloud_ms);
/* The timed launches one by one, in microseconds, in the order they were read back (at most 8192 since the last reset): *n = how many
 * there are, the first min(*n, capacity) are copied.  What a min / median / max over the sampled launches is made from. */
int  pr_profile_launches(float *launch_us, uint32_t capacity, uint32_t *n);
/* A kd-tree correspondence pass is four kernels (search, bound + window, task walk, reduce over the winners: Scene_nn::query of
 * pcd_scene.h:60-136 inside thrust__pcd2Ab, icp.h:128-209).  Timed asynchronous batches (profile 3) put events between them:
 * part_ms[0..3] = accumulated time of each kernel, *passes = passes that contributed.  Reset by pr_profile_reset. */
int  pr_profile_nn(double part_ms[4], uint64_t *passes);
/* Audit entry: the 29 per-point terms of ONE correspondence pass -- thrust__pcd2Ab's Vec29f, icp.h:128-209: the 21
 * upper-triangle products J_i J_j row by row, the 6 products J_i r, r^2, and 1 -- for every point of one cloud, zeros where the query
 * found no correspondence.  update16 (a row-major 4x4, or NULL) is applied to the cloud first and written back, exactly as the fused pass
 * applies the pending update of the previous iteration (transform_pcd_cuda, icp.cu:142-153).  want_packed != 0 looks the projective scene
 * up through the packed 16-byte record the fused path uses.  contrib_host receives n_points x 29 floats.  A caller that adds the rows in
 * point order reproduces the reference's single-thread summation (icp.cpp:139-148), which the product kernel replaces by its fixed tree. */
int  pr_debug_contrib29(pr_vec3 *cloud_dev, uint32_t n_points, int scene_kind, const void *scene, const float *update16, int want_packed,
                        float *contrib_host);
/* Two things the library does silently for correctness, counted per context since it was created: asynchronous batches that
 * pr_refine_wait ran a SECOND time because the device-side checks found a stale model box or a scene array that no longer matches its
 * cached form (a caller who sees this grow writes to its buffers behind the library's back: pr_invalidate is the cheap cure), and timed
 * spans / batches whose HIP-event timing was dropped because an event could not be created or recorded (the work itself ran). */
int  pr_stats(uint64_t *batches_repeated, uint64_t *timings_dropped);
/* HIP-event time of the pr_gather_results exchanges issued while option "profile" was non-zero (events on the context's stream around
 * the grouped send / receive; accumulated since pr_profile_reset).  Waits for the stream.  New in this library -- the reference has
 * no multi-device code (test.cpp:14). */
int  pr_gather_profile(double *gather_ms, uint64_t *gathers);

#ifdef __cplusplus
}
#endif
#endif /* POSE_REFINE_H */
