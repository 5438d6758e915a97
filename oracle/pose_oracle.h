/*
 * pose_oracle.h -- CPU ORACLE for the render -> cloud -> point-to-plane ICP hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (pose_refine_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference algorithm (meiqua/pose_refine); every
 * function cites the reference file:line it follows.  Paths are relative to the
 * reference root.
 *
 * Parity pin, stated plainly:
 *   - po_mat4_mul (cuda_icp/geometry.h) is pinned by the REFERENCE ITSELF: oracle/_ref compiles that header verbatim and
 *     tests/golden/geometry_h.json holds what it computes (tests/test_geometry_golden.py).
 *   - Everything else is PARITY UNPINNED in the strict sense: the reference holds no golden vectors or known-answer tests
 *     for this path and none of its other translation units builds here (OpenCV, Eigen, assimp are absent; stand-ins are not
 *     allowed).  What it is checked against are numbers the SURVEY transcribed from a throw-away build of the verbatim
 *     reference CPU sources (OMP_NUM_THREADS=1) -- render checksums, cloud size, kd-tree shape and the
 *     four ICP results: tests/golden/survey_8c.json, tests/test_oracle_golden.py.
 *   - The 6x6 solver follows Eigen's published algorithm (pivoted LDLT, AngleAxis -> quaternion product); Eigen is absent
 *     from the reference tree and un-versioned (cuda_icp/CMakeLists.txt:18), so it is held to the 1e-4 transform tolerance only.
 */
#ifndef POSE_ORACLE_H
#define POSE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } po_vec3;                 /* geometry.h:83-103 vec<3,float>          */
typedef struct { po_vec3 v0, v1, v2; } po_tri;              /* renderer.h:58-68  Model::Triangle       */
typedef struct { int x, y, width, height; } po_roi;        /* renderer.h:43-48  Model::ROI            */

typedef struct {                                           /* pcd_scene.h:5-19  Node_kdtree (52 B)    */
    int parent, child1, child2;
    float split_v;
    float bbox[6];
    int split_dim;
    int left, right;
} po_kdnode;

typedef struct {                                           /* icp.h:26-36  RegistrationResult (72 B)  */
    float T[16];
    float inlier_rmse;
    float fitness;
} po_result;

typedef struct {                                           /* icp.h:38-50  ICPConvergenceCriteria     */
    float relative_fitness;
    float relative_rmse;
    int max_iteration;
} po_criteria;

/* Scene descriptors: non-owning views like the reference's (depth_scene.h:7-15, pcd_scene.h:48-52). */
typedef struct {
    size_t width, height;
    float max_dist_diff;
    float K[9];
    const po_vec3 *pcd;
    const po_vec3 *normal;
    size_t tl_x, tl_y;           /* pcd2dep's offsets (common.h:63-73): 0 for a full-frame scene; the arrays of a cropped scene
                                    start at frame pixel (tl_x, tl_y) -- the reference declares the parameters, never passes them */
} po_scene_proj;

typedef struct {
    float max_dist_diff;
    const po_vec3 *pcd;
    const po_vec3 *normal;
    const po_kdnode *nodes;
} po_scene_nn;

enum { PO_SCENE_PROJ = 0, PO_SCENE_NN = 1 };
/* reduction order of the 29-float accumulator */
enum { PO_SUM_SEQUENTIAL = 0,   /* index order, = reference at OMP_NUM_THREADS=1 (icp.cpp:139-148) */
       PO_SUM_CANONICAL  = 1 }; /* the fixed tree the HIP kernels use (DESIGN.md "canonical tree")  */

/* ---- renderer ------------------------------------------------------------------------------- */
void po_compute_proj(const float K[9], int width, int height, float near_, float far_, float proj[16]);
void po_render(const po_tri *tris, size_t n_tris, const float *poses16, size_t n_poses,
               size_t width, size_t height, const float proj[16], po_roi roi, int32_t *depth_out);
size_t po_ply_count(const char *path, size_t *n_vertices);
int    po_ply_load(const char *path, po_tri *tris_out, size_t cap_tris);

/* ---- depth -> cloud ------------------------------------------------------------------------- */
size_t po_depth2cloud_i32(const int32_t *depth, uint32_t width, uint32_t height, const float K[9],
                          uint32_t stride, uint32_t tl_x, uint32_t tl_y, po_vec3 *cloud_out);
size_t po_depth2cloud_u16(const uint16_t *depth, uint32_t width, uint32_t height, const float K[9],
                          uint32_t stride, uint32_t tl_x, uint32_t tl_y, po_vec3 *cloud_out);

/* ---- scene preparation ---------------------------------------------------------------------- */
void po_depth_i32_to_u16(const int32_t *in, uint16_t *out, size_t n);
void po_get_normal(const uint16_t *depth16, int width, int height, const float K[9], po_vec3 *normals_out);
void po_scene_proj_init(const void *depth, int depth_is_i32, const float K[9], size_t width, size_t height,
                        po_vec3 *pcd_out, po_vec3 *normal_out);
size_t po_scene_nn_gather(const void *depth, int depth_is_i32, const float K[9], int width, int height,
                          po_vec3 *pcd_out, po_vec3 *normal_out);
size_t po_kd_build(po_vec3 *pcd, po_vec3 *normal, size_t n, int max_leaf, po_kdnode *nodes_out, size_t cap_nodes);

/* ---- per-point math ------------------------------------------------------------------------- */
int  po_query_proj(const po_scene_proj *s, po_vec3 src, po_vec3 *dst, po_vec3 *nrm);
int  po_query_nn(const po_scene_nn *s, po_vec3 src, po_vec3 *dst, po_vec3 *nrm, int *winner, float *dist_sq,
                 uint32_t *node_visits);
void po_contrib29(po_vec3 src, po_vec3 dst, po_vec3 nrm, float out[29]);
void po_solve666(const float A[36], const float b[6], float T_out[16]);
void po_mat4_mul(const float A[16], const float B[16], float C[16]);
void po_transform_cloud(po_vec3 *cloud, size_t n, const float T[16]);

/* ---- ICP -------------------------------------------------------------------------------------
 * scene_kind: PO_SCENE_PROJ -> scene is po_scene_proj*, PO_SCENE_NN -> po_scene_nn*.
 * sum_mode / points_per_block: reduction order (points_per_block only used by PO_SUM_CANONICAL).
 * trace29 (optional): receives the 29 sums of every pass, (max_iteration+1)*29 floats.
 * Returns the number of correspondence passes executed.  Mutates cloud like icp.cpp:125-188. */
int po_icp(po_vec3 *cloud, size_t n, int scene_kind, const void *scene, po_criteria crit,
           int sum_mode, uint32_t points_per_block, po_result *result, float *trace29);
void po_sum29(const po_vec3 *cloud, size_t n, int scene_kind, const void *scene,
              int sum_mode, uint32_t points_per_block, float out29[29]);

/* whole path for a batch of hypotheses (render -> cloud -> ICP), OpenMP over poses; used for
 * known answers at small P and as bench.py's cpu_baseline.  Returns threads used. */
int po_set_threads(int n);   /* OpenMP threads of po_refine_batch (0 = query only); returns the setting in force */
int po_refine_batch(const po_tri *tris, size_t n_tris, const float *poses16, size_t n_poses,
                    size_t width, size_t height, const float proj[16], const float K[9],
                    int scene_kind, const void *scene, po_criteria crit,
                    int sum_mode, uint32_t points_per_block, po_roi roi /* width<=0: none */, po_result *results, uint32_t *cloud_sizes);

#ifdef __cplusplus
}
#endif
#endif
