/*
 * pose_oracle.c -- CPU ORACLE (test infrastructure only; see pose_oracle.h).
 *
 * Plain-C restatement of the reference hot path.  Written from the behaviour of the
 * reference, not copied: same arithmetic, operand order and integer conversions, different
 * code.  All float expressions are evaluated in binary32 in the order the reference writes
 * them; build with -ffp-contract=off (see Makefile) so no FMA is formed.
 *
 * Float->integer conversions that are undefined behaviour in the reference (NaN / out of
 * range) are given x86-64 cvttss2si semantics here (result = INT_MIN / 2^63), which is what
 * the reference CPU build does in practice; the HIP kernels emulate the same.
 */
#include "pose_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* conversions                                                                                 */
/* ------------------------------------------------------------------------------------------ */
static inline int32_t f2i_x86(float v)
{   /* int32_t(float): truncation; NaN/out-of-range -> INT_MIN like cvttss2si */
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT32_MIN;
    return (int32_t)v;
}
static inline uint64_t f2u64_x86(float v)
{   /* size_t(float) as gcc emits it on x86-64 for in-range values; anything else -> 2^63 */
    if (!(v > -1.0f && v < 9223372036854775808.0f)) return (uint64_t)1 << 63;
    return (uint64_t)v;
}

/* ------------------------------------------------------------------------------------------ */
/* renderer                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* cuda_renderer/renderer.cpp:161-185 compute_proj */
void po_compute_proj(const float K[9], int width, int height, float near_, float far_, float p[16])
{
    float t;
    p[0] = 2 * K[0] / width;
    t = -2 * K[1] / width;         p[1] = -t;
    t = -2 * K[2] / width + 1;     p[2] = -t;
    p[3] = 0;
    p[4] = 0;
    t = 2 * K[4] / height;         p[5] = -t;
    t = 2 * K[5] / height - 1;     p[6] = -t;
    p[7] = 0;
    p[8] = 0; p[9] = 0;
    t = -(far_ + near_) / (far_ - near_); p[10] = -t;
    p[11] = -2 * far_ * near_ / (far_ - near_);
    p[12] = 0; p[13] = 0; p[14] = 1; p[15] = 0;
}

/* renderer.h:296-303 mat_mul_v: rows a,b,c of a row-major 4x4 applied to (v,1) */
static inline po_vec3 xform3(const float *m, po_vec3 v)
{
    po_vec3 r;
    r.x = m[0] * v.x + m[1] * v.y + m[2]  * v.z + m[3];
    r.y = m[4] * v.x + m[5] * v.y + m[6]  * v.z + m[7];
    r.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11];
    return r;
}
static inline float sel_max(float a, float b) { return (a > b) ? a : b; }   /* renderer.h:335-336 */
static inline float sel_min(float a, float b) { return (a < b) ? a : b; }   /* renderer.h:337-338 */

/* renderer.h:315-318 calculateSignedArea */
static inline float signed_area(const float *A, const float *B, const float *C)
{
    return 0.5f * ((C[0] - A[0]) * (B[1] - A[1]) - (B[0] - A[0]) * (C[1] - A[1]));
}

/* cuda_renderer/renderer.cpp:190-257 rasterization (CPU flavour: plain min instead of atomicMin) */
static void raster_one(const po_vec3 clip[3], const float w3[3], int32_t *depth_entry,
                       size_t width, size_t height, po_roi roi)
{
    float pts[3][2];
    for (int i = 0; i < 3; i++) {
        const po_vec3 *c = &clip[i];
        pts[i][0] = c->x / w3[i] * (float)width  / 2.0f + (float)width  / 2.0f;
        pts[i][1] = c->y / w3[i] * (float)height / 2.0f + (float)height / 2.0f;
    }
    float lo[2] = { FLT_MAX, FLT_MAX }, hi[2] = { -FLT_MAX, -FLT_MAX };
    float cmax[2] = { (float)(width - 1), (float)(height - 1) };
    float cmin[2] = { 0, 0 };
    size_t real_width = width;
    if (roi.width > 0 && roi.height > 0) {
        cmin[0] = (float)roi.x;
        cmin[1] = (float)(height - 1 - (size_t)(roi.y + roi.height - 1));
        cmax[0] = (float)((roi.x + roi.width) - 1);
        cmax[1] = (float)(height - 1 - (size_t)roi.y);
        real_width = (size_t)roi.width;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) {
            lo[j] = sel_max(cmin[j], sel_min(lo[j], pts[i][j]));
            hi[j] = sel_min(cmax[j], sel_max(hi[j], pts[i][j]));
        }

    /* DOCUMENTED DEVIATION: a zero-area triangle makes 1/area = inf and NaN
     * barycentrics that slip through all six rejects; int32(NaN) then differs between the
     * reference's own CPU and GPU builds.  Both this oracle and the HIP raster skip them. */
    float area = signed_area(pts[0], pts[1], pts[2]);
    if (!(area != 0.0f)) return;
    float base_inv = 1 / area;

    for (uint64_t py = f2u64_x86(lo[1] + 0.5f); (float)py <= hi[1]; py++) {
        for (uint64_t px = f2u64_x86(lo[0] + 0.5f); (float)px <= hi[0]; px++) {
            float P[2] = { (float)px, (float)py };
            /* renderer.h:320-333 barycentric */
            float beta  = signed_area(pts[0], P, pts[2]) * base_inv;
            float gamma = signed_area(pts[0], pts[1], P) * base_inv;
            float alpha = 1.0f - beta - gamma;
            if (alpha < -0.0f || beta < -0.0f || gamma < -0.0f ||
                alpha > 1.0f || beta > 1.0f || gamma > 1.0f) continue;
            float az = alpha / w3[0], bz = beta / w3[1], gz = gamma / w3[2];
            float frag = (alpha + beta + gamma) / (az + bz + gz);
            size_t xw = (size_t)px - (size_t)(long)roi.x;
            size_t yw = height - 1 - (size_t)py - (size_t)(long)roi.y;
            int32_t d = f2i_x86(frag + 0.5f);
            int32_t *cell = &depth_entry[xw + yw * real_width];
            if (d < *cell) *cell = d;
        }
    }
}

/* cuda_renderer/renderer.cpp:259-298 render_cpu (one image per pose, INT_MAX -> 0 at the end) */
void po_render(const po_tri *tris, size_t n_tris, const float *poses16, size_t n_poses,
               size_t width, size_t height, const float proj[16], po_roi roi, int32_t *out)
{
    size_t rw = width, rh = height;
    if (roi.width > 0 && roi.height > 0) { rw = (size_t)roi.width; rh = (size_t)roi.height; }
    size_t total = n_poses * rw * rh;
    for (size_t i = 0; i < total; i++) out[i] = INT32_MAX;

#pragma omp parallel for schedule(dynamic, 1)
    for (long ip = 0; ip < (long)n_poses; ip++) {
        const float *pose = poses16 + 16 * (size_t)ip;
        int32_t *img = out + (size_t)ip * rw * rh;
        for (size_t t = 0; t < n_tris; t++) {
            po_vec3 cam[3] = { xform3(pose, tris[t].v0), xform3(pose, tris[t].v1), xform3(pose, tris[t].v2) };
            float w3[3] = { cam[0].z, cam[1].z, cam[2].z };
            po_vec3 clip[3] = { xform3(proj, cam[0]), xform3(proj, cam[1]), xform3(proj, cam[2]) };
            raster_one(clip, w3, img, width, height, roi);
        }
    }
    for (size_t i = 0; i < total; i++) if (out[i] == INT32_MAX) out[i] = 0;
}

/* ASCII PLY reader standing in for the assimp import of renderer.cpp:16-104: the reference
 * only consumes the triangle list (tris) on this path; faces are emitted in file order. */
static int ply_header(FILE *f, size_t *nv, size_t *nf, int *vprops)
{
    char line[512];
    int in_vertex = 0;
    *nv = *nf = 0; *vprops = 0;
    if (!fgets(line, sizeof line, f) || strncmp(line, "ply", 3) != 0) return -1;
    while (fgets(line, sizeof line, f)) {
        if (strncmp(line, "end_header", 10) == 0) return 0;
        if (strncmp(line, "format", 6) == 0 && !strstr(line, "ascii")) return -2;
        if (strncmp(line, "element vertex", 14) == 0) { *nv = strtoull(line + 14, NULL, 10); in_vertex = 1; }
        else if (strncmp(line, "element face", 12) == 0) { *nf = strtoull(line + 12, NULL, 10); in_vertex = 0; }
        else if (strncmp(line, "element", 7) == 0) in_vertex = 0;
        else if (strncmp(line, "property", 8) == 0 && in_vertex) (*vprops)++;
    }
    return -3;
}
size_t po_ply_count(const char *path, size_t *n_vertices)
{
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    size_t nv, nf; int vp;
    int rc = ply_header(f, &nv, &nf, &vp);
    fclose(f);
    if (rc) return 0;
    if (n_vertices) *n_vertices = nv;
    return nf;
}
int po_ply_load(const char *path, po_tri *tris_out, size_t cap_tris)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    size_t nv, nf; int vp;
    if (ply_header(f, &nv, &nf, &vp) || vp < 3) { fclose(f); return -2; }
    po_vec3 *v = (po_vec3 *)malloc(nv * sizeof(po_vec3));
    char line[1024];
    for (size_t i = 0; i < nv; i++) {
        if (!fgets(line, sizeof line, f)) { free(v); fclose(f); return -3; }
        char *p = line;
        v[i].x = strtof(p, &p); v[i].y = strtof(p, &p); v[i].z = strtof(p, &p);
    }
    size_t n = 0;
    for (size_t i = 0; i < nf; i++) {
        if (!fgets(line, sizeof line, f)) break;
        char *p = line;
        long cnt = strtol(p, &p, 10);
        if (cnt < 3) continue;                       /* renderer.cpp:78 skips degenerate faces */
        if (cnt != 3) { free(v); fclose(f); return -4; }
        long a = strtol(p, &p, 10), b = strtol(p, &p, 10), c = strtol(p, &p, 10);
        if (n < cap_tris) { tris_out[n].v0 = v[a]; tris_out[n].v1 = v[b]; tris_out[n].v2 = v[c]; }
        n++;
    }
    free(v); fclose(f);
    return (int)n;
}

/* ------------------------------------------------------------------------------------------ */
/* depth -> cloud   (cuda_icp/icp.cpp:73-117 depth2cloud_cpu<T>)                                */
/* ------------------------------------------------------------------------------------------ */
#define DEPTH2CLOUD_BODY(T)                                                                        \
    uint32_t gw = width / stride, gh = height / stride;                                            \
    size_t n = 0;                                                                                  \
    /* output order = exclusive scan over mask[x + y*width], i.e. row-major (icp.cpp:85-95).      \
     * stride>1 overruns the mask in the reference; here the strided grid is       \
     * walked row-major, which equals the reference for stride==1. */                              \
    for (uint32_t y = 0; y < gh; y++)                                                              \
        for (uint32_t x = 0; x < gw; x++) {                                                        \
            T d = depth[(size_t)x * stride + (size_t)y * stride * width];                          \
            if (d <= 0) continue;                                                                  \
            float z = d / 1000.0f;                                                                 \
            po_vec3 p;                                                                             \
            p.x = ((float)(x + tl_x) - K[2]) / K[0] * z;                                           \
            p.y = ((float)(y + tl_y) - K[5]) / K[4] * z;                                           \
            p.z = z;                                                                               \
            if (cloud_out) cloud_out[n] = p;                                                       \
            n++;                                                                                   \
        }                                                                                          \
    return n;

size_t po_depth2cloud_i32(const int32_t *depth, uint32_t width, uint32_t height, const float K[9],
                          uint32_t stride, uint32_t tl_x, uint32_t tl_y, po_vec3 *cloud_out)
{ DEPTH2CLOUD_BODY(int32_t) }
size_t po_depth2cloud_u16(const uint16_t *depth, uint32_t width, uint32_t height, const float K[9],
                          uint32_t stride, uint32_t tl_x, uint32_t tl_y, po_vec3 *cloud_out)
{ DEPTH2CLOUD_BODY(uint16_t) }

/* ------------------------------------------------------------------------------------------ */
/* scene preparation                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* cv::Mat::convertTo(CV_32S -> CV_16U) = saturate_cast<ushort>(int)  (common.cpp:23, pcd_scene.cpp:11) */
void po_depth_i32_to_u16(const int32_t *in, uint16_t *out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        int32_t v = in[i];
        out[i] = (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
    }
}

/* cuda_icp/scene/common.cpp:3-15 accumBilateral */
static inline void bilateral_tap(long delta, long i, long j, long A[4], long b[2], long thr)
{
    long f = (labs(delta) < thr) ? 1 : 0;
    long fi = f * i, fj = f * j;
    A[0] += fi * i; A[1] += fi * j; A[3] += fj * j;
    b[0] += fi * delta; b[1] += fj * delta;
}

/* cuda_icp/scene/common.cpp:17-107 get_normal (LINEMOD depth-modality normals, radius 5) */
void po_get_normal(const uint16_t *dep, int W, int H, const float K[9], po_vec3 *normals)
{
    const int r = 5;
    const long dist_thr = 2000, diff_thr = 50;
    memset(normals, 0, (size_t)W * H * sizeof(po_vec3));
    static const int tap[8][2] = { {-1,-1},{0,-1},{1,-1},{-1,0},{1,0},{-1,1},{0,1},{1,1} };
    for (int y = r; y < H - r - 1; y++)
        for (int x = r; x < W - r - 1; x++) {
            const uint16_t *c = dep + (size_t)y * W + x;
            long d = c[0];
            if (!(d < dist_thr)) continue;
            long A[4] = { 0, 0, 0, 0 }, b[2] = { 0, 0 };
            for (int k = 0; k < 8; k++) {
                long i = tap[k][0] * r, j = tap[k][1] * r;
                bilateral_tap((long)c[i + j * (long)W] - d, i, j, A, b, diff_thr);
            }
            long det = A[0] * A[3] - A[1] * A[1];
            long ddx = A[3] * b[0] - A[1] * b[1];
            long ddy = -A[1] * b[0] + A[0] * b[1];
            float nx = K[0] * (float)ddx;
            float ny = K[4] * (float)ddy;
            float nz = (float)(-det * d);
            float len = sqrtf(nx * nx + ny * ny + nz * nz);
            if (len > 0) {
                float inv = 1.0f / len;
                po_vec3 *o = normals + (size_t)y * W + x;
                o->x = nx * inv; o->y = ny * inv; o->z = nz * inv;
            }
        }
}

/* cuda_icp/scene/common.h:47-61 dep2pcd (dep already widened to an unsigned/int value) */
static inline po_vec3 dep2pcd_f(size_t x, size_t y, float dep_as_float, int is_zero, const float K[9])
{
    po_vec3 p = { 0, 0, 0 };
    if (is_zero) return p;
    float z = dep_as_float / 1000.0f;
    p.x = ((float)x - K[2]) / K[0] * z;
    p.y = ((float)y - K[5]) / K[4] * z;
    p.z = z;
    return p;
}

/* cuda_icp/scene/depth_scene/depth_scene.cpp:3-35 init_Scene_projective_cpu (buffers only) */
void po_scene_proj_init(const void *depth, int is_i32, const float K[9], size_t W, size_t H,
                        po_vec3 *pcd, po_vec3 *nrm)
{
    size_t n = W * H;
    uint16_t *d16 = (uint16_t *)malloc(n * sizeof(uint16_t));
    if (is_i32) {
        const int32_t *d = (const int32_t *)depth;
        for (size_t r = 0; r < H; r++)
            for (size_t c = 0; c < W; c++) {
                uint32_t v = (uint32_t)d[c + r * W];          /* read through at<uint32_t> (:26) */
                pcd[c + r * W] = dep2pcd_f(c, r, (float)v, v == 0, K);
            }
        po_depth_i32_to_u16(d, d16, n);
    } else {
        const uint16_t *d = (const uint16_t *)depth;
        for (size_t r = 0; r < H; r++)
            for (size_t c = 0; c < W; c++) {
                uint16_t v = d[c + r * W];
                pcd[c + r * W] = dep2pcd_f(c, r, (float)(int)v, v == 0, K);
            }
        memcpy(d16, d, n * sizeof(uint16_t));
    }
    po_get_normal(d16, (int)W, (int)H, K, nrm);
    free(d16);
}

/* cuda_icp/scene/pcd_scene/pcd_scene.cpp:4-31 init_Scene_nn_cpu up to build_tree():
 * uint16 conversion, normals, valid pixels gathered row-major */
size_t po_scene_nn_gather(const void *depth, int is_i32, const float K[9], int W, int H,
                          po_vec3 *pcd, po_vec3 *nrm)
{
    size_t n = (size_t)W * H, cnt = 0;
    uint16_t *d16 = (uint16_t *)malloc(n * sizeof(uint16_t));
    if (is_i32) po_depth_i32_to_u16((const int32_t *)depth, d16, n);
    else memcpy(d16, depth, n * sizeof(uint16_t));
    po_vec3 *all = (po_vec3 *)malloc(n * sizeof(po_vec3));
    po_get_normal(d16, W, H, K, all);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            uint16_t v = d16[(size_t)r * W + c];
            if (v > 0) {
                pcd[cnt] = dep2pcd_f((size_t)c, (size_t)r, (float)(int)v, 0, K);
                nrm[cnt] = all[(size_t)r * W + c];
                cnt++;
            }
        }
    free(all); free(d16);
    return cnt;
}

static inline float axis_of(po_vec3 p, int d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }

/* cuda_icp/scene/pcd_scene/pcd_scene.cpp:45-184 KDTree_cpu::build_tree -- level-order, no
 * recursion; reorders pcd/normal in place by the final index permutation (:173-183). */
size_t po_kd_build(po_vec3 *pcd, po_vec3 *nrm, size_t n, int max_leaf, po_kdnode *nodes, size_t cap)
{
    if (n == 0 || cap == 0) return 0;
    int *idx = (int *)malloc(n * sizeof(int)), *tmp = (int *)malloc(n * sizeof(int));
    for (size_t i = 0; i < n; i++) idx[i] = (int)i;
    memset(nodes, 0, cap * sizeof(po_kdnode));
    for (size_t i = 0; i < cap; i++) { nodes[i].parent = nodes[i].child1 = nodes[i].child2 = -1; }
    nodes[0].left = 0; nodes[0].right = (int)n;

    size_t count = 1, lvl_lo = 0, lvl_hi = 1;
    for (;;) {
        int grew = 0;
        size_t next_lo = lvl_hi, this_hi = lvl_hi;
        for (size_t ni = lvl_lo; ni < this_hi; ni++) {
            int L = nodes[ni].left, R = nodes[ni].right;
            if (R - L <= max_leaf) continue;
            if (count + 2 > cap) { free(idx); free(tmp); return 0; }
            float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            for (int k = L; k < R; k++) {
                po_vec3 p = pcd[idx[k]];
                float c3[3] = { p.x, p.y, p.z };
                for (int d = 0; d < 3; d++) {
                    if (c3[d] > mx[d]) mx[d] = c3[d];
                    if (c3[d] < mn[d]) mn[d] = c3[d];
                }
            }
            int dim = 0; float split = 0, best_span = -FLT_MAX;
            for (int d = 0; d < 3; d++) {
                float span = mx[d] - mn[d], mid = (mn[d] + mx[d]) / 2;
                if (span > best_span) { best_span = span; dim = d; split = mid; }
            }
            /* two-ended partition with the alternating tie switch (:113-133) */
            int lo = L, hi = R - 1, toggle = 1;
            float below = -FLT_MAX, above = FLT_MAX;
            for (int k = L; k < R; k++) {
                float v = axis_of(pcd[idx[k]], dim);
                if (v == split) toggle = !toggle;
                if (v < split || (v == split && toggle)) { tmp[lo++] = idx[k]; if (v > below) below = v; }
                else                                     { tmp[hi--] = idx[k]; if (v < above) above = v; }
            }
            split = (below + above) / 2;                       /* :135 */
            for (int k = L; k < R; k++) idx[k] = tmp[k];

            po_kdnode *nd = &nodes[ni];
            nd->child1 = (int)count; nd->child2 = (int)count + 1;
            nd->split_v = split; nd->split_dim = dim;
            for (int d = 0; d < 3; d++) { nd->bbox[2 * d] = mn[d]; nd->bbox[2 * d + 1] = mx[d]; }
            nodes[count].left = L;      nodes[count].right = lo;     nodes[count].parent = (int)ni;
            nodes[count + 1].left = lo; nodes[count + 1].right = R;  nodes[count + 1].parent = (int)ni;
            count += 2; grew = 1;
        }
        if (!grew) break;
        lvl_lo = next_lo; lvl_hi = count;
    }
    po_vec3 *buf = (po_vec3 *)malloc(n * sizeof(po_vec3));
    for (size_t i = 0; i < n; i++) buf[i] = pcd[idx[i]];
    memcpy(pcd, buf, n * sizeof(po_vec3));
    for (size_t i = 0; i < n; i++) buf[i] = nrm[idx[i]];
    memcpy(nrm, buf, n * sizeof(po_vec3));
    free(buf); free(idx); free(tmp);
    return count;
}

/* ------------------------------------------------------------------------------------------ */
/* correspondence queries                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* cuda_icp/scene/common.h:63-73 pcd2dep + depth_scene.h:29-48 Scene_projective::query */
int po_query_proj(const po_scene_proj *s, po_vec3 src, po_vec3 *dst, po_vec3 *nrm)
{
    const float *K = s->K;
    int x = f2i_x86(src.x / src.z * K[0] + K[2] - (float)s->tl_x + 0.5f);   /* size_t tl_x enters the float expression as float */
    int y = f2i_x86(src.y / src.z * K[4] + K[5] - (float)s->tl_y + 0.5f);
    if (x < 0 || y < 0 || (size_t)x >= s->width || (size_t)y >= s->height) return 0;
    size_t i = (size_t)x + (size_t)y * s->width;
    po_vec3 d = s->pcd[i];
    float dz = src.z - d.z;
    float adz = (dz > 0) ? dz : -dz;                            /* std__abs common.h:75-77 */
    if (d.z <= 0 || adz > s->max_dist_diff) return 0;
    *dst = d; *nrm = s->normal[i];
    return 1;
}

static inline float sq(float v) { return v * v; }

/* cuda_icp/scene/pcd_scene/pcd_scene.h:60-136 Scene_nn::query -- stackless kd traversal */
int po_query_nn(const po_scene_nn *s, po_vec3 src, po_vec3 *dst, po_vec3 *nrm, int *winner,
                float *dist_sq, uint32_t *node_visits)
{
    const po_kdnode *nodes = s->nodes;
    int cur = 0, prev = -1, climbing = 0, best_i = 0;
    float best = FLT_MAX;
    uint32_t visits = 0;
    while (cur >= 0) {
        const po_kdnode *nd = &nodes[cur];
        visits++;
        float diff = 0;
        if (nd->split_dim == 0) diff = src.x - nd->split_v;
        if (nd->split_dim == 1) diff = src.y - nd->split_v;
        if (nd->split_dim == 2) diff = src.z - nd->split_v;
        int near_c = (diff < 0) ? nd->child1 : nd->child2;
        int far_c  = (diff < 0) ? nd->child2 : nd->child1;
        int leaf = (nd->child1 < 0 || nd->child2 < 0);
        if (!climbing) {
            if (leaf) {
                for (int i = nd->left; i < nd->right; i++) {
                    float d2 = sq(src.x - s->pcd[i].x) + sq(src.y - s->pcd[i].y) + sq(src.z - s->pcd[i].z);
                    if (d2 < best) { best = d2; best_i = i; }
                }
                climbing = 1; prev = cur; cur = nd->parent;
            } else { prev = cur; cur = near_c; }
        } else {
            float lb = 0;
            if (src.x < nd->bbox[0]) lb += sq(nd->bbox[0] - src.x); else if (src.x > nd->bbox[1]) lb += sq(nd->bbox[1] - src.x);
            if (src.y < nd->bbox[2]) lb += sq(nd->bbox[2] - src.y); else if (src.y > nd->bbox[3]) lb += sq(nd->bbox[3] - src.y);
            if (src.z < nd->bbox[4]) lb += sq(nd->bbox[4] - src.z); else if (src.z > nd->bbox[5]) lb += sq(nd->bbox[5] - src.z);
            if (prev == near_c && lb <= best) { prev = cur; cur = far_c; climbing = 0; }
            else { prev = cur; cur = nd->parent; }
        }
    }
    if (node_visits) *node_visits = visits;
    if (winner) *winner = best_i;
    if (dist_sq) *dist_sq = best;
    if (best < sq(s->max_dist_diff)) { *dst = s->pcd[best_i]; *nrm = s->normal[best_i]; return 1; }
    return 0;
}

/* cuda_icp/icp.h:138-206 thrust__pcd2Ab::operator() for a VALID correspondence */
void po_contrib29(po_vec3 s, po_vec3 d, po_vec3 n, float out[29])
{
    float dx = d.x - s.x, dy = d.y - s.y, dz = d.z - s.z;
    float r = dx * n.x + dy * n.y + dz * n.z;
    float J[6];
    J[0] = n.z * s.y - n.y * s.z;
    J[1] = n.x * s.z - n.z * s.x;
    J[2] = n.y * s.x - n.x * s.y;
    J[3] = n.x; J[4] = n.y; J[5] = n.z;
    int k = 0;
    for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) out[k++] = J[a] * J[b];
    for (int a = 0; a < 6; a++) out[21 + a] = J[a] * r;
    out[27] = dx * dx + dy * dy + dz * dz;
    out[28] = 1;
}

/* ------------------------------------------------------------------------------------------ */
/* 6x6 solve (cuda_icp/icp.cpp:7-45) -- THIRD-PARTY arithmetic: Eigen (unpinned version,
 * cuda_icp/CMakeLists.txt:18), restated from its published algorithm:
 *   LDLT<Matrix6d>: symmetric pivoting on the largest |diagonal| entry, unblocked, lower;
 *   solve: P, L^-1, D^+ (pseudo-inverse with tolerance 1/highest), L^-T, P^T;
 *   AngleAxisd*AngleAxisd -> Quaterniond product; Quaterniond::toRotationMatrix.
 * "parity unpinned": no reference test holds a vector for it.                                   */
/* ------------------------------------------------------------------------------------------ */
static void ldlt6_solve(double M[6][6], const double rhs[6], double x[6])
{
    int perm[6];
    for (int k = 0; k < 6; k++) {
        int piv = k; double big = fabs(M[k][k]);
        for (int i = k + 1; i < 6; i++) if (fabs(M[i][i]) > big) { big = fabs(M[i][i]); piv = i; }
        perm[k] = piv;
        if (piv != k) {                     /* symmetric row/column swap on the lower triangle */
            for (int j = 0; j < k; j++) { double t = M[k][j]; M[k][j] = M[piv][j]; M[piv][j] = t; }
            for (int i = piv + 1; i < 6; i++) { double t = M[i][k]; M[i][k] = M[i][piv]; M[i][piv] = t; }
            for (int i = k + 1; i < piv; i++) { double t = M[i][k]; M[i][k] = M[piv][i]; M[piv][i] = t; }
            double t = M[k][k]; M[k][k] = M[piv][piv]; M[piv][piv] = t;
        }
        double tmp[6];
        for (int j = 0; j < k; j++) tmp[j] = M[j][j] * M[k][j];
        double acc = 0; for (int j = 0; j < k; j++) acc += M[k][j] * tmp[j];
        M[k][k] -= acc;
        for (int i = k + 1; i < 6; i++) {
            double a2 = 0; for (int j = 0; j < k; j++) a2 += M[i][j] * tmp[j];
            M[i][k] -= a2;
        }
        double d = M[k][k];
        if (fabs(d) > 0) for (int i = k + 1; i < 6; i++) M[i][k] /= d;
    }
    double y[6];
    for (int i = 0; i < 6; i++) y[i] = rhs[i];
    for (int k = 0; k < 6; k++) if (perm[k] != k) { double t = y[k]; y[k] = y[perm[k]]; y[perm[k]] = t; }
    for (int i = 0; i < 6; i++) for (int j = 0; j < i; j++) y[i] -= M[i][j] * y[j];
    const double tol = 1.0 / DBL_MAX;
    for (int i = 0; i < 6; i++) y[i] = (fabs(M[i][i]) > tol) ? y[i] / M[i][i] : 0.0;
    for (int i = 5; i >= 0; i--) for (int j = i + 1; j < 6; j++) y[i] -= M[j][i] * y[j];
    for (int k = 5; k >= 0; k--) if (perm[k] != k) { double t = y[k]; y[k] = y[perm[k]]; y[perm[k]] = t; }
    for (int i = 0; i < 6; i++) x[i] = y[i];
}

typedef struct { double w, x, y, z; } quatd;
static quatd quat_mul(quatd a, quatd b)
{
    quatd r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

void po_solve666(const float A[36], const float b[6], float T[16])
{
    double M[6][6], rhs[6], u[6];
    /* Eigen maps A column-major (icp.cpp:31); A is symmetric so the transpose is harmless */
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) M[r][c] = (double)A[c * 6 + r] + (r == c ? 0.01 * 1.0 : 0.01 * 0.0);
    for (int i = 0; i < 6; i++) rhs[i] = (double)b[i];
    ldlt6_solve(M, rhs, u);
    /* icp.cpp:7-17: R = Rz(u2) * Ry(u1) * Rx(u0), composed as quaternions by Eigen */
    quatd qz = { cos(0.5 * u[2]), 0, 0, sin(0.5 * u[2]) };
    quatd qy = { cos(0.5 * u[1]), 0, sin(0.5 * u[1]), 0 };
    quatd qx = { cos(0.5 * u[0]), sin(0.5 * u[0]), 0, 0 };
    quatd q = quat_mul(quat_mul(qz, qy), qx);
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    double R[9] = { 1 - (tyy + tzz), txy - twz, txz + twy,
                    txy + twz, 1 - (txx + tzz), tyz - twx,
                    txz - twy, tyz + twx, 1 - (txx + tyy) };
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[r * 4 + c] = (float)R[r * 3 + c];
        T[r * 4 + 3] = (float)u[3 + r];
    }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

/* cuda_icp/geometry.h:106-111 + :292-298: each entry = dot(row, col) summed in index order 3,2,1,0 */
void po_mat4_mul(const float A[16], const float B[16], float C[16])
{
    float out[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = 0;
            for (int k = 3; k >= 0; k--) acc += A[i * 4 + k] * B[k * 4 + j];
            out[i * 4 + j] = acc;
        }
    memcpy(C, out, sizeof out);
}

/* cuda_icp/icp.cpp:47-59 transform_pcd */
void po_transform_cloud(po_vec3 *cloud, size_t n, const float T[16])
{
    for (size_t i = 0; i < n; i++) {
        po_vec3 p = cloud[i], q;
        q.x = T[0] * p.x + T[1] * p.y + T[2]  * p.z + T[3];
        q.y = T[4] * p.x + T[5] * p.y + T[6]  * p.z + T[7];
        q.z = T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11];
        cloud[i] = q;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* transform-reduce of the 29-float accumulator                                                */
/* ------------------------------------------------------------------------------------------ */
static inline void point29(const po_vec3 *cloud, size_t j, int kind, const void *scene, float out[29])
{
    po_vec3 d, nr;
    int ok = (kind == PO_SCENE_PROJ) ? po_query_proj((const po_scene_proj *)scene, cloud[j], &d, &nr)
                                     : po_query_nn((const po_scene_nn *)scene, cloud[j], &d, &nr, NULL, NULL, NULL);
    if (ok) po_contrib29(cloud[j], d, nr, out);
    else memset(out, 0, 29 * sizeof(float));
}

/* The canonical tree the HIP kernels implement:
 *   workgroup = 256 lanes = 4 wavefronts; workgroup g owns points [g*PPB, (g+1)*PPB), PPB = 1024*S;
 *   lane t accumulates, starting from 0, points g*PPB + s*1024 + 256*i + t for s=0..S-1, i=0..3 in
 *   that order (adjacent lanes hold adjacent points); each wavefront is reduced by a balanced pairwise tree over its 64 lanes in lane
 *   order (adjacent pairs first); the 4 wavefront sums are added as ((w0+w1)+w2)+w3; workgroup
 *   sums are added sequentially in workgroup order starting from 0. */
static void sum29_canonical(const po_vec3 *cloud, size_t n, int kind, const void *scene,
                            uint32_t ppb, float out[29])
{
    if (ppb < 1024) ppb = 1024;
    uint32_t steps = ppb / 1024;
    size_t groups = (n + ppb - 1) / ppb;
    float total[29]; memset(total, 0, sizeof total);
    float (*lane)[29] = (float (*)[29])malloc(256 * 29 * sizeof(float));
    for (size_t g = 0; g < groups; g++) {
        for (int t = 0; t < 256; t++) {
            float *acc = lane[t];
            for (int c = 0; c < 29; c++) acc[c] = 0;
            for (uint32_t s = 0; s < steps; s++)
                for (int i = 0; i < 4; i++) {
                    size_t j = g * (size_t)ppb + (size_t)s * 1024 + (size_t)i * 256 + (size_t)t;
                    if (j >= n) continue;
                    float c29[29];
                    point29(cloud, j, kind, scene, c29);
                    for (int c = 0; c < 29; c++) acc[c] += c29[c];
                }
        }
        float wsum[4][29];
        for (int w = 0; w < 4; w++) {
            for (int stride = 1; stride < 64; stride *= 2)
                for (int l = 0; l < 64; l += 2 * stride)
                    for (int c = 0; c < 29; c++) lane[64 * w + l][c] = lane[64 * w + l][c] + lane[64 * w + l + stride][c];
            memcpy(wsum[w], lane[64 * w], 29 * sizeof(float));
        }
        for (int c = 0; c < 29; c++) {
            float b = ((wsum[0][c] + wsum[1][c]) + wsum[2][c]) + wsum[3][c];
            total[c] += b;
        }
    }
    free(lane);
    memcpy(out, total, sizeof total);
}

void po_sum29(const po_vec3 *cloud, size_t n, int kind, const void *scene, int sum_mode,
              uint32_t ppb, float out[29])
{
    if (sum_mode == PO_SUM_CANONICAL) { sum29_canonical(cloud, n, kind, scene, ppb, out); return; }
    /* icp.cpp:139-148 with one thread: reducer starts at 0 and adds points in index order */
    float acc[29]; memset(acc, 0, sizeof acc);
    for (size_t j = 0; j < n; j++) {
        float c29[29];
        point29(cloud, j, kind, scene, c29);
        for (int c = 0; c < 29; c++) acc[c] += c29[c];
    }
    memcpy(out, acc, sizeof acc);
}

/* cuda_icp/icp.cpp:125-188 ICP_Point2Plane_cpu */
int po_icp(po_vec3 *cloud, size_t n, int kind, const void *scene, po_criteria crit,
           int sum_mode, uint32_t ppb, po_result *res, float *trace29)
{
    po_result cur, prev;
    memset(&cur, 0, sizeof cur);
    cur.T[0] = cur.T[5] = cur.T[10] = cur.T[15] = 1;
    int passes = 0;
    for (uint32_t it = 0; it <= (uint32_t)crit.max_iteration; it++) {
        float Ab[29];
        po_sum29(cloud, n, kind, scene, sum_mode, ppb, Ab);
        if (trace29) memcpy(trace29 + 29 * (size_t)it, Ab, sizeof Ab);
        passes++;
        prev = cur;
        float count = Ab[28], err = Ab[27];
        if (count == 0) break;
        cur.fitness = count / (float)n;
        cur.inlier_rmse = sqrtf(err / count);
        if (it == (uint32_t)crit.max_iteration) break;
        if (fabsf(cur.fitness - prev.fitness) < crit.relative_fitness &&
            fabsf(cur.inlier_rmse - prev.inlier_rmse) < crit.relative_rmse) break;
        float A[36], b[6];
        for (int i = 0; i < 6; i++) b[i] = Ab[21 + i];
        int k = 0;
        for (int y = 0; y < 6; y++) for (int x = y; x < 6; x++) { A[x + y * 6] = Ab[k]; A[y + x * 6] = Ab[k]; k++; }
        float E[16];
        po_solve666(A, b, E);
        po_transform_cloud(cloud, n, E);
        po_mat4_mul(E, cur.T, cur.T);
    }
    *res = cur;
    return passes;
}

/* number of OpenMP threads the batch loop below uses (0: leave the runtime's choice); returns the setting in force.  bench.py sets
 * it to what the host really grants the process -- min(affinity mask, cgroup cpu.max quota) -- because omp_get_max_threads() reports
 * every online CPU of the box, and a cgroup with a 16-CPU quota runs 256 such threads on 16 CPUs' worth of time. */
int po_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

/* whole path per hypothesis: render_cpu(1 pose) -> depth2cloud_cpu -> ICP_Point2Plane_cpu
 * (the per-pose loop of BASELINE.md section 3), OpenMP across hypotheses. */
int po_refine_batch(const po_tri *tris, size_t n_tris, const float *poses16, size_t n_poses,
                    size_t width, size_t height, const float proj[16], const float K[9],
                    int kind, const void *scene, po_criteria crit, int sum_mode, uint32_t ppb, po_roi roi,
                    po_result *results, uint32_t *cloud_sizes)
{
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
#endif
    /* with an ROI (renderer.h:199): the hypothesis is rendered into a roi.width x roi.height image and the cloud is
       extracted with tl = (roi.x, roi.y) (icp.h:57-60), which restores full-frame coordinates */
    const int has_roi = roi.width > 0 && roi.height > 0;
    const size_t rw = has_roi ? (size_t)roi.width : width, rh = has_roi ? (size_t)roi.height : height;
    /* every thread keeps ONE depth image and ONE cloud for all its hypotheses (the reference's per-pose loop returns fresh
       std::vectors, but 128 threads each mmap-ing and faulting 4.9 MB per hypothesis measure the kernel's page-fault path,
       not the algorithm: a baseline should not be slowed by that) */
#pragma omp parallel
    {
        int32_t *depth = (int32_t *)malloc(width * height * sizeof(int32_t));
        po_vec3 *cloud = (po_vec3 *)malloc(width * height * sizeof(po_vec3));
#pragma omp for schedule(dynamic, 1)
        for (long ip = 0; ip < (long)n_poses; ip++) {
            po_render(tris, n_tris, poses16 + 16 * (size_t)ip, 1, width, height, proj, roi, depth);
            size_t n = po_depth2cloud_i32(depth, (uint32_t)rw, (uint32_t)rh, K, 1, has_roi ? (uint32_t)roi.x : 0u, has_roi ? (uint32_t)roi.y : 0u, cloud);
            po_icp(cloud, n, kind, scene, crit, sum_mode, ppb, &results[ip], NULL);
            if (cloud_sizes) cloud_sizes[ip] = (uint32_t)n;
        }
        free(cloud); free(depth);
    }
    return threads;
}
