// pr_host.cpp -- host-side (CPU) parts of the C ABI: model import, projection matrix, scene
// preparation (normals, back-projection, kd-tree build) and the 6x6 update solver.
// These run on the CPU in the reference as well (init_Scene_*_cuda = CPU preparation + H2D copy,
// depth_scene.cu:3-20, pcd_scene.cu:3-20; eigen_slover_666 is host code called from icp.cu:207).
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "pr_internal.h"
#include "pr_solver.inl"

namespace prh {
void solve_666(const float A[36], const float b[6], float T[16]) { prs::solve_666_impl(A, b, T); }
void mat4_mul(const float A[16], const float B[16], float C[16]) { prs::mat4_mul_impl(A, B, C); }
void set_error(const char *fmt, ...);   // pr_api.cpp
}  // namespace prh

namespace {

struct PlyHeader {
    size_t n_vertices = 0, n_faces = 0;
    int vertex_props = 0;
    bool ascii = false, ok = false;
};

PlyHeader read_ply_header(std::ifstream &in)
{
    PlyHeader h;
    std::string line;
    if (!std::getline(in, line) || line.compare(0, 3, "ply") != 0) return h;
    enum { kNone, kVertex, kFace, kOther } section = kNone;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "end_header") { h.ok = true; break; }
        if (tok == "format") { std::string f; ss >> f; h.ascii = (f == "ascii"); }
        else if (tok == "element") {
            std::string name; size_t n = 0;
            ss >> name >> n;
            if (name == "vertex") { h.n_vertices = n; section = kVertex; }
            else if (name == "face") { h.n_faces = n; section = kFace; }
            else section = kOther;
        } else if (tok == "property" && section == kVertex) h.vertex_props++;
    }
    return h;
}

inline uint16_t saturate_u16(int32_t v) { return (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v)); }

// dep2pcd (cuda_icp/scene/common.h:47-61): depth in mm already converted to float
inline pr_vec3 back_project(size_t col, size_t row, float depth_mm, bool zero, const float K[9])
{
    if (zero) return pr_vec3{ 0.0f, 0.0f, 0.0f };
    const float z = depth_mm / 1000.0f;
    return pr_vec3{ ((float)col - K[2]) / K[0] * z, ((float)row - K[5]) / K[4] * z, z };
}

}  // namespace

extern "C" {

int pr_ply_count(const char *path, size_t *n_triangles, size_t *n_vertices)
{
    std::ifstream in(path);
    if (!in) { prh::set_error("pr_ply_count: cannot open %s", path); return PR_ERR_IO; }
    PlyHeader h = read_ply_header(in);
    if (!h.ok || !h.ascii) { prh::set_error("pr_ply_count: %s is not an ASCII PLY", path); return PR_ERR_IO; }
    if (n_triangles) *n_triangles = h.n_faces;
    if (n_vertices) *n_vertices = h.n_vertices;
    return PR_OK;
}

// Replaces Model::LoadModel/recursive_render (cuda_renderer/renderer.cpp:16-104) for plain PLY files:
// the path only consumes `tris`; faces with < 3 indices are dropped (:78), others must be triangles (:79).
int pr_ply_load(const char *path, pr_triangle *tris_out, size_t cap, size_t *n_triangles)
{
    std::ifstream in(path);
    if (!in) { prh::set_error("pr_ply_load: cannot open %s", path); return PR_ERR_IO; }
    PlyHeader h = read_ply_header(in);
    if (!h.ok || !h.ascii || h.vertex_props < 3) { prh::set_error("pr_ply_load: unsupported PLY %s", path); return PR_ERR_IO; }
    std::vector<pr_vec3> verts(h.n_vertices);
    std::string line;
    for (size_t i = 0; i < h.n_vertices; ++i) {
        if (!std::getline(in, line)) { prh::set_error("pr_ply_load: truncated vertex list"); return PR_ERR_IO; }
        const char *p = line.c_str(); char *e = nullptr;
        verts[i].x = strtof(p, &e); p = e;
        verts[i].y = strtof(p, &e); p = e;
        verts[i].z = strtof(p, &e);
    }
    size_t n = 0;
    for (size_t f = 0; f < h.n_faces; ++f) {
        if (!std::getline(in, line)) break;
        const char *p = line.c_str(); char *e = nullptr;
        long k = strtol(p, &e, 10); p = e;
        if (k < 3) continue;
        if (k != 3) { prh::set_error("pr_ply_load: face %zu is not a triangle", f); return PR_ERR_INVALID; }
        long idx[3];
        for (int j = 0; j < 3; ++j) { idx[j] = strtol(p, &e, 10); p = e; }
        for (int j = 0; j < 3; ++j)
            if (idx[j] < 0 || (size_t)idx[j] >= h.n_vertices) { prh::set_error("pr_ply_load: bad vertex index"); return PR_ERR_INVALID; }
        if (n < cap && tris_out) tris_out[n] = pr_triangle{ verts[idx[0]], verts[idx[1]], verts[idx[2]] };
        ++n;
    }
    if (n_triangles) *n_triangles = n;
    return PR_OK;
}

// compute_proj (cuda_renderer/renderer.cpp:161-185), including its sign flips
void pr_compute_proj(const float K[9], int width, int height, float near_, float far_, pr_mat4 *out)
{
    float *p = out->m;
    for (int i = 0; i < 16; ++i) p[i] = 0.0f;
    p[0] = 2 * K[0] / width;
    p[1] = -(-2 * K[1] / width);
    p[2] = -(-2 * K[2] / width + 1);
    p[5] = -(2 * K[4] / height);
    p[6] = -(2 * K[5] / height - 1);
    p[10] = -(-(far_ + near_) / (far_ - near_));
    p[11] = -2 * far_ * near_ / (far_ - near_);
    p[14] = -(-1.0f);
}

// get_normal (cuda_icp/scene/common.cpp:17-107): LINEMOD-style normals from 8 taps at radius 5,
// integer normal equations in 64-bit, |delta| < 50 gate per tap, depth < 2000 gate per pixel.
int pr_get_normal(const uint16_t *depth, int W, int H, const float K[9], pr_vec3 *normals)
{
    if (!depth || !normals || W <= 0 || H <= 0) { prh::set_error("pr_get_normal: bad arguments"); return PR_ERR_INVALID; }
    std::memset(normals, 0, sizeof(pr_vec3) * (size_t)W * H);
    const int R = 5;
    const long kMaxDepth = 2000, kMaxJump = 50;
    const long ox[8] = { -R, 0, R, -R, R, -R, 0, R };
    const long oy[8] = { -R, -R, -R, 0, 0, R, R, R };
    for (int y = R; y < H - R - 1; ++y) {
        for (int x = R; x < W - R - 1; ++x) {
            const size_t at = (size_t)y * W + x;
            const long d0 = depth[at];
            if (d0 >= kMaxDepth) continue;
            long sxx = 0, sxy = 0, syy = 0, bx = 0, by = 0;
            for (int k = 0; k < 8; ++k) {
                const long delta = (long)depth[at + ox[k] + oy[k] * (long)W] - d0;
                if (std::labs(delta) >= kMaxJump) continue;       // f = 0 contributes nothing
                sxx += ox[k] * ox[k]; sxy += ox[k] * oy[k]; syy += oy[k] * oy[k];
                bx += ox[k] * delta;  by += oy[k] * delta;
            }
            const long det = sxx * syy - sxy * sxy;
            const long gx = syy * bx - sxy * by;
            const long gy = -sxy * bx + sxx * by;
            float nx = K[0] * (float)gx;
            float ny = K[4] * (float)gy;
            float nz = (float)(-det * d0);
            const float len = sqrtf(nx * nx + ny * ny + nz * nz);
            if (len > 0) {
                const float inv = 1.0f / len;
                normals[at] = pr_vec3{ nx * inv, ny * inv, nz * inv };
            }
        }
    }
    return PR_OK;
}

// init_Scene_projective_cpu (cuda_icp/scene/depth_scene/depth_scene.cpp:3-35), buffers only
int pr_scene_proj_prepare(const void *depth, int is_i32, const float K[9], size_t W, size_t H,
                          pr_vec3 *pcd_out, pr_vec3 *normal_out)
{
    if (!depth || !pcd_out || !normal_out || W == 0 || H == 0) { prh::set_error("pr_scene_proj_prepare: bad arguments"); return PR_ERR_INVALID; }
    std::vector<uint16_t> d16(W * H);
    if (is_i32) {
        const int32_t *d = static_cast<const int32_t *>(depth);
        for (size_t r = 0; r < H; ++r)
            for (size_t c = 0; c < W; ++c) {
                const uint32_t v = (uint32_t)d[c + r * W];                 // CV_32S read through at<uint32_t>
                pcd_out[c + r * W] = back_project(c, r, (float)v, v == 0, K);
                d16[c + r * W] = saturate_u16(d[c + r * W]);               // get_normal's convertTo(CV_16U)
            }
    } else {
        const uint16_t *d = static_cast<const uint16_t *>(depth);
        for (size_t r = 0; r < H; ++r)
            for (size_t c = 0; c < W; ++c) {
                const uint16_t v = d[c + r * W];
                pcd_out[c + r * W] = back_project(c, r, (float)(int)v, v == 0, K);
                d16[c + r * W] = v;
            }
    }
    return pr_get_normal(d16.data(), (int)W, (int)H, K, normal_out);
}

// KDTree_cpu::build_tree (cuda_icp/scene/pcd_scene/pcd_scene.cpp:45-184): breadth-first, one tree
// level per sweep, children appended pairwise, bbox-midpoint split on the widest axis, stable
// two-ended partition with the alternating tie rule, split value re-centred between the halves;
// finally points/normals are permuted into tree order.
int pr_kdtree_build(pr_vec3 *pcd, pr_vec3 *normal, size_t n, int max_leaf, pr_kdnode *nodes, size_t cap, uint32_t *n_nodes)
{
    if (!pcd || !normal || !nodes || n == 0 || cap == 0) { prh::set_error("pr_kdtree_build: bad arguments"); return PR_ERR_INVALID; }
    std::vector<int> order(n), scratch(n);
    std::iota(order.begin(), order.end(), 0);
    auto blank = [](pr_kdnode &nd) { std::memset(&nd, 0, sizeof nd); nd.parent = nd.child1 = nd.child2 = -1; };
    blank(nodes[0]);
    nodes[0].left = 0; nodes[0].right = (int)n;
    size_t total = 1, sweep_begin = 0, sweep_end = 1;
    while (sweep_begin < sweep_end) {
        const size_t total_before = total;
        for (size_t id = sweep_begin; id < sweep_end; ++id) {
            const int lo = nodes[id].left, hi = nodes[id].right;
            if (hi - lo <= max_leaf) continue;
            if (total + 2 > cap) { prh::set_error("pr_kdtree_build: node capacity %zu too small", cap); return PR_ERR_NOMEM; }
            float bmin[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, bmax[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            for (int k = lo; k < hi; ++k) {
                const float *p = &pcd[order[k]].x;
                for (int a = 0; a < 3; ++a) { if (p[a] > bmax[a]) bmax[a] = p[a]; if (p[a] < bmin[a]) bmin[a] = p[a]; }
            }
            int axis = 0; float cut = 0.0f, widest = -FLT_MAX;
            for (int a = 0; a < 3; ++a) {
                const float extent = bmax[a] - bmin[a];
                if (extent > widest) { widest = extent; axis = a; cut = (bmin[a] + bmax[a]) / 2; }
            }
            int head = lo, tail = hi - 1;
            bool tie_goes_left = true;
            float left_max = -FLT_MAX, right_min = FLT_MAX;
            for (int k = lo; k < hi; ++k) {
                const float v = (&pcd[order[k]].x)[axis];
                if (v == cut) tie_goes_left = !tie_goes_left;
                if (v < cut || (v == cut && tie_goes_left)) { scratch[head++] = order[k]; if (v > left_max) left_max = v; }
                else { scratch[tail--] = order[k]; if (v < right_min) right_min = v; }
            }
            std::copy(scratch.begin() + lo, scratch.begin() + hi, order.begin() + lo);
            pr_kdnode &nd = nodes[id];
            nd.child1 = (int)total; nd.child2 = (int)total + 1;
            nd.split_dim = axis; nd.split_v = (left_max + right_min) / 2;
            for (int a = 0; a < 3; ++a) { nd.bbox[2 * a] = bmin[a]; nd.bbox[2 * a + 1] = bmax[a]; }
            blank(nodes[total]);     nodes[total].parent = (int)id;     nodes[total].left = lo;       nodes[total].right = head;
            blank(nodes[total + 1]); nodes[total + 1].parent = (int)id; nodes[total + 1].left = head; nodes[total + 1].right = hi;
            total += 2;
        }
        sweep_begin = sweep_end;
        sweep_end = (total == total_before) ? sweep_begin : total;
    }
    std::vector<pr_vec3> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = pcd[order[i]];
    std::copy(tmp.begin(), tmp.end(), pcd);
    for (size_t i = 0; i < n; ++i) tmp[i] = normal[order[i]];
    std::copy(tmp.begin(), tmp.end(), normal);
    if (n_nodes) *n_nodes = (uint32_t)total;
    return PR_OK;
}

// init_Scene_nn_cpu (cuda_icp/scene/pcd_scene/pcd_scene.cpp:4-37)
int pr_scene_nn_prepare(const void *depth, int is_i32, const float K[9], int W, int H, int max_leaf,
                        pr_vec3 *pcd_out, pr_vec3 *normal_out, pr_kdnode *nodes_out, size_t cap_nodes,
                        uint32_t *n_points, uint32_t *n_nodes)
{
    if (!depth || W <= 0 || H <= 0) { prh::set_error("pr_scene_nn_prepare: bad arguments"); return PR_ERR_INVALID; }
    const size_t px = (size_t)W * H;
    std::vector<uint16_t> d16(px);
    if (is_i32) { const int32_t *d = static_cast<const int32_t *>(depth); for (size_t i = 0; i < px; ++i) d16[i] = saturate_u16(d[i]); }
    else std::memcpy(d16.data(), depth, px * sizeof(uint16_t));
    std::vector<pr_vec3> all_normals(px);
    int rc = pr_get_normal(d16.data(), W, H, K, all_normals.data());
    if (rc) return rc;
    size_t n = 0;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const uint16_t v = d16[(size_t)r * W + c];
            if (v == 0) continue;
            pcd_out[n] = back_project((size_t)c, (size_t)r, (float)(int)v, false, K);
            normal_out[n] = all_normals[(size_t)r * W + c];
            ++n;
        }
    if (n_points) *n_points = (uint32_t)n;
    if (n == 0) { if (n_nodes) *n_nodes = 0; return PR_OK; }
    return pr_kdtree_build(pcd_out, normal_out, n, max_leaf, nodes_out, cap_nodes, n_nodes);
}

void pr_solve_666(const float A[36], const float b[6], pr_mat4 *T_out) { prs::solve_666_impl(A, b, T_out->m); }

void pr_shard_range(uint32_t n_items, uint32_t rank, uint32_t world, uint32_t *first, uint32_t *count)
{
    // contiguous blocks; the first (n % world) ranks take one extra item
    if (world == 0) world = 1;
    const uint32_t base = n_items / world, extra = n_items % world;
    const uint32_t f = rank * base + (rank < extra ? rank : extra);
    if (first) *first = f;
    if (count) *count = base + (rank < extra ? 1u : 0u);
}

}  // extern "C"
