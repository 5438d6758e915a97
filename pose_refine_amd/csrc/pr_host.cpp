// pr_host.cpp -- host-side (CPU) parts of the C ABI: model import, projection matrix, scene
// preparation (normals, back-projection, kd-tree build) and the 6x6 update solver.
// These run on the CPU in the reference as well (init_Scene_*_cuda = CPU preparation + H2D copy,
// depth_scene.cu:3-20, pcd_scene.cu:3-20; eigen_slover_666 is host code called from icp.cu:207).
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "pr_internal.h"
#include "pr_solver.inl"

namespace prh {
void solve_666(const float A[36], const float b[6], float T[16]) { prs::solve_666_impl(A, b, T); }
void mat4_mul(const float A[16], const float B[16], float C[16]) { prs::mat4_mul_impl(A, B, C); }
void set_error(const char *fmt, ...);   // pr_context.cpp
}  // namespace prh

namespace {

// ---- mesh import: what Model::LoadModel takes from assimp (cuda_renderer/renderer.cpp:16-104) ----------------------------
// The reference hands the file to assimp (aiProcessPreset_TargetRealtime_Quality, which triangulates) and walks the scene graph,
// multiplying node transforms into the vertices (recursive_render).  The formats this path is fed are single-mesh files without
// a node hierarchy, so the graph walk reduces to the identity transform; what remains is the file parsing, done here for PLY
// (ASCII, binary little / big endian, any scalar property types, extra vertex properties skipped) and Wavefront OBJ:
//   tris      every face as triangles (faces with < 3 indices dropped, renderer.cpp:78; polygons fanned like aiProcess_Triangulate)
//   vertices  the file's vertex list, faces the index triples (renderer.cpp:87-100)
//   bbox      component-wise min / max over the vertices (get_bounding_box, renderer.cpp:106-150)
struct Mesh {
    std::vector<pr_vec3> vertices;       // as in the file, untransformed (renderer.cpp:93-99)
    std::vector<int32_t> faces;          // 3 per triangle: indices into the vertex list of the MESH the face belongs to (renderer.cpp:87-91)
    std::vector<pr_triangle> tris;       // hierarchical formats: triangles with the node transforms multiplied in (recursive_render);
                                         // empty for the flat formats, whose triangles are vertices[faces]
    bool has_box = false; float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };   // hierarchical formats: box of the TRANSFORMED vertices (get_bounding_box_for_node)
    std::string error;
};

enum PlyType { kI8, kU8, kI16, kU16, kI32, kU32, kF32, kF64, kBad };
PlyType ply_type(const std::string &t)
{
    if (t == "char" || t == "int8") return kI8;       if (t == "uchar" || t == "uint8") return kU8;
    if (t == "short" || t == "int16") return kI16;    if (t == "ushort" || t == "uint16") return kU16;
    if (t == "int" || t == "int32") return kI32;      if (t == "uint" || t == "uint32") return kU32;
    if (t == "float" || t == "float32") return kF32;  if (t == "double" || t == "float64") return kF64;
    return kBad;
}
size_t ply_size(PlyType t) { static const size_t sz[] = { 1, 1, 2, 2, 4, 4, 4, 8, 0 }; return sz[t]; }
struct PlyProp { std::string name; bool is_list = false; PlyType count_type = kBad, type = kBad; };
struct PlyElement { std::string name; size_t count = 0; std::vector<PlyProp> props; };

double ply_read_binary(std::istream &in, PlyType t, bool swap, bool &ok)
{
    unsigned char b[8] = { 0 };
    const size_t n = ply_size(t);
    if (!in.read(reinterpret_cast<char *>(b), (std::streamsize)n)) { ok = false; return 0; }
    if (swap) for (size_t i = 0; i < n / 2; ++i) std::swap(b[i], b[n - 1 - i]);
    switch (t) {
    case kI8: return (double)*reinterpret_cast<int8_t *>(b);     case kU8: return (double)b[0];
    case kI16: { int16_t v; std::memcpy(&v, b, 2); return v; }   case kU16: { uint16_t v; std::memcpy(&v, b, 2); return v; }
    case kI32: { int32_t v; std::memcpy(&v, b, 4); return v; }   case kU32: { uint32_t v; std::memcpy(&v, b, 4); return v; }
    case kF32: { float v; std::memcpy(&v, b, 4); return v; }     case kF64: { double v; std::memcpy(&v, b, 8); return v; }
    default: ok = false; return 0;
    }
}

void add_polygon(Mesh &m, const std::vector<long> &idx, size_t n_vertices)
{
    if (idx.size() < 3) return;                                   // renderer.cpp:78
    for (long i : idx) if (i < 0 || (size_t)i >= n_vertices) { m.error = "face refers to a vertex that does not exist"; return; }
    for (size_t k = 1; k + 1 < idx.size(); ++k) { m.faces.push_back((int32_t)idx[0]); m.faces.push_back((int32_t)idx[k]); m.faces.push_back((int32_t)idx[k + 1]); }
}

bool load_ply(const char *path, Mesh &m)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) { m.error = std::string("cannot open ") + path; return false; }
    in.seekg(0, std::ios::end);
    const size_t file_size = (size_t)std::max<std::streamoff>(0, in.tellg());   // no element can have more entries than the file has bytes
    in.seekg(0, std::ios::beg);
    std::string line;
    if (!std::getline(in, line) || line.compare(0, 3, "ply") != 0) { m.error = "not a PLY file"; return false; }
    bool ascii = false, big = false, header_ok = false;
    std::vector<PlyElement> elems;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "end_header") { header_ok = true; break; }
        if (tok == "format") { std::string f; ss >> f; ascii = (f == "ascii"); big = (f == "binary_big_endian"); if (!ascii && !big && f != "binary_little_endian") { m.error = "unknown PLY format " + f; return false; } }
        else if (tok == "element") {
            PlyElement e; double cnt = -1;
            ss >> e.name >> cnt;
            if (!(cnt >= 0) || cnt > (double)file_size) { m.error = "PLY element count that the file cannot hold: " + line; return false; }
            e.count = (size_t)cnt; elems.push_back(e);
        }
        else if (tok == "property" && !elems.empty()) {
            PlyProp p; std::string t; ss >> t;
            if (t == "list") { std::string ct, it; ss >> ct >> it >> p.name; p.is_list = true; p.count_type = ply_type(ct); p.type = ply_type(it); if (p.count_type == kBad) { m.error = "unknown PLY type " + ct; return false; } }
            else { ss >> p.name; p.type = ply_type(t); }
            if (p.type == kBad) { m.error = "unknown PLY property type in: " + line; return false; }
            elems.back().props.push_back(p);
        }
    }
    if (!header_ok) { m.error = "PLY header without end_header"; return false; }
    const uint16_t probe = 1;
    const bool host_little = *reinterpret_cast<const unsigned char *>(&probe) == 1;
    const bool swap = !ascii && (big == host_little);
    size_t n_vertices = 0;
    for (const PlyElement &e : elems) {
        const bool is_vertex = (e.name == "vertex"), is_face = (e.name == "face");
        int ix = -1, iy = -1, iz = -1, ilist = -1;
        for (size_t k = 0; k < e.props.size(); ++k) {
            if (is_vertex && !e.props[k].is_list) { if (e.props[k].name == "x") ix = (int)k; else if (e.props[k].name == "y") iy = (int)k; else if (e.props[k].name == "z") iz = (int)k; }
            if (is_face && e.props[k].is_list && ilist < 0 && (e.props[k].name == "vertex_indices" || e.props[k].name == "vertex_index")) ilist = (int)k;
        }
        if (is_vertex && (ix < 0 || iy < 0 || iz < 0)) { m.error = "PLY vertex element without x / y / z"; return false; }
        if (is_vertex) { n_vertices = e.count; m.vertices.resize(e.count); }
        std::vector<long> poly;
        for (size_t i = 0; i < e.count; ++i) {
            std::istringstream ls;
            if (ascii) { do { if (!std::getline(in, line)) { m.error = "truncated PLY body"; return false; } } while (line.find_first_not_of(" \t\r") == std::string::npos); ls.str(line); }
            for (size_t k = 0; k < e.props.size(); ++k) {
                const PlyProp &p = e.props[k];
                bool ok = true;
                auto next = [&](PlyType t) { double v = 0; if (ascii) { if (!(ls >> v)) ok = false; } else v = ply_read_binary(in, t, swap, ok); return v; };
                if (!p.is_list) {
                    const double v = next(p.type);
                    if (!ok) { m.error = "truncated PLY body"; return false; }
                    if (is_vertex) { if ((int)k == ix) m.vertices[i].x = (float)v; else if ((int)k == iy) m.vertices[i].y = (float)v; else if ((int)k == iz) m.vertices[i].z = (float)v; }
                } else {
                    const double cnt = next(p.count_type);
                    if (!ok || cnt < 0 || cnt > 1e6) { m.error = "bad PLY list"; return false; }
                    poly.clear();
                    for (long j = 0; j < (long)cnt; ++j) {
                        const double v = next(p.type);
                        if (!ok) { m.error = "truncated PLY body"; return false; }
                        poly.push_back((v >= 0 && v < 9.0e15) ? (long)v : -1L);      // (a negative, NaN or absurd index is refused by add_polygon)
                    }
                    if (is_face && (int)k == ilist) { add_polygon(m, poly, n_vertices); if (!m.error.empty()) return false; }
                }
            }
        }
    }
    return true;
}

bool load_obj(const char *path, Mesh &m)
{
    std::ifstream in(path);
    if (!in) { m.error = std::string("cannot open ") + path; return false; }
    std::string line;
    struct Pending { std::vector<long> idx; };
    std::vector<Pending> polys;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "v") { pr_vec3 v{ 0, 0, 0 }; ss >> v.x >> v.y >> v.z; m.vertices.push_back(v); }
        else if (tok == "f") {
            Pending p; std::string ref;
            while (ss >> ref) { long i = strtol(ref.c_str(), nullptr, 10); p.idx.push_back(i < 0 ? (long)m.vertices.size() + i : i - 1); }   // v, v/vt, v/vt/vn, v//vn; negative = relative
            polys.push_back(std::move(p));
        }
    }
    for (const Pending &p : polys) { add_polygon(m, p.idx, m.vertices.size()); if (!m.error.empty()) return false; }
    return true;
}


// ---- glTF 2.0 (.gltf + external / embedded buffers, .glb): the hierarchical format of this importer -------------------------------
// What the reference gets from assimp for such a file and does with it (renderer.cpp:69-104 recursive_render): every node's matrix
// is multiplied onto its parent's (m = m * node), every mesh of the node contributes its triangles with m applied to the three
// vertices (mat_mul_vec: a1*x + a2*y + a3*z + a4 per row, in float), the vertex list stays untransformed, faces keep their per-mesh
// indices, and the bounding box is taken over the transformed vertices (get_bounding_box_for_node).  A glTF primitive is an assimp
// mesh; triangle lists, strips and fans are read (assimp triangulates the latter two), points and lines are skipped like the
// reference skips faces with fewer than three indices.  assimp itself is absent from the reference tree (parity unpinned): node
// matrices given as translation / rotation / scale are composed as T * R * S with the usual quaternion-to-matrix formula.
struct Json {
    enum Kind { kNull, kBool, kNum, kStr, kArr, kObj } kind = kNull;
    double num = 0; bool b = false; std::string str;
    std::vector<Json> arr; std::vector<std::pair<std::string, Json>> obj;
    const Json *get(const char *key) const { for (const auto &kv : obj) if (kv.first == key) return &kv.second; return nullptr; }
    double number(const char *key, double dflt) const { const Json *j = get(key); return (j && j->kind == kNum) ? j->num : dflt; }
    // a field that must be a non-negative integer (an index, a count, a byte offset); false if it is present and is not one
    bool index(const char *key, size_t dflt, size_t &out) const
    {
        const Json *j = get(key);
        if (!j) { out = dflt; return true; }
        if (j->kind != kNum || !(j->num >= 0.0) || !(j->num <= 4503599627370496.0) || j->num != std::floor(j->num)) return false;
        out = (size_t)j->num; return true;
    }
};
bool json_index(const Json &j, size_t &out)
{
    if (j.kind != Json::kNum || !(j.num >= 0.0) || !(j.num <= 4503599627370496.0) || j.num != std::floor(j.num)) return false;
    out = (size_t)j.num; return true;
}
struct JsonParser {
    const char *p, *end; std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    bool value(Json &out, int depth = 0)
    {
        if (depth > 64) return fail("JSON nested too deeply");
        ws();
        if (p >= end) return fail("unexpected end of JSON");
        if (*p == '{') {
            ++p; out.kind = Json::kObj; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                Json k; ws();
                if (p >= end || *p != '"' || !string(k.str)) return fail("JSON object key expected");
                ws(); if (p >= end || *p != ':') return fail("':' expected"); ++p;
                Json v; if (!value(v, depth + 1)) return false;
                out.obj.emplace_back(std::move(k.str), std::move(v));
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail("',' or '}' expected");
            }
        }
        if (*p == '[') {
            ++p; out.kind = Json::kArr; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                Json v; if (!value(v, depth + 1)) return false;
                out.arr.push_back(std::move(v));
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("',' or ']' expected");
            }
        }
        if (*p == '"') { out.kind = Json::kStr; return string(out.str); }
        if (end - p >= 4 && !strncmp(p, "true", 4)) { p += 4; out.kind = Json::kBool; out.b = true; return true; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; out.kind = Json::kBool; out.b = false; return true; }
        if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; out.kind = Json::kNull; return true; }
        char *q = nullptr;
        const std::string tmp(p, (size_t)std::min<ptrdiff_t>(end - p, 64));
        out.num = strtod(tmp.c_str(), &q);
        if (q == tmp.c_str()) return fail("JSON value expected");
        p += q - tmp.c_str(); out.kind = Json::kNum; return true;
    }
    bool string(std::string &out)
    {
        ++p;                                                      // opening quote
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) { case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                              case 'u': { if (end - p < 5) return fail("bad \\u escape"); const unsigned c = (unsigned)strtoul(std::string(p + 1, 4).c_str(), nullptr, 16); out += (c < 128) ? (char)c : '?'; p += 4; break; }
                              default: out += *p; }
                ++p;
            } else out += *p++;
        }
        if (p >= end) return fail("unterminated JSON string");
        ++p; return true;
    }
};
bool base64_decode(const std::string &in, size_t from, std::vector<unsigned char> &out)
{
    auto val = [](char c) -> int { if (c >= 'A' && c <= 'Z') return c - 'A'; if (c >= 'a' && c <= 'z') return c - 'a' + 26; if (c >= '0' && c <= '9') return c - '0' + 52; if (c == '+') return 62; if (c == '/') return 63; return -1; };
    unsigned acc = 0; int bits = 0;
    for (size_t i = from; i < in.size(); ++i) {
        if (in[i] == '=') break;
        const int v = val(in[i]);
        if (v < 0) continue;
        acc = (acc << 6) | (unsigned)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((unsigned char)((acc >> bits) & 0xffu)); }
    }
    return true;
}
struct Mat4 { float m[16]; };                                      // row-major a1..d4 like aiMatrix4x4
Mat4 mat4_identity() { Mat4 r; for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f; return r; }
Mat4 mat4_mul(const Mat4 &a, const Mat4 &t)                        // aiMultiplyMatrix4(&a, &t): a = a * t, sums in index order
{
    Mat4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
        r.m[4 * i + j] = a.m[4 * i] * t.m[j] + a.m[4 * i + 1] * t.m[4 + j] + a.m[4 * i + 2] * t.m[8 + j] + a.m[4 * i + 3] * t.m[12 + j];
    return r;
}
pr_vec3 mat4_apply(const Mat4 &m, const pr_vec3 &v)               // renderer.cpp:60-67 mat_mul_vec
{
    return pr_vec3{ m.m[0] * v.x + m.m[1] * v.y + m.m[2] * v.z + m.m[3], m.m[4] * v.x + m.m[5] * v.y + m.m[6] * v.z + m.m[7],
                    m.m[8] * v.x + m.m[9] * v.y + m.m[10] * v.z + m.m[11] };
}
struct Gltf {
    Json root; std::vector<std::vector<unsigned char>> buffers; std::string dir; Mesh *out = nullptr; size_t visits = 0;
    bool fail(const std::string &e) { if (out->error.empty()) out->error = e; return false; }
    const Json *item(const char *array, size_t i) const { const Json *a = root.get(array); return (a && a->kind == Json::kArr && i < a->arr.size()) ? &a->arr[i] : nullptr; }
    // accessor -> doubles (count x components)
    bool accessor(size_t index, int want_comps, std::vector<double> &vals)
    {
        const Json *acc = item("accessors", index);
        if (!acc) return fail("glTF: accessor out of range");
        size_t count = 0, ctype = 0, view_i = 0, acc_off = 0, view_off = 0, stride = 0, buf_i = 0;
        if (!acc->index("count", 0, count) || !acc->index("componentType", 0, ctype) || !acc->index("bufferView", (size_t)-1, view_i) || !acc->index("byteOffset", 0, acc_off))
            return fail("glTF: accessor field is not a non-negative integer");
        const Json *ty = acc->get("type");
        const int comps = (ty && ty->str == "VEC3") ? 3 : ((ty && ty->str == "SCALAR") ? 1 : 0);
        if (comps != want_comps) return fail("glTF: accessor of an unexpected type");
        if (acc->get("sparse")) return fail("glTF: sparse accessors are not read");
        const Json *view = item("bufferViews", view_i);
        if (!view) return fail("glTF: accessor without a buffer view");
        if (!view->index("buffer", (size_t)-1, buf_i) || !view->index("byteOffset", 0, view_off) || !view->index("byteStride", 0, stride))
            return fail("glTF: buffer view field is not a non-negative integer");
        if (buf_i >= buffers.size()) return fail("glTF: buffer out of range");
        const std::vector<unsigned char> &buf = buffers[buf_i];
        size_t csize = 0;
        switch (ctype) { case 5120: case 5121: csize = 1; break; case 5122: case 5123: csize = 2; break; case 5125: case 5126: csize = 4; break; default: return fail("glTF: unknown component type"); }
        // positions are floats, indices unsigned integers (glTF 2.0 3.6.2, 3.7.2.1; quantised positions are an extension that is not read)
        if (want_comps == 3 ? ctype != 5126 : (ctype != 5121 && ctype != 5123 && ctype != 5125)) return fail("glTF: accessor of an unexpected component type");
        const size_t off = view_off + acc_off, elem = csize * (size_t)comps;
        if (stride == 0) stride = elem;
        if (stride > 65536 || off > buf.size() || elem > buf.size() - off || (count > 0 && (count - 1) > (buf.size() - off - elem) / stride)) return fail("glTF: accessor reaches beyond its buffer");
        vals.resize(count * (size_t)comps);
        for (size_t i = 0; i < count; ++i) for (int c = 0; c < comps; ++c) {
            const unsigned char *src = buf.data() + off + i * stride + (size_t)c * csize;
            double v = 0;
            switch (ctype) { case 5120: v = *reinterpret_cast<const int8_t *>(src); break; case 5121: v = *src; break;
                             case 5122: { int16_t t; std::memcpy(&t, src, 2); v = t; break; } case 5123: { uint16_t t; std::memcpy(&t, src, 2); v = t; break; }
                             case 5125: { uint32_t t; std::memcpy(&t, src, 4); v = t; break; } case 5126: { float t; std::memcpy(&t, src, 4); v = t; break; } }
            vals[i * (size_t)comps + (size_t)c] = v;
        }
        return true;
    }
    Mat4 local(const Json &node) const
    {
        const Json *mj = node.get("matrix");
        Mat4 r = mat4_identity();
        if (mj && mj->kind == Json::kArr && mj->arr.size() == 16) { for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 4; ++rr) r.m[4 * rr + c] = (float)mj->arr[(size_t)(4 * c + rr)].num; return r; }   // column-major in the file
        float t[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 1 }, sc[3] = { 1, 1, 1 };
        auto read = [&](const char *k, float *dst, size_t n) { const Json *a = node.get(k); if (a && a->kind == Json::kArr && a->arr.size() == n) for (size_t i = 0; i < n; ++i) dst[i] = (float)a->arr[i].num; };
        read("translation", t, 3); read("rotation", q, 4); read("scale", sc, 3);
        const float x = q[0], y = q[1], z = q[2], w = q[3];
        const float R[9] = { 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y) };
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) r.m[4 * i + j] = R[3 * i + j] * sc[j]; r.m[4 * i + 3] = t[i]; }
        return r;
    }
    bool node(size_t index, Mat4 m, int depth)
    {
        if (depth > 256) return fail("glTF: node hierarchy too deep (a cycle?)");
        if (++visits > (1u << 20)) return fail("glTF: more than 2^20 node visits (nodes shared between parents?)");
        const Json *nd = item("nodes", index);
        if (!nd) return fail("glTF: node out of range");
        m = mat4_mul(m, local(*nd));                                // renderer.cpp:71
        const Json *mesh_ref = nd->get("mesh");
        if (mesh_ref) {
            size_t mesh_i = 0;
            if (!json_index(*mesh_ref, mesh_i)) return fail("glTF: mesh reference is not a non-negative integer");
            const Json *mesh = item("meshes", mesh_i);
            const Json *prims = mesh ? mesh->get("primitives") : nullptr;
            if (!prims || prims->kind != Json::kArr) return fail("glTF: mesh without primitives");
            for (const Json &pr : prims->arr) {                     // one assimp mesh per primitive
                size_t mode = 4, pos_i = 0, ind_i = 0;
                if (!pr.index("mode", 4, mode)) return fail("glTF: primitive mode is not a non-negative integer");
                const Json *attrs = pr.get("attributes");
                const Json *pos = attrs ? attrs->get("POSITION") : nullptr;
                if (!pos) continue;
                if (!json_index(*pos, pos_i)) return fail("glTF: POSITION is not an accessor index");
                std::vector<double> pv, iv;
                if (!accessor(pos_i, 3, pv)) return false;
                const size_t nv = pv.size() / 3;
                const Json *ind = pr.get("indices");
                if (ind) { if (!json_index(*ind, ind_i)) return fail("glTF: indices is not an accessor index"); if (!accessor(ind_i, 1, iv)) return false; }
                else { iv.resize(nv); for (size_t i = 0; i < nv; ++i) iv[i] = (double)i; }
                std::vector<pr_vec3> verts(nv);
                for (size_t i = 0; i < nv; ++i) verts[i] = pr_vec3{ (float)pv[3 * i], (float)pv[3 * i + 1], (float)pv[3 * i + 2] };
                auto tri = [&](size_t a, size_t b, size_t c) -> bool {
                    if (a >= nv || b >= nv || c >= nv) return fail("glTF: index refers to a vertex that does not exist");
                    out->tris.push_back(pr_triangle{ mat4_apply(m, verts[a]), mat4_apply(m, verts[b]), mat4_apply(m, verts[c]) });
                    out->faces.push_back((int32_t)a); out->faces.push_back((int32_t)b); out->faces.push_back((int32_t)c);
                    return true;
                };
                if (mode == 4) { for (size_t k = 0; k + 2 < iv.size(); k += 3) if (!tri((size_t)iv[k], (size_t)iv[k + 1], (size_t)iv[k + 2])) return false; }
                else if (mode == 5) { for (size_t k = 0; k + 2 < iv.size(); ++k) if (!((k & 1) ? tri((size_t)iv[k + 1], (size_t)iv[k], (size_t)iv[k + 2]) : tri((size_t)iv[k], (size_t)iv[k + 1], (size_t)iv[k + 2]))) return false; }
                else if (mode == 6) { for (size_t k = 1; k + 1 < iv.size(); ++k) if (!tri((size_t)iv[0], (size_t)iv[k], (size_t)iv[k + 1])) return false; }
                // (points, lines: faces with fewer than three indices give no triangle, renderer.cpp:78 -- but assimp keeps such a mesh's vertices,
                // and the reference appends the vertices of EVERY mesh and walks them for the box, so they count here as well)
                for (const pr_vec3 &v : verts) {                    // renderer.cpp:93-99, and get_bounding_box_for_node on the transformed copy
                    out->vertices.push_back(v);
                    const pr_vec3 tv = mat4_apply(m, v);
                    const float c3[3] = { tv.x, tv.y, tv.z };
                    for (int a = 0; a < 3; ++a) { out->lo[a] = std::min(out->lo[a], c3[a]); out->hi[a] = std::max(out->hi[a], c3[a]); }
                }
            }
        }
        const Json *kids = nd->get("children");
        if (kids && kids->kind == Json::kArr) for (const Json &k : kids->arr) { size_t ki = 0; if (!json_index(k, ki)) return fail("glTF: child is not a node index"); if (!node(ki, m, depth + 1)) return false; }
        return true;
    }
};
// percent-decoded relative path of a buffer / image uri; false for anything that could leave the model's directory
bool uri_to_relative_path(const std::string &uri, std::string &out)
{
    out.clear();
    for (size_t i = 0; i < uri.size(); ++i) {
        if (uri[i] == '%') {
            auto hex = [](char c) -> int { return (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1; };
            if (i + 2 >= uri.size()) return false;                // '%' needs two hex digits behind it
            const int hi = hex(uri[i + 1]), lo = hex(uri[i + 2]);
            if (hi < 0 || lo < 0 || (hi == 0 && lo == 0)) return false;
            out.push_back((char)(hi * 16 + lo));
            i += 2;
        } else if (uri[i] == '?' || uri[i] == '#') return false;   // queries and fragments make no sense for a file
        else out.push_back(uri[i]);
    }
    if (out.empty() || out[0] == '/' || out[0] == '\\') return false;
    if (out.find(':') != std::string::npos) return false;          // scheme or drive prefix
    size_t start = 0;
    while (start <= out.size()) {                                   // no ".." segment, with either separator
        size_t end = out.find_first_of("/\\", start);
        if (end == std::string::npos) end = out.size();
        if (out.compare(start, end - start, "..") == 0) return false;
        start = end + 1;
    }
    return true;
}
bool read_file(const std::string &path, std::vector<unsigned char> &out)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    out.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
    return true;
}
bool load_gltf(const char *path, Mesh &m, bool binary)
{
    std::vector<unsigned char> file;
    if (!read_file(path, file)) { m.error = std::string("cannot open ") + path; return false; }
    Gltf g; g.out = &m;
    const std::string sp(path);
    const size_t slash = sp.find_last_of("/\\");
    g.dir = slash == std::string::npos ? "" : sp.substr(0, slash + 1);
    std::string json;
    std::vector<unsigned char> glb_bin; bool have_glb_bin = false;
    if (binary) {
        if (file.size() < 20 || std::memcmp(file.data(), "glTF", 4) != 0) { m.error = "not a GLB file"; return false; }
        size_t off = 12;
        while (off + 8 <= file.size()) {
            uint32_t len, type; std::memcpy(&len, file.data() + off, 4); std::memcpy(&type, file.data() + off + 4, 4);
            if (off + 8 + len > file.size()) { m.error = "GLB chunk reaches beyond the file"; return false; }
            if (type == 0x4E4F534Au) json.assign(reinterpret_cast<const char *>(file.data() + off + 8), len);
            else if (type == 0x004E4942u && !have_glb_bin) { glb_bin.assign(file.begin() + (long)(off + 8), file.begin() + (long)(off + 8 + len)); have_glb_bin = true; }
            off += 8 + ((len + 3u) & ~3u);
        }
    } else json.assign(file.begin(), file.end());
    JsonParser jp{ json.data(), json.data() + json.size(), std::string() };
    if (!jp.value(g.root) || g.root.kind != Json::kObj) { m.error = "glTF: " + (jp.err.empty() ? std::string("not a JSON object") : jp.err); return false; }
    const Json *bufs = g.root.get("buffers");
    if (bufs && bufs->kind == Json::kArr) for (size_t i = 0; i < bufs->arr.size(); ++i) {
        std::vector<unsigned char> data;
        const Json *uri = bufs->arr[i].get("uri");
        if (!uri || uri->kind != Json::kStr) { if (i == 0 && have_glb_bin) data = glb_bin; else { m.error = "glTF: buffer without a uri"; return false; } }
        else if (uri->str.compare(0, 5, "data:") == 0) { const size_t comma = uri->str.find(','); if (comma == std::string::npos) { m.error = "glTF: malformed data uri"; return false; } base64_decode(uri->str, comma + 1, data); }
        else {
            // a relative file reference (glTF 2.0 2.8: RFC 3986 relative path, percent-encoded): decode %XX, and stay inside the model's
            // directory -- no absolute paths, no drive or scheme prefixes, no ".." segments (a crafted file must not read arbitrary files)
            std::string rel;
            if (!uri_to_relative_path(uri->str, rel)) { m.error = "glTF: buffer uri is not a relative path inside the model's directory: " + uri->str; return false; }
            if (!read_file(g.dir + rel, data)) { m.error = "glTF: cannot open buffer " + g.dir + rel; return false; }
        }
        g.buffers.push_back(std::move(data));
    }
    m.has_box = true;
    for (int a = 0; a < 3; ++a) { m.lo[a] = 1e10f; m.hi[a] = -1e10f; }     // renderer.cpp:147-148
    const Json *scenes = g.root.get("scenes");
    size_t scene_i = 0;
    if (!g.root.index("scene", 0, scene_i)) { m.error = "glTF: scene is not a non-negative integer"; return false; }
    const Json *scene = g.item("scenes", scene_i);
    if (scenes && scene) {
        const Json *roots = scene->get("nodes");
        if (roots && roots->kind == Json::kArr) for (const Json &r : roots->arr) { size_t ri = 0; if (!json_index(r, ri)) { m.error = "glTF: scene root is not a node index"; return false; } if (!g.node(ri, mat4_identity(), 0)) return false; }
    } else {                                                      // no scene: every node that is nobody's child is a root
        const Json *nodes = g.root.get("nodes");
        if (nodes && nodes->kind == Json::kArr) {
            std::vector<char> is_child(nodes->arr.size(), 0);
            for (const Json &n : nodes->arr) { const Json *k = n.get("children"); if (k && k->kind == Json::kArr) for (const Json &c : k->arr) { size_t ci = 0; if (json_index(c, ci) && ci < is_child.size()) is_child[ci] = 1; } }
            for (size_t i = 0; i < nodes->arr.size(); ++i) if (!is_child[i] && !g.node(i, mat4_identity(), 0)) return false;
        }
    }
    return m.error.empty();
}

bool load_mesh(const char *path, Mesh &m)
{
    if (!path) { m.error = "null path"; return false; }
    const std::string s(path);
    const size_t dot = s.rfind('.');
    std::string ext = dot == std::string::npos ? "" : s.substr(dot + 1);
    for (char &c : ext) c = (char)tolower((unsigned char)c);
    if (ext == "obj") return load_obj(path, m);
    if (ext == "gltf") return load_gltf(path, m, false);
    if (ext == "glb") return load_gltf(path, m, true);
    return load_ply(path, m);                                     // .ply and anything that starts with the PLY magic
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

// Model::Model(fileName) / LoadModel (cuda_renderer/renderer.cpp:11-58)
// load_mesh behind the C boundary: whatever a malformed file makes the containers throw becomes an error code
static bool load_mesh_noexcept(const char *path, Mesh &m)
{
    try { return load_mesh(path, m); }
    catch (const std::bad_alloc &) { m.error = "out of memory while reading the file (a count in it is absurd?)"; }
    catch (const std::exception &e) { m.error = std::string("malformed file: ") + e.what(); }
    catch (...) { m.error = "malformed file"; }
    return false;
}
int pr_mesh_count(const char *path, size_t *n_triangles, size_t *n_vertices)
{
    Mesh m;
    if (!path) { prh::set_error("pr_mesh_count: null path"); return PR_ERR_INVALID; }
    if (!load_mesh_noexcept(path, m)) { prh::set_error("pr_mesh_count: %s", m.error.c_str()); return PR_ERR_IO; }
    if (n_triangles) *n_triangles = m.faces.size() / 3;
    if (n_vertices) *n_vertices = m.vertices.size();
    return PR_OK;
}
int pr_mesh_load(const char *path, pr_triangle *tris_out, size_t cap_triangles, size_t *n_triangles,
                 pr_vec3 *vertices_out, size_t cap_vertices, size_t *n_vertices, int32_t *faces_out,
                 float bbox_min[3], float bbox_max[3])
{
    Mesh m;
    if (!path) { prh::set_error("pr_mesh_load: null path"); return PR_ERR_INVALID; }
    if (!load_mesh_noexcept(path, m)) { prh::set_error("pr_mesh_load: %s", m.error.c_str()); return PR_ERR_IO; }
    const size_t nt = m.faces.size() / 3;
    for (size_t t = 0; t < nt && t < cap_triangles; ++t) {
        const int32_t *f = &m.faces[3 * t];
        if (tris_out) tris_out[t] = m.tris.empty() ? pr_triangle{ m.vertices[(size_t)f[0]], m.vertices[(size_t)f[1]], m.vertices[(size_t)f[2]] } : m.tris[t];
        if (faces_out) { faces_out[3 * t] = f[0]; faces_out[3 * t + 1] = f[1]; faces_out[3 * t + 2] = f[2]; }
    }
    if (vertices_out) for (size_t v = 0; v < m.vertices.size() && v < cap_vertices; ++v) vertices_out[v] = m.vertices[v];
    if (n_triangles) *n_triangles = nt;
    if (n_vertices) *n_vertices = m.vertices.size();
    // get_bounding_box (renderer.cpp:143-150) starts from +-1e10 and folds every vertex in
    float lo[3] = { 1e10f, 1e10f, 1e10f }, hi[3] = { -1e10f, -1e10f, -1e10f };
    for (const pr_vec3 &v : m.vertices) {
        lo[0] = std::min(lo[0], v.x); lo[1] = std::min(lo[1], v.y); lo[2] = std::min(lo[2], v.z);
        hi[0] = std::max(hi[0], v.x); hi[1] = std::max(hi[1], v.y); hi[2] = std::max(hi[2], v.z);
    }
    if (m.has_box) for (int a = 0; a < 3; ++a) { lo[a] = m.lo[a]; hi[a] = m.hi[a]; }      // a hierarchical file: the box of the transformed vertices
    if (bbox_min) for (int a = 0; a < 3; ++a) bbox_min[a] = lo[a];
    if (bbox_max) for (int a = 0; a < 3; ++a) bbox_max[a] = hi[a];
    return PR_OK;
}
// the two original entry points (triangles only)
int pr_ply_count(const char *path, size_t *n_triangles, size_t *n_vertices) { return pr_mesh_count(path, n_triangles, n_vertices); }
int pr_ply_load(const char *path, pr_triangle *tris_out, size_t cap, size_t *n_triangles)
{
    return pr_mesh_load(path, tris_out, cap, n_triangles, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
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
// mat<4,4,float> * mat<4,4,float> (cuda_icp/geometry.h:292-298): the product the ICP loop accumulates its result with (icp.cu:212);
// the same source (pr_solver.inl) is compiled for the device-side loop
void pr_mat4_mul(const pr_mat4 *A, const pr_mat4 *B, pr_mat4 *C_out) { prs::mat4_mul_impl(A->m, B->m, C_out->m); }

void pr_shard_range(uint32_t n_items, uint32_t rank, uint32_t world, uint32_t *first, uint32_t *count)
{
    // contiguous blocks; the first (n % world) ranks take one extra item
    if (world == 0) world = 1;
    if (rank >= world) { if (first) *first = n_items; if (count) *count = 0; return; }      // no such rank: an empty block behind the last one
    const uint32_t base = n_items / world, extra = n_items % world;
    const uint32_t f = rank * base + (rank < extra ? rank : extra);
    if (first) *first = f;
    if (count) *count = base + (rank < extra ? 1u : 0u);
}

}  // extern "C"
