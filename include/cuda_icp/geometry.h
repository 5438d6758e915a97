// geometry.h -- host-side small vector / matrix algebra carrying the NAMES, members and memory layout user code of the
// reference's cuda_icp/geometry.h (lines 22-334) relies on: vec<N,T> (x/y/z members for N = 2, 3), mat<R,C,T> (rows of vec),
// the Vec3f / Mat3x3f / Mat4x4f typedef family, embed / proj / cross, det / cofactor / adjugate / invert.  Layout: Vec3f = 12 B,
// Mat3x3f = 36 B, Mat4x4f = 64 B, row-major plain floats, which is what the C ABI (include/pose_refine.h) takes.
//
// The implementation is this repo's own, built on the three small folding helpers in pr_geom below.  Only the ORDER of float
// operations is dictated by the reference, because it is observable in the last bit: every dot product -- and so mat * vec,
// mat * mat, det -- accumulates from the HIGHEST index down to 0, starting from T() (geometry.h:106-111,164-170,285-298).
// tests/golden/geometry_h.json holds bit patterns the reference header computed (oracle/Makefile `ref fixtures`);
// tests/test_geometry_golden.py runs this header through the same driver and compares bit for bit.
#pragma once
#include <cassert>
#include <cmath>
#include <cstddef>
#include <ostream>
#include <iostream>

template <size_t N, typename T> struct vec;
template <size_t R, size_t C, typename T> class mat;

namespace pr_geom {

// s = T(); s += term(N-1); ... ; s += term(0)      -- the one accumulation order everything numeric below goes through
template <size_t N, typename T, typename Term>
inline T fold_high_to_low(Term term)
{
    T s = T();
    size_t i = N;
    while (i != 0) { --i; s += term(i); }
    return s;
}

// out[i] = f(i) for every i; element-wise, so the visiting order is not observable
template <size_t N, typename Out, typename F>
inline void fill_each(Out &out, F f)
{
    for (size_t i = 0; i != N; ++i) out[i] = f(i);
}

// index of the k-th row/column that survives deleting `gone`
inline size_t skip(size_t k, size_t gone) { return k < gone ? k : k + 1; }

// Laplace expansion along row 0; the 1x1 overload ends the recursion
template <typename T> inline T determinant(const mat<1, 1, T> &m);
template <size_t N, typename T> inline T determinant(const mat<N, N, T> &m);

}  // namespace pr_geom

// ---------------------------------------------------------------------------------------------------------------------------------
// vectors
// ---------------------------------------------------------------------------------------------------------------------------------
template <size_t N, typename T>
struct vec {
    vec() { for (T &e : v_) e = T(); }
    T &operator[](size_t i) { assert(i < N); return v_[i]; }
    const T &operator[](size_t i) const { assert(i < N); return v_[i]; }
    static vec Zero() { return vec(); }
    vec &operator+=(const vec &o)
    {
        pr_geom::fill_each<N>(*this, [&](size_t i) { return v_[i] + o.v_[i]; });
        return *this;
    }
    vec operator+(const vec &o) { vec sum(*this); sum += o; return sum; }

private:
    T v_[N];
};

template <typename T>
struct vec<2, T> {
    T x, y;
    vec() : x(T()), y(T()) {}
    vec(T x_, T y_) : x(x_), y(y_) {}
    template <class U> vec(const vec<2, U> &other);                 // only the float <-> int pairs below exist
    T &operator[](size_t i) { assert(i < 2); return i ? y : x; }
    const T &operator[](size_t i) const { assert(i < 2); return i ? y : x; }
};

template <typename T>
struct vec<3, T> {
    T x, y, z;
    vec() : x(T()), y(T()), z(T()) {}
    vec(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    template <class U> vec(const vec<3, U> &other);
    T &operator[](size_t i)
    {
        assert(i < 3);
        switch (i) { case 0: return x; case 1: return y; default: return z; }
    }
    const T &operator[](size_t i) const
    {
        assert(i < 3);
        switch (i) { case 0: return x; case 1: return y; default: return z; }
    }
    float norm() { return std::sqrt(x * x + y * y + z * z); }
    vec &normalize(T length = 1)
    {
        const T scale = length / norm();
        x *= scale; y *= scale; z *= scale;
        return *this;
    }
};

template <size_t N, typename T>
inline T operator*(const vec<N, T> &a, const vec<N, T> &b)
{
    return pr_geom::fold_high_to_low<N, T>([&](size_t i) { return a[i] * b[i]; });
}
template <size_t N, typename T>
inline vec<N, T> operator+(vec<N, T> a, const vec<N, T> &b)
{
    pr_geom::fill_each<N>(a, [&](size_t i) { return a[i] + b[i]; });
    return a;
}
template <size_t N, typename T>
inline vec<N, T> operator-(vec<N, T> a, const vec<N, T> &b)
{
    pr_geom::fill_each<N>(a, [&](size_t i) { return a[i] - b[i]; });
    return a;
}
template <size_t N, typename T, typename S>
inline vec<N, T> operator*(vec<N, T> a, const S &s)
{
    pr_geom::fill_each<N>(a, [&](size_t i) { T e = a[i]; e *= s; return e; });
    return a;
}
template <size_t N, typename T, typename S>
inline vec<N, T> operator/(vec<N, T> a, const S &s)
{
    pr_geom::fill_each<N>(a, [&](size_t i) { T e = a[i]; e /= s; return e; });
    return a;
}

// embed<LEN>(v, fill): v extended with `fill`;  proj<LEN>(v): the first LEN components
template <size_t LEN, size_t N, typename T>
inline vec<LEN, T> embed(const vec<N, T> &v, T fill = 1)
{
    vec<LEN, T> wide;
    pr_geom::fill_each<LEN>(wide, [&](size_t i) { return i < N ? v[i] : fill; });
    return wide;
}
template <size_t LEN, size_t N, typename T>
inline vec<LEN, T> proj(const vec<N, T> &v)
{
    vec<LEN, T> narrow;
    pr_geom::fill_each<LEN>(narrow, [&](size_t i) { return v[i]; });
    return narrow;
}
template <typename T>
inline vec<3, T> cross(vec<3, T> a, vec<3, T> b)
{
    const T cx = a.y * b.z - a.z * b.y;
    const T cy = a.z * b.x - a.x * b.z;
    const T cz = a.x * b.y - a.y * b.x;
    return vec<3, T>(cx, cy, cz);
}
template <size_t N, typename T>
inline std::ostream &operator<<(std::ostream &os, const vec<N, T> &v)
{
    for (size_t i = 0; i != N; ++i) os << v[i] << " ";
    return os;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// matrices: R rows, each a vec<C,T>
// ---------------------------------------------------------------------------------------------------------------------------------
template <size_t R, size_t C, typename T>
class mat {
public:
    typedef vec<C, T> row_type;

    mat() {}
    mat(const T *row_major)
    {
        for (size_t k = 0; k != R * C; ++k) row_[k / C][k % C] = row_major[k];
    }
    static mat identity()
    {
        mat m;
        for (size_t k = 0; k != R * C; ++k) m.row_[k / C][k % C] = T(k / C == k % C);
        return m;
    }

    row_type &operator[](size_t r) { assert(r < R); return row_[r]; }
    const row_type &operator[](size_t r) const { assert(r < R); return row_[r]; }

    vec<R, T> col(size_t c) const
    {
        assert(c < C);
        vec<R, T> out;
        pr_geom::fill_each<R>(out, [&](size_t r) { return row_[r][c]; });
        return out;
    }
    void set_col(size_t c, vec<R, T> v)
    {
        assert(c < C);
        for (size_t r = 0; r != R; ++r) row_[r][c] = v[r];
    }
    mat<C, R, T> transpose()
    {
        mat<C, R, T> t;
        for (size_t c = 0; c != C; ++c) t[c] = col(c);
        return t;
    }

    // the matrix without row `row` and column `col`
    mat<R - 1, C - 1, T> get_minor(size_t row, size_t col) const
    {
        mat<R - 1, C - 1, T> sub;
        for (size_t r = 0; r + 1 != R; ++r)
            for (size_t c = 0; c + 1 != C; ++c) sub[r][c] = row_[pr_geom::skip(r, row)][pr_geom::skip(c, col)];
        return sub;
    }
    T det() const { return pr_geom::determinant(*this); }
    T cofactor(size_t row, size_t col) const
    {
        const T sign = ((row + col) & 1) ? -1 : 1;
        return get_minor(row, col).det() * sign;
    }
    mat adjugate() const
    {
        mat adj;
        for (size_t r = 0; r != R; ++r)
            for (size_t c = 0; c != C; ++c) adj[r][c] = cofactor(r, c);
        return adj;
    }
    // adjugate / (adjugate row 0 . row 0): the transposed inverse (what the reference uses for normal matrices)
    mat invert_transpose()
    {
        mat adj = adjugate();
        const T d = adj[0] * row_[0];
        for (size_t r = 0; r != R; ++r) adj[r] = adj[r] / d;
        return adj;
    }
    mat invert() { return invert_transpose().transpose(); }

    // not in the reference: the adapters hand matrices to the C ABI as plain floats
    const T *data() const { return &row_[0][0]; }
    T *data() { return &row_[0][0]; }

private:
    row_type row_[R];
};

namespace pr_geom {
template <typename T> inline T determinant(const mat<1, 1, T> &m) { return m[0][0]; }
template <size_t N, typename T> inline T determinant(const mat<N, N, T> &m)
{
    return fold_high_to_low<N, T>([&](size_t c) { return m[0][c] * m.cofactor(0, c); });
}
}  // namespace pr_geom

template <size_t R, size_t C, typename T>
inline vec<R, T> operator*(const mat<R, C, T> &m, const vec<C, T> &v)
{
    vec<R, T> out;
    pr_geom::fill_each<R>(out, [&](size_t r) { return m[r] * v; });
    return out;
}
// `result.transformation_ = extrinsic * result.transformation_` (icp.cu:212) is this product: entry (r,c) = row r . column c,
// summed k = K-1 .. 0
template <size_t R, size_t K, size_t C, typename T>
inline mat<R, C, T> operator*(const mat<R, K, T> &a, const mat<K, C, T> &b)
{
    mat<R, C, T> out;
    for (size_t c = 0; c != C; ++c) {
        const vec<K, T> column = b.col(c);
        for (size_t r = 0; r != R; ++r) out[r][c] = a[r] * column;
    }
    return out;
}
template <size_t R, size_t C, typename T>
inline mat<R, C, T> operator/(mat<R, C, T> m, const T &s)
{
    for (size_t r = 0; r != R; ++r) m[r] = m[r] / s;
    return m;
}
template <size_t R, size_t C, typename T>
inline std::ostream &operator<<(std::ostream &os, const mat<R, C, T> &m)
{
    for (size_t r = 0; r != R; ++r) os << m[r] << std::endl;
    return os;
}

typedef vec<2, float> Vec2f;
typedef vec<2, int> Vec2i;
typedef vec<3, float> Vec3f;
typedef vec<3, int> Vec3i;
typedef vec<4, float> Vec4f;
typedef vec<4, float> Vec4i;               // float, as in the reference (geometry.h:319)
typedef mat<4, 4, float> Mat4x4f;
typedef mat<3, 3, float> Mat3x3f;
typedef vec<3, float> Vec6f;               // 3 floats, as in the reference (geometry.h:323)
typedef mat<6, 6, float> Mat6x6f;

// float -> int conversions round half up through int(v + .5f) (geometry.h:326-333)
namespace pr_geom { inline int round_half_up(float v) { return int(v + .5f); } }
template <> template <> inline vec<2, int>::vec(const vec<2, float> &f) : x(pr_geom::round_half_up(f.x)), y(pr_geom::round_half_up(f.y)) {}
template <> template <> inline vec<2, float>::vec(const vec<2, int> &n) : x(float(n.x)), y(float(n.y)) {}
template <> template <> inline vec<3, int>::vec(const vec<3, float> &f)
    : x(pr_geom::round_half_up(f.x)), y(pr_geom::round_half_up(f.y)), z(pr_geom::round_half_up(f.z)) {}
template <> template <> inline vec<3, float>::vec(const vec<3, int> &n) : x(float(n.x)), y(float(n.y)), z(float(n.z)) {}

static_assert(sizeof(Vec3f) == 12 && sizeof(Mat3x3f) == 36 && sizeof(Mat4x4f) == 64, "layout must match the C ABI");
