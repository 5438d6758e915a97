// geometry.h -- the fixed-size vector / matrix types of cuda_icp/geometry.h:22-324 with the reference's names, members and
// memory layout (Vec3f = 12 B {x,y,z}, Mat3x3f = 36 B, Mat4x4f = 64 B, row-major plain floats), so user code written against
// the reference compiles unchanged on the host.  Own implementation.  Where the order of float operations is observable it
// follows the reference: dot products (and therefore mat * vec and mat * mat) sum from the HIGHEST index down
// (geometry.h:106-111,285-298).  tests/golden/geometry_h.json holds bit patterns computed by the reference header itself
// (oracle/Makefile `ref`); tests/test_geometry_golden.py compares this header and the oracle with them bit for bit.
#pragma once
#include <cassert>
#include <cmath>
#include <cstddef>
#include <iostream>

template <size_t R, size_t C, typename T> class mat;

template <size_t N, typename T> struct vec {
    T v_[N];
    vec() { for (size_t i = 0; i < N; ++i) v_[i] = T(); }
    T &operator[](size_t i) { assert(i < N); return v_[i]; }
    const T &operator[](size_t i) const { assert(i < N); return v_[i]; }
    static vec Zero() { return vec(); }
    vec &operator+=(const vec &o) { for (size_t i = 0; i < N; ++i) v_[i] += o.v_[i]; return *this; }
    vec operator+(const vec &o) { vec r(*this); r += o; return r; }
};
template <typename T> struct vec<2, T> {
    T x, y;
    vec() : x(T()), y(T()) {}
    vec(T X, T Y) : x(X), y(Y) {}
    template <class U> vec(const vec<2, U> &v);
    T &operator[](size_t i) { assert(i < 2); return i == 0 ? x : y; }
    const T &operator[](size_t i) const { assert(i < 2); return i == 0 ? x : y; }
};
template <typename T> struct vec<3, T> {
    T x, y, z;
    vec() : x(T()), y(T()), z(T()) {}
    vec(T X, T Y, T Z) : x(X), y(Y), z(Z) {}
    template <class U> vec(const vec<3, U> &v);
    T &operator[](size_t i) { assert(i < 3); return i == 0 ? x : (i == 1 ? y : z); }
    const T &operator[](size_t i) const { assert(i < 3); return i == 0 ? x : (i == 1 ? y : z); }
    float norm() { return std::sqrt(x * x + y * y + z * z); }
    vec &normalize(T l = 1);
};

// dot product, summed from the highest index down (geometry.h:106-111)
template <size_t N, typename T> T operator*(const vec<N, T> &a, const vec<N, T> &b) { T s = T(); for (size_t i = N; i--;) s += a[i] * b[i]; return s; }
template <size_t N, typename T> vec<N, T> operator+(vec<N, T> a, const vec<N, T> &b) { for (size_t i = N; i--;) a[i] += b[i]; return a; }
template <size_t N, typename T> vec<N, T> operator-(vec<N, T> a, const vec<N, T> &b) { for (size_t i = N; i--;) a[i] -= b[i]; return a; }
template <size_t N, typename T, typename U> vec<N, T> operator*(vec<N, T> a, const U &s) { for (size_t i = N; i--;) a[i] *= s; return a; }
template <size_t N, typename T, typename U> vec<N, T> operator/(vec<N, T> a, const U &s) { for (size_t i = N; i--;) a[i] /= s; return a; }
template <typename T> vec<3, T> &vec<3, T>::normalize(T l) { *this = (*this) * (l / norm()); return *this; }
template <size_t LEN, size_t N, typename T> vec<LEN, T> embed(const vec<N, T> &v, T fill = 1) { vec<LEN, T> r; for (size_t i = LEN; i--;) r[i] = (i < N ? v[i] : fill); return r; }
template <size_t LEN, size_t N, typename T> vec<LEN, T> proj(const vec<N, T> &v) { vec<LEN, T> r; for (size_t i = LEN; i--;) r[i] = v[i]; return r; }
template <typename T> vec<3, T> cross(vec<3, T> a, vec<3, T> b) { return vec<3, T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <size_t N, typename T> std::ostream &operator<<(std::ostream &o, const vec<N, T> &v) { for (size_t i = 0; i < N; ++i) o << v[i] << " "; return o; }

// determinant by cofactor expansion along row 0, highest column first (geometry.h:164-179)
template <size_t N, typename T> struct dt { static T det(const mat<N, N, T> &m) { T r = 0; for (size_t i = N; i--;) r += m[0][i] * m.cofactor(0, i); return r; } };
template <typename T> struct dt<1, T> { static T det(const mat<1, 1, T> &m) { return m[0][0]; } };

template <size_t R, size_t C, typename T> class mat {
    vec<C, T> rows_[R];
public:
    mat() {}
    mat(const T *d) { for (size_t i = 0; i < R; ++i) for (size_t j = 0; j < C; ++j) rows_[i][j] = d[i * C + j]; }
    vec<C, T> &operator[](size_t i) { assert(i < R); return rows_[i]; }
    const vec<C, T> &operator[](size_t i) const { assert(i < R); return rows_[i]; }
    vec<R, T> col(size_t j) const { assert(j < C); vec<R, T> c; for (size_t i = R; i--;) c[i] = rows_[i][j]; return c; }
    void set_col(size_t j, vec<R, T> v) { assert(j < C); for (size_t i = R; i--;) rows_[i][j] = v[i]; }
    static mat identity() { mat m; for (size_t i = 0; i < R; ++i) for (size_t j = 0; j < C; ++j) m[i][j] = T(i == j); return m; }
    T det() const { return dt<C, T>::det(*this); }
    mat<R - 1, C - 1, T> get_minor(size_t row, size_t col) const
    { mat<R - 1, C - 1, T> r; for (size_t i = R - 1; i--;) for (size_t j = C - 1; j--;) r[i][j] = rows_[i < row ? i : i + 1][j < col ? j : j + 1]; return r; }
    T cofactor(size_t row, size_t col) const { return get_minor(row, col).det() * ((row + col) % 2 ? -1 : 1); }
    mat adjugate() const { mat r; for (size_t i = R; i--;) for (size_t j = C; j--;) r[i][j] = cofactor(i, j); return r; }
    mat invert_transpose() { mat r = adjugate(); T d = r[0] * rows_[0]; return r / d; }
    mat invert() { return invert_transpose().transpose(); }
    mat<C, R, T> transpose() { mat<C, R, T> r; for (size_t i = C; i--;) r[i] = this->col(i); return r; }
    const T *data() const { return &rows_[0][0]; }       // (not in the reference: the adapters hand matrices to the C ABI as plain floats)
    T *data() { return &rows_[0][0]; }
};
template <size_t R, size_t C, typename T> vec<R, T> operator*(const mat<R, C, T> &m, const vec<C, T> &v) { vec<R, T> r; for (size_t i = R; i--;) r[i] = m[i] * v; return r; }
template <size_t R, size_t K, size_t C, typename T> mat<R, C, T> operator*(const mat<R, K, T> &a, const mat<K, C, T> &b)
{   // result.transformation_ = extrinsic * result.transformation_ (icp.cu:212) relies on this order (geometry.h:292-298)
    mat<R, C, T> r;
    for (size_t i = R; i--;) for (size_t j = C; j--;) r[i][j] = a[i] * b.col(j);
    return r;
}
template <size_t R, size_t C, typename T> mat<C, R, T> operator/(mat<R, C, T> m, const T &s) { for (size_t i = R; i--;) m[i] = m[i] / s; return m; }
template <size_t R, size_t C, typename T> std::ostream &operator<<(std::ostream &o, const mat<R, C, T> &m) { for (size_t i = 0; i < R; ++i) o << m[i] << std::endl; return o; }

typedef vec<2, float> Vec2f;
typedef vec<2, int> Vec2i;
typedef vec<3, float> Vec3f;
typedef vec<3, int> Vec3i;
typedef vec<4, float> Vec4f;
typedef vec<4, float> Vec4i;               // sic (geometry.h:319)
typedef mat<4, 4, float> Mat4x4f;
typedef mat<3, 3, float> Mat3x3f;
typedef vec<3, float> Vec6f;               // sic (geometry.h:323)
typedef mat<6, 6, float> Mat6x6f;

// geometry.h:326-333: float -> int rounds half up by int(v + .5f)
template <> template <> inline vec<3, int>::vec(const vec<3, float> &v) : x(int(v.x + .5f)), y(int(v.y + .5f)), z(int(v.z + .5f)) {}
template <> template <> inline vec<3, float>::vec(const vec<3, int> &v) : x(float(v.x)), y(float(v.y)), z(float(v.z)) {}
template <> template <> inline vec<2, int>::vec(const vec<2, float> &v) : x(int(v.x + .5f)), y(int(v.y + .5f)) {}
template <> template <> inline vec<2, float>::vec(const vec<2, int> &v) : x(float(v.x)), y(float(v.y)) {}

static_assert(sizeof(Vec3f) == 12 && sizeof(Mat3x3f) == 36 && sizeof(Mat4x4f) == 64, "layout must match the C ABI");
