// geometry.h -- small fixed-size vector / matrix types with the reference's names and memory
// layout (cuda_icp/geometry.h:314-324: Vec3f = 12 B {x,y,z}, Mat3x3f = 36 B, Mat4x4f = 64 B, all
// row-major plain floats), so user code written against the reference compiles unchanged.
// Own implementation; only the members the path and test.cpp use are provided.
#pragma once
#include <cstddef>
#include <iostream>

template <size_t N, typename T> struct vec {
    T v_[N];
    vec() { for (size_t i = 0; i < N; ++i) v_[i] = T(); }
    T &operator[](size_t i) { return v_[i]; }
    const T &operator[](size_t i) const { return v_[i]; }
    static vec Zero() { return vec(); }
    vec &operator+=(const vec &o) { for (size_t i = 0; i < N; ++i) v_[i] += o.v_[i]; return *this; }
    vec operator+(const vec &o) const { vec r(*this); r += o; return r; }
};
template <typename T> struct vec<3, T> {
    T x, y, z;
    vec() : x(T()), y(T()), z(T()) {}
    vec(T X, T Y, T Z) : x(X), y(Y), z(Z) {}
    T &operator[](size_t i) { return i == 0 ? x : (i == 1 ? y : z); }
    const T &operator[](size_t i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <size_t N, typename T> vec<N, T> operator-(vec<N, T> a, const vec<N, T> &b) { for (size_t i = N; i--;) a[i] -= b[i]; return a; }
template <size_t N, typename T> vec<N, T> operator+(vec<N, T> a, const vec<N, T> &b) { for (size_t i = N; i--;) a[i] += b[i]; return a; }
// dot product, summed from the highest index down like the reference (geometry.h:106-111)
template <size_t N, typename T> T operator*(const vec<N, T> &a, const vec<N, T> &b) { T s = T(); for (size_t i = N; i--;) s += a[i] * b[i]; return s; }

template <size_t R, size_t C, typename T> class mat {
    vec<C, T> rows_[R];
public:
    mat() {}
    explicit mat(const T *d) { for (size_t i = 0; i < R; ++i) for (size_t j = 0; j < C; ++j) rows_[i][j] = d[i * C + j]; }
    vec<C, T> &operator[](size_t i) { return rows_[i]; }
    const vec<C, T> &operator[](size_t i) const { return rows_[i]; }
    vec<R, T> col(size_t j) const { vec<R, T> c; for (size_t i = 0; i < R; ++i) c[i] = rows_[i][j]; return c; }
    static mat identity() { mat m; for (size_t i = 0; i < R; ++i) for (size_t j = 0; j < C; ++j) m[i][j] = T(i == j); return m; }
    const T *data() const { return &rows_[0][0]; }
    T *data() { return &rows_[0][0]; }
};
template <size_t R, size_t K, size_t C, typename T> mat<R, C, T> operator*(const mat<R, K, T> &a, const mat<K, C, T> &b)
{   // result.T = extrinsic * result.T relies on this order (geometry.h:292-298)
    mat<R, C, T> r;
    for (size_t i = 0; i < R; ++i) for (size_t j = 0; j < C; ++j) r[i][j] = a[i] * b.col(j);
    return r;
}
template <size_t N, typename T> std::ostream &operator<<(std::ostream &o, const vec<N, T> &v) { for (size_t i = 0; i < N; ++i) o << v[i] << " "; return o; }
template <size_t R, size_t C, typename T> std::ostream &operator<<(std::ostream &o, const mat<R, C, T> &m) { for (size_t i = 0; i < R; ++i) o << m[i] << std::endl; return o; }

typedef vec<3, float> Vec3f;
typedef vec<3, int> Vec3i;
typedef mat<4, 4, float> Mat4x4f;
typedef mat<3, 3, float> Mat3x3f;
static_assert(sizeof(Vec3f) == 12 && sizeof(Mat3x3f) == 36 && sizeof(Mat4x4f) == 64, "layout must match the C ABI");
