// pr_solver.inl -- the per-iteration rigid update: (A + 0.01 I) x = b in double, x -> 4x4.
//
// Replaces cuda_icp::eigen_slover_666 / TransformVector6dToMatrix4d / eigen_to_custom
// (cuda_icp/icp.cpp:7-45).  The reference calls Eigen (absent from its tree, version unpinned);
// this file restates Eigen's published algorithms:
//   * LDLT<Matrix6d>::compute/solve -- unblocked lower LDL^T with symmetric pivoting on the largest
//     |diagonal| entry, pseudo-inverse of D with tolerance 1/DBL_MAX;
//   * AngleAxisd(z)*AngleAxisd(y)*AngleAxisd(x) -- composed as unit quaternions, then
//     Quaterniond::toRotationMatrix.
// The same source is compiled for the host (PR_SOLVE_HOST, pr_solve_666) and for the device
// (PR_SOLVE_DEVICE finalize kernel).  sin/cos come from the fixed polynomial below rather than
// libm/ocml so that both builds produce bit-identical updates (all operations are IEEE double
// add/mul/div with contraction disabled); it is accurate to < 1 ulp on |x| <= pi/4 and uses a
// two-term Cody-Waite reduction beyond.
#pragma once

#ifndef PR_HD
#define PR_HD
#endif

namespace prs {

PR_HD inline double dabs(double v) { return v < 0 ? -v : v; }

// minimax kernels on [-pi/4, pi/4] (the classic fdlibm coefficient set)
PR_HD inline double ksin(double x)
{
    const double z = x * x;
    double p = 1.58969099521155010221e-10;
    p = p * z + -2.50507602534068634195e-08;
    p = p * z + 2.75573137070700676789e-06;
    p = p * z + -1.98412698298579493134e-04;
    p = p * z + 8.33333333332248946124e-03;
    p = p * z + -1.66666666666666324348e-01;
    return x + (x * z) * p;
}
PR_HD inline double kcos(double x)
{
    const double z = x * x;
    double p = -1.13596475577881948265e-11;
    p = p * z + 2.08757232129817482790e-09;
    p = p * z + -2.75573143513906633035e-07;
    p = p * z + 2.48015872894767294178e-05;
    p = p * z + -1.38888888888741095749e-03;
    p = p * z + 4.16666666666666019037e-02;
    return (1.0 - 0.5 * z) + (z * z) * p;
}
PR_HD inline void sincos_d(double x, double *s, double *c)
{
    const double quarter_pi = 0.78539816339744830962;
    if (dabs(x) <= quarter_pi) { *s = ksin(x); *c = kcos(x); return; }
    if (!(dabs(x) < 1.0e9)) { *s = 0.0; *c = 1.0; return; }            // non-finite / absurd update
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00;                 // first 33 bits of pi/2
    const double pio2_lo = 6.07710050650619224932e-11;                 // pi/2 - pio2_hi
    double t = x * two_over_pi;
    long long n = (long long)(t < 0 ? t - 0.5 : t + 0.5);
    double fn = (double)n;
    double r = (x - fn * pio2_hi) - fn * pio2_lo;
    double sr = ksin(r), cr = kcos(r);
    switch ((int)(n & 3)) {
        case 0:  *s = sr;  *c = cr;  break;
        case 1:  *s = cr;  *c = -sr; break;
        case 2:  *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr;  break;
    }
}

// lower-triangular LDL^T with diagonal pivoting, in place on the 6x6 (row-major, lower part used).
// Every array index below is a compile-time constant after unrolling (the data-dependent pivot is
// applied through a chain of constant-index conditional swaps), so on the device the whole 6x6
// lives in registers instead of scratch memory.
template <int K, int P> PR_HD inline void sym_swap(double *m)
{
#define MM(r, c) m[(r) * 6 + (c)]
#pragma unroll
    for (int j = 0; j < K; ++j)       { double t = MM(K, j); MM(K, j) = MM(P, j); MM(P, j) = t; }
#pragma unroll
    for (int i = P + 1; i < 6; ++i)   { double t = MM(i, K); MM(i, K) = MM(i, P); MM(i, P) = t; }
#pragma unroll
    for (int i = K + 1; i < P; ++i)   { double t = MM(i, K); MM(i, K) = MM(P, i); MM(P, i) = t; }
    double t = MM(K, K); MM(K, K) = MM(P, P); MM(P, P) = t;
#undef MM
}
template <int K> PR_HD inline int ldlt6_step(double *m)
{
#define MM(r, c) m[(r) * 6 + (c)]
    int p = K;
    double top = dabs(MM(K, K));
#pragma unroll
    for (int i = K + 1; i < 6; ++i) {
        double a = dabs(MM(i, i));
        if (a > top) { top = a; p = i; }
    }
    if (K + 1 < 6 && p == K + 1) sym_swap<K, (K + 1 < 6 ? K + 1 : 5)>(m);
    if (K + 2 < 6 && p == K + 2) sym_swap<K, (K + 2 < 6 ? K + 2 : 5)>(m);
    if (K + 3 < 6 && p == K + 3) sym_swap<K, (K + 3 < 6 ? K + 3 : 5)>(m);
    if (K + 4 < 6 && p == K + 4) sym_swap<K, (K + 4 < 6 ? K + 4 : 5)>(m);
    if (K + 5 < 6 && p == K + 5) sym_swap<K, (K + 5 < 6 ? K + 5 : 5)>(m);
    double w[6];
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < K; ++j) { w[j] = MM(j, j) * MM(K, j); dot += MM(K, j) * w[j]; }
    MM(K, K) -= dot;
    const double dk = MM(K, K);
#pragma unroll
    for (int i = K + 1; i < 6; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) acc += MM(i, j) * w[j];
        double v = MM(i, K) - acc;
        MM(i, K) = (dabs(dk) > 0.0) ? v / dk : v;
    }
    return p;
#undef MM
}
template <int K> PR_HD inline void perm_apply(double *y, int p)
{
    if (K + 1 < 6 && p == K + 1) { double t = y[K]; y[K] = y[K + 1 < 6 ? K + 1 : 5]; y[K + 1 < 6 ? K + 1 : 5] = t; }
    if (K + 2 < 6 && p == K + 2) { double t = y[K]; y[K] = y[K + 2 < 6 ? K + 2 : 5]; y[K + 2 < 6 ? K + 2 : 5] = t; }
    if (K + 3 < 6 && p == K + 3) { double t = y[K]; y[K] = y[K + 3 < 6 ? K + 3 : 5]; y[K + 3 < 6 ? K + 3 : 5] = t; }
    if (K + 4 < 6 && p == K + 4) { double t = y[K]; y[K] = y[K + 4 < 6 ? K + 4 : 5]; y[K + 4 < 6 ? K + 4 : 5] = t; }
    if (K + 5 < 6 && p == K + 5) { double t = y[K]; y[K] = y[K + 5 < 6 ? K + 5 : 5]; y[K + 5 < 6 ? K + 5 : 5] = t; }
}
PR_HD inline void ldlt6(double *m /*36*/, const double *rhs, double *x)
{
#define MM(r, c) m[(r) * 6 + (c)]
    const int s0 = ldlt6_step<0>(m), s1 = ldlt6_step<1>(m), s2 = ldlt6_step<2>(m);
    const int s3 = ldlt6_step<3>(m), s4 = ldlt6_step<4>(m);
    (void)ldlt6_step<5>(m);
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = rhs[i];
    perm_apply<0>(y, s0); perm_apply<1>(y, s1); perm_apply<2>(y, s2); perm_apply<3>(y, s3); perm_apply<4>(y, s4);
#pragma unroll
    for (int i = 1; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < i; ++j) y[i] -= MM(i, j) * y[j];
    }
    const double tiny = 1.0 / 1.7976931348623157e308;
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = (dabs(MM(i, i)) > tiny) ? y[i] / MM(i, i) : 0.0;
#pragma unroll
    for (int i = 4; i >= 0; --i) {
#pragma unroll
        for (int j = i + 1; j < 6; ++j) y[i] -= MM(j, i) * y[j];
    }
    perm_apply<4>(y, s4); perm_apply<3>(y, s3); perm_apply<2>(y, s2); perm_apply<1>(y, s1); perm_apply<0>(y, s0);
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = y[i];
#undef MM
}

// u = (rx, ry, rz, tx, ty, tz) and the sines / cosines of the three half angles -> T: row-major 4x4 float
// (TransformVector6dToMatrix4d, icp.cpp:7-27)
PR_HD inline void compose_update(const double *u, double sx, double cx, double sy, double cy, double sz, double cz, float *T)
{
    // q = qz * qy * qx with qz=(cz;0,0,sz), qy=(cy;0,sy,0), qx=(cx;sx,0,0), general Hamilton products
    // first qzy = qz*qy
    double aw = cz * cy - 0.0 * 0.0 - 0.0 * sy - sz * 0.0;
    double ax = cz * 0.0 + 0.0 * cy + 0.0 * 0.0 - sz * sy;
    double ay = cz * sy + 0.0 * cy + sz * 0.0 - 0.0 * 0.0;
    double az = cz * 0.0 + sz * cy + 0.0 * sy - 0.0 * 0.0;
    // then q = qzy*qx
    double qw = aw * cx - ax * sx - ay * 0.0 - az * 0.0;
    double qx = aw * sx + ax * cx + ay * 0.0 - az * 0.0;
    double qy = aw * 0.0 + ay * cx + az * sx - ax * 0.0;
    double qz = aw * 0.0 + az * cx + ax * 0.0 - ay * sx;

    const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    T[0] = (float)(1.0 - (tyy + tzz)); T[1] = (float)(txy - twz);         T[2]  = (float)(txz + twy);         T[3]  = (float)u[3];
    T[4] = (float)(txy + twz);         T[5] = (float)(1.0 - (txx + tzz)); T[6]  = (float)(tyz - twx);         T[7]  = (float)u[4];
    T[8] = (float)(txz - twy);         T[9] = (float)(tyz + twx);         T[10] = (float)(1.0 - (txx + tyy)); T[11] = (float)u[5];
    T[12] = 0.0f; T[13] = 0.0f; T[14] = 0.0f; T[15] = 1.0f;
}

// A: 36 floats (symmetric, any major), b: 6 floats -> T: row-major 4x4 float
PR_HD inline void solve_666_impl(const float *A, const float *b, float *T)
{
    double m[36], rhs[6], u[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) m[r * 6 + c] = (double)A[c * 6 + r] + (r == c ? 0.01 : 0.0);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) rhs[i] = (double)b[i];
    ldlt6(m, rhs, u);

    double sx, cx, sy, cy, sz, cz;
    sincos_d(0.5 * u[0], &sx, &cx);
    sincos_d(0.5 * u[1], &sy, &cy);
    sincos_d(0.5 * u[2], &sz, &cz);
    compose_update(u, sx, cx, sy, cy, sz, cz, T);
}

// result.T = E * result.T with the reference's summation order: index 3,2,1,0 from 0
// (cuda_icp/geometry.h:106-111 dot product, :292-298 mat*mat)
PR_HD inline void mat4_mul_impl(const float *A, const float *B, float *C)
{
    float tmp[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
            acc += A[i * 4 + 3] * B[12 + j];
            acc += A[i * 4 + 2] * B[8 + j];
            acc += A[i * 4 + 1] * B[4 + j];
            acc += A[i * 4 + 0] * B[j];
            tmp[i * 4 + j] = acc;
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) C[i] = tmp[i];
}

}  // namespace prs
