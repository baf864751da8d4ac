// smallmat.h — small dense fp64 linear algebra: per-lane (serial, on registers / local
// arrays) and per-wave (64 lanes cooperating on a matrix in LDS or HBM).
#pragma once
#include "wave.h"

namespace dekf {

// ---------------------------------------------------------------- per lane, serial
// y = R(3x3 row-major) * v
DEKF_FN void mv3(const double* R, const double* v, double* y) {
    for (int i = 0; i < 3; ++i) y[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
DEKF_FN void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
// Eigen::Quaterniond(w,x,y,z).normalized().toRotationMatrix(), row-major
DEKF_FN void quat_to_rot(const double* q, double* R) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double n = sqrt(w * w + x * x + y * y + z * z);
    if (n > 0) { w /= n; x /= n; y /= n; z /= n; }
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// in-place inverse of a small n x n row-major matrix by Gauss-Jordan with partial
// pivoting (the role of Eigen's PartialPivLU-based inverse()).  Returns false on a zero pivot.
template <int NMAX>
DEKF_FN bool inv_small(double* A, int n) {
    double I[NMAX * NMAX];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) I[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int p = 0; p < n; ++p) {
        int piv = p;
        double best = fabs(A[p * n + p]);
        for (int i = p + 1; i < n; ++i)
            if (fabs(A[i * n + p]) > best) { best = fabs(A[i * n + p]); piv = i; }
        if (best == 0.0) return false;
        if (piv != p)
            for (int j = 0; j < n; ++j) {
                double t = A[p * n + j]; A[p * n + j] = A[piv * n + j]; A[piv * n + j] = t;
                t = I[p * n + j]; I[p * n + j] = I[piv * n + j]; I[piv * n + j] = t;
            }
        double d = 1.0 / A[p * n + p];
        for (int j = 0; j < n; ++j) { A[p * n + j] *= d; I[p * n + j] *= d; }
        for (int i = 0; i < n; ++i) {
            if (i == p) continue;
            double f = A[i * n + p];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) { A[i * n + j] -= f * A[p * n + j]; I[i * n + j] -= f * I[p * n + j]; }
        }
    }
    for (int i = 0; i < n * n; ++i) A[i] = I[i];
    return true;
}

// in-place inverse of an SPD N x N row-major matrix, Gauss-Jordan without pivoting, fully
// unrolled so that every index is static and the matrix stays in registers
template <int N>
DEKF_FN void inv_spd_unrolled(double* A) {
#pragma unroll
    for (int p = 0; p < N; ++p) {
        double d = 1.0 / A[p * N + p];
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j != p) A[p * N + j] *= d;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i == p) continue;
            double f = A[i * N + p];
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (j != p) A[i * N + j] -= f * A[p * N + j];
            A[i * N + p] = -f * d;
        }
        A[p * N + p] = d;
    }
}

// inverse of a symmetric 3x3 given packed (00 01 02 11 12 22) -> packed, by cofactors
DEKF_FN bool inv3_sym(const double* s, double* o) {
    double a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5];
    double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    double det = a * c00 + b * c01 + c * c02;
    if (det == 0.0) return false;
    double id = 1.0 / det;
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id;
    o[5] = (a * d - b * b) * id;
    return true;
}

// ---------------------------------------------------------------- per wave, on memory
// C (m x n) = op(A) (m x k) * op(B) (k x n), row-major with leading dimensions
template <bool TA, bool TB>
DEKF_FN void wmatmul(double* C, int ldc, const double* A, int lda, const double* B, int ldb, int m, int k, int n,
                     double alpha = 1.0, double beta = 0.0) {
    wfor(m * n, [&](int e) {
        int i = e / n, j = e - i * n;
        double s = 0.0;
        for (int t = 0; t < k; ++t) {
            double a = TA ? A[t * lda + i] : A[i * lda + t];
            double b = TB ? B[j * ldb + t] : B[t * ldb + j];
            s += a * b;
        }
        C[i * ldc + j] = alpha * s + (beta != 0.0 ? beta * C[i * ldc + j] : 0.0);
    });
}

// In-place inverse of A (n x n, leading dimension n) by Gauss-Jordan with row pivoting.
// scratch: n*n (the growing inverse) + n (saved pivot column).  pivoting = false for SPD.
DEKF_FN bool winverse(double* A, int n, double* scratch, bool pivoting) {
    double* Inv = scratch;
    double* col = scratch + n * n;
    wfor(n * n, [&](int e) { Inv[e] = (e / n == e % n) ? 1.0 : 0.0; });
    bool ok = true;
    for (int p = 0; p < n; ++p) {
        int piv = p;
        if (pivoting) {
            double best;
            piv = p + wred_argmax(n - p, [&](int i) { return fabs(A[(p + i) * n + p]); }, &best);
            if (best == 0.0) ok = false;
        } else if (A[p * n + p] == 0.0) {
            ok = false;
        }
        if (!ok) break;  // wave-uniform
        if (piv != p) {
            wfor(2 * n, [&](int e) {
                double* Mx = e < n ? A : Inv;
                int j = e < n ? e : e - n;
                double t = Mx[p * n + j]; Mx[p * n + j] = Mx[piv * n + j]; Mx[piv * n + j] = t;
            });
        }
        double d = 1.0 / A[p * n + p];
        DEKF_SYNC();  // everyone has read the pivot before row p is rescaled
        wfor(2 * n, [&](int e) {
            if (e < n) { A[p * n + e] *= d; col[e] = (e == p) ? 0.0 : A[e * n + p]; }
            else Inv[p * n + (e - n)] *= d;
        });
        wfor(2 * n * n, [&](int e) {
            double* Mx = e < n * n ? A : Inv;
            int q = e < n * n ? e : e - n * n;
            int i = q / n, j = q - i * n;
            if (i != p) Mx[q] -= col[i] * Mx[p * n + j];
        });
    }
    if (ok) wfor(n * n, [&](int e) { A[e] = Inv[e]; });
    return ok;
}

// In-place inverse of a DEFINITE symmetric matrix (positive or negative: no pivoting needed) by the
// Gauss-Jordan sweep: n pivots, each one copy of the pivot row / column and one pass over the n*n
// elements — half the elements of the augmented [A | I] form above, and no pivot search.
// scratch: 2n (pivot row, pivot column).
DEKF_FN bool winverse_definite(double* A, int n, double* scratch) {
    double* rowp = scratch;
    double* colp = scratch + n;
    bool ok = true;
    for (int p = 0; p < n; ++p) {
        const double piv = A[p * n + p];
        if (!(fabs(piv) > 0.0) || !(fabs(piv) < 1e300)) { ok = false; break; }  // group-uniform
        wfor(2 * n, [&](int e) {
            if (e < n) rowp[e] = A[p * n + e];
            else colp[e - n] = A[(e - n) * n + p];
        });
        const double d = 1.0 / piv;
        wfor(n * n, [&](int e) {
            int i = e / n, j = e - i * n;
            double v;
            if (i == p) v = (j == p) ? d : rowp[j] * d;
            else if (j == p) v = -colp[i] * d;
            else v = A[e] - colp[i] * rowp[j] * d;
            A[e] = v;
        });
    }
    return ok;
}

}  // namespace dekf
