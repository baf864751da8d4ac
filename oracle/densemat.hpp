// TEST INFRASTRUCTURE ONLY — part of the CPU oracle (see oracle/README.md).
// Minimal dynamic dense matrix in fp64 standing in for the Eigen::MatrixXd /
// SparseMatrix<double> objects the reference manipulates.  Only the operations the
// reference path uses are provided; semantics follow Eigen's (row, col) indexing.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdio>
#include <vector>

namespace orc {

using Vec = std::vector<double>;

struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
    static Mat identity(int n) {
        Mat m(n, n);
        for (int i = 0; i < n; ++i) m(i, i) = 1.0;
        return m;
    }
    static Mat diag(const Vec& d) {
        Mat m((int)d.size(), (int)d.size());
        for (size_t i = 0; i < d.size(); ++i) m((int)i, (int)i) = d[i];
        return m;
    }
    void set_block(int i0, int j0, const Mat& b) {
        assert(i0 + b.r <= r && j0 + b.c <= c);
        for (int i = 0; i < b.r; ++i)
            for (int j = 0; j < b.c; ++j) (*this)(i0 + i, j0 + j) = b(i, j);
    }
    void add_block(int i0, int j0, const Mat& b) {
        assert(i0 + b.r <= r && j0 + b.c <= c);
        for (int i = 0; i < b.r; ++i)
            for (int j = 0; j < b.c; ++j) (*this)(i0 + i, j0 + j) += b(i, j);
    }
    Mat block(int i0, int j0, int nr, int nc) const {
        assert(i0 + nr <= r && j0 + nc <= c);
        Mat b(nr, nc);
        for (int i = 0; i < nr; ++i)
            for (int j = 0; j < nc; ++j) b(i, j) = (*this)(i0 + i, j0 + j);
        return b;
    }
    // Eigen conservativeResize: keep the top-left overlap, new entries zero
    void conservative_resize(int nr, int nc) {
        Mat m(nr, nc);
        for (int i = 0; i < (nr < r ? nr : r); ++i)
            for (int j = 0; j < (nc < c ? nc : c); ++j) m(i, j) = (*this)(i, j);
        *this = m;
    }
    Mat T() const {
        Mat t(c, r);
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < c; ++j) t(j, i) = (*this)(i, j);
        return t;
    }
};

inline Mat operator*(const Mat& A, const Mat& B) {
    assert(A.c == B.r);
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; ++i)
        for (int k = 0; k < A.c; ++k) {
            double aik = A(i, k);
            if (aik == 0.0) continue;
            for (int j = 0; j < B.c; ++j) C(i, j) += aik * B(k, j);
        }
    return C;
}
inline Mat operator+(const Mat& A, const Mat& B) {
    assert(A.r == B.r && A.c == B.c);
    Mat C = A;
    for (size_t i = 0; i < C.a.size(); ++i) C.a[i] += B.a[i];
    return C;
}
inline Mat operator-(const Mat& A, const Mat& B) {
    assert(A.r == B.r && A.c == B.c);
    Mat C = A;
    for (size_t i = 0; i < C.a.size(); ++i) C.a[i] -= B.a[i];
    return C;
}
inline Mat operator*(double s, const Mat& A) {
    Mat C = A;
    for (auto& v : C.a) v *= s;
    return C;
}
inline Mat operator-(const Mat& A) { return -1.0 * A; }

inline Vec operator*(const Mat& A, const Vec& x) {
    assert(A.c == (int)x.size());
    Vec y(A.r, 0.0);
    for (int i = 0; i < A.r; ++i) {
        double s = 0;
        for (int j = 0; j < A.c; ++j) s += A(i, j) * x[j];
        y[i] = s;
    }
    return y;
}
inline Vec operator+(const Vec& a, const Vec& b) {
    assert(a.size() == b.size());
    Vec c = a;
    for (size_t i = 0; i < c.size(); ++i) c[i] += b[i];
    return c;
}
inline Vec operator-(const Vec& a, const Vec& b) {
    assert(a.size() == b.size());
    Vec c = a;
    for (size_t i = 0; i < c.size(); ++i) c[i] -= b[i];
    return c;
}
inline Vec operator*(double s, const Vec& a) {
    Vec c = a;
    for (auto& v : c) v *= s;
    return c;
}
inline Vec operator-(const Vec& a) { return -1.0 * a; }
inline Vec segment(const Vec& v, int i0, int n) { return Vec(v.begin() + i0, v.begin() + i0 + n); }
inline void set_segment(Vec& v, int i0, const Vec& s) {
    for (size_t i = 0; i < s.size(); ++i) v[i0 + i] = s[i];
}
inline double norm2(const Vec& v) {
    double s = 0;
    for (double x : v) s += x * x;
    return std::sqrt(s);
}
inline Vec cross3(const Vec& a, const Vec& b) {
    return Vec{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
// EigenUtils::vector3dSkew (EigenUtils.hpp:91-97)
inline Mat skew3(const Vec& v) {
    Mat s(3, 3);
    s(0, 1) = -v[2]; s(0, 2) = v[1];
    s(1, 0) = v[2];  s(1, 2) = -v[0];
    s(2, 0) = -v[1]; s(2, 1) = v[0];
    return s;
}

// Inverse through LU with partial pivoting: what Eigen's MatrixXd::inverse() does for
// dynamic sizes (PartialPivLU).  Returns false on an exactly zero pivot.
inline bool lu_inverse(const Mat& A, Mat& Ainv) {
    int n = A.r;
    assert(A.r == A.c);
    Mat LU = A;
    std::vector<int> piv(n);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(LU(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(LU(i, k)) > best) { best = std::fabs(LU(i, k)); p = i; }
        if (best == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(LU(k, j), LU(p, j));
            std::swap(piv[k], piv[p]);
        }
        for (int i = k + 1; i < n; ++i) {
            LU(i, k) /= LU(k, k);
            double l = LU(i, k);
            if (l != 0.0)
                for (int j = k + 1; j < n; ++j) LU(i, j) -= l * LU(k, j);
        }
    }
    Ainv = Mat(n, n);
    Vec y(n);
    for (int col = 0; col < n; ++col) {
        for (int i = 0; i < n; ++i) {
            double s = (piv[i] == col) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) s -= LU(i, j) * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < n; ++j) s -= LU(i, j) * Ainv(j, col);
            Ainv(i, col) = s / LU(i, i);
        }
    }
    return true;
}
inline Mat inverse(const Mat& A) {
    Mat Ai;
    bool ok = lu_inverse(A, Ai);
    if (!ok) std::fprintf(stderr, "[oracle] singular matrix in inverse()\n");
    return Ai;
}

// TEST KNOB for the one inverse whose conditioning limits parity: the 21/24/33/36-dim saddle matrix of marginalizeQP
// (MheSrb.cpp:588,640), diagonal entries from 1e-11 to 1e14 with leg_odom_type 1.  How far do two equally legitimate fp64
// evaluations of the reference's formula lie apart?  (tests/test_foot_states.py measures it and ties the device's allowance to it.)
//   0  as above: Eigen's PartialPivLU in double — what every parity test compares with
//   1  the same elimination in long double (x87 80-bit: 11 more mantissa bits), result rounded to double
//   2  the same double elimination on the symmetrically REVERSED matrix P A P (another, equally valid, pivot sequence)
//   3  as 0, but the SPD inverses M^-1, Q^-1, R^-1 that S is built from run their Cholesky solve in the reversed elimination order
//      (est_oracle.hpp: marginalize): S itself then differs in its last bits, as it does in any other implementation
//   4  3 and 2 together
//   5  as 0 on S with every entry moved by one unit in the last place (a fixed symmetric pattern)
inline int& marg_inverse_variant() { static int v = 0; return v; }
inline Mat lu_inverse_long(const Mat& A) {
    const int n = A.r;
    std::vector<long double> LU((size_t)n * n), X((size_t)n * n), y(n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) LU[(size_t)i * n + j] = A(i, j);
    std::vector<int> piv(n);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        long double best = fabsl(LU[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabsl(LU[(size_t)i * n + k]) > best) { best = fabsl(LU[(size_t)i * n + k]); p = i; }
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(LU[(size_t)k * n + j], LU[(size_t)p * n + j]);
            std::swap(piv[k], piv[p]);
        }
        for (int i = k + 1; i < n; ++i) {
            LU[(size_t)i * n + k] /= LU[(size_t)k * n + k];
            const long double l = LU[(size_t)i * n + k];
            for (int j = k + 1; j < n; ++j) LU[(size_t)i * n + j] -= l * LU[(size_t)k * n + j];
        }
    }
    Mat Ai(n, n);
    for (int col = 0; col < n; ++col) {
        for (int i = 0; i < n; ++i) {
            long double s = (piv[i] == col) ? 1.0L : 0.0L;
            for (int j = 0; j < i; ++j) s -= LU[(size_t)i * n + j] * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            long double s = y[i];
            for (int j = i + 1; j < n; ++j) s -= LU[(size_t)i * n + j] * X[(size_t)j * n + col];
            X[(size_t)i * n + col] = s / LU[(size_t)i * n + i];
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ai(i, j) = (double)X[(size_t)i * n + j];
    return Ai;
}
inline Mat inverse_marg(const Mat& A) {
    const int v = marg_inverse_variant(), n = A.r;
    if (v == 1) return lu_inverse_long(A);
    if (v == 5) {
        // every entry of S moved by one unit in the last place, up or down by a fixed symmetric pattern: what another order of the
        // sums that BUILD S (or a fused multiply-add in one of them) does to it
        Mat P = A;
        for (int i = 0; i < n; ++i)
            for (int j = i; j < n; ++j) {
                const unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(j * 40503u + 0x9e3779b9u);
                const double f = ((h >> 7) & 1u) ? 1.0 + 0x1p-52 : 1.0 - 0x1p-53;
                P(i, j) = A(i, j) * f;
                P(j, i) = A(j, i) * f;
            }
        return inverse(P);
    }
    if (v == 2 || v == 4) {
        Mat R(n, n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) R(i, j) = A(n - 1 - i, n - 1 - j);
        Mat Ri = inverse(R), Ai(n, n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ai(i, j) = Ri(n - 1 - i, n - 1 - j);
        return Ai;
    }
    return inverse(A);
}

// SPD inverse by Cholesky solve against I: what the reference's
// SimplicialLLT::solve(I) computes (MheSrb.cpp:524-525,555-559,610-614).
inline Mat spd_inverse(const Mat& A) {
    int n = A.r;
    Mat L(n, n);
    for (int j = 0; j < n; ++j) {
        double d = A(j, j);
        for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
        if (!(d > 0)) {
            std::fprintf(stderr, "[oracle] spd_inverse: matrix not positive definite (d=%g)\n", d);
            return inverse(A);
        }
        L(j, j) = std::sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double s = A(i, j);
            for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
            L(i, j) = s / L(j, j);
        }
    }
    Mat X(n, n);
    Vec y(n);
    for (int col = 0; col < n; ++col) {
        for (int i = 0; i < n; ++i) {
            double s = (i == col) ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k) s -= L(i, k) * y[k];
            y[i] = s / L(i, i);
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = y[i];
            for (int k = i + 1; k < n; ++k) s -= L(k, i) * X(k, col);
            X(i, col) = s / L(i, i);
        }
    }
    return X;
}

// Eigen::Quaterniond(w,x,y,z).normalized().toRotationMatrix()
inline Mat quat_to_rot_normalized(double w, double x, double y, double z) {
    double n = std::sqrt(w * w + x * x + y * y + z * z);
    if (n > 0) { w /= n; x /= n; y /= n; z /= n; }
    Mat R(3, 3);
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz;       R(0, 2) = txz + twy;
    R(1, 0) = txy + twz;       R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy;       R(2, 1) = tyz + twx;       R(2, 2) = 1 - (txx + tyy);
    return R;
}

}  // namespace orc
