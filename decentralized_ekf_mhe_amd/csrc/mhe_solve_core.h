// mhe_solve_core.h — per-instance (one wavefront) OSQP-style ADMM solve of the MHE window QP.
//
// Replaces  MHEproblem::initQP + solveQP + getsolution  (src/decentral_legged_est/src/MheSrb.cpp:
// 272-349, 715-723), i.e. osqp_setup + osqp_solve on
//     min 1/2 x'Hx + g'x   s.t.  l <= Ax <= u
// with H, g, A, l, u as MHEproblem::updateQP / marginalizeQP leave them (:351-447, :475-713),
// and the tail of DecentralizedEstimation::update (DecentralEst.cpp:179-185).
//
// Same algorithm as OSQP (Ruiz equilibration, per-row rho, sigma/alpha-relaxed ADMM,
// termination every check_termination iterations on unscaled residuals, adaptive rho with
// refactorisation, cold start), different linear algebra: the QP is never assembled.
// Per window step k the variables are [x_k(9) v_k(3L) w_k(9) c_k(3)] and the rows
// [Meas_k(3L) Dyn_k(9) VO_k(3)], every slack v/w/c appears in exactly
// one row with coefficient -1, and H is block diagonal.  The ADMM linear system
//     (P + sigma I + A' diag(rho) A) xt = sigma x - q + A'(diag(rho) z - y),   zt = A xt
// (OSQP's KKT system with nu eliminated) therefore reduces, after eliminating the slack blocks
// (3x3 per leg, 6x6+3 for w, 3x3 for c), to a block-tridiagonal SPD system in the x_k with
// 9x9 blocks — a fixed-interval smoother — solved by block LDL': S_k^-1 and W_k = C_k S_k^-1
// are the "banded KKT factor", refreshed whenever rho changes.
//
// Memory (SolveMem): when it fits 2 workgroups per CU (Go1/Cassie at N = 20: <= 80 KiB) the
// ADMM iterates, the scaling vectors, the bounds, the slack-block inverses, the block
// tridiagonal factor and the rotations all live in LDS, so an ADMM iteration touches HBM only
// at the termination checks (window records, every 25 iterations).  Longer windows (PogoX,
// N = 100) keep the iterates in LDS and stream the factor from the workgroup's HBM scratch slab.
#pragma once
#include "cfg.h"
#include "mhe_assemble_core.h"
#include "smallmat.h"

namespace dekf {

// Section stamps for the DIAGNOSTIC build only (-DDEKF_PROFILE -> libdekf_prof.so); the product
// kernel executes none of this.  Stamp values go to DevState::prof and nowhere else.
#if defined(DEKF_PROFILE) && DEKF_DEVICE_BUILD
#define DEKF_PROF_MARK(q, sec)                                                  \
    do {                                                                        \
        if (DEKF_LANE() == 0) {                                                 \
            long long now_ = clock64();                                         \
            (q).prof[sec] += (double)(now_ - (q).prof_last);                    \
            (q).prof_last = now_;                                               \
        }                                                                       \
    } while (0)
#else
#define DEKF_PROF_MARK(q, sec) ((void)0)
#endif

constexpr int SOLVE_TMP = 344;  // [162,171) scaled q; at factor time [0,162) and [176,338): two Gauss-Jordan ping-pong pairs

// how many doubles of LDS a solve needs in each placement mode
struct SolveLayout {
    int n_pad, m_pad, K;
    int vec;        // iterates: x z y xt zt at xs xd tmp
    int resident;   // D E lo hi Sv Sw Sc Sinv Wk R
    DEKF_HD void init(int N, int L) {
        int nm = 3 * L;
        K = N;
        n_pad = N * (9 + nm + 12);
        m_pad = N * (nm + 12);
        vec = 2 * n_pad + 4 * m_pad + 18 * N + SOLVE_TMP;
        resident = n_pad + 3 * m_pad + N * (6 * L + 24 + 6) + N * (45 + 81) + 9 * N;
    }
    // LDS-resident factor only if two workgroups still fit in one CU's 160 KiB
    DEKF_HD bool factor_in_lds() const { return (size_t)(vec + resident) * 8 <= 80 * 1024; }
    // the factor-time product P A_dyn can alias the (then dead) xt|zt|at vectors when they are big enough
    DEKF_HD bool pa_in_lds() const { return factor_in_lds() && (n_pad + 2 * m_pad >= (K - 1) * 81); }
    DEKF_HD size_t lds_bytes() const { return (size_t)(factor_in_lds() ? vec + resident : vec) * 8; }
};

// variable / row indices with the leg count known at compile time: every / and % by the
// per-step block sizes becomes a multiply-shift instead of a ~40-instruction software divide
template <int L>
struct IdxT {
    static constexpr int nm = 3 * L, SV = 21 + 3 * L, SC = 12 + 3 * L;
    DEKF_FN static int x(int k, int j) { return k * SV + j; }
    DEKF_FN static int v(int k, int r) { return k * SV + 9 + r; }
    DEKF_FN static int w(int k, int r) { return k * SV + 9 + nm + r; }
    DEKF_FN static int c(int k, int a) { return k * SV + 18 + nm + a; }
    DEKF_FN static int rm(int k, int r) { return k * SC + r; }
    DEKF_FN static int rd(int k, int r) { return k * SC + nm + r; }
    DEKF_FN static int rv(int k, int a) { return k * SC + nm + 9 + a; }
};

template <int L>
struct SolveCtx {
    static constexpr int LEGS = L;
    const DevCfg& c;
    const DevState& s;
    int b, K, kstart, n, m;
    IdxT<L> ix;
    // LDS always
    double *x, *z, *y, *xt, *zt, *at, *xs, *xd, *tmp;
    // LDS or HBM scratch
    double *D, *E, *lo, *hi, *Sv, *Sw, *Sc, *Sinv, *Wk, *R;
    // factor-time temporaries
    double *Wm, *Wd, *Wc, *PA;
    const double *Mp, *np;
    double cc;   // cost scaling c
    double rho;  // current scalar rho
    double* Pst;  // unscaled P blocks staged by solve_scale (aliases Sinv | Wk until the first factorisation)
    bool staged;  // Pst valid
    double* prof;         // diagnostic build only
    long long prof_last;  // diagnostic build only

    DEKF_FN const double* rec(int k) const {
        return s.rec + ((size_t)b * c.wcap + ((kstart + k) % c.wcap)) * c.rec;
    }
    DEKF_FN double adyn(int k, int r, int j) const { return adyn_entry(R + 9 * k, c.dt, r, j); }
    // per-row rho from the scaled bounds (OSQP set_rho_vec / osqp_update_rho)
    DEKF_FN double rho_at(int r) const {
        double lb = lo[r], ub = hi[r];
        if (lb < -OSQP_INFTY * MIN_SCALING && ub > OSQP_INFTY * MIN_SCALING) return RHO_MIN;
        return (ub - lb < RHO_TOL) ? RHO_EQ_OVER_RHO_INEQ * rho : rho;
    }
    // unscaled bound of row (k, kind, o): kind 0 Meas, 1 Dyn, 2 VO
    DEKF_FN void bounds(int k, int kind, int o, double& lb, double& ub) const {
        const double* r = rec(k);
        if (kind == 0) lb = ub = r[Rec::BM + o];
        else if (kind == 1) lb = ub = (o < 3 ? -0.5 * c.dt * c.dt * r[Rec::AS + o] : (o < 6 ? -c.dt * r[Rec::AS + o - 3] : 0.0));
        else if (r[Rec::VOF] != 0.0) lb = ub = r[Rec::VOB + o];
        else { lb = -OSQP_INFTY; ub = OSQP_INFTY; }
    }
    DEKF_FN void dec_var(int i, int& k, int& kind, int& o) const {
        k = i / ix.SV;
        int q = i - k * ix.SV;
        if (q < 9) { kind = 0; o = q; }
        else if (q < 9 + ix.nm) { kind = 1; o = q - 9; }
        else if (q < 18 + ix.nm) { kind = 2; o = q - 9 - ix.nm; }
        else { kind = 3; o = q - 18 - ix.nm; }
    }
    DEKF_FN void dec_row(int r, int& k, int& kind, int& o) const {
        k = r / ix.SC;
        int q = r - k * ix.SC;
        if (q < ix.nm) { kind = 0; o = q; }
        else if (q < ix.nm + 9) { kind = 1; o = q - ix.nm; }
        else { kind = 2; o = q - ix.nm - 9; }
    }
    // row that slack variable (k, kind 1..3, o) lives in, and vice versa
    DEKF_FN int slack_row(int k, int kind, int o) const { return kind == 1 ? ix.rm(k, o) : (kind == 2 ? ix.rd(k, o) : ix.rv(k, o)); }
    DEKF_FN int row_slack(int k, int kind, int o) const { return kind == 0 ? ix.v(k, o) : (kind == 1 ? ix.w(k, o) : ix.c(k, o)); }

    // (scaled A restricted to the x blocks)' * vec, component x_k[j]
    DEKF_FN double gather_x(int k, int j, const double* vec) const {
        double dj = D[ix.x(k, j)];
        double acc = 0.0;
        if (j >= 3 && j < 6)
            for (int leg = 0; leg < L; ++leg) { int r = ix.rm(k, 3 * leg + j - 3); acc += E[r] * vec[r]; }
        if (k < K - 1) {
            for (int rr = 0; rr < 9; ++rr) {
                double a = adyn(k, rr, j);
                if (a != 0.0) { int r = ix.rd(k, rr); acc += E[r] * a * vec[r]; }
            }
            if (j < 3) { int r = ix.rv(k, j); acc += E[r] * vec[r]; }
        }
        if (k > 0) {
            int r = ix.rd(k - 1, j);
            acc -= E[r] * vec[r];
            if (j < 3) { int r2 = ix.rv(k - 1, j); acc -= E[r2] * vec[r2]; }
        }
        return acc * dj;
    }
    // (scaled A restricted to the x blocks) * xv, row (k, kind, o); xv(k, j) returns x_k[j]
    template <class XF>
    DEKF_FN double row_dot_x(int k, int kind, int o, XF xv) const {
        double acc;
        if (kind == 0) {
            acc = D[ix.x(k, 3 + o % 3)] * xv(k, 3 + o % 3);
        } else if (kind == 1) {
            acc = 0.0;
            for (int j = 0; j < 9; ++j) {
                double a = adyn(k, o, j);
                if (a != 0.0) acc += a * D[ix.x(k, j)] * xv(k, j);
            }
            acc -= D[ix.x(k + 1, o)] * xv(k + 1, o);
        } else {
            acc = D[ix.x(k, o)] * xv(k, o) - D[ix.x(k + 1, o)] * xv(k + 1, o);
        }
        int r = kind == 0 ? ix.rm(k, o) : (kind == 1 ? ix.rd(k, o) : ix.rv(k, o));
        return E[r] * acc;
    }
    // (P_scaled x)_i for variable i, or the inf-norm of column i of P_scaled (reads HBM records)
    DEKF_FN double p_apply(int i, const double* xv, bool norm_only) const {
        int k, kind, o;
        dec_var(i, k, kind, o);
        double di = D[i];
        double acc = 0.0;
        auto term = [&](double pij, int i2) {
            double v = cc * di * pij * D[i2];
            if (norm_only) acc = dmax(acc, fabs(v));
            else acc += v * xv[i2];
        };
        if (kind == 0) {
            if (k == 0)
                for (int t = 0; t < 9; ++t) term(o <= t ? Mp[9 * o + t] : Mp[9 * t + o], ix.x(0, t));
        } else if (kind == 1) {
            int leg = o / 3, a = o - 3 * leg;
            const double* q6 = rec(k) + Rec::qm(ix.nm) + 6 * leg;
            for (int t = 0; t < 3; ++t) term(symget(q6, a, t, 3), ix.v(k, 3 * leg + t));
        } else if (kind == 2) {
            if (o < 6) {
                const double* q21 = rec(k) + Rec::QD;
                for (int t = 0; t < 6; ++t) term(symget(q21, o, t, 6), ix.w(k, t));
            } else term(c.Q_bias_dt2[o - 6], i);
        } else {
            const double* q6 = rec(k) + Rec::QC;
            for (int t = 0; t < 3; ++t) term(symget(q6, o, t, 3), ix.c(k, t));
        }
        return acc;
    }
    // inf-norm of column i of the scaled A
    DEKF_FN double a_colnorm(int i) const {
        int k, kind, o;
        dec_var(i, k, kind, o);
        double di = D[i];
        if (kind != 0) return E[slack_row(k, kind, o)] * di;
        double acc = 0.0;
        if (o >= 3 && o < 6)
            for (int leg = 0; leg < L; ++leg) acc = dmax(acc, E[ix.rm(k, 3 * leg + o - 3)]);
        if (k < K - 1) {
            for (int rr = 0; rr < 9; ++rr) acc = dmax(acc, E[ix.rd(k, rr)] * fabs(adyn(k, rr, o)));
            if (o < 3) acc = dmax(acc, E[ix.rv(k, o)]);
        }
        if (k > 0) {
            acc = dmax(acc, E[ix.rd(k - 1, o)]);
            if (o < 3) acc = dmax(acc, E[ix.rv(k - 1, o)]);
        }
        return acc * di;
    }
    // inf-norm of row r of the scaled A
    DEKF_FN double a_rownorm(int r) const {
        int k, kind, o;
        dec_row(r, k, kind, o);
        double acc = D[row_slack(k, kind, o)];
        if (kind == 0) acc = dmax(acc, D[ix.x(k, 3 + o % 3)]);
        else if (kind == 1) {
            for (int j = 0; j < 9; ++j) acc = dmax(acc, fabs(adyn(k, o, j)) * D[ix.x(k, j)]);
            acc = dmax(acc, D[ix.x(k + 1, o)]);
        } else {
            acc = dmax(acc, dmax(D[ix.x(k, o)], D[ix.x(k + 1, o)]));
        }
        return acc * E[r];
    }
};

DEKF_FN double limit_scaling(double v) {
    v = v < MIN_SCALING ? 1.0 : v;
    v = v > MAX_SCALING ? MAX_SCALING : v;
    return v;
}

// Ruiz equilibration + cost scaling (OSQP scale_data), on the structured QP
//
// The unscaled P blocks (Q_meas, Q_dyn, Q_cam per step, M on x_0) are staged once into `q.Pst`
// (it aliases the S^-1 / W arrays, which are dead until the factorisation), the P column norms are
// cached in `q.x` (dead until the cold start) and every norm loop is kind-homogeneous, so the ten
// passes touch LDS only.  Pst stays valid for step 3a of the FIRST factorisation.
// stage the unscaled P blocks from the HBM window records into q.Pst:
// per step [Qm 6L | Qd 21 | Qc 6], then M (upper triangle, packed 45)
template <class Q>
DEKF_FN void stage_p(Q& q) {
    constexpr int L = Q::LEGS, NM = 3 * L, PS = 6 * L + 27;
    const int K = q.K;
    double* Pst = q.Pst;
    wfor(K * PS + 45, [&](int e) {
        if (e < K * PS) {
            int k = e / PS, o = e - k * PS;
            const double* r = q.rec(k);
            Pst[e] = o < 6 * L ? r[Rec::qm(NM) + o] : (o < 6 * L + 21 ? r[Rec::QD + o - 6 * L] : r[Rec::QC + o - 6 * L - 21]);
        } else {
            int p = e - K * PS, i = 0;
            while (p >= 9 - i) { p -= 9 - i; ++i; }
            Pst[e] = q.Mp[9 * i + i + p];
        }
    });
    q.staged = true;
}

template <class Q>
DEKF_FN void solve_scale(Q& q) {
    constexpr int L = Q::LEGS, NM = 3 * L, SV = 21 + NM, SC = 12 + NM, PS = 6 * L + 27;
    const int n = q.n, m = q.m, K = q.K, K1 = q.K - 1;
    const double dt = q.c.dt, hdt2 = 0.5 * dt * dt;
    double *D = q.D, *E = q.E, *Pst = q.Pst, *pc = q.x, *Dt = q.xt, *Et = q.zt;
    const double* g = q.np;
    stage_p(q);
    wfor(n + m, [&](int e) { if (e < n) D[e] = 1.0; else E[e - n] = 1.0; });
    q.cc = 1.0;
    const double* Mst = Pst + K * PS;
    // inf-norm of every column of c D P D -> pc[] (cc excluded: multiplied in where it is used)
    auto pnorms = [&]() {
        wfor_nosync(K * 9, [&](int e) {  // x columns: only x_0 carries a Hessian (the arrival cost)
            int k = e / 9, j = e - 9 * k;
            double v = 0.0;
            if (k == 0)
                for (int t = 0; t < 9; ++t) v = dmax(v, fabs(D[j] * symget(Mst, j, t, 9) * D[t]));
            pc[k * SV + j] = v;
        });
        wfor_nosync(K * NM, [&](int e) {  // v columns
            int k = e / NM, o = e - k * NM, leg = o / 3, a = o - 3 * leg;
            const double* q6 = Pst + k * PS + 6 * leg;
            const double* d = D + k * SV + 9 + 3 * leg;
            double v = dmax(fabs(symget(q6, a, 0, 3) * d[0]), dmax(fabs(symget(q6, a, 1, 3) * d[1]), fabs(symget(q6, a, 2, 3) * d[2])));
            pc[k * SV + 9 + o] = v * d[a];
        });
        wfor_nosync(K1 * 9, [&](int e) {  // w columns
            int k = e / 9, o = e - 9 * k;
            const double* d = D + k * SV + 9 + NM;
            double v;
            if (o < 6) {
                const double* q21 = Pst + k * PS + 6 * L;
                v = 0.0;
                for (int t = 0; t < 6; ++t) v = dmax(v, fabs(symget(q21, o, t, 6) * d[t]));
                v *= d[o];
            } else v = d[o] * q.c.Q_bias_dt2[o - 6] * d[o];
            pc[k * SV + 9 + NM + o] = v;
        });
        wfor(K1 * 3, [&](int e) {  // c columns
            int k = e / 3, a = e - 3 * k;
            const double* q6 = Pst + k * PS + 6 * L + 21;
            const double* d = D + k * SV + 18 + NM;
            double v = dmax(fabs(symget(q6, a, 0, 3) * d[0]), dmax(fabs(symget(q6, a, 1, 3) * d[1]), fabs(symget(q6, a, 2, 3) * d[2])));
            pc[k * SV + 18 + NM + a] = v * d[a];
        });
    };
    pnorms();
    for (int it = 0; it < q.c.scaling; ++it) {
        const double cc = q.cc;
        // ---- column norms of [P; A] -> Dt, row norms of A -> Et
        wfor_nosync(K * 3, [&](int e) {  // position columns
            int k = e / 3, a = e - 3 * k, i = k * SV + a;
            double an = 0.0;
            if (k < K1) an = dmax(E[k * SC + NM + a], E[k * SC + NM + 9 + a]);
            if (k > 0) an = dmax(an, dmax(E[(k - 1) * SC + NM + a], E[(k - 1) * SC + NM + 9 + a]));
            Dt[i] = 1.0 / sqrt(limit_scaling(dmax(cc * pc[i], an * D[i])));
        });
        wfor_nosync(K * 3, [&](int e) {  // velocity columns
            int k = e / 3, a = e - 3 * k, i = k * SV + 3 + a;
            double an = 0.0;
            for (int leg = 0; leg < L; ++leg) an = dmax(an, E[k * SC + 3 * leg + a]);
            if (k < K1) an = dmax(an, dmax(E[k * SC + NM + 3 + a], dt * E[k * SC + NM + a]));
            if (k > 0) an = dmax(an, E[(k - 1) * SC + NM + 3 + a]);
            Dt[i] = 1.0 / sqrt(limit_scaling(dmax(cc * pc[i], an * D[i])));
        });
        wfor_nosync(K * 3, [&](int e) {  // bias columns
            int k = e / 3, a = e - 3 * k, i = k * SV + 6 + a;
            double an = 0.0;
            if (k < K1) {
                const double* R = q.R + 9 * k;
                an = E[k * SC + NM + 6 + a];
                for (int r = 0; r < 3; ++r) {
                    double ra = fabs(R[3 * r + a]);
                    an = dmax(an, dmax(hdt2 * ra * E[k * SC + NM + r], dt * ra * E[k * SC + NM + 3 + r]));
                }
            }
            if (k > 0) an = dmax(an, E[(k - 1) * SC + NM + 6 + a]);
            Dt[i] = 1.0 / sqrt(limit_scaling(dmax(cc * pc[i], an * D[i])));
        });
        wfor_nosync(m, [&](int r) {  // slack columns: one entry -1 in their own row
            int k = r / SC, i = k * SV + 9 + (r - k * SC);
            Dt[i] = 1.0 / sqrt(limit_scaling(dmax(cc * pc[i], E[r] * D[i])));
        });
        wfor_nosync(K * NM, [&](int e) {  // Meas rows
            int k = e / NM, o = e - k * NM, r = k * SC + o;
            Et[r] = 1.0 / sqrt(limit_scaling(E[r] * dmax(D[k * SV + 9 + o], D[k * SV + 3 + o % 3])));
        });
        wfor_nosync(K1 * 3, [&](int e) {  // Dyn position rows
            int k = e / 3, a = e - 3 * k, r = k * SC + NM + a;
            const double* R = q.R + 9 * k + 3 * a;
            const double* d = D + k * SV;
            double v = dmax(dmax(d[9 + NM + a], d[a]), dmax(dt * d[3 + a], d[SV + a]));
            for (int j = 0; j < 3; ++j) v = dmax(v, hdt2 * fabs(R[j]) * d[6 + j]);
            Et[r] = 1.0 / sqrt(limit_scaling(E[r] * v));
        });
        wfor_nosync(K1 * 3, [&](int e) {  // Dyn velocity rows
            int k = e / 3, a = e - 3 * k, r = k * SC + NM + 3 + a;
            const double* R = q.R + 9 * k + 3 * a;
            const double* d = D + k * SV;
            double v = dmax(dmax(d[9 + NM + 3 + a], d[3 + a]), d[SV + 3 + a]);
            for (int j = 0; j < 3; ++j) v = dmax(v, dt * fabs(R[j]) * d[6 + j]);
            Et[r] = 1.0 / sqrt(limit_scaling(E[r] * v));
        });
        wfor_nosync(K1 * 3, [&](int e) {  // Dyn bias rows
            int k = e / 3, a = e - 3 * k, r = k * SC + NM + 6 + a;
            const double* d = D + k * SV;
            Et[r] = 1.0 / sqrt(limit_scaling(E[r] * dmax(d[9 + NM + 6 + a], dmax(d[6 + a], d[SV + 6 + a]))));
        });
        wfor(K1 * 3, [&](int e) {  // VO rows
            int k = e / 3, a = e - 3 * k, r = k * SC + NM + 9 + a;
            const double* d = D + k * SV;
            Et[r] = 1.0 / sqrt(limit_scaling(E[r] * dmax(d[18 + NM + a], dmax(d[a], d[SV + a]))));
        });
        wfor(n + m, [&](int e) { if (e < n) D[e] *= Dt[e]; else E[e - n] *= Et[e - n]; });
        // ---- cost normalisation: mean column norm of the re-scaled P against |q|_inf
        pnorms();
        double psum = cc * wred_sum(n, [&](int i) { return pc[i]; });
        double qn = 0.0;
        for (int j = 0; j < 9; ++j) qn = dmax(qn, fabs(cc * D[j] * g[j]));
        double ct = 1.0 / limit_scaling(dmax(psum / (double)n, limit_scaling(qn)));
        q.cc = cc * ct;
        DEKF_SYNC();
    }
}

// numeric factorisation for the current rho: slack-block inverses, effective row weights,
// block-tridiagonal LDL' (S_k^-1 packed symmetric, W_k)
template <class Q>
DEKF_FN bool solve_factor(Q& q) {
    const DevCfg& c = q.c;
    constexpr int L = Q::LEGS, nm = 3 * Q::LEGS;
    const int K = q.K;
    const auto& ix = q.ix;
    const double sigma = c.sigma, cc = q.cc;
    // 3a. slack blocks: one lane per block, P blocks from the staged copy (Sinv | Wk are dead here:
    //     the previous factor is being replaced)
    constexpr int PS = 6 * L + 27;
    if (!q.staged) stage_p(q);
    q.staged = false;  // 3c overwrites the staging area
    wfor(K * (L + 2), [&](int e) {
        int k = e / (L + 2), blk = e - k * (L + 2);
        const double* pk = q.Pst + k * PS;  // [Qm 6L | Qd 21 | Qc 6]
        if (blk < L) {
            const double* q6 = pk + 6 * blk;
            double gv[3], rr[3], S6[6], Si[6];
            for (int a = 0; a < 3; ++a) {
                int row = ix.rm(k, 3 * blk + a);
                rr[a] = q.rho_at(row);
                gv[a] = rr[a] * q.E[row] * q.D[ix.v(k, 3 * blk + a)];
            }
            for (int a = 0; a < 3; ++a)
                for (int d = a; d < 3; ++d)
                    S6[symidx(a, d, 3)] = cc * q.D[ix.v(k, 3 * blk + a)] * q6[symidx(a, d, 3)] * q.D[ix.v(k, 3 * blk + d)];
            for (int a = 0; a < 3; ++a) S6[symidx(a, a, 3)] += sigma + gv[a] * q.E[ix.rm(k, 3 * blk + a)] * q.D[ix.v(k, 3 * blk + a)];
            inv3_sym(S6, Si);
            for (int t = 0; t < 6; ++t) q.Sv[(k * L + blk) * 6 + t] = Si[t];
            for (int a = 0; a < 3; ++a)
                for (int d = a; d < 3; ++d)
                    q.Wm[(k * L + blk) * 6 + symidx(a, d, 3)] = (a == d ? rr[a] : 0.0) - gv[a] * Si[symidx(a, d, 3)] * gv[d];
        } else if (k < K - 1 && blk == L) {
            const double* q21 = pk + 6 * L;
            double S[36], gv[9], rr[9];
#pragma unroll
            for (int a = 0; a < 9; ++a) {
                int row = ix.rd(k, a);
                rr[a] = q.rho_at(row);
                gv[a] = rr[a] * q.E[row] * q.D[ix.w(k, a)];
            }
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int d = 0; d < 6; ++d)
                    S[6 * a + d] = cc * q.D[ix.w(k, a)] * symget(q21, a, d, 6) * q.D[ix.w(k, d)] +
                                   (a == d ? sigma + gv[a] * q.E[ix.rd(k, a)] * q.D[ix.w(k, a)] : 0.0);
            inv_spd_unrolled<6>(S);
            double* sw = q.Sw + k * 24;
            double* wd = q.Wd + k * 24;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int d = a; d < 6; ++d) {
                    double si = 0.5 * (S[6 * a + d] + S[6 * d + a]);
                    sw[symidx(a, d, 6)] = si;
                    wd[symidx(a, d, 6)] = (a == d ? rr[a] : 0.0) - gv[a] * si * gv[d];
                }
#pragma unroll
            for (int a = 6; a < 9; ++a) {
                double dw = q.D[ix.w(k, a)];
                double sdiag = cc * dw * c.Q_bias_dt2[a - 6] * dw + sigma + gv[a] * q.E[ix.rd(k, a)] * dw;
                sw[21 + a - 6] = 1.0 / sdiag;
                wd[21 + a - 6] = rr[a] - gv[a] * gv[a] / sdiag;
            }
        } else if (k < K - 1 && blk == L + 1) {
            const double* q6 = pk + 6 * L + 21;
            double gv[3], rr[3], S6[6], Si[6];
            for (int a = 0; a < 3; ++a) {
                int row = ix.rv(k, a);
                rr[a] = q.rho_at(row);
                gv[a] = rr[a] * q.E[row] * q.D[ix.c(k, a)];
            }
            for (int a = 0; a < 3; ++a)
                for (int d = a; d < 3; ++d)
                    S6[symidx(a, d, 3)] = cc * q.D[ix.c(k, a)] * q6[symidx(a, d, 3)] * q.D[ix.c(k, d)];
            for (int a = 0; a < 3; ++a) S6[symidx(a, a, 3)] += sigma + gv[a] * q.E[ix.rv(k, a)] * q.D[ix.c(k, a)];
            inv3_sym(S6, Si);
            for (int t = 0; t < 6; ++t) q.Sc[k * 6 + t] = Si[t];
            for (int a = 0; a < 3; ++a)
                for (int d = a; d < 3; ++d)
                    q.Wc[k * 6 + symidx(a, d, 3)] = (a == d ? rr[a] : 0.0) - gv[a] * Si[symidx(a, d, 3)] * gv[d];
        }
    });
    // effective dyn-row weight as a 9x9 accessor
    auto wd_at = [&](int k, int a, int d) -> double {
        const double* wd = q.Wd + k * 24;
        if (a < 6 && d < 6) return symget(wd, a, d, 6);
        return (a == d) ? wd[21 + a - 6] : 0.0;
    };
    // 3b. PA_k = Wd_k * (E_D A_dyn D_x)
    wfor((K - 1) * 81, [&](int e) {
        int k = e / 81, p = e - 81 * k, i = p / 9, j = p - 9 * i;
        double acc = 0.0;
        for (int t = 0; t < 9; ++t) {
            double w = wd_at(k, i, t);
            if (w != 0.0) acc += w * q.E[ix.rd(k, t)] * q.adyn(k, t, j);
        }
        q.PA[e] = acc * q.D[ix.x(k, j)];
    });
    // 3c. T_kk (upper triangle, packed) -> Sinv[k], C_k -> Wk[k]
    wfor(K * 45 + (K - 1) * 81, [&](int e) {
        if (e < K * 45) {
            int k = e / 45, p = e - 45 * k;
            int i = 0;
            while (p >= 9 - i) { p -= 9 - i; ++i; }
            int j = i + p;
            double di = q.D[ix.x(k, i)], dj = q.D[ix.x(k, j)];
            double acc = (i == j) ? sigma : 0.0;
            if (k == 0) acc += cc * di * q.Mp[9 * i + j] * dj;
            if (i >= 3 && j < 6)
                for (int leg = 0; leg < L; ++leg)
                    acc += q.E[ix.rm(k, 3 * leg + i - 3)] * di * symget(q.Wm + (k * L + leg) * 6, i - 3, j - 3, 3) *
                           q.E[ix.rm(k, 3 * leg + j - 3)] * dj;
            if (k < K - 1) {
                if (j < 3) acc += q.E[ix.rv(k, i)] * di * symget(q.Wc + k * 6, i, j, 3) * q.E[ix.rv(k, j)] * dj;
                for (int t = 0; t < 9; ++t) {
                    double a = q.adyn(k, t, i);
                    if (a != 0.0) acc += q.E[ix.rd(k, t)] * a * di * q.PA[k * 81 + 9 * t + j];
                }
            }
            if (k > 0) {
                acc += q.E[ix.rd(k - 1, i)] * di * wd_at(k - 1, i, j) * q.E[ix.rd(k - 1, j)] * dj;
                if (j < 3)
                    acc += q.E[ix.rv(k - 1, i)] * di * symget(q.Wc + (k - 1) * 6, i, j, 3) * q.E[ix.rv(k - 1, j)] * dj;
            }
            q.Sinv[e] = acc;
        } else {
            int e2 = e - K * 45;
            int k = e2 / 81, p = e2 - 81 * k, i = p / 9, j = p - 9 * i;
            double d1 = q.D[ix.x(k + 1, i)];
            double acc = -q.E[ix.rd(k, i)] * d1 * q.PA[k * 81 + p];
            if (i < 3 && j < 3)
                acc -= q.E[ix.rv(k, i)] * d1 * symget(q.Wc + k * 6, i, j, 3) * q.E[ix.rv(k, j)] * q.D[ix.x(k, j)];
            q.Wk[e2] = acc;
        }
    });
    // C_k moves to PA[k]; W_k will be written to Wk[k]
    wfor((K - 1) * 81, [&](int e) { q.PA[e] = q.Wk[e]; });
    // 3d. TWO-SIDED block LDL' ("burn at both ends"): blocks 0..mid-1 are eliminated downwards,
    //     blocks K-1..mid+1 upwards, both fronts in the same phases (two 9x9 problems per phase),
    //     and they meet in block mid.  Same storage as a one-sided factorisation, half the depth:
    //       top     S_k = T_kk - W_{k-1} C_{k-1}',      W_k     = C_k S_k^-1        (k <  mid)
    //       bottom  S_k = T_kk - What_k C_k,            What_{k-1} = C_{k-1}' S_k^-1 (k >  mid)
    //       middle  S_m = T_mm - W_{m-1} C_{m-1}' - What_m C_m
    //     W_k lives in Wk[k] for k < mid, What_k in Wk[k] for k >= mid (it couples block k+1 to k).
    bool ok = true;
    const int mid = K / 2;
    const int nph = (mid > K - 1 - mid ? mid : K - 1 - mid);
    double* bufs[2][2] = {{q.tmp, q.tmp + 81}, {q.tmp + 176, q.tmp + 257}};
    // S of block k into dst; use_top / use_bot select the Schur terms
    auto build_s = [&](int k, bool use_top, bool use_bot, int p, double* dst) {
        int i = p / 9, j = p - 9 * i;
        int lo_ = i < j ? i : j, hi_ = i < j ? j : i;
        double acc = q.Sinv[k * 45 + symidx(lo_, hi_, 9)];
        if (use_top) {
            const double* Wp = q.Wk + (k - 1) * 81;
            const double* Cp = q.PA + (k - 1) * 81;
            double s1 = 0.0, s2 = 0.0;
            for (int t = 0; t < 9; ++t) { s1 += Wp[9 * i + t] * Cp[9 * j + t]; s2 += Wp[9 * j + t] * Cp[9 * i + t]; }
            acc -= 0.5 * (s1 + s2);
        }
        if (use_bot) {
            const double* Wh = q.Wk + k * 81;
            const double* Ck = q.PA + k * 81;
            double s1 = 0.0, s2 = 0.0;
            for (int t = 0; t < 9; ++t) { s1 += Wh[9 * i + t] * Ck[9 * t + j]; s2 += Wh[9 * j + t] * Ck[9 * t + i]; }
            acc -= 0.5 * (s1 + s2);
        }
        dst[p] = acc;
    };
    auto gj_step = [&](const double* src, double* dst, int pv, int p) {
        int i = p / 9, j = p - 9 * i;
        double d = 1.0 / src[pv * 9 + pv];
        double v;
        if (i == pv) v = (j == pv) ? d : src[p] * d;
        else if (j == pv) v = -src[p] * d;
        else v = src[p] - src[i * 9 + pv] * src[pv * 9 + j] * d;
        dst[p] = v;
    };
    auto pivot_ok = [&](const double* src, int pv) { double piv = src[pv * 9 + pv]; return (fabs(piv) > 0.0) && (fabs(piv) < 1e300); };
    auto store_sinv = [&](int k, const double* src, int e) {
        int p = e, i = 0;
        while (p >= 9 - i) { p -= 9 - i; ++i; }
        int j = i + p;
        q.Sinv[k * 45 + e] = 0.5 * (src[9 * i + j] + src[9 * j + i]);
    };
    for (int t = 0; t <= nph; ++t) {
        const bool last = t == nph;             // the meeting block
        const int kt = last ? mid : t, kb = K - 1 - t;
        const bool vt = last || t < mid;        // side 0 active
        const bool vb = !last && kb > mid;      // side 1 active
        wfor(162, [&](int e) {
            int side = e / 81, p = e - 81 * side;
            if (side == 0 && vt) build_s(kt, kt > 0, last && mid < K - 1, p, bufs[0][0]);
            if (side == 1 && vb) build_s(kb, false, kb < K - 1, p, bufs[1][0]);
        });
        int cur = 0;
        for (int pv = 0; pv < 9; ++pv) {
            if (vt && !pivot_ok(bufs[0][cur], pv)) ok = false;
            if (vb && !pivot_ok(bufs[1][cur], pv)) ok = false;
            wfor(162, [&](int e) {
                int side = e / 81, p = e - 81 * side;
                if ((side == 0 && vt) || (side == 1 && vb)) gj_step(bufs[side][cur], bufs[side][cur ^ 1], pv, p);
            });
            cur ^= 1;
        }
        const double* st = bufs[0][cur];
        const double* sb = bufs[1][cur];
        wfor(252, [&](int e) {
            int side = e / 126, o = e - 126 * side;
            if (side == 0 && vt) {
                if (o < 45) store_sinv(kt, st, o);
                else if (!last) {  // W_kt = C_kt S^-1
                    int p = o - 45, i = p / 9, j = p - 9 * i;
                    const double* Ck = q.PA + kt * 81;
                    double acc = 0.0;
                    for (int u = 0; u < 9; ++u) acc += Ck[9 * i + u] * st[9 * u + j];
                    q.Wk[kt * 81 + p] = acc;
                }
            } else if (side == 1 && vb) {
                if (o < 45) store_sinv(kb, sb, o);
                else {  // What_{kb-1} = C_{kb-1}' S^-1
                    int p = o - 45, i = p / 9, j = p - 9 * i;
                    const double* Ck = q.PA + (kb - 1) * 81;
                    double acc = 0.0;
                    for (int u = 0; u < 9; ++u) acc += Ck[9 * u + i] * sb[9 * u + j];
                    q.Wk[(kb - 1) * 81 + p] = acc;
                }
            }
        });
    }
    return ok;
}

// ---------------------------------------------------------------- one ADMM linear solve
// Every loop below runs over ONE kind of row / variable (Meas rows, Dyn p/v rows, Dyn bias rows,
// VO rows; x position / velocity / bias columns), so the 64 lanes of a wavefront execute the same
// straight-line code instead of diverging three ways, and all index arithmetic is by constants.
// Meas and Dyn rows are equalities by construction (l == u), so their rho is rho_eq without a
// look-up; VO rows go through rho_at() because their bounds switch between +-inf and equality.

// sum over the rows that touch x_k[a] / x_k[3+a] / x_k[6+a] of A(row, col) * w(row), w(row) already
// carrying the row scaling E[row]
template <class Q, class WF>
DEKF_FN double gather_pcol(const Q& q, int k, int a, WF w) {
    constexpr int NM = 3 * Q::LEGS, SC = 12 + NM;
    double g = 0.0;
    if (k < q.K - 1) g += w(k * SC + NM + a) + w(k * SC + NM + 9 + a);
    if (k > 0) g -= w((k - 1) * SC + NM + a) + w((k - 1) * SC + NM + 9 + a);
    return g;
}
template <class Q, class WF>
DEKF_FN double gather_vcol(const Q& q, int k, int a, WF w) {
    constexpr int L = Q::LEGS, NM = 3 * L, SC = 12 + NM;
    double g = 0.0;
#pragma unroll
    for (int leg = 0; leg < L; ++leg) g += w(k * SC + 3 * leg + a);
    if (k < q.K - 1) g += w(k * SC + NM + 3 + a) + q.c.dt * w(k * SC + NM + a);
    if (k > 0) g -= w((k - 1) * SC + NM + 3 + a);
    return g;
}
template <class Q, class WF>
DEKF_FN double gather_bcol(const Q& q, int k, int a, WF w) {
    constexpr int NM = 3 * Q::LEGS, SC = 12 + NM;
    double g = 0.0;
    if (k < q.K - 1) {
        const double dt = q.c.dt, hdt2 = 0.5 * dt * dt;
        const double* R = q.R + 9 * k;
        g += w(k * SC + NM + 6 + a);
#pragma unroll
        for (int r = 0; r < 3; ++r) g -= R[3 * r + a] * (hdt2 * w(k * SC + NM + r) + dt * w(k * SC + NM + 3 + r));
    }
    if (k > 0) g -= w((k - 1) * SC + NM + 6 + a);
    return g;
}

#if DEKF_DEVICE_BUILD
// Block-tridiagonal forward / backward sweeps with the running 9-vector held in registers of
// lanes 0..8 of the first wavefront and broadcast with v_readlane: the dependent chain of one
// step is 18 v_readlane + 9 FMA, with no LDS store->load round trip and therefore no exposure to
// the LDS queue the other wavefronts keep busy.  Same arithmetic as the w0for form in
// admm_linear (tests/hostsim runs that one); lanes >= 9 mirror lane 8 and never store.
DEKF_FN double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
template <class Q>
DEKF_FN void tri_sweeps_registers(Q& q) {
    constexpr int SV = 21 + 3 * Q::LEGS;
    const int K = q.K;
    double *xs = q.xs, *xd = q.xd;
    const int lane = DEKF_LANE();
    const int i = lane < 9 ? lane : 8;
    const bool act = lane < 9;
    double f = xs[i];
    for (int k = 1; k < K; ++k) {
        const double* wr = q.Wk + (k - 1) * 81 + 9 * i;
        double w[9], ft[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = wr[t];
        double b = xs[9 * k + i];
#pragma unroll
        for (int t = 0; t < 9; ++t) ft[t] = readlane_f64(f, t);
        double a0 = w[0] * ft[0] + w[3] * ft[3] + w[6] * ft[6];
        double a1 = w[1] * ft[1] + w[4] * ft[4] + w[7] * ft[7];
        double a2 = w[2] * ft[2] + w[5] * ft[5] + w[8] * ft[8];
        f = b - (a0 + a1 + a2);
        if (act) xs[9 * k + i] = f;
    }
    wave_sync();
    for (int e = lane; e < K * 9; e += WAVE) {  // g_k = S_k^-1 f_k, all k at once
        int k = e / 9, r = e - 9 * k;
        const double* Si = q.Sinv + k * 45;
        const double* fk = xs + 9 * k;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int t = 0; t < 9; t += 3) {
            a0 += symget(Si, r, t, 9) * fk[t];
            a1 += symget(Si, r, t + 1, 9) * fk[t + 1];
            a2 += symget(Si, r, t + 2, 9) * fk[t + 2];
        }
        xd[e] = a0 + a1 + a2;
    }
    wave_sync();
    double u = xd[9 * (K - 1) + i];
    if (act) { xs[9 * (K - 1) + i] = u; xd[9 * (K - 1) + i] = q.D[(K - 1) * SV + i] * u; }
    for (int k = K - 2; k >= 0; --k) {
        const double* W = q.Wk + k * 81;
        double w[9], ut[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = W[9 * t + i];
        double g = xd[9 * k + i];
        double dk = q.D[k * SV + i];
#pragma unroll
        for (int t = 0; t < 9; ++t) ut[t] = readlane_f64(u, t);
        double a0 = w[0] * ut[0] + w[3] * ut[3] + w[6] * ut[6];
        double a1 = w[1] * ut[1] + w[4] * ut[4] + w[7] * ut[7];
        double a2 = w[2] * ut[2] + w[5] * ut[5] + w[8] * ut[8];
        u = g - (a0 + a1 + a2);
        if (act) { xs[9 * k + i] = u; xd[9 * k + i] = dk * u; }
    }
}
#endif

#if DEKF_DEVICE_BUILD
// ALTERNATIVE (compiled only with -DDEKF_SWEEP_MFMA; measured and rejected in round 1, see below):
// block-tridiagonal forward / backward sweeps on the matrix cores.  One step of either recurrence
// is a 9x9 mat-vec plus a vector, f_k = b_k - W f_{k-1}: as D = C + A*B with A = -W (padded to
// 16 x 12, three k-steps of v_mfma_f64_16x16x4_f64), B = the previous vector in column 0 and
// C = b_k in column 0.  On gfx950 the f64 accumulator map is D[row = (lane>>4) + 4*reg][col = lane&15]
// and the B operand map is B[k = lane>>4][col = lane&15] (probe: tools/probes/mfma_f64_probe.hip), so
// register `s` of the previous result IS the B operand of k-step s in the same lane: the dependent
// chain of a step is three MFMAs with no cross-lane traffic at all (the hardware does the broadcast
// that cost 18 v_readlane per step before).  Columns 1..15 stay identically zero.  Same arithmetic
// as the w0for form in admm_linear (tests/hostsim runs that one).
// Measured on MI355X (tools/profile_sections.py, Go1 B=4096): correct (all GPU parity tests pass) but
// ~980 ticks per step against ~450 for the v_readlane form: the dependent f64 MFMA costs 64 (D->C) to
// 96 (D->B) ticks on an idle chip (tools/probes/mfma_f64_latency.hip), three per step, plus the
// MFMA->VALU hazards and the operand loads in the chain; with only 81 of 3072 MACs useful it does not pay.
typedef double dekf_v4d __attribute__((ext_vector_type(4)));
template <class Q>
DEKF_FN void tri_sweeps_mfma(Q& q) {
    constexpr int SV = 21 + 3 * Q::LEGS;
    const int K = q.K;
    double *xs = q.xs, *xd = q.xd;
    const int l = DEKF_LANE();
    const int ci = l & 15, kq = l >> 4;          // A row / B,C,D column ; k-quad = accumulator row group
    const int cr = ci < 9 ? ci : 0;              // clamped row for the (masked) A loads
    const bool arow = ci < 9, col0 = ci == 0;
    const bool r2 = kq == 0;                     // row 8 = kq + 4*2 exists only for kq == 0
    // accumulator rows held by this lane: kq, kq+4, kq+8 (reg 3 = rows 12..15 is always zero)
    // every load below is unconditional (in-bounds for all lanes) and masked by a multiply: a
    // branch per operand would serialise one LDS round trip per branch inside the dependent chain
    const double am = arow ? -1.0 : 0.0, am2 = (arow && r2) ? -1.0 : 0.0;
    const double cm = col0 ? 1.0 : 0.0, cm2 = (col0 && r2) ? 1.0 : 0.0;
    dekf_v4d f = {cm * xs[kq], cm * xs[kq + 4], cm2 * xs[8], 0.0};
    for (int k = 1; k < K; ++k) {
        const double* W = q.Wk + (k - 1) * 81 + 9 * cr;
        double a0 = am * W[kq];
        double a1 = am * W[4 + kq];
        double a2 = am2 * W[8];
        dekf_v4d c = {cm * xs[9 * k + kq], cm * xs[9 * k + kq + 4], cm2 * xs[9 * k + 8], 0.0};
        dekf_v4d acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, f[0], c, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, f[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, f[2], acc, 0, 0, 0);
        f = acc;
        if (col0) { xs[9 * k + kq] = f[0]; xs[9 * k + kq + 4] = f[1]; if (r2) xs[9 * k + 8] = f[2]; }
    }
    wave_sync();
    for (int e = l; e < K * 9; e += WAVE) {  // g_k = S_k^-1 f_k, all k at once
        int k = e / 9, r = e - 9 * k;
        const double* Si = q.Sinv + k * 45;
        const double* fk = xs + 9 * k;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int t = 0; t < 9; t += 3) {
            s0 += symget(Si, r, t, 9) * fk[t];
            s1 += symget(Si, r, t + 1, 9) * fk[t + 1];
            s2 += symget(Si, r, t + 2, 9) * fk[t + 2];
        }
        xd[e] = s0 + s1 + s2;
    }
    wave_sync();
    dekf_v4d u = {0.0, 0.0, 0.0, 0.0};
    if (col0) {
        const int o = 9 * (K - 1);
        u[0] = xd[o + kq]; u[1] = xd[o + kq + 4]; if (r2) u[2] = xd[o + 8];
        const double* Dk = q.D + (K - 1) * SV;
        xs[o + kq] = u[0]; xs[o + kq + 4] = u[1];
        xd[o + kq] = Dk[kq] * u[0]; xd[o + kq + 4] = Dk[kq + 4] * u[1];
        if (r2) { xs[o + 8] = u[2]; xd[o + 8] = Dk[8] * u[2]; }
    }
    for (int k = K - 2; k >= 0; --k) {
        const double* W = q.Wk + k * 81 + cr;   // A = -W': A[ci][kk] = -W[kk][ci]
        double a0 = am * W[9 * kq];
        double a1 = am * W[9 * (4 + kq)];
        double a2 = am2 * W[72];
        dekf_v4d c = {cm * xd[9 * k + kq], cm * xd[9 * k + kq + 4], cm2 * xd[9 * k + 8], 0.0};
        dekf_v4d acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, u[0], c, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, u[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, u[2], acc, 0, 0, 0);
        u = acc;
        if (col0) {
            const double* Dk = q.D + k * SV;
            xs[9 * k + kq] = u[0]; xs[9 * k + kq + 4] = u[1];
            xd[9 * k + kq] = Dk[kq] * u[0]; xd[9 * k + kq + 4] = Dk[kq + 4] * u[1];
            if (r2) { xs[9 * k + 8] = u[2]; xd[9 * k + 8] = Dk[8] * u[2]; }
        }
    }
}
#endif

// One leg of the two-sided block-tridiagonal solve: a chain of `steps` dependent 9x9 mat-vecs
//     v_new = rhs[k_new] - M v_prev,   k_new = k_prev + dk,   M = Wk[k_new + wofs] (TR: transposed)
// starting from the vector stored at block k0.  Forward legs (BWD = false) read rhs from xs and
// overwrite it; outward legs (BWD = true) read rhs = g from xd and leave xs = u, xd = D .* u.
// Device: the running vector sits in registers of lanes 0..8 of the calling wavefront and is
// broadcast with v_readlane (18 per step), so the dependent chain never touches LDS or a barrier;
// all 64 lanes execute it (lanes >= 9 mirror lane 8 and never store).  Host build: plain loops.
template <bool TR, bool BWD, class Q>
DEKF_FN void sweep_chain(Q& q, int k0, int dk, int steps, int wofs) {
    constexpr int SV = 21 + 3 * Q::LEGS;
    double *xs = q.xs, *xd = q.xd;
#if DEKF_DEVICE_BUILD
    const int lane = DEKF_LANE() & 63;
    const int i = lane < 9 ? lane : 8;
    const bool act = lane < 9;
    double v = xs[9 * k0 + i];
    for (int s = 1; s <= steps; ++s) {
        const int kn = k0 + s * dk;
        const double* W = q.Wk + (kn + wofs) * 81;
        double w[9], vt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = TR ? W[9 * t + i] : W[9 * i + t];
        const double rhs = BWD ? xd[9 * kn + i] : xs[9 * kn + i];
        const double dsc = BWD ? q.D[kn * SV + i] : 0.0;
#pragma unroll
        for (int t = 0; t < 9; ++t) vt[t] = readlane_f64(v, t);
        double a0 = w[0] * vt[0] + w[3] * vt[3] + w[6] * vt[6];
        double a1 = w[1] * vt[1] + w[4] * vt[4] + w[7] * vt[7];
        double a2 = w[2] * vt[2] + w[5] * vt[5] + w[8] * vt[8];
        v = rhs - (a0 + a1 + a2);
        if (act) {
            xs[9 * kn + i] = v;
            if (BWD) xd[9 * kn + i] = dsc * v;
        }
    }
#else
    double v[9], nv[9];
    for (int i = 0; i < 9; ++i) v[i] = xs[9 * k0 + i];
    for (int s = 1; s <= steps; ++s) {
        const int kn = k0 + s * dk;
        const double* W = q.Wk + (kn + wofs) * 81;
        for (int i = 0; i < 9; ++i) {
            double w[9];
            for (int t = 0; t < 9; ++t) w[t] = TR ? W[9 * t + i] : W[9 * i + t];
            double a0 = w[0] * v[0] + w[3] * v[3] + w[6] * v[6];
            double a1 = w[1] * v[1] + w[4] * v[4] + w[7] * v[7];
            double a2 = w[2] * v[2] + w[5] * v[5] + w[8] * v[8];
            nv[i] = (BWD ? xd[9 * kn + i] : xs[9 * kn + i]) - (a0 + a1 + a2);
        }
        for (int i = 0; i < 9; ++i) {
            v[i] = nv[i];
            xs[9 * kn + i] = v[i];
            if (BWD) xd[9 * kn + i] = q.D[kn * SV + i] * v[i];
        }
    }
#endif
}

// In: xt = right-hand side (n), at = u (consumed by the caller).  Out: xs = xt on the x blocks
// (K*9), at[row] = xt of that row's slack, zt = A xt.
template <class Q>
DEKF_FN void admm_linear(Q& q) {
    constexpr int L = Q::LEGS, NM = 3 * L, SV = 21 + NM, SC = 12 + NM;
    const int K = q.K, K1 = q.K - 1;
    double *xt = q.xt, *zt = q.zt, *at = q.at, *xs = q.xs, *xd = q.xd;
    const double *D = q.D, *E = q.E;
    const double dt = q.c.dt, hdt2 = 0.5 * dt * dt;
    const double rho_eq = RHO_EQ_OVER_RHO_INEQ * q.rho;

    // B. slack forward elimination: t = S^-1 rhs_s -> zt[row];  E rho beta t -> at[row]
    wfor_nosync(K * NM, [&](int e) {  // Meas rows, 3x3 block per leg
        int k = e / NM, o = e - k * NM, leg = o / 3, a = o - 3 * leg;
        const double* si = q.Sv + (k * L + leg) * 6;
        const double* in = xt + k * SV + 9 + 3 * leg;
        double t = symget(si, a, 0, 3) * in[0] + symget(si, a, 1, 3) * in[1] + symget(si, a, 2, 3) * in[2];
        int r = k * SC + o;
        zt[r] = t;
        at[r] = rho_eq * E[r] * E[r] * D[k * SV + 9 + o] * t;
    });
    wfor_nosync(K1 * 6, [&](int e) {  // Dyn rows, position / velocity: 6x6 block
        int k = e / 6, o = e - 6 * k;
        const double* sw = q.Sw + k * 24;
        const double* in = xt + k * SV + 9 + NM;
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 6; ++u) t += symget(sw, o, u, 6) * in[u];
        int r = k * SC + NM + o;
        zt[r] = t;
        at[r] = rho_eq * E[r] * E[r] * D[k * SV + 9 + NM + o] * t;
    });
    wfor_nosync(K1 * 3, [&](int e) {  // Dyn rows, bias: diagonal
        int k = e / 3, a = e - 3 * k;
        int r = k * SC + NM + 6 + a, sv = k * SV + 9 + NM + 6 + a;
        double t = q.Sw[k * 24 + 21 + a] * xt[sv];
        zt[r] = t;
        at[r] = rho_eq * E[r] * E[r] * D[sv] * t;
    });
    wfor(K1 * 3, [&](int e) {  // VO rows, 3x3 block
        int k = e / 3, a = e - 3 * k;
        const double* si = q.Sc + k * 6;
        const double* in = xt + k * SV + 18 + NM;
        double t = symget(si, a, 0, 3) * in[0] + symget(si, a, 1, 3) * in[1] + symget(si, a, 2, 3) * in[2];
        int r = k * SC + NM + 9 + a;
        zt[r] = t;
        at[r] = q.rho_at(r) * E[r] * E[r] * D[k * SV + 18 + NM + a] * t;
    });
    DEKF_PROF_MARK(q, 3);
    // C. reduced right-hand side on the x blocks
    auto wh = [&](int r) { return at[r]; };
    wfor_nosync(K * 3, [&](int e) {
        int k = e / 3, a = e - 3 * k;
        xs[9 * k + a] = xt[k * SV + a] + D[k * SV + a] * gather_pcol(q, k, a, wh);
    });
    wfor_nosync(K * 3, [&](int e) {
        int k = e / 3, a = e - 3 * k;
        xs[9 * k + 3 + a] = xt[k * SV + 3 + a] + D[k * SV + 3 + a] * gather_vcol(q, k, a, wh);
    });
    wfor(K * 3, [&](int e) {
        int k = e / 3, a = e - 3 * k;
        xs[9 * k + 6 + a] = xt[k * SV + 6 + a] + D[k * SV + 6 + a] * gather_bcol(q, k, a, wh);
    });
    DEKF_PROF_MARK(q, 4);
    // D. two-sided block-tridiagonal solve (factorisation: solve_factor 3d).  Wavefront 0 eliminates
    //    blocks 0..mid-1 downwards while wavefront 1 eliminates K-1..mid+1 upwards; they meet in block
    //    mid; g_k = S_k^-1 f_k for all k at once (it is outside both recursions); then both wavefronts
    //    substitute outwards from the middle.  Half the sequential depth of a one-sided sweep, no
    //    workgroup barrier inside a leg.  Leaves xs = u and xd = D .* u.
    {
        const int mid = K / 2;
#if DEKF_DEVICE_BUILD
        __builtin_amdgcn_s_setprio(3);  // the legs are the critical path and share their SIMDs
#endif
        two_waves([&] { sweep_chain<false, false>(q, 0, 1, mid - 1, -1); },
                  [&] { sweep_chain<false, false>(q, K - 1, -1, K - 2 - mid, 0); });
#if DEKF_DEVICE_BUILD
        __builtin_amdgcn_s_setprio(0);
#endif
        DEKF_SYNC();
        wfor(9, [&](int i) {  // the meeting block
            double acc = 0.0;
            if (mid > 0) {
                const double* W = q.Wk + (mid - 1) * 81 + 9 * i;
                const double* f = xs + 9 * (mid - 1);
                for (int t = 0; t < 9; ++t) acc += W[t] * f[t];
            }
            if (mid < K - 1) {
                const double* W = q.Wk + mid * 81 + 9 * i;
                const double* f = xs + 9 * (mid + 1);
                for (int t = 0; t < 9; ++t) acc += W[t] * f[t];
            }
            xs[9 * mid + i] -= acc;
        });
        wfor(K * 9, [&](int e) {
            int k = e / 9, i = e - 9 * k;
            const double* Si = q.Sinv + k * 45;
            const double* f = xs + 9 * k;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int t = 0; t < 9; t += 3) {
                a0 += symget(Si, i, t, 9) * f[t];
                a1 += symget(Si, i, t + 1, 9) * f[t + 1];
                a2 += symget(Si, i, t + 2, 9) * f[t + 2];
            }
            xd[e] = a0 + a1 + a2;
        });
        wfor(9, [&](int i) {
            double u = xd[9 * mid + i];
            xs[9 * mid + i] = u;
            xd[9 * mid + i] = D[mid * SV + i] * u;
        });
#if DEKF_DEVICE_BUILD
        __builtin_amdgcn_s_setprio(3);
#endif
        two_waves([&] { sweep_chain<true, true>(q, mid, -1, mid, 0); },
                  [&] { sweep_chain<true, true>(q, mid, 1, K - 1 - mid, -1); });
#if DEKF_DEVICE_BUILD
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 5);
    // F. a = A_x xt_x -> at[row];  rho beta a -> xt[slack of the row]
    wfor_nosync(K * NM, [&](int e) {  // Meas
        int k = e / NM, o = e - k * NM, a = o % 3;
        int r = k * SC + o, sv = k * SV + 9 + o;
        double ar = E[r] * xd[9 * k + 3 + a];
        at[r] = ar;
        xt[sv] = rho_eq * E[r] * D[sv] * ar;
    });
    wfor_nosync(K1 * 3, [&](int e) {  // Dyn position rows
        int k = e / 3, a = e - 3 * k;
        const double* R = q.R + 9 * k + 3 * a;
        const double* xk = xd + 9 * k;
        int r = k * SC + NM + a, sv = k * SV + 9 + NM + a;
        double ar = E[r] * (xk[a] + dt * xk[3 + a] - hdt2 * (R[0] * xk[6] + R[1] * xk[7] + R[2] * xk[8]) - xk[9 + a]);
        at[r] = ar;
        xt[sv] = rho_eq * E[r] * D[sv] * ar;
    });
    wfor_nosync(K1 * 3, [&](int e) {  // Dyn velocity rows
        int k = e / 3, a = e - 3 * k;
        const double* R = q.R + 9 * k + 3 * a;
        const double* xk = xd + 9 * k;
        int r = k * SC + NM + 3 + a, sv = k * SV + 9 + NM + 3 + a;
        double ar = E[r] * (xk[3 + a] - dt * (R[0] * xk[6] + R[1] * xk[7] + R[2] * xk[8]) - xk[12 + a]);
        at[r] = ar;
        xt[sv] = rho_eq * E[r] * D[sv] * ar;
    });
    wfor_nosync(K1 * 3, [&](int e) {  // Dyn bias rows
        int k = e / 3, a = e - 3 * k;
        const double* xk = xd + 9 * k;
        int r = k * SC + NM + 6 + a, sv = k * SV + 9 + NM + 6 + a;
        double ar = E[r] * (xk[6 + a] - xk[15 + a]);
        at[r] = ar;
        xt[sv] = rho_eq * E[r] * D[sv] * ar;
    });
    wfor(K1 * 3, [&](int e) {  // VO rows
        int k = e / 3, a = e - 3 * k;
        const double* xk = xd + 9 * k;
        int r = k * SC + NM + 9 + a, sv = k * SV + 18 + NM + a;
        double ar = E[r] * (xk[a] - xk[9 + a]);
        at[r] = ar;
        xt[sv] = q.rho_at(r) * E[r] * D[sv] * ar;
    });
    DEKF_PROF_MARK(q, 7);
    // G. slack back-substitution s = t + S^-1 (rho beta a) -> at[row];  zt = a - beta s
    wfor_nosync(K * NM, [&](int e) {
        int k = e / NM, o = e - k * NM, leg = o / 3, a = o - 3 * leg;
        const double* si = q.Sv + (k * L + leg) * 6;
        const double* in = xt + k * SV + 9 + 3 * leg;
        int r = k * SC + o;
        double sl = zt[r] + symget(si, a, 0, 3) * in[0] + symget(si, a, 1, 3) * in[1] + symget(si, a, 2, 3) * in[2];
        zt[r] = at[r] - E[r] * D[k * SV + 9 + o] * sl;
        at[r] = sl;
    });
    wfor_nosync(K1 * 6, [&](int e) {
        int k = e / 6, o = e - 6 * k;
        const double* sw = q.Sw + k * 24;
        const double* in = xt + k * SV + 9 + NM;
        int r = k * SC + NM + o;
        double sl = zt[r];
#pragma unroll
        for (int u = 0; u < 6; ++u) sl += symget(sw, o, u, 6) * in[u];
        zt[r] = at[r] - E[r] * D[k * SV + 9 + NM + o] * sl;
        at[r] = sl;
    });
    wfor_nosync(K1 * 3, [&](int e) {
        int k = e / 3, a = e - 3 * k;
        int r = k * SC + NM + 6 + a, sv = k * SV + 9 + NM + 6 + a;
        double sl = zt[r] + q.Sw[k * 24 + 21 + a] * xt[sv];
        zt[r] = at[r] - E[r] * D[sv] * sl;
        at[r] = sl;
    });
    wfor(K1 * 3, [&](int e) {
        int k = e / 3, a = e - 3 * k;
        const double* si = q.Sc + k * 6;
        const double* in = xt + k * SV + 18 + NM;
        int r = k * SC + NM + 9 + a;
        double sl = zt[r] + symget(si, a, 0, 3) * in[0] + symget(si, a, 1, 3) * in[1] + symget(si, a, 2, 3) * in[2];
        zt[r] = at[r] - E[r] * D[k * SV + 18 + NM + a] * sl;
        at[r] = sl;
    });
    DEKF_PROF_MARK(q, 8);
}

struct SolveInfo {
    int iters, status, rho_updates;
    double pri_res, dua_res, rho;
};

// osqp_setup + osqp_solve + extraction.  window = steps kstart .. kstart+K-1 (newest = T).
// FACTOR_LDS / PA_LDS are compile-time so that every pointer has a provable address space
// (ds_read/ds_write instead of flat_load) — see SolveLayout::factor_in_lds / pa_in_lds.
template <int L, bool FACTOR_LDS, bool PA_LDS, int NFIX = 0>
DEKF_FN SolveInfo solve_window(const DevCfg& c, const DevState& s, int b, int kstart, int K, double* lds, double* gws) {
    // NFIX != 0: the horizon is a compile-time constant, so every LDS array sits at a constant offset
    // (folded into the ds_read/ds_write immediates instead of living in scalar registers)
    const int NH = NFIX ? NFIX : c.N;
    SolveLayout lay;
    lay.init(NH, L);
    Gws g;
    g.init(NH, L);
    SolveCtx<L> q{c, s, b, K, kstart, 0, 0, IdxT<L>{}};
    {   // carve LDS: iterates first; xt, zt, at adjacent so PA can alias them at factor time
        double* p = lds;
        q.x = p; p += lay.n_pad;
        q.z = p; p += lay.m_pad;
        q.y = p; p += lay.m_pad;
        q.xt = p; p += lay.n_pad;
        q.zt = p; p += lay.m_pad;
        q.at = p; p += lay.m_pad;
        q.xs = p; p += 9 * NH;
        q.xd = p; p += 9 * NH;
        q.tmp = p; p += SOLVE_TMP;
        if constexpr (FACTOR_LDS) {
            q.D = p; p += lay.n_pad;
            q.E = p; p += lay.m_pad;
            q.lo = p; p += lay.m_pad;
            q.hi = p; p += lay.m_pad;
            q.Sv = p; p += NH * 6 * L;
            q.Sw = p; p += NH * 24;
            q.Sc = p; p += NH * 6;
            q.Sinv = p; p += NH * 45;
            q.Wk = p; p += NH * 81;
            q.R = p; p += NH * 9;
        } else {
            q.D = gws + g.D; q.E = gws + g.E; q.lo = gws + g.lo; q.hi = gws + g.hi;
            q.Sv = gws + g.Sv; q.Sw = gws + g.Sw; q.Sc = gws + g.Sc;
            q.Sinv = gws + g.Sinv; q.Wk = gws + g.Wk; q.R = gws + g.rho;  // rho slot is unused: R (9K <= m_pad)
        }
        q.Wm = gws + g.Wm; q.Wd = gws + g.Wd; q.Wc = gws + g.Wc;
        if constexpr (PA_LDS) q.PA = q.xt;
        else q.PA = gws + g.PA;
    }
    q.n = (K - 1) * IdxT<L>::SV + 9 + IdxT<L>::nm;
    q.m = (K - 1) * IdxT<L>::SC + IdxT<L>::nm;
    q.Mp = s.Mp + 81 * (size_t)b;
    q.np = s.np_ + 9 * (size_t)b;
    q.cc = 1.0;
    q.Pst = q.Sinv;  // Sinv | Wk are adjacent in both placements and dead until a factorisation writes them
    q.staged = false;
    const int n = q.n, m = q.m;
    const auto& ix = q.ix;
    SolveInfo info{0, DEKF_SOLVE_MAX_ITER, 0, 0.0, 0.0, c.rho0};

    q.prof = s.prof + 16 * (size_t)b;
#if defined(DEKF_PROFILE) && DEKF_DEVICE_BUILD
    q.prof_last = clock64();
    const long long prof_t0 = q.prof_last;
    if (DEKF_LANE() == 0)
        for (int i = 0; i < 16; ++i) q.prof[i] = 0.0;
#endif
    wfor(K * 9, [&](int e) { q.R[e] = q.rec(e / 9)[Rec::R + e % 9]; });
    if (c.scaling > 0) solve_scale(q);
    else { wfor(n + m, [&](int e) { if (e < n) q.D[e] = 1.0; else q.E[e - n] = 1.0; }); }
    DEKF_PROF_MARK(q, 0);
    q.rho = dmin(dmax(c.rho0, RHO_MIN), RHO_MAX);
    // scaled bounds, cold start
    double *x = q.x, *z = q.z, *y = q.y, *xt = q.xt, *zt = q.zt, *at = q.at;
    wfor(n + m, [&](int e) {
        if (e < n) { x[e] = 0.0; return; }
        int r = e - n, k, kind, o;
        q.dec_row(r, k, kind, o);
        double lb, ub;
        q.bounds(k, kind, o, lb, ub);
        q.lo[r] = lb * q.E[r];
        q.hi[r] = ub * q.E[r];
        z[r] = 0.0;
        y[r] = 0.0;
        at[r] = 0.0;  // u = rho z - y of the cold start
    });
    bool ok = solve_factor(q);
    wfor(n + m, [&](int e) {  // cold start: u = 0 and a zero right-hand side (PA may have aliased xt | zt | at)
        if (e < n) xt[e] = 0.0;
        else at[e - n] = 0.0;
    });
    double qs[9];  // scaled linear cost on x_0 (registers for the checks, LDS copy for the per-lane look-ups)
    for (int j = 0; j < 9; ++j) qs[j] = q.cc * q.D[ix.x(0, j)] * q.np[j];
    wfor(9, [&](int j) { q.tmp[162 + j] = q.cc * q.D[ix.x(0, j)] * q.np[j]; });
    const double sigma = c.sigma, alpha = c.alpha;
    const double cinv = 1.0 / q.cc;
    int iter = 0;
    bool done = false;
    DEKF_PROF_MARK(q, 1);
    while (ok && !done && iter < c.max_iter) {
        ++iter;
        // A. right-hand side sigma x - q + A'u on the x columns (u = rho z - y sits in `at`; the slack
        //    entries of the right-hand side were written together with u by the previous update)
        {
            constexpr int NM = 3 * L, SV = 21 + NM;
            const double *D = q.D, *E = q.E;
            const double* qsl = q.tmp + 162;
            auto wu = [&](int r) { return E[r] * at[r]; };
            wfor_nosync(K * 3, [&](int e) {
                int k = e / 3, a = e - 3 * k, i = k * SV + a;
                xt[i] = sigma * x[i] - (k == 0 ? qsl[a] : 0.0) + D[i] * gather_pcol(q, k, a, wu);
            });
            wfor_nosync(K * 3, [&](int e) {
                int k = e / 3, a = e - 3 * k, i = k * SV + 3 + a;
                xt[i] = sigma * x[i] - (k == 0 ? qsl[3 + a] : 0.0) + D[i] * gather_vcol(q, k, a, wu);
            });
            wfor(K * 3, [&](int e) {
                int k = e / 3, a = e - 3 * k, i = k * SV + 6 + a;
                xt[i] = sigma * x[i] - (k == 0 ? qsl[6 + a] : 0.0) + D[i] * gather_bcol(q, k, a, wu);
            });
        }
        DEKF_PROF_MARK(q, 2);
        admm_linear(q);
        // H. x, z, y updates (alpha relaxation, projection onto [lo, hi]); u and the slack part of
        //    the next right-hand side (every slack lives in exactly one row: slack = k*SV + 9 + q)
        {
            constexpr int NM = 3 * L, SV = 21 + NM, SC = 12 + NM;
            const double rho_eq = RHO_EQ_OVER_RHO_INEQ * q.rho, rho_eq_inv = 1.0 / rho_eq;
            wfor_nosync(K * 9, [&](int e) {
                int k = e / 9, i = k * SV + (e - 9 * k);
                x[i] = alpha * q.xs[e] + (1.0 - alpha) * x[i];
            });
            wfor(m, [&](int r) {
                int k = r / SC, qq = r - k * SC, sv = k * SV + 9 + qq;
                x[sv] = alpha * at[r] + (1.0 - alpha) * x[sv];
                bool eq = qq < NM + 9;
                double rv = eq ? rho_eq : q.rho_at(r);
                double rinv = eq ? rho_eq_inv : 1.0 / rv;
                double zh = alpha * zt[r] + (1.0 - alpha) * z[r];
                double zn = dmin(dmax(zh + rinv * y[r], q.lo[r]), q.hi[r]);
                double yn = y[r] + rv * (zh - zn);
                y[r] = yn;
                z[r] = zn;
                double un = rv * zn - yn;
                at[r] = un;
                xt[sv] = sigma * x[sv] - q.E[r] * q.D[sv] * un;
            });
        }
        DEKF_PROF_MARK(q, 9);
        bool can_check = c.check_termination > 0 && (iter % c.check_termination == 0);
        bool adapt_now = c.adaptive_rho && c.adaptive_rho_interval > 0 && (iter % c.adaptive_rho_interval == 0);
        if (can_check || adapt_now || iter == c.max_iter) {
            double ra[6], va[8];
            auto xv = [&](int k, int j) { return x[ix.x(k, j)]; };
            wred_maxn<6>(m, ra, [&](int r, double* acc) {
                int k, kind, o;
                q.dec_row(r, k, kind, o);
                int sv = q.row_slack(k, kind, o);
                double Ax = q.row_dot_x(k, kind, o, xv) - q.E[r] * q.D[sv] * x[sv];
                double pr = Ax - z[r];
                double ei = 1.0 / q.E[r];
                acc[0] = dmax(acc[0], fabs(pr) * ei);
                acc[1] = dmax(acc[1], fabs(z[r]) * ei);
                acc[2] = dmax(acc[2], fabs(Ax) * ei);
                acc[3] = dmax(acc[3], fabs(pr));
                acc[4] = dmax(acc[4], fabs(z[r]));
                acc[5] = dmax(acc[5], fabs(Ax));
            });
            wred_maxn<8>(n, va, [&](int i, double* acc) {
                int k, kind, o;
                q.dec_var(i, k, kind, o);
                double Px = q.p_apply(i, x, false);
                double Aty, qv = 0.0;
                if (kind == 0) {
                    Aty = q.gather_x(k, o, y);
                    if (k == 0) qv = qs[o];
                } else {
                    int r = q.slack_row(k, kind, o);
                    Aty = -q.E[r] * q.D[i] * y[r];
                }
                double dr = qv + Px + Aty;
                double di = 1.0 / q.D[i];
                acc[0] = dmax(acc[0], fabs(dr) * di);
                acc[1] = dmax(acc[1], fabs(qv) * di);
                acc[2] = dmax(acc[2], fabs(Aty) * di);
                acc[3] = dmax(acc[3], fabs(Px) * di);
                acc[4] = dmax(acc[4], fabs(dr));
                acc[5] = dmax(acc[5], fabs(qv));
                acc[6] = dmax(acc[6], fabs(Aty));
                acc[7] = dmax(acc[7], fabs(Px));
            });
            info.pri_res = ra[0];
            info.dua_res = cinv * va[0];
            if (can_check || iter == c.max_iter) {
                double eps_pri = c.eps_abs + c.eps_rel * dmax(ra[1], ra[2]);
                double eps_dua = c.eps_abs + c.eps_rel * cinv * dmax(va[1], dmax(va[2], va[3]));
                if (info.pri_res < eps_pri && info.dua_res < eps_dua) { info.status = DEKF_SOLVE_OK; done = true; }
            }
            DEKF_PROF_MARK(q, 10);
            if (!done && adapt_now) {
                double pr = ra[3] / (dmax(ra[4], ra[5]) + 1e-10);
                double du = va[4] / (dmax(va[5], dmax(va[6], va[7])) + 1e-10);
                double rho_new = dmin(dmax(q.rho * sqrt(pr / (du + 1e-10)), RHO_MIN), RHO_MAX);
                if (rho_new > q.rho * c.adaptive_rho_tolerance || rho_new < q.rho / c.adaptive_rho_tolerance) {
                    q.rho = rho_new;
                    info.rho_updates++;
                    DEKF_SYNC();
                    ok = solve_factor(q);
                    wfor(m, [&](int r) {  // the factorisation scratch may alias xt | zt | at
                        constexpr int SVc = 21 + 3 * L, SCc = 12 + 3 * L;
                        int k = r / SCc, sv = k * SVc + 9 + (r - k * SCc);
                        double un = q.rho_at(r) * z[r] - y[r];
                        at[r] = un;
                        xt[sv] = sigma * x[sv] - q.E[r] * q.D[sv] * un;
                    });
                    DEKF_PROF_MARK(q, 11);
                }
            }
        }
    }
    info.iters = iter;
    info.rho = q.rho;
    // store_solution + update() tail: x_T = D x ; v_b = R (x_T[3:6] + gyro x p_imu_2_opti)
    const double* rT = q.rec(K - 1);
    double xT[9];
    bool finite = ok;
    for (int j = 0; j < 9; ++j) {
        xT[j] = q.D[ix.x(K - 1, j)] * x[ix.x(K - 1, j)];
        if (!(fabs(xT[j]) <= 1e300)) finite = false;
    }
    if (!finite) info.status = DEKF_SOLVE_NUMERIC;
    if (DEKF_LANE() == 0) {
        const double p_opti[3] = {0.016041, 0.089061, 0.0579875};
        double wxp[3], t[3], vb[3];
        cross3(rT + Rec::GY, p_opti, wxp);
        for (int a = 0; a < 3; ++a) t[a] = xT[3 + a] + wxp[a];
        mv3(rT + Rec::R, t, vb);
        for (int j = 0; j < 9; ++j) s.x_mhe[9 * (size_t)b + j] = xT[j];
        for (int a = 0; a < 3; ++a) s.v_b[3 * (size_t)b + a] = vb[a];
        s.status[b] = info.status;
        s.iters[b] = info.iters;
        s.rho_updates[b] = info.rho_updates;
        s.pri_res[b] = info.pri_res;
        s.dua_res[b] = info.dua_res;
    }
    DEKF_SYNC();
#if defined(DEKF_PROFILE) && DEKF_DEVICE_BUILD
    DEKF_PROF_MARK(q, 12);
    if (DEKF_LANE() == 0) q.prof[13] = (double)(clock64() - prof_t0);
#endif
    return info;
}

}  // namespace dekf
