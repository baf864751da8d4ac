// TEST INFRASTRUCTURE ONLY — part of the CPU oracle (see oracle/README.md).
//
// CPU restatement of the OSQP algorithm the reference calls through osqp-eigen
//   MHEproblem::initQP   (src/decentral_legged_est/src/MheSrb.cpp:272-338)  -> osqp_setup
//   MHEproblem::solveQP  (src/decentral_legged_est/src/MheSrb.cpp:340-349)  -> osqp_solve
// with the settings passed in DecentralizedEstimation::InitializeMHE
//   (src/decentral_legged_est/src/DecentralEst.cpp:204-217).
//
// OSQP and osqp-eigen are un-vendored, un-pinned dependencies of the reference
// (find_package(OsqpEigen), no version anywhere; API usage implies osqp-eigen >= 0.7 on
// OSQP 0.6.x).  They are NOT under /root/reference and are not installed here, so this
// file restates the published algorithm: B. Stellato, G. Banjac, P. Goulart, A. Bemporad,
// S. Boyd, "OSQP: an operator splitting solver for quadratic programs", Math. Prog. Comp.
// 12 (2020) — Algorithm 1, sec. 5.1 (linear system), 5.2 (rho selection / adaptive rho),
// 3.4 (termination), 5.1 Ruiz equilibration, 4 (polishing) — with OSQP 0.6.x's constants and update order
// (RHO_MIN 1e-6, RHO_MAX 1e6, RHO_EQ_OVER_RHO_INEQ 1e3, RHO_TOL 1e-4, MIN/MAX_SCALING
// 1e-4/1e4, OSQP_INFTY 1e30, scaled_termination off, polish_refine_iter 3).
//
// Deliberate, documented deviations (both make runs deterministic):
//  * adaptive_rho_interval: OSQP 0.6 derives it from wall-clock (0.4 x setup time, rounded
//    to a multiple of check_termination); here it is a fixed iteration count.
//  * time_limit is ignored.
// The KKT system is solved by a sparse LDL^T (up-looking, elimination-tree based — the
// textbook algorithm QDLDL also follows) under a reverse Cuthill-McKee ordering instead of
// AMD; the ordering changes round-off only.
//
// parity: there is no reference test or golden vector at this boundary ("parity unpinned" by
// the reference).  The pins are tests/test_oracle_*.py: ADMM result vs exact KKT solve, and
// the MHE == KF identity.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <queue>
#include <vector>

#include "densemat.hpp"

namespace orc {

struct Csc {
    int m = 0, n = 0;  // rows, cols
    std::vector<int> p, i;
    std::vector<double> x;
};

// column-compressed copy of a dense matrix; `upper` keeps only i <= j
inline Csc csc_from_dense(const Mat& D, bool upper) {
    Csc S;
    S.m = D.r; S.n = D.c;
    S.p.assign(D.c + 1, 0);
    for (int j = 0; j < D.c; ++j) {
        for (int i = 0; i < D.r; ++i) {
            if (upper && i > j) break;
            double v = D(i, j);
            if (v != 0.0) { S.i.push_back(i); S.x.push_back(v); }
        }
        S.p[j + 1] = (int)S.i.size();
    }
    return S;
}

struct OsqpSettings {
    double rho = 0.1, sigma = 1e-6, alpha = 1.6;
    double eps_abs = 1e-3, eps_rel = 1e-3, eps_prim_inf = 1e-4, eps_dual_inf = 1e-4;
    int max_iter = 4000;
    int scaling = 10;
    int check_termination = 25;
    int adaptive_rho = 1;
    int adaptive_rho_interval = 25;
    double adaptive_rho_tolerance = 5.0;
    // solution polishing (OSQP paper sec. 4; osqp.polish / osqp.delta of the reference, EstSub.cpp:184-188, DecentralEst.cpp:207,211)
    int polish = 0;
    double delta = 1e-6;
    int polish_refine_iter = 3;  // OSQP default; the reference does not set it
};

enum OsqpStatus {
    OSQP_R_UNSOLVED = 0,
    OSQP_R_SOLVED = 1,
    OSQP_R_SOLVED_INACCURATE = 2,
    OSQP_R_MAX_ITER = 3,
    OSQP_R_PRIMAL_INFEASIBLE = 4,
    OSQP_R_DUAL_INFEASIBLE = 5,
    OSQP_R_NON_CVX = 6,
    OSQP_R_NUMERIC = 7
};

// ---------------------------------------------------------------- sparse LDL^T
struct SparseLDL {
    int n = 0;
    std::vector<int> perm, iperm;  // perm[new] = old
    std::vector<int> Ap, Ai;       // permuted upper-triangular pattern
    std::vector<double> Ax;
    std::vector<int> Lp, Li, Parent, Lnz;
    std::vector<double> Lx, D;
    bool ok = false;

    // reverse Cuthill-McKee on the symmetric pattern given by (row, col) pairs
    static std::vector<int> rcm(int n, const std::vector<std::pair<int, int>>& edges) {
        std::vector<std::vector<int>> adj(n);
        for (auto& e : edges)
            if (e.first != e.second) { adj[e.first].push_back(e.second); adj[e.second].push_back(e.first); }
        std::vector<int> deg(n);
        for (int v = 0; v < n; ++v) {
            std::sort(adj[v].begin(), adj[v].end());
            adj[v].erase(std::unique(adj[v].begin(), adj[v].end()), adj[v].end());
            deg[v] = (int)adj[v].size();
        }
        std::vector<char> seen(n, 0);
        std::vector<int> order;
        order.reserve(n);
        std::vector<int> nodes(n);
        for (int v = 0; v < n; ++v) nodes[v] = v;
        std::stable_sort(nodes.begin(), nodes.end(), [&](int a, int b) { return deg[a] < deg[b]; });
        for (int s : nodes) {
            if (seen[s]) continue;
            // pseudo-peripheral start: two BFS sweeps from the min-degree node of the component
            int start = s;
            for (int sweep = 0; sweep < 2; ++sweep) {
                std::vector<int> dist(n, -1);
                std::queue<int> q;
                q.push(start); dist[start] = 0;
                int last = start;
                while (!q.empty()) {
                    int v = q.front(); q.pop();
                    last = v;
                    for (int w : adj[v]) if (dist[w] < 0 && !seen[w]) { dist[w] = dist[v] + 1; q.push(w); }
                }
                start = last;
            }
            std::queue<int> q;
            q.push(start); seen[start] = 1;
            while (!q.empty()) {
                int v = q.front(); q.pop();
                order.push_back(v);
                std::vector<int> nb;
                for (int w : adj[v]) if (!seen[w]) { nb.push_back(w); seen[w] = 1; }
                std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] < deg[b]; });
                for (int w : nb) q.push(w);
            }
        }
        std::reverse(order.begin(), order.end());
        return order;
    }

    // triplets: upper-triangular entries (i <= j) of the symmetric matrix, original indexing
    void analyze(int n_, const std::vector<int>& ti, const std::vector<int>& tj) {
        n = n_;
        std::vector<std::pair<int, int>> edges(ti.size());
        for (size_t k = 0; k < ti.size(); ++k) edges[k] = {ti[k], tj[k]};
        perm = rcm(n, edges);
        iperm.assign(n, 0);
        for (int k = 0; k < n; ++k) iperm[perm[k]] = k;
    }

    bool factor(const std::vector<int>& ti, const std::vector<int>& tj, const std::vector<double>& tv) {
        // permuted upper-triangular CSC
        std::vector<int> cnt(n + 1, 0);
        std::vector<int> pi(ti.size()), pj(ti.size());
        for (size_t k = 0; k < ti.size(); ++k) {
            int a = iperm[ti[k]], b = iperm[tj[k]];
            if (a > b) std::swap(a, b);
            pi[k] = a; pj[k] = b;
            cnt[b + 1]++;
        }
        Ap.assign(n + 1, 0);
        for (int j = 0; j < n; ++j) Ap[j + 1] = Ap[j] + cnt[j + 1];
        Ai.assign(ti.size(), 0);
        Ax.assign(ti.size(), 0.0);
        std::vector<int> fill(Ap.begin(), Ap.end() - 1);
        for (size_t k = 0; k < ti.size(); ++k) {
            int dst = fill[pj[k]]++;
            Ai[dst] = pi[k];
            Ax[dst] = tv[k];
        }
        // symbolic: elimination tree and column counts
        Parent.assign(n, -1);
        Lnz.assign(n, 0);
        std::vector<int> Flag(n, 0);
        for (int k = 0; k < n; ++k) {
            Parent[k] = -1; Flag[k] = k; Lnz[k] = 0;
            for (int p = Ap[k]; p < Ap[k + 1]; ++p) {
                int i = Ai[p];
                if (i < k)
                    for (; Flag[i] != k; i = Parent[i]) {
                        if (Parent[i] == -1) Parent[i] = k;
                        Lnz[i]++;
                        Flag[i] = k;
                    }
            }
        }
        Lp.assign(n + 1, 0);
        for (int k = 0; k < n; ++k) Lp[k + 1] = Lp[k] + Lnz[k];
        Li.assign(Lp[n], 0);
        Lx.assign(Lp[n], 0.0);
        D.assign(n, 0.0);
        // numeric: up-looking, one row of L per step
        std::vector<double> Y(n, 0.0);
        std::vector<int> Pattern(n, 0);
        for (int k = 0; k < n; ++k) {
            Y[k] = 0.0;
            int top = n;
            Flag[k] = k;
            Lnz[k] = 0;
            for (int p = Ap[k]; p < Ap[k + 1]; ++p) {
                int i = Ai[p];
                if (i <= k) {
                    Y[i] += Ax[p];
                    int len = 0;
                    for (; Flag[i] != k; i = Parent[i]) { Pattern[len++] = i; Flag[i] = k; }
                    while (len > 0) Pattern[--top] = Pattern[--len];
                }
            }
            D[k] = Y[k];
            Y[k] = 0.0;
            for (; top < n; ++top) {
                int i = Pattern[top];
                double yi = Y[i];
                Y[i] = 0.0;
                int p2 = Lp[i] + Lnz[i];
                for (int p = Lp[i]; p < p2; ++p) Y[Li[p]] -= Lx[p] * yi;
                double lki = yi / D[i];
                D[k] -= lki * yi;
                Li[p2] = k;
                Lx[p2] = lki;
                Lnz[i]++;
            }
            if (D[k] == 0.0 || !std::isfinite(D[k])) { ok = false; return false; }
        }
        ok = true;
        return true;
    }

    void solve(std::vector<double>& b) const {  // in place, original indexing
        std::vector<double> y(n);
        for (int k = 0; k < n; ++k) y[k] = b[perm[k]];
        for (int j = 0; j < n; ++j)
            for (int p = Lp[j]; p < Lp[j + 1]; ++p) y[Li[p]] -= Lx[p] * y[j];
        for (int j = 0; j < n; ++j) y[j] /= D[j];
        for (int j = n - 1; j >= 0; --j)
            for (int p = Lp[j]; p < Lp[j + 1]; ++p) y[j] -= Lx[p] * y[Li[p]];
        for (int k = 0; k < n; ++k) b[perm[k]] = y[k];
    }
    int nnzL() const { return Lp.empty() ? 0 : Lp[n]; }
};

// ---------------------------------------------------------------- OSQP restatement
class OsqpRestate {
  public:
    static constexpr double RHO_MIN = 1e-6, RHO_MAX = 1e6, RHO_EQ_OVER_RHO_INEQ = 1e3, RHO_TOL = 1e-4;
    static constexpr double MIN_SCALING = 1e-4, MAX_SCALING = 1e4, INFTY = 1e30;

    OsqpSettings s;
    int n = 0, m = 0;
    Csc P, A;  // scaled in place; P upper triangular
    Vec q, l, u;
    Vec D, E, Dinv, Einv;
    double c = 1.0, cinv = 1.0;
    Vec rho_vec, rho_inv_vec;
    std::vector<int> constr_type;
    SparseLDL ldl;
    std::vector<int> kti, ktj;  // KKT triplets (upper)
    std::vector<double> ktv;
    std::vector<int> kkt_rho_pos;  // index in ktv of each -1/rho diagonal
    // iterates
    Vec x, z, y, x_prev, z_prev, xz_tilde, delta_x, delta_y, Ax, Px, Aty;
    // info
    int iters = 0, status = OSQP_R_UNSOLVED, rho_updates = 0, factorizations = 0;
    double pri_res = 0, dua_res = 0;
    Vec x_sol, y_sol;
    int polish_status = 0;  // 0 not run, 1 successful (the polished point replaced the ADMM iterate), -1 unsuccessful
    double pol_pri_res = 0, pol_dua_res = 0;

    static double limit1(double v) {
        v = v < MIN_SCALING ? 1.0 : v;
        v = v > MAX_SCALING ? MAX_SCALING : v;
        return v;
    }
    static double norm_inf(const Vec& v) {
        double r = 0;
        for (double e : v) r = std::max(r, std::fabs(e));
        return r;
    }
    static double scaled_norm_inf(const Vec& S, const Vec& v) {
        double r = 0;
        for (size_t i = 0; i < v.size(); ++i) r = std::max(r, std::fabs(S[i] * v[i]));
        return r;
    }
    // y (+)= A x for CSC A
    static void mat_vec(const Csc& M, const Vec& xx, Vec& yy, bool accumulate) {
        if (!accumulate) std::fill(yy.begin(), yy.end(), 0.0);
        for (int j = 0; j < M.n; ++j)
            for (int p = M.p[j]; p < M.p[j + 1]; ++p) yy[M.i[p]] += M.x[p] * xx[j];
    }
    // y (+)= A' x ; skip_diag used for the strictly-lower part of symmetric P
    static void mat_tpose_vec(const Csc& M, const Vec& xx, Vec& yy, bool accumulate, bool skip_diag) {
        if (!accumulate) std::fill(yy.begin(), yy.end(), 0.0);
        for (int j = 0; j < M.n; ++j)
            for (int p = M.p[j]; p < M.p[j + 1]; ++p) {
                if (skip_diag && M.i[p] == j) continue;
                yy[j] += M.x[p] * xx[M.i[p]];
            }
    }

    void scale_data() {
        D.assign(n, 1.0); E.assign(m, 1.0); c = 1.0;
        Vec Dt(n), Et(m);
        for (int it = 0; it < s.scaling; ++it) {
            // inf-norms of the columns of [P A'; A 0]
            std::fill(Dt.begin(), Dt.end(), 0.0);
            std::fill(Et.begin(), Et.end(), 0.0);
            for (int j = 0; j < n; ++j)
                for (int p = P.p[j]; p < P.p[j + 1]; ++p) {
                    double v = std::fabs(P.x[p]);
                    int i = P.i[p];
                    Dt[j] = std::max(Dt[j], v);
                    if (i != j) Dt[i] = std::max(Dt[i], v);
                }
            for (int j = 0; j < n; ++j)
                for (int p = A.p[j]; p < A.p[j + 1]; ++p) {
                    double v = std::fabs(A.x[p]);
                    Dt[j] = std::max(Dt[j], v);
                    Et[A.i[p]] = std::max(Et[A.i[p]], v);
                }
            for (auto& v : Dt) v = 1.0 / std::sqrt(limit1(v));
            for (auto& v : Et) v = 1.0 / std::sqrt(limit1(v));
            for (int j = 0; j < n; ++j)
                for (int p = P.p[j]; p < P.p[j + 1]; ++p) P.x[p] *= Dt[P.i[p]] * Dt[j];
            for (int j = 0; j < n; ++j)
                for (int p = A.p[j]; p < A.p[j + 1]; ++p) A.x[p] *= Et[A.i[p]] * Dt[j];
            for (int j = 0; j < n; ++j) { q[j] *= Dt[j]; D[j] *= Dt[j]; }
            for (int i = 0; i < m; ++i) E[i] *= Et[i];
            // cost normalisation
            std::fill(Dt.begin(), Dt.end(), 0.0);
            for (int j = 0; j < n; ++j)
                for (int p = P.p[j]; p < P.p[j + 1]; ++p) {
                    double v = std::fabs(P.x[p]);
                    int i = P.i[p];
                    Dt[j] = std::max(Dt[j], v);
                    if (i != j) Dt[i] = std::max(Dt[i], v);
                }
            double mean = 0;
            for (double v : Dt) mean += v;
            mean /= n;
            double nq = limit1(norm_inf(q));
            double ct = 1.0 / limit1(std::max(mean, nq));
            for (auto& v : P.x) v *= ct;
            for (auto& v : q) v *= ct;
            c *= ct;
        }
        cinv = 1.0 / c;
        Dinv.resize(n); Einv.resize(m);
        for (int j = 0; j < n; ++j) Dinv[j] = 1.0 / D[j];
        for (int i = 0; i < m; ++i) { Einv[i] = 1.0 / E[i]; l[i] *= E[i]; u[i] *= E[i]; }
    }

    void set_rho_vec() {
        rho_vec.resize(m); rho_inv_vec.resize(m); constr_type.resize(m);
        s.rho = std::min(std::max(s.rho, RHO_MIN), RHO_MAX);
        for (int i = 0; i < m; ++i) {
            if (l[i] < -INFTY * MIN_SCALING && u[i] > INFTY * MIN_SCALING) {
                constr_type[i] = -1; rho_vec[i] = RHO_MIN;
            } else if (u[i] - l[i] < RHO_TOL) {
                constr_type[i] = 1; rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * s.rho;
            } else {
                constr_type[i] = 0; rho_vec[i] = s.rho;
            }
            rho_inv_vec[i] = 1.0 / rho_vec[i];
        }
    }

    bool build_and_factor_kkt(bool first) {
        if (first) {
            kti.clear(); ktj.clear(); ktv.clear(); kkt_rho_pos.assign(m, -1);
            // P + sigma I (upper); make sure every diagonal exists
            std::vector<char> has_diag(n, 0);
            for (int j = 0; j < n; ++j)
                for (int p = P.p[j]; p < P.p[j + 1]; ++p) {
                    double v = P.x[p];
                    if (P.i[p] == j) { v += s.sigma; has_diag[j] = 1; }
                    kti.push_back(P.i[p]); ktj.push_back(j); ktv.push_back(v);
                }
            for (int j = 0; j < n; ++j)
                if (!has_diag[j]) { kti.push_back(j); ktj.push_back(j); ktv.push_back(s.sigma); }
            // A' in the upper-right block: entry A(i,j) sits at (j, n+i)
            for (int j = 0; j < n; ++j)
                for (int p = A.p[j]; p < A.p[j + 1]; ++p) {
                    kti.push_back(j); ktj.push_back(n + A.i[p]); ktv.push_back(A.x[p]);
                }
            for (int i = 0; i < m; ++i) {
                kkt_rho_pos[i] = (int)ktv.size();
                kti.push_back(n + i); ktj.push_back(n + i); ktv.push_back(-rho_inv_vec[i]);
            }
            ldl.analyze(n + m, kti, ktj);
        } else {
            for (int i = 0; i < m; ++i) ktv[kkt_rho_pos[i]] = -rho_inv_vec[i];
        }
        factorizations++;
        return ldl.factor(kti, ktj, ktv);
    }

    // osqp_setup: copies the data, scales, chooses rho per constraint, factors, cold start
    bool setup(const Csc& P_upper, const Vec& q_, const Csc& A_, const Vec& l_, const Vec& u_,
               const OsqpSettings& st) {
        s = st;
        P = P_upper; A = A_; q = q_; l = l_; u = u_;
        n = P.n; m = A.m;
        iters = 0; status = OSQP_R_UNSOLVED; rho_updates = 0; factorizations = 0;
        if (s.scaling) scale_data();
        else { D.assign(n, 1.0); E.assign(m, 1.0); Dinv = D; Einv = E; c = cinv = 1.0; }
        set_rho_vec();
        x.assign(n, 0.0); z.assign(m, 0.0); y.assign(m, 0.0);
        x_prev.assign(n, 0.0); z_prev.assign(m, 0.0);
        xz_tilde.assign(n + m, 0.0);
        delta_x.assign(n, 0.0); delta_y.assign(m, 0.0);
        Ax.assign(m, 0.0); Px.assign(n, 0.0); Aty.assign(n, 0.0);
        if (!build_and_factor_kkt(true)) { status = OSQP_R_NUMERIC; return false; }
        return true;
    }

    double compute_pri_res() {  // leaves the scaled residual vector in z_prev
        mat_vec(A, x, Ax, false);
        for (int i = 0; i < m; ++i) z_prev[i] = Ax[i] - z[i];
        return s.scaling ? scaled_norm_inf(Einv, z_prev) : norm_inf(z_prev);
    }
    double compute_pri_tol(double eps_abs, double eps_rel) {
        double mx = s.scaling ? std::max(scaled_norm_inf(Einv, z), scaled_norm_inf(Einv, Ax))
                              : std::max(norm_inf(z), norm_inf(Ax));
        return eps_abs + eps_rel * mx;
    }
    double compute_dua_res() {  // leaves the scaled residual vector in x_prev
        mat_vec(P, x, Px, false);
        mat_tpose_vec(P, x, Px, true, true);
        mat_tpose_vec(A, y, Aty, false, false);
        for (int j = 0; j < n; ++j) x_prev[j] = q[j] + Px[j] + Aty[j];
        return s.scaling ? cinv * scaled_norm_inf(Dinv, x_prev) : norm_inf(x_prev);
    }
    double compute_dua_tol(double eps_abs, double eps_rel) {
        double mx;
        if (s.scaling) {
            mx = std::max(scaled_norm_inf(Dinv, q), std::max(scaled_norm_inf(Dinv, Aty), scaled_norm_inf(Dinv, Px)));
            mx *= cinv;
        } else {
            mx = std::max(norm_inf(q), std::max(norm_inf(Aty), norm_inf(Px)));
        }
        return eps_abs + eps_rel * mx;
    }
    void update_info() {
        pri_res = compute_pri_res();
        dua_res = compute_dua_res();
    }
    bool is_primal_infeasible(double eps) {
        Vec dy = delta_y;
        // project onto the polar of the recession cone of [l,u]
        for (int i = 0; i < m; ++i) {
            if (u[i] > INFTY * MIN_SCALING) {
                if (l[i] < -INFTY * MIN_SCALING) dy[i] = 0.0;
                else dy[i] = std::min(dy[i], 0.0);
            } else if (l[i] < -INFTY * MIN_SCALING) {
                dy[i] = std::max(dy[i], 0.0);
            }
        }
        double nrm = s.scaling ? scaled_norm_inf(E, dy) : norm_inf(dy);
        if (nrm > eps) {
            double lhs = 0;
            for (int i = 0; i < m; ++i) lhs += u[i] * std::max(dy[i], 0.0) + l[i] * std::min(dy[i], 0.0);
            if (lhs < -eps * nrm) {
                Vec Atdy(n, 0.0);
                mat_tpose_vec(A, dy, Atdy, false, false);
                double v = s.scaling ? scaled_norm_inf(Dinv, Atdy) : norm_inf(Atdy);
                return v < eps * nrm;
            }
        }
        return false;
    }
    bool is_dual_infeasible(double eps) {
        double nrm = s.scaling ? scaled_norm_inf(D, delta_x) : norm_inf(delta_x);
        double cost_scaling = s.scaling ? c : 1.0;
        if (nrm > eps) {
            double qdx = 0;
            for (int j = 0; j < n; ++j) qdx += q[j] * delta_x[j];
            if (qdx < -cost_scaling * eps * nrm) {
                Vec Pdx(n, 0.0);
                mat_vec(P, delta_x, Pdx, false);
                mat_tpose_vec(P, delta_x, Pdx, true, true);
                double v = s.scaling ? scaled_norm_inf(Dinv, Pdx) : norm_inf(Pdx);
                if (v < cost_scaling * eps * nrm) {
                    Vec Adx(m, 0.0);
                    mat_vec(A, delta_x, Adx, false);
                    if (s.scaling) for (int i = 0; i < m; ++i) Adx[i] *= Einv[i];
                    for (int i = 0; i < m; ++i) {
                        if ((u[i] < INFTY * MIN_SCALING && Adx[i] > eps * nrm) ||
                            (l[i] > -INFTY * MIN_SCALING && Adx[i] < -eps * nrm))
                            return false;
                    }
                    return true;
                }
            }
        }
        return false;
    }
    bool check_termination(bool approximate) {
        double eps_abs = s.eps_abs, eps_rel = s.eps_rel, epi = s.eps_prim_inf, edi = s.eps_dual_inf;
        if (approximate) { eps_abs *= 10; eps_rel *= 10; epi *= 10; edi *= 10; }
        if (!(pri_res <= INFTY) || !(dua_res <= INFTY)) { status = OSQP_R_NON_CVX; return true; }
        bool prim_ok = false, dual_ok = false, prim_inf = false, dual_inf = false;
        if (m == 0) prim_ok = true;
        else {
            double eps_prim = compute_pri_tol(eps_abs, eps_rel);
            if (pri_res < eps_prim) prim_ok = true;
            else prim_inf = is_primal_infeasible(epi);
        }
        double eps_dual = compute_dua_tol(eps_abs, eps_rel);
        if (dua_res < eps_dual) dual_ok = true;
        else dual_inf = is_dual_infeasible(edi);
        if (prim_ok && dual_ok) { status = approximate ? OSQP_R_SOLVED_INACCURATE : OSQP_R_SOLVED; return true; }
        if (prim_inf) { status = OSQP_R_PRIMAL_INFEASIBLE; return true; }
        if (dual_inf) { status = OSQP_R_DUAL_INFEASIBLE; return true; }
        return false;
    }
    double compute_rho_estimate() {
        double pr = norm_inf(z_prev), du = norm_inf(x_prev);  // scaled residual vectors
        double prn = std::max(norm_inf(z), norm_inf(Ax));
        pr /= (prn + 1e-10);
        double dun = std::max(norm_inf(q), std::max(norm_inf(Aty), norm_inf(Px)));
        du /= (dun + 1e-10);
        double est = s.rho * std::sqrt(pr / (du + 1e-10));
        return std::min(std::max(est, RHO_MIN), RHO_MAX);
    }
    bool adapt_rho() {
        double rho_new = compute_rho_estimate();
        if (rho_new > s.rho * s.adaptive_rho_tolerance || rho_new < s.rho / s.adaptive_rho_tolerance) {
            s.rho = std::min(std::max(rho_new, RHO_MIN), RHO_MAX);
            for (int i = 0; i < m; ++i) {
                if (constr_type[i] == 0) rho_vec[i] = s.rho;
                else if (constr_type[i] == 1) rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * s.rho;
                rho_inv_vec[i] = 1.0 / rho_vec[i];
            }
            rho_updates++;
            return build_and_factor_kkt(false);
        }
        return true;
    }

    // Solution polishing (OSQP paper sec. 4, Algorithm 2 / polish.c of OSQP 0.6): guess the active set from the signs of the
    // dual iterate, solve the equality-constrained QP on it through the regularised KKT system
    //     [P + delta I, A_act'; A_act, -delta I] [x; y_act] = [-q; b_act]
    // with polish_refine_iter steps of iterative refinement against the unregularised matrix, and keep the result if it
    // improves the residuals.  Works on the scaled problem, like OSQP.
    void polish() {
        std::vector<int> act;        // row -> index in the reduced system, or -1
        std::vector<double> bact;
        act.assign(m, -1);
        int mr = 0;
        for (int i = 0; i < m; ++i) {
            if (z[i] - l[i] < -y[i]) { act[i] = mr++; bact.push_back(l[i]); }       // lower-active
            else if (u[i] - z[i] < y[i]) { act[i] = mr++; bact.push_back(u[i]); }   // upper-active
        }
        const int N = n + mr;
        std::vector<int> ti, tj;
        std::vector<double> tv, tv0;  // regularised / plain values on the same pattern
        std::vector<char> has_diag(n, 0);
        for (int j = 0; j < n; ++j)
            for (int p = P.p[j]; p < P.p[j + 1]; ++p) {
                double v = P.x[p];
                ti.push_back(P.i[p]); tj.push_back(j); tv0.push_back(v);
                if (P.i[p] == j) { v += s.delta; has_diag[j] = 1; }
                tv.push_back(v);
            }
        for (int j = 0; j < n; ++j)
            if (!has_diag[j]) { ti.push_back(j); tj.push_back(j); tv.push_back(s.delta); tv0.push_back(0.0); }
        for (int j = 0; j < n; ++j)
            for (int p = A.p[j]; p < A.p[j + 1]; ++p)
                if (act[A.i[p]] >= 0) { ti.push_back(j); tj.push_back(n + act[A.i[p]]); tv.push_back(A.x[p]); tv0.push_back(A.x[p]); }
        for (int i = 0; i < mr; ++i) { ti.push_back(n + i); tj.push_back(n + i); tv.push_back(-s.delta); tv0.push_back(0.0); }
        SparseLDL pl;
        pl.analyze(N, ti, tj);
        if (!pl.factor(ti, tj, tv)) { polish_status = -1; return; }
        Vec rhs(N), sol(N), res(N);
        for (int j = 0; j < n; ++j) rhs[j] = -q[j];
        for (int i = 0; i < mr; ++i) rhs[n + i] = bact[i];
        sol = rhs;
        pl.solve(sol);
        for (int it = 0; it < s.polish_refine_iter; ++it) {
            // res = rhs - K sol, K the unregularised symmetric matrix given by its upper triplets
            res = rhs;
            for (size_t k = 0; k < ti.size(); ++k) {
                const int a = ti[k], b = tj[k];
                res[a] -= tv0[k] * sol[b];
                if (a != b) res[b] -= tv0[k] * sol[a];
            }
            pl.solve(res);
            for (int k = 0; k < N; ++k) sol[k] += res[k];
        }
        // the polished point: x, z = proj(A x), y (zero on the inactive rows)
        Vec xp(sol.begin(), sol.begin() + n), zp(m, 0.0), yp(m, 0.0);
        mat_vec(A, xp, zp, false);
        for (int i = 0; i < m; ++i) {
            zp[i] = std::min(std::max(zp[i], l[i]), u[i]);
            if (act[i] >= 0) yp[i] = sol[n + act[i]];
        }
        // its residuals, computed like those of the iterate
        Vec xs = x, zs = z, ys = y;
        const double pr0 = pri_res, du0 = dua_res;
        x = xp; z = zp; y = yp;
        pol_pri_res = compute_pri_res();
        pol_dua_res = compute_dua_res();
        const bool good = (pol_pri_res < pr0 && pol_dua_res < du0) || (pol_pri_res < pr0 && du0 < 1e-10) || (pol_dua_res < du0 && pr0 < 1e-10);
        if (good) { polish_status = 1; pri_res = pol_pri_res; dua_res = pol_dua_res; }
        else { polish_status = -1; x = xs; z = zs; y = ys; pri_res = pr0; dua_res = du0; }
    }

    // osqp_solve
    int solve() {
        if (status == OSQP_R_NUMERIC) return status;
        int iter;
        bool can_check = false;
        for (iter = 1; iter <= s.max_iter; ++iter) {
            std::swap(x, x_prev);
            std::swap(z, z_prev);
            // update_xz_tilde
            for (int j = 0; j < n; ++j) xz_tilde[j] = s.sigma * x_prev[j] - q[j];
            for (int i = 0; i < m; ++i) xz_tilde[n + i] = z_prev[i] - rho_inv_vec[i] * y[i];
            ldl.solve(xz_tilde);
            for (int i = 0; i < m; ++i) xz_tilde[n + i] = z_prev[i] + rho_inv_vec[i] * (xz_tilde[n + i] - y[i]);
            // update_x
            for (int j = 0; j < n; ++j) {
                x[j] = s.alpha * xz_tilde[j] + (1.0 - s.alpha) * x_prev[j];
                delta_x[j] = x[j] - x_prev[j];
            }
            // update_z (projection onto [l,u])
            for (int i = 0; i < m; ++i) {
                double v = s.alpha * xz_tilde[n + i] + (1.0 - s.alpha) * z_prev[i] + rho_inv_vec[i] * y[i];
                z[i] = std::min(std::max(v, l[i]), u[i]);
            }
            // update_y
            for (int i = 0; i < m; ++i) {
                delta_y[i] = rho_vec[i] * (s.alpha * xz_tilde[n + i] + (1.0 - s.alpha) * z_prev[i] - z[i]);
                y[i] += delta_y[i];
            }
            can_check = s.check_termination && (iter % s.check_termination == 0);
            if (can_check) {
                update_info();
                if (check_termination(false)) break;
            }
            if (s.adaptive_rho && s.adaptive_rho_interval && (iter % s.adaptive_rho_interval == 0)) {
                if (!can_check) update_info();
                if (!adapt_rho()) { status = OSQP_R_NUMERIC; break; }
            }
        }
        if (iter > s.max_iter) iter = s.max_iter;
        iters = iter;
        if (status == OSQP_R_UNSOLVED) {
            if (!can_check) { update_info(); check_termination(false); }
            if (status == OSQP_R_UNSOLVED && !check_termination(true)) status = OSQP_R_MAX_ITER;
        }
        polish_status = 0;
        if (s.polish && status == OSQP_R_SOLVED) polish();
        // store_solution: unscale
        x_sol.resize(n); y_sol.resize(m);
        for (int j = 0; j < n; ++j) x_sol[j] = D[j] * x[j];
        for (int i = 0; i < m; ++i) y_sol[i] = cinv * E[i] * y[i];
        return status;
    }
};

}  // namespace orc
