// TEST INFRASTRUCTURE ONLY — part of the CPU oracle (see oracle/README.md).
//
// fp64, single-instance restatement of the decentral_legged_est estimator core:
//   DecentralizedEstimation::initialize        src/decentral_legged_est/src/DecentralEst.cpp:9-150
//   ::update                                   :152-198
//   ::InitializeMHE / ::UpdateMHE              :200-351 / :353-585
//   ::InitializeKF / ::UpdateKF                :592-700 / :702-861
//   ::GetMeasurement                           :864-985
//   ::UpdateVOConstraints                      :987-1009
//   MHEproblem::updateQP                       src/decentral_legged_est/src/MheSrb.cpp:351-447
//   MHEproblem::marginalizeQP                  :475-713
//   MHEproblem::initQP/solveQP/getsolution     :272-349, :715-723   (OSQP: osqp_restate.hpp)
//   Bezier                                     src/decentral_legged_est/src/Spline/Bezier_simple.cpp:12-82
// The QP (H, g, A, l, u) is kept as generic dense matrices and grown / sliced exactly like
// the reference's sparse ones, so this file knows nothing about the block structure the HIP
// kernels exploit.  Legs / joints-per-leg are parameters (the reference hard-codes 4 x 3).
// parity: unpinned by the reference (no tests); pins: tests/test_oracle_mhe.py.
#pragma once
#include <algorithm>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../include/dekf.h"
#include "densemat.hpp"
#include "osqp_restate.hpp"

namespace orc {

// ---- Bezier_simple.cpp ---------------------------------------------------------------
struct BezierOracle {
    std::vector<Vec> way_points;
    std::vector<double> way_times;
    std::vector<Vec> distances, nodes;
    double t_interval = 0, t_start = 0, u_inc = 0, num = 0;
    Vec node_pre{0, 0, 0};

    void add_way_point(const Vec& p, double t_end) {
        way_points.push_back(p);
        way_times.push_back(t_end);
        if (way_points.size() > 4) {
            way_points.erase(way_points.begin());
            way_times.erase(way_times.begin());
        }
        t_interval = way_times.back() - way_times.front();
    }
    void set_interval(double t_interpolate_start, int interpolate_num, double dt) {
        t_start = t_interpolate_start;
        u_inc = dt / t_interval;
        num = interpolate_num;
        node_pre = Vec{0, 0, 0};
    }
    static Vec cubic(double u, const Vec& P0, const Vec& P1, const Vec& P2, const Vec& P3) {
        Vec pt(3);
        for (int a = 0; a < 3; ++a) {
            double v = u * u * u * ((-1) * P0[a] + 3 * P1[a] - 3 * P2[a] + P3[a]);
            v += u * u * (3 * P0[a] - 6 * P1[a] + 3 * P2[a]);
            v += u * ((-3) * P0[a] + 3 * P1[a]);
            v += P0[a];
            pt[a] = v;
        }
        return pt;
    }
    void interpolate_waypoint() {
        distances.clear();
        nodes.clear();
        if (way_points.size() < 4) return;
        int pt = (int)way_points.size() - 4;
        double u0 = (t_start - way_times.front()) / t_interval;
        for (double i = 0; i < num; i++) {
            double u = u0 + u_inc * i;
            Vec node = cubic(u, way_points[pt], way_points[pt + 1], way_points[pt + 2], way_points[pt + 3]);
            distances.push_back(node - node_pre);
            node_pre = node;
            nodes.push_back(node);
        }
    }
    int node_count() const { return (int)nodes.size(); }
};

// ---- MheSrb.cpp bookkeeping ----------------------------------------------------------
typedef std::pair<char, int> Key;  // (kind, discrete time)

struct CostItemO {
    Vec b;
    Mat Q;
    std::map<Key, Mat> dep;
};
struct ConstraintO {
    Vec lb, ub;
    std::map<Key, Mat> dep;
    bool equality = false;
};

class QpBook {
  public:
    int N = 0, ns = 0, nm = 0, nc = 0;
    int nVar = 0, nVarStart = 0, nVarEnd = 0;
    int nCon = 0, nCon_new = 0;
    std::map<Key, int> var_idx;
    std::map<Key, CostItemO> cost, cost_new;
    std::map<Key, ConstraintO> con, con_new;
    std::vector<Key> con_new_order;
    Mat H, A;
    Vec g, lb, ub;
    Mat M_p;
    Vec n_p;
    OsqpSettings settings;
    OsqpRestate osqp;
    Vec solution, x_solution_now;
    bool solver_ready = false;

    void set_horizon(int N_, int ns_, int nm_, int nc_) { N = N_; ns = ns_; nm = nm_; nc = nc_; }

    void add_variable(Key k, int size) {
        var_idx[k] = nVarEnd;
        nVarEnd += size;
        nVar = nVarEnd - nVarStart;
    }
    void add_cost(Key k, const Vec& b, const Mat& Q) {
        CostItemO c;
        c.b = b; c.Q = Q;
        cost_new[k] = c;
    }
    void add_cost_dep(Key k, Key var, const Mat& Ax) {
        if (cost_new.count(k) && var_idx.count(var)) cost_new[k].dep[var] = Ax;
        else std::fprintf(stderr, "[oracle] cost dependency on unknown cost/variable\n");
    }
    void add_constraint(Key k, const Vec& l, const Vec& u) {
        nCon += (int)u.size();
        nCon_new += (int)u.size();
        ConstraintO c;
        c.lb = l; c.ub = u;
        con_new[k] = c;
        con_new_order.push_back(k);
    }
    void add_constraint_dep(Key k, Key var, const Mat& Ax) {
        if (con_new.count(k) && var_idx.count(var)) con_new[k].dep[var] = Ax;
        else std::fprintf(stderr, "[oracle] constraint dependency on unknown constraint/variable\n");
    }
    void update_constraint_bound(Key k, const Vec& l, const Vec& u, bool equality) {
        if (con.count(k)) { con[k].lb = l; con[k].ub = u; con[k].equality = equality; }
        else std::fprintf(stderr, "[oracle] bound update on unknown constraint (%c,%d)\n", k.first, k.second);
    }
    static void resize_vec(Vec& v, int n) { v.resize(n, 0.0); }

    // append the blocks registered since the last call (MheSrb.cpp:351-447)
    void update_qp() {
        H.conservative_resize(nVar, nVar);
        resize_vec(g, nVar);
        for (auto& kv : cost_new) {
            cost[kv.first] = kv.second;
            const CostItemO& c = kv.second;
            for (auto& di : c.dep) {
                int oi = var_idx[di.first] - nVarStart;
                for (auto& dj : c.dep) {
                    int oj = var_idx[dj.first] - nVarStart;
                    H.add_block(oi, oj, di.second.T() * c.Q * dj.second);
                }
                Vec gi = -(di.second.T() * (c.Q * c.b));
                for (size_t e = 0; e < gi.size(); ++e) g[oi + e] += gi[e];
            }
        }
        cost_new.clear();
        A.conservative_resize(nCon, nVar);
        resize_vec(lb, nCon);
        resize_vec(ub, nCon);
        int row = nCon - nCon_new;
        for (const Key& k : con_new_order) {
            const ConstraintO& c = con_new[k];
            con[k] = c;
            set_segment(lb, row, c.lb);
            set_segment(ub, row, c.ub);
            for (auto& d : c.dep) A.add_block(row, var_idx[d.first] - nVarStart, d.second);
            row += (int)c.lb.size();
        }
        con_new.clear();
        con_new_order.clear();
        nCon_new = 0;
    }

    // bound rewrite on already assembled VO rows (MheSrb.cpp:449-473)
    void update_image_bound(const std::map<int, Vec>& rows) {
        for (auto& kv : rows)
            for (int a = 0; a < 3; ++a) { lb[kv.first + a] = -kv.second[a]; ub[kv.first + a] = -kv.second[a]; }
    }

    // drop step T from the window, fold it into the arrival cost (MheSrb.cpp:475-713)
    void marginalize(int T) {
        Key kx{'x', T}, kv{'v', T}, kw{'w', T}, kc{'c', T};
        Key meas{'M', T}, dyn{'D', T}, cam{'V', T}, prior{'P', 0};
        int vsz = ns + ns + nc + nm;
        var_idx.erase(kx); var_idx.erase(kv); var_idx.erase(kw); var_idx.erase(kc);
        nVarStart += vsz;
        nVar = nVarEnd - nVarStart;
        Mat H_new = H.block(vsz, vsz, H.r - vsz, H.c - vsz);
        Vec g_new = segment(g, vsz, (int)g.size() - vsz);
        if (cost.count(dyn)) {
            if (cost.count(prior)) {
                M_p = cost[prior].Q;
                n_p = -(M_p * cost[prior].b);
                cost.erase(prior);
            }
            // (test knob, densemat.hpp: variants >= 3 take the three SPD inverses that feed S in another elimination order — S then
            // differs from the reference's in its last bits, which is what any other implementation's does)
            const bool alt_in = marg_inverse_variant() >= 3;
            auto spd_inv = [&](const Mat& X) {
                if (!alt_in) return spd_inverse(X);
                // the same Cholesky solve in the opposite elimination order, on the symmetrised matrix (one triangle must stay the
                // only source: fed from both triangles of M — e.g. through a pivoted LU — the recursion amplifies their rounding
                // asymmetry and diverges within three swing phases)
                const int n = X.r;
                Mat Rv(n, n);
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) Rv(i, j) = X(n - 1 - (i < j ? i : j), n - 1 - (i < j ? j : i));  // LOWER triangle of X (row >= column), reversed
                Mat Ri = spd_inverse(Rv), Xi(n, n);
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) Xi(i, j) = Ri(n - 1 - i, n - 1 - j);
                return Xi;
            };
            Mat M_inv = spd_inv(M_p);
            const Mat& R_meas = cost[meas].Q;
            const Mat& H_meas = con[meas].dep[kx];
            const Vec& y_meas = con[meas].lb;
            const Mat& Q_dyn = cost[dyn].Q;
            const Mat& A_dyn = con[dyn].dep[kx];
            Vec b_dyn = -con[dyn].lb;
            Mat R_meas_inv = spd_inv(R_meas);
            int na;          // rows of the stacked [dyn; cam] block
            Mat A_m, Q_m_inv;
            Vec b_m;
            if (con[cam].equality) {
                const Mat& Q_cam = cost[cam].Q;
                const Mat& A_cam = con[cam].dep[kx];
                Vec b_cam = -con[cam].lb;
                na = ns + nc;
                A_m = Mat(na, ns);
                A_m.set_block(0, 0, A_dyn);
                A_m.set_block(ns, 0, A_cam);
                Mat Q_m(na, na);
                Q_m.set_block(0, 0, Q_dyn);
                Q_m.set_block(ns, ns, Q_cam);
                b_m = Vec(na, 0.0);
                set_segment(b_m, 0, b_dyn);
                set_segment(b_m, ns, b_cam);
                Q_m_inv = spd_inv(Q_m);
            } else {
                na = ns;
                A_m = A_dyn;
                b_m = b_dyn;
                Q_m_inv = spd_inv(Q_dyn);
            }
            int dim = na + nm;
            Mat S(dim, dim);
            Mat S11 = -(A_m * M_inv * A_m.T());
            S11 = S11 - Q_m_inv;
            Mat S22 = -(H_meas * M_inv * H_meas.T());
            S22 = S22 - R_meas_inv;
            Mat S12 = -(A_m * M_inv * H_meas.T());
            S.set_block(0, 0, S11);
            S.set_block(na, na, S22);
            S.set_block(0, na, S12);
            S.set_block(na, 0, S12.T());
            Mat B(dim, ns);
            for (int i = 0; i < ns; ++i) B(i, i) = -1.0;
            if (na > ns)
                for (int i = 0; i < nc; ++i) B(ns + i, i) = -1.0;
            Mat C = B.T();
            Mat S_inv = inverse_marg(S);  // = inverse(S) unless a test turned the knob (densemat.hpp)
            Vec u(dim, 0.0);
            set_segment(u, 0, -b_m + A_m * (M_inv * n_p));
            set_segment(u, na, y_meas + H_meas * (M_inv * n_p));
            Mat M_next = -(C * S_inv * B);
            Vec n_next = C * (S_inv * u);
            M_p = M_next;
            n_p = n_next;
            H = Mat(nVar, nVar);
            H.add_block(0, 0, H_new);
            H.add_block(0, 0, M_p);
            g = g_new;
            for (int i = 0; i < ns; ++i) g[i] += n_p[i];
            cost.erase(dyn); cost.erase(cam); cost.erase(meas);
        } else {
            std::fprintf(stderr, "[oracle] marginalize: missing dynamic cost at %d\n", T);
        }
        if (nCon != 0) {
            int csz = (int)(con[dyn].lb.size() + con[cam].lb.size() + con[meas].lb.size());
            nCon -= csz;
            A = A.block(csz, vsz, A.r - csz, A.c - vsz);
            lb = segment(lb, csz, (int)lb.size() - csz);
            ub = segment(ub, csz, (int)ub.size() - csz);
            con.erase(dyn); con.erase(cam); con.erase(meas);
        }
    }

    bool init_qp() {
        Csc Pc = csc_from_dense(H, true);
        Csc Ac = csc_from_dense(A, false);
        solver_ready = osqp.setup(Pc, g, Ac, lb, ub, settings);
        return solver_ready;
    }
    void solve_qp() {
        osqp.solve();
        solution = osqp.x_sol;
    }
    void get_solution(int T) {
        int o = var_idx[Key{'x', T}] - nVarStart;
        x_solution_now = segment(solution, o, ns);
    }
    void reset() {
        nVar = 0; nCon = 0;
        var_idx.clear(); con.clear(); cost.clear(); cost_new.clear();
    }
};

// ---- DecentralEst.cpp ----------------------------------------------------------------
struct RobotStoreO {  // robot_store (DecentralEst.hpp:65-94)
    double imu_time = 0;
    Vec accel_b{0, 0, 0}, angular_b{0, 0, 0};
    Vec joint_velocity;  // L*nj
    Vec contact;         // L
    Vec p_imu_2_foot;    // 3L
    Mat J_imu_2_foot;    // 3L x nj
    double vo_time_pre = 0, vo_time_now = 0;
    bool vo_new = false;
    Vec vo_p{0, 0, 0};
    Vec quat{1, 0, 0, 0};  // w x y z
};

class EstOracle {
  public:
    dekf_params prm;
    RobotStoreO store;
    QpBook qp;
    BezierOracle curve;
    int L, nj, type, ns, nm, nc, N, est_type;
    double dt;
    Vec gravity{0, 0, -9.81};
    Mat C_p, C_accel, C_accel_bias, C_foot_slide, C_foot_swing, C_gyro;
    Mat C_enc_pos, C_enc_vel;  // nj x nj
    Mat Q_accel_bias, Q_foot_slide, Q_foot_swing, Q_vo_p;
    Mat Q_prior, A_meas, I_dyn, I_meas, I_cam, A_cam;
    Vec x_prior;
    // stacks
    std::vector<int> dtime_stack;
    std::vector<double> time_stack;
    std::vector<Vec> accel_s_stack, angular_stack, p_foot_stack, qdot_stack, contact_stack;
    std::vector<Mat> R_stack, J_stack;
    std::vector<int> vo_insert_idx, vo_insert_dtime;
    bool vo_to_be_processed = false;
    // public results
    Mat R_sb;
    Vec p_vo_accumulate{0, 0, 0};
    Vec x_MHE, v_MHE_b{0, 0, 0};
    Vec x_KF, v_KF_b{0, 0, 0};
    Mat C_KF, K_KF;
    bool initialized = false;

    static Mat cov3(const double* s) { Mat m(3, 3); for (int i = 0; i < 3; ++i) m(i, i) = s[i] * s[i]; return m; }
    static Mat gain3(const double* s) { Mat m(3, 3); for (int i = 0; i < 3; ++i) m(i, i) = 1.0 / (s[i] * s[i]); return m; }

    explicit EstOracle(const dekf_params& p) : prm(p) {
        L = p.num_legs; nj = p.joints_per_leg; type = p.leg_odom_type;
        est_type = p.est_type; N = p.N; dt = 1.0 / p.rate;
        ns = 9 + 3 * type * L; nm = 3 * L; nc = 3;
        store.joint_velocity.assign(L * nj, 0.0);
        store.contact.assign(L, 0.0);
        store.p_imu_2_foot.assign(3 * L, 0.0);
        store.J_imu_2_foot = Mat(3 * L, nj);
    }

    void initialize() {
        qp = QpBook();
        qp.set_horizon(N, ns, nm, nc);
        C_p = cov3(prm.p_process_std); C_accel = cov3(prm.accel_input_std);
        C_accel_bias = cov3(prm.accel_bias_std); C_gyro = cov3(prm.gyro_input_std);
        C_foot_slide = cov3(prm.foot_slide_std); C_foot_swing = cov3(prm.foot_swing_std);
        C_enc_pos = Mat(nj, nj); C_enc_vel = Mat(nj, nj);
        for (int i = 0; i < nj; ++i) {
            C_enc_pos(i, i) = prm.joint_position_std[i] * prm.joint_position_std[i];
            C_enc_vel(i, i) = prm.joint_velocity_std[i] * prm.joint_velocity_std[i];
        }
        Q_accel_bias = gain3(prm.accel_bias_std); Q_foot_slide = gain3(prm.foot_slide_std);
        Q_foot_swing = gain3(prm.foot_swing_std); Q_vo_p = gain3(prm.vo_p_std);
        Q_prior = Mat(ns, ns);
        x_prior = Vec(ns, 0.0);
        I_dyn = Mat::identity(ns); I_meas = Mat::identity(nm); I_cam = Mat::identity(nc);
        A_meas = Mat(nm, ns);
        for (int i = 0; i < L; ++i) {
            if (type == 0) {
                for (int a = 0; a < 3; ++a) A_meas(3 * i + a, 3 + a) = 1.0;
            } else {
                for (int a = 0; a < 3; ++a) { A_meas(3 * i + a, a) = -1.0; A_meas(3 * i + a, 9 + 3 * i + a) = 1.0; }
            }
        }
        A_cam = Mat(nc, ns);
        for (int a = 0; a < 3; ++a) A_cam(a, a) = 1.0;
        if (est_type == 0) initialize_mhe();
        else { initialize_kf(); update_kf(); }
        initialized = true;
    }

    Mat leg_J(int i) const { return J_stack.back().block(3 * i, 0, 3, nj); }
    Vec leg_p(int i) const { return segment(p_foot_stack.back(), 3 * i, 3); }
    Vec leg_qdot(int i) const { return segment(qdot_stack.back(), nj * i, nj); }

    // covariance of the foot-velocity pseudo-measurement of leg i, world frame
    Mat leg_velocity_cov(int i, const Mat& R) const {
        Mat J = leg_J(i);
        Mat G(3, 2 * nj + 3);
        G.set_block(0, 0, -J);
        G.set_block(0, nj, -(skew3(angular_stack.back()) * J));
        G.set_block(0, 2 * nj, skew3(leg_p(i)));
        Mat Cm(2 * nj + 3, 2 * nj + 3);
        Cm.set_block(0, 0, C_enc_vel);
        Cm.set_block(nj, nj, C_enc_pos);
        Cm.set_block(2 * nj, 2 * nj, C_gyro);
        return R * G * Cm * G.T() * R.T();
    }
    // b_meas and (gain or covariance) of the measurement at the newest stack entry
    void measurement_terms(Vec& b_meas, Mat& W, bool as_gain) const {
        const Mat& R = R_stack.back();
        b_meas = Vec(nm, 0.0);
        W = Mat(nm, nm);
        for (int i = 0; i < L; ++i) {
            if (type == 0) {
                Vec b = -(R * (leg_J(i) * leg_qdot(i))) - R * cross3(angular_stack.back(), leg_p(i));
                set_segment(b_meas, 3 * i, b);
                if (contact_stack.back()[i] == 0.0) {
                    W.set_block(3 * i, 3 * i, as_gain ? Q_foot_swing : C_foot_swing);
                } else {
                    Mat Cm = leg_velocity_cov(i, R);
                    W.set_block(3 * i, 3 * i, as_gain ? inverse(Cm) : Cm);
                }
            } else {
                set_segment(b_meas, 3 * i, R * leg_p(i));
                Mat JCJ = leg_J(i) * C_enc_pos * leg_J(i).T();
                if (as_gain) W.set_block(3 * i, 3 * i, R * inverse(JCJ) * R.T());
                else W.set_block(3 * i, 3 * i, R * JCJ * R.T());
            }
        }
    }

    void initialize_mhe() {
        qp.settings.rho = prm.rho; qp.settings.sigma = prm.sigma; qp.settings.alpha = prm.alpha;
        qp.settings.eps_abs = prm.abs_tol; qp.settings.eps_rel = prm.rel_tol;
        qp.settings.eps_prim_inf = prm.prim_tol; qp.settings.eps_dual_inf = prm.dual_tol;
        qp.settings.max_iter = prm.max_qp_iter; qp.settings.adaptive_rho = prm.adapt_rho;
        qp.settings.scaling = prm.scaling_iters; qp.settings.check_termination = prm.check_termination;
        qp.settings.adaptive_rho_interval = prm.adaptive_rho_interval;
        qp.settings.adaptive_rho_tolerance = prm.adaptive_rho_tolerance;
        qp.settings.polish = prm.polish; qp.settings.delta = prm.delta;  // DecentralEst.cpp:207,211
        qp.settings.polish_refine_iter = prm.polish_refine_iter;
        get_measurement(0);
        Q_prior = Mat(ns, ns);
        Q_prior.set_block(0, 0, gain3(prm.p_init_std));
        Q_prior.set_block(3, 3, gain3(prm.v_init_std));
        Q_prior.set_block(6, 6, gain3(prm.accel_bias_init_std));
        Vec b_meas; Mat Q_meas;
        measurement_terms(b_meas, Q_meas, true);
        if (type == 1)
            for (int i = 0; i < L; ++i) {
                set_segment(x_prior, 9 + 3 * i, segment(b_meas, 3 * i, 3));
                Q_prior.set_block(9 + 3 * i, 9 + 3 * i, gain3(prm.foot_init_std));
            }
        qp.add_variable({'x', 0}, ns);
        qp.add_cost({'P', 0}, x_prior, Q_prior);
        qp.add_cost_dep({'P', 0}, {'x', 0}, I_dyn);
        qp.add_variable({'v', 0}, nm);
        qp.add_constraint({'M', 0}, b_meas, b_meas);
        qp.add_constraint_dep({'M', 0}, {'x', 0}, A_meas);
        qp.add_constraint_dep({'M', 0}, {'v', 0}, -I_meas);
        qp.add_cost({'M', 0}, Vec(nm, 0.0), Q_meas);
        qp.add_cost_dep({'M', 0}, {'v', 0}, I_meas);
        qp.update_qp();
    }

    void dynamics_terms(Mat& A_dyn, Vec& b_dyn, const Mat& R, const Vec& accel_s) const {
        b_dyn = Vec(ns, 0.0);
        set_segment(b_dyn, 0, (-dt * dt / 2) * accel_s);
        set_segment(b_dyn, 3, (-dt) * accel_s);
        A_dyn = Mat::identity(ns);
        A_dyn.set_block(0, 3, dt * Mat::identity(3));
        A_dyn.set_block(0, 6, (-dt * dt / 2) * R);
        A_dyn.set_block(3, 6, (-dt) * R);
    }

    void update_mhe(int T) {
        qp.add_variable({'w', T - 1}, ns);
        qp.add_variable({'c', T - 1}, nc);
        qp.add_variable({'x', T}, ns);
        qp.add_variable({'v', T}, nm);
        Mat R = R_stack.back();
        Vec accel_s = accel_s_stack.back();
        Mat A_dyn; Vec b_dyn;
        dynamics_terms(A_dyn, b_dyn, R, accel_s);
        Mat Q_dyn(ns, ns);
        Mat G_pv(6, 6);
        G_pv.set_block(0, 0, dt * R);
        G_pv.set_block(0, 3, (0.5 * dt * dt) * R);
        G_pv.set_block(3, 3, dt * R);
        Mat C_pv(6, 6);
        C_pv.set_block(0, 0, C_p);
        C_pv.set_block(3, 3, C_accel);
        Q_dyn.set_block(0, 0, inverse(G_pv * C_pv * G_pv.T()));
        Q_dyn.set_block(6, 6, (1 / (dt * dt)) * Q_accel_bias);
        if (type == 1)
            for (int i = 0; i < L; ++i) {
                const Mat& Qf = contact_stack.back()[i] ? Q_foot_slide : Q_foot_swing;
                Q_dyn.set_block(9 + 3 * i, 9 + 3 * i, (1 / (dt * dt)) * (R * Qf * R.T()));
            }
        qp.add_constraint({'D', T - 1}, b_dyn, b_dyn);
        qp.add_constraint_dep({'D', T - 1}, {'w', T - 1}, -I_dyn);
        qp.add_constraint_dep({'D', T - 1}, {'x', T}, -I_dyn);
        qp.add_constraint_dep({'D', T - 1}, {'x', T - 1}, A_dyn);
        qp.add_cost({'D', T - 1}, Vec(ns, 0.0), Q_dyn);
        qp.add_cost_dep({'D', T - 1}, {'w', T - 1}, I_dyn);
        // camera placeholder: unbounded until a VO interval covers this step
        Vec b_cam(nc, OsqpRestate::INFTY);
        Mat Q_cam = R * Q_vo_p * R.T();
        qp.add_constraint({'V', T - 1}, -b_cam, b_cam);
        qp.add_constraint_dep({'V', T - 1}, {'x', T - 1}, A_cam);
        qp.add_constraint_dep({'V', T - 1}, {'x', T}, -A_cam);
        qp.add_constraint_dep({'V', T - 1}, {'c', T - 1}, -I_cam);
        qp.add_cost({'V', T - 1}, Vec(nc, 0.0), Q_cam);
        qp.add_cost_dep({'V', T - 1}, {'c', T - 1}, I_cam);

        get_measurement(T);

        Vec b_meas; Mat Q_meas;
        measurement_terms(b_meas, Q_meas, true);
        qp.add_constraint({'M', T}, b_meas, b_meas);
        qp.add_constraint_dep({'M', T}, {'x', T}, A_meas);
        qp.add_constraint_dep({'M', T}, {'v', T}, -I_meas);
        qp.add_cost({'M', T}, Vec(nm, 0.0), Q_meas);
        qp.add_cost_dep({'M', T}, {'v', T}, I_meas);
        qp.update_qp();
    }

    void initialize_kf() {
        get_measurement(0);
        Mat C_prior(ns, ns);
        C_prior.set_block(0, 0, cov3(prm.p_init_std));
        C_prior.set_block(3, 3, cov3(prm.v_init_std));
        C_prior.set_block(6, 6, cov3(prm.accel_bias_init_std));
        Vec b_meas; Mat C_meas;
        measurement_terms(b_meas, C_meas, false);
        if (type == 1) {
            for (int i = 0; i < L; ++i) C_prior.set_block(9 + 3 * i, 9 + 3 * i, cov3(prm.foot_init_std));
            set_segment(x_prior, 9, b_meas);
        }
        x_KF = x_prior;
        C_KF = C_prior;
        kf_correct(b_meas, C_meas);
    }
    void kf_correct(const Vec& b_meas, const Mat& C_meas) {
        K_KF = C_KF * A_meas.T() * inverse(A_meas * C_KF * A_meas.T() + C_meas);
        x_KF = x_KF + K_KF * (b_meas - A_meas * x_KF);
        C_KF = (Mat::identity(ns) - K_KF * A_meas) * C_KF;
    }
    void update_kf() {
        Mat R = R_stack.back();
        Vec accel_s = accel_s_stack.back();
        Mat A_dyn; Vec b_dyn;
        dynamics_terms(A_dyn, b_dyn, R, accel_s);
        Mat G(ns, ns);
        G.set_block(0, 0, dt * R);
        G.set_block(0, 3, (-0.5 * dt * dt) * R);
        G.set_block(3, 3, (-dt) * R);
        G.set_block(6, 6, dt * Mat::identity(3));
        Mat C_in(ns, ns);
        C_in.set_block(0, 0, C_p);
        C_in.set_block(3, 3, C_accel);
        C_in.set_block(6, 6, C_accel_bias);
        if (type == 1)
            for (int i = 0; i < L; ++i) {
                G.set_block(9 + 3 * i, 9 + 3 * i, dt * R);
                C_in.set_block(9 + 3 * i, 9 + 3 * i, contact_stack.back()[i] == 0.0 ? C_foot_swing : C_foot_slide);
            }
        x_KF = A_dyn * x_KF - b_dyn;
        C_KF = A_dyn * C_KF * A_dyn.T() + G * C_in * G.T();
        get_measurement(0);
        Vec b_meas; Mat C_meas;
        measurement_terms(b_meas, C_meas, false);
        kf_correct(b_meas, C_meas);
    }

    void get_measurement(int T) {
        R_sb = quat_to_rot_normalized(store.quat[0], store.quat[1], store.quat[2], store.quat[3]);
        double imu_time = store.imu_time;
        Vec accel_s = R_sb * store.accel_b + gravity;
        if (store.vo_new && !time_stack.empty()) {
            Vec vo_p = store.vo_p;
            double t_pre = store.vo_time_pre, t_now = store.vo_time_now;
            store.vo_new = false;
            auto it_pre = std::upper_bound(time_stack.begin(), time_stack.end(), t_pre);
            if (it_pre != time_stack.begin()) {
                int idx_pre = (int)(it_pre - time_stack.begin()) - 1;
                Mat R_pre = R_stack[idx_pre];
                auto it_now = std::upper_bound(time_stack.begin(), time_stack.end(), t_now);
                int idx_now = (int)(it_now - time_stack.begin()) - 1;
                p_vo_accumulate = p_vo_accumulate + R_pre * vo_p;
                int win_start = (int)time_stack.size() - std::min(N, T);
                int interp_start = std::max(win_start, idx_pre);
                double t_interp_start = time_stack[interp_start];
                int dtime_interp_start = dtime_stack[interp_start];
                curve.add_way_point(p_vo_accumulate, t_now);
                if (idx_now > win_start && curve.way_points.size() >= 4) {
                    int insert_rel = interp_start - win_start;
                    int num = idx_now - interp_start + 1;
                    curve.set_interval(t_interp_start, num, dt);
                    curve.interpolate_waypoint();
                    vo_insert_idx.push_back(insert_rel);
                    vo_insert_dtime.push_back(dtime_interp_start);
                    vo_to_be_processed = true;
                }
            }
        }
        time_stack.push_back(imu_time);
        dtime_stack.push_back(T);
        R_stack.push_back(R_sb);
        accel_s_stack.push_back(accel_s);
        p_foot_stack.push_back(store.p_imu_2_foot);
        J_stack.push_back(store.J_imu_2_foot);
        contact_stack.push_back(store.contact);
        qdot_stack.push_back(store.joint_velocity);
        angular_stack.push_back(store.angular_b);
        if ((int)accel_s_stack.size() > 4 * N + 1) {
            dtime_stack.erase(dtime_stack.begin());
            R_stack.erase(R_stack.begin());
            accel_s_stack.erase(accel_s_stack.begin());
            p_foot_stack.erase(p_foot_stack.begin());
            J_stack.erase(J_stack.begin());
            contact_stack.erase(contact_stack.begin());
            time_stack.erase(time_stack.begin());
            qdot_stack.erase(qdot_stack.begin());
            angular_stack.erase(angular_stack.begin());
        }
        if ((int)vo_insert_idx.size() >= N + 1) {
            vo_insert_idx.erase(vo_insert_idx.begin());
            vo_insert_dtime.erase(vo_insert_dtime.begin());
        }
    }

    void update_vo_constraints() {
        std::map<int, Vec> rows;
        int blk = nm + ns + nc;
        for (int i = 0; i < curve.node_count() - 1; ++i) {
            const Vec& d = curve.distances[i + 1];
            int idx = (vo_insert_idx.back() + i) * blk + nm + ns;
            rows[idx] = d;
            qp.update_constraint_bound({'V', vo_insert_dtime.back() + i}, -d, -d, true);
        }
        qp.update_image_bound(rows);
    }

    void update(int T) {
        static const double p_imu_2_opti[3] = {0.016041, 0.089061, 0.0579875};
        Vec p_opti(p_imu_2_opti, p_imu_2_opti + 3);
        if (est_type == 0) {
            update_mhe(T);
            if (vo_to_be_processed) {
                update_vo_constraints();
                vo_to_be_processed = false;
            }
            if (T >= N) qp.marginalize(T - N);
            qp.init_qp();
            qp.solve_qp();
            qp.get_solution(T);
            x_MHE = qp.x_solution_now;
            v_MHE_b = R_stack.back() * (segment(x_MHE, 3, 3) + cross3(angular_stack.back(), p_opti));
        } else {
            update_kf();
            v_KF_b = R_stack.back() * (segment(x_KF, 3, 3) + cross3(angular_stack.back(), p_opti));
        }
    }
};

}  // namespace orc
