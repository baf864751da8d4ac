// host_common.h — host-side pieces shared by the C ABI (dekf_capi.hip) and the test-only
// lane-sequential build (tests/hostsim): parameter validation, DevCfg fill, state allocation.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>

#include "cfg.h"
#include "ekf_core.h"

namespace dekf {

inline void default_params(dekf_params* p) {
    // src/go1_example/config/parameters_go1.yaml of the reference + OSQP defaults
    std::memset(p, 0, sizeof(*p));
    auto set3 = [](double* a, double x, double y, double z) { a[0] = x; a[1] = y; a[2] = z; };
    set3(p->p_init_std, 0.001, 0.001, 0.001);
    set3(p->v_init_std, 0.001, 0.001, 0.001);
    set3(p->foot_init_std, 0.001, 0.001, 0.001);
    set3(p->accel_bias_init_std, 0.0001, 0.0001, 0.0001);
    set3(p->p_process_std, 0.001, 0.001, 0.001);
    set3(p->accel_input_std, 0.025, 0.025, 0.02);
    set3(p->gyro_input_std, 0.03, 0.03, 0.03);
    set3(p->accel_bias_std, 0.07, 0.02, 0.03);
    p->quaternion_ib[0] = 1.0;
    set3(p->p_ib, 0.01592, 0.06659, 0.00617);
    p->num_legs = 4; p->joints_per_leg = 3; p->leg_odom_type = 0;
    for (int i = 0; i < DEKF_MAX_JOINTS; ++i) { p->joint_position_std[i] = 0.04; p->joint_velocity_std[i] = 0.22; }
    set3(p->foot_slide_std, 0.003, 0.003, 0.003);
    set3(p->foot_swing_std, 1.0e7, 1.0e7, 1.0e7);
    p->contact_effort_threshold = 150.0;
    set3(p->vo_p_std, 0.000015, 0.000015, 0.000015);
    p->rate = 200; p->N = 20; p->est_type = 0;
    p->rho = 0.1; p->alpha = 1.6; p->delta = 1e-5; p->sigma = 1e-5;
    p->verbose = 0; p->adapt_rho = 1; p->polish = 0; p->max_qp_iter = 4000;
    p->rel_tol = p->abs_tol = p->prim_tol = p->dual_tol = 1e-6;
    p->time_limit = 0.0028;
    p->scaling_iters = 10; p->check_termination = 25; p->adaptive_rho_interval = 25;
    p->adaptive_rho_tolerance = 5.0;
    for (int i = 0; i < 4; ++i) { p->ekf_init_std[i] = 0.001; p->ekf_vo_meas_std[i] = 0.0001; }
    set3(p->ekf_process_std, 0.1, 0.1, 0.1);
    set3(p->ekf_gravity_meas_std, 4.0, 4.0, 4.0);
    p->ekf_quaternion_init[0] = 1.0;
    p->ekf_rate = 500; p->ekf_history = 256;
    p->polish_refine_iter = 3;
    p->arrival_cost_form = 0; p->solve_pipeline = 0; p->solve_workgroups_per_cu = 0; p->polish_accept_osqp = 0;
}

// returns nullptr when ok, else a message
inline const char* fill_cfg(const dekf_params& p, int B, DevCfg& c) {
    if (B < 1) return "batch must be >= 1";
    if (p.num_legs < 1 || p.num_legs > DEKF_MAX_LEGS) return "num_legs out of range";
    if (p.joints_per_leg < 1 || p.joints_per_leg > DEKF_MAX_JOINTS) return "joints_per_leg out of range";
    if (p.leg_odom_type != 0 && p.leg_odom_type != 1) return "leg_odom_type must be 0 (foot velocity) or 1 (foot position)";
    if (p.est_type != 0 && p.est_type != 1) return "est_type must be 0 (MHE) or 1 (KF)";
    if (p.N < 2 || p.N > 128) return "N out of range [2,128]";
    if (p.rate < 1 || p.ekf_rate < 1) return "rate must be positive";
    if (p.ekf_history < 4) return "ekf_history must be >= 4";
    if (p.polish != 0 && p.polish != 1) return "osqp.polish must be 0 or 1";
    if (p.polish && !(p.delta > 0)) return "osqp.delta must be positive when osqp.polish is on";
    // the polishing solve runs as an ADMM step with 1 / rho = delta on the equality rows, and rho lives in [RHO_MIN, RHO_MAX] * 1e3:
    // outside this range the device would silently solve a differently regularised system than OSQP does
    if (p.polish && !(p.delta >= 1e-9 && p.delta <= 1e3)) return "osqp.delta must lie in [1e-9, 1e3] when osqp.polish is on";
    if (p.polish_refine_iter < 0 || p.polish_refine_iter > 100) return "polish_refine_iter out of range [0,100]";
    if (p.arrival_cost_form != 0 && p.arrival_cost_form != 1) return "arrival_cost_form must be 0 (reference form) or 1 (information form)";
    if (p.solve_pipeline != 0 && p.solve_pipeline != 1) return "solve_pipeline must be 0 (in order) or 1 (consecutive steps overlap)";
    if (p.solve_workgroups_per_cu < 0 || p.solve_workgroups_per_cu > 8) return "solve_workgroups_per_cu out of range [0,8]";
    if (p.leg_odom_type == 1) {  // these become gains 1 / std^2 (DecentralEst.cpp:47-51): a zero would be an infinite weight
        for (int i = 0; i < 3; ++i)
            if (!(p.foot_slide_std[i] > 0) || !(p.foot_init_std[i] > 0) || !(p.foot_swing_std[i] > 0))
                return "leg_odom_type 1 needs positive foot_slide_std, foot_init_std and foot_swing_std";
    }
    if (!(p.sigma > 0) || !(p.rho > 0) || !(p.alpha > 0 && p.alpha < 2)) return "rho/sigma/alpha out of range";
    std::memset(&c, 0, sizeof(c));
    c.B = B; c.L = p.num_legs; c.nj = p.joints_per_leg; c.N = p.N; c.nm = 3 * p.num_legs;
    c.ft = p.leg_odom_type; c.ns = 9 + 3 * c.ft * c.L;
    c.SV = 2 * c.ns + c.nm + 3; c.SC = c.nm + c.ns + 3;
    // window records: N + 1 slots (the N of a window and the one the next step fills); one more when steps are pipelined, because
    // the term construction of step T then runs while the solve of step T - 2 may still be reading its window (dekf_capi.hip)
    c.ring = 4 * p.N + 1; c.wcap = p.N + 1 + ((p.solve_pipeline == 1 && p.est_type == 0) ? 1 : 0); c.rec = Rec::len(c.L, c.ft);
    c.snap_len = c.ns * c.ns + c.ns + 4 * c.wcap;
    c.est_type = p.est_type;
    c.marg_info = p.leg_odom_type == 1 && p.arrival_cost_form == 1;
    c.dt = 1.0 / (double)p.rate;
    c.hdt2 = 0.5 * c.dt * c.dt;
    c.gws_wt = 0;  // (dekf_create sets it once the placement of the factor is known)
    c.inf_thr = OSQP_INFTY * MIN_SCALING;
    auto sq = [](double v) { return v * v; };
    for (int i = 0; i < 3; ++i) {
        c.C_p[i] = sq(p.p_process_std[i]);
        c.C_accel[i] = sq(p.accel_input_std[i]);
        c.C_accel_bias[i] = sq(p.accel_bias_std[i]);
        c.C_gyro[i] = sq(p.gyro_input_std[i]);
        c.C_swing[i] = sq(p.foot_swing_std[i]);
        c.Q_swing[i] = 1.0 / sq(p.foot_swing_std[i]);
        c.C_slide[i] = sq(p.foot_slide_std[i]);
        c.Q_slide[i] = 1.0 / sq(p.foot_slide_std[i]);
        c.C_foot_init[i] = sq(p.foot_init_std[i]);
        c.Q_foot_init[i] = 1.0 / sq(p.foot_init_std[i]);
        c.Q_vo[i] = 1.0 / sq(p.vo_p_std[i]);
        c.Q_bias_dt2[i] = 1 / (c.dt * c.dt) * (1.0 / sq(p.accel_bias_std[i]));
        c.Q_prior[i] = 1.0 / sq(p.p_init_std[i]);
        c.Q_prior[3 + i] = 1.0 / sq(p.v_init_std[i]);
        c.Q_prior[6 + i] = 1.0 / sq(p.accel_bias_init_std[i]);
        c.C_prior[i] = sq(p.p_init_std[i]);
        c.C_prior[3 + i] = sq(p.v_init_std[i]);
        c.C_prior[6 + i] = sq(p.accel_bias_init_std[i]);
        c.ekf_Cgyro[i] = sq(p.ekf_process_std[i]);
        c.ekf_Caccel[i] = sq(p.ekf_gravity_meas_std[i]);
    }
    for (int i = 0; i < c.nj; ++i) { c.C_enc_pos[i] = sq(p.joint_position_std[i]); c.C_enc_vel[i] = sq(p.joint_velocity_std[i]); }
    c.rho0 = p.rho; c.rho0c = p.rho < RHO_MIN ? RHO_MIN : (p.rho > RHO_MAX ? RHO_MAX : p.rho); c.sigma = p.sigma; c.alpha = p.alpha; c.eps_abs = p.abs_tol; c.eps_rel = p.rel_tol;
    c.max_iter = p.max_qp_iter; c.scaling = p.scaling_iters; c.check_termination = p.check_termination;
    c.adaptive_rho = p.adapt_rho; c.adaptive_rho_interval = p.adaptive_rho_interval;
    c.adaptive_rho_tolerance = p.adaptive_rho_tolerance;
    c.polish = p.polish; c.polish_refine_iter = p.polish_refine_iter; c.delta = p.delta;
    c.polish_accept_osqp = p.polish_accept_osqp != 0;
    c.ekf_dt = 1.0 / (double)p.ekf_rate;
    for (int i = 0; i < 4; ++i) {
        c.ekf_Cvo[i] = sq(p.ekf_vo_meas_std[i]);
        c.ekf_P0[i] = sq(p.ekf_init_std[i]);
        c.ekf_q0[i] = p.ekf_quaternion_init[i];
    }
    c.ekf_hist = p.ekf_history;
    return nullptr;
}

// every persistent array: alloc(bytes) must return zero-filled memory
// `copies` (1 or 2): sets of the per-solve data (input snapshot, outputs, solver scratch, section stamps); set 1 follows set 0
// in each allocation (second_set() moves a DevState's pointers over)
constexpr int DEKF_SNAP_SETS = 3;  // input snapshots of a pipelined handle (by T mod 3; outputs and solver scratch: by T mod 2)
template <class Alloc>
inline void alloc_state(const DevCfg& c, DevState& s, int solve_slots, Alloc alloc, int copies = 1) {
    const size_t B = (size_t)c.B, L = (size_t)c.L, nj = (size_t)c.nj;
    auto D = [&](size_t n) { return (double*)alloc(n * sizeof(double)); };
    auto I = [&](size_t n) { return (int*)alloc(n * sizeof(int)); };
    s.imu_t = D(B); s.accel = D(3 * B); s.gyro = D(3 * B);
    s.p_foot = D(3 * L * B); s.J = D(3 * L * nj * B); s.qdot = D(L * nj * B); s.contact = D(L * B);
    s.quat = D(4 * B);
    s.vo_flag = I(B); s.vo_tpre = D(B); s.vo_tnow = D(B); s.vo_dp = D(3 * B);
    s.ekf_vo_flag = I(B); s.ekf_vo_t = D(B); s.ekf_vo_q = D(4 * B);
    s.ekf_q = D(4 * B); s.ekf_P = D(16 * B); s.ekf_hist = D((size_t)c.ekf_hist * EKF_HIST_REC * B);
    s.st_time = D((size_t)c.ring * B); s.st_R = D((size_t)c.ring * 9 * B); s.st_dtime = I((size_t)c.ring * B);
    const size_t ns = (size_t)c.ns;
    s.rec = D((size_t)c.wcap * c.rec * B); s.Mp = D(ns * ns * B); s.np_ = D(ns * B);
    s.Mp_next = D(ns * ns * B); s.np_next = D(ns * B); s.marg_tag = I(B);
    s.snap = D((size_t)(copies > 1 ? DEKF_SNAP_SETS : 1) * c.snap_len * B);
    s.wp = D(12 * B); s.wpt = D(4 * B); s.wp_count = I(B);
    s.p_vo = D(3 * B); s.vo_ins_idx = I(B); s.vo_ins_dtime = I(B);
    Gws g;
    g.init(c.N, c.L, c.ft, c.gws_wt);
    const size_t cp = (size_t)copies;
    s.gws = D(cp * solve_slots * g.total);
    s.queue = I(cp * 2);
    s.kf_x = D(ns * B); s.kf_C = D(ns * ns * B);
    s.x_mhe = D(cp * ns * B); s.v_b = D(cp * 3 * B);
    s.status = I(cp * B); s.iters = I(cp * B); s.rho_updates = I(cp * B); s.polish_status = I(cp * B);
    s.pri_res = D(cp * B); s.dua_res = D(cp * B);
    s.prof = D(cp * DEKF_PROF_SLOTS * B);
}
inline DevState second_set(const DevCfg& c, const DevState& s, int solve_slots) {
    const size_t B = (size_t)c.B, ns = (size_t)c.ns;
    Gws g;
    g.init(c.N, c.L, c.ft, c.gws_wt);
    DevState t = s;
    // (snap: one of DEKF_SNAP_SETS copies by T mod 3, chosen per launch — dekf_update)
    t.gws += (size_t)solve_slots * g.total;
    t.queue += 2;
    t.x_mhe += ns * B; t.v_b += 3 * B;
    t.status += B; t.iters += B; t.rho_updates += B; t.polish_status += B;
    t.pri_res += B; t.dua_res += B;
    t.prof += DEKF_PROF_SLOTS * B;
    return t;
}

}  // namespace dekf
