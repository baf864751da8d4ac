// TEST INFRASTRUCTURE ONLY — C entry points of the CPU oracle, loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
// decentralized_ekf_mhe_amd/ may link or load this library.
#include <malloc.h>

#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "ekf_oracle.hpp"
#include "est_oracle.hpp"

using namespace orc;

namespace {
EkfParams ekf_params_from(const dekf_params& p) {
    EkfParams e;
    std::memcpy(e.init_std, p.ekf_init_std, sizeof e.init_std);
    std::memcpy(e.process_std, p.ekf_process_std, sizeof e.process_std);
    std::memcpy(e.gravity_meas_std, p.ekf_gravity_meas_std, sizeof e.gravity_meas_std);
    std::memcpy(e.vo_meas_std, p.ekf_vo_meas_std, sizeof e.vo_meas_std);
    std::memcpy(e.quaternion_init, p.ekf_quaternion_init, sizeof e.quaternion_init);
    e.rate = p.ekf_rate;
    return e;
}

// EKF + estimator chained the way the three reference processes are chained through
// imu/filter: the EKF output quaternion is what the estimator reads next.
struct Pipe {
    EkfOracle ekf;
    EstOracle est;
    explicit Pipe(const dekf_params& p) : ekf(ekf_params_from(p)), est(p) {}
    void step(int T) {
        ekf.step();
        est.store.quat = ekf.quat;
        if (T == 0) est.initialize();
        else est.update(T);
    }
};

void set_leg(EstOracle* e, const double* p_foot, const double* J, const double* qdot, const double* contact) {
    int L = e->L, nj = e->nj;
    e->store.p_imu_2_foot.assign(p_foot, p_foot + 3 * L);
    for (int i = 0; i < 3 * L; ++i)
        for (int j = 0; j < nj; ++j) e->store.J_imu_2_foot(i, j) = J[i * nj + j];
    e->store.joint_velocity.assign(qdot, qdot + L * nj);
    e->store.contact.assign(contact, contact + L);
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------- Bezier (Bezier_simple.cpp:12-82)
// way points P[n][3] with their stamps t[n] are added one by one (only the last four survive), then
// set_interval(t_start, num, dt) + interpolate_waypoint(); returns the node count (0 with fewer than four points)
int orc_bezier(int n, const double* P, const double* t, double t_start, int num, double dt, double* nodes, double* distances) {
    BezierOracle c;
    for (int i = 0; i < n; ++i) c.add_way_point(Vec{P[3 * i], P[3 * i + 1], P[3 * i + 2]}, t[i]);
    c.set_interval(t_start, num, dt);
    c.interpolate_waypoint();
    for (int i = 0; i < c.node_count(); ++i)
        for (int a = 0; a < 3; ++a) { nodes[3 * i + a] = c.nodes[i][a]; distances[3 * i + a] = c.distances[i][a]; }
    return c.node_count();
}

// ---------------------------------------------------------------- EKF
void* orc_ekf_create(const dekf_params* p) { return new EkfOracle(ekf_params_from(*p)); }
void orc_ekf_destroy(void* h) { delete (EkfOracle*)h; }
void orc_ekf_set_imu(void* h, double t, const double* accel, const double* gyro) { ((EkfOracle*)h)->set_imu(t, accel, gyro); }
void orc_ekf_set_vo(void* h, double t, const double* q) { ((EkfOracle*)h)->set_vo(t, q); }
void orc_ekf_step(void* h) { ((EkfOracle*)h)->step(); }
int orc_ekf_last_replay(void* h) { return ((EkfOracle*)h)->last_replay; }
void orc_ekf_get(void* h, double* q, double* cov) {
    EkfOracle* e = (EkfOracle*)h;
    if (q) std::memcpy(q, e->quat.data(), 4 * sizeof(double));
    if (cov) std::memcpy(cov, e->Cov_q.a.data(), 16 * sizeof(double));
}
// the three public methods, stateless (orien_ekf.hpp:79-81)
void orc_ekf_predict(void* h, const double* q, const double* gyro, const double* cov, double* q_out, double* cov_out) {
    EkfOracle* e = (EkfOracle*)h;
    Mat C(4, 4), Co;
    std::memcpy(C.a.data(), cov, 16 * sizeof(double));
    Vec qo;
    e->predict(qo, Vec(q, q + 4), Vec(gyro, gyro + 3), C, Co);
    std::memcpy(q_out, qo.data(), 4 * sizeof(double));
    std::memcpy(cov_out, Co.a.data(), 16 * sizeof(double));
}
void orc_ekf_correct(void* h, const double* q, const double* accel, const double* cov, double* q_out, double* cov_out) {
    EkfOracle* e = (EkfOracle*)h;
    Mat C(4, 4), Co;
    std::memcpy(C.a.data(), cov, 16 * sizeof(double));
    Vec qo;
    e->correct(qo, Vec(q, q + 4), Vec(accel, accel + 3), C, Co);
    std::memcpy(q_out, qo.data(), 4 * sizeof(double));
    std::memcpy(cov_out, Co.a.data(), 16 * sizeof(double));
}
void orc_ekf_vo_correct(void* h, const double* q, const double* q_vo, const double* cov, double* q_out, double* cov_out) {
    EkfOracle* e = (EkfOracle*)h;
    Mat C(4, 4), Co;
    std::memcpy(C.a.data(), cov, 16 * sizeof(double));
    Vec qo;
    e->vo_correct(qo, Vec(q, q + 4), Vec(q_vo, q_vo + 4), C, Co);
    std::memcpy(q_out, qo.data(), 4 * sizeof(double));
    std::memcpy(cov_out, Co.a.data(), 16 * sizeof(double));
}

// ---------------------------------------------------------------- estimator
// test knob (densemat.hpp: marg_inverse_variant): 0 the reference's evaluation, 1 long double, 2 reversed pivot order; process-wide
void orc_set_marg_inverse_variant(int v) { marg_inverse_variant() = v; }
void* orc_est_create(const dekf_params* p) { return new EstOracle(*p); }
void orc_est_destroy(void* h) { delete (EstOracle*)h; }
void orc_est_set_imu(void* h, double t, const double* accel, const double* gyro) {
    EstOracle* e = (EstOracle*)h;
    e->store.imu_time = t;
    e->store.accel_b.assign(accel, accel + 3);
    e->store.angular_b.assign(gyro, gyro + 3);
}
void orc_est_set_quat(void* h, const double* q) { ((EstOracle*)h)->store.quat.assign(q, q + 4); }
void orc_est_set_leg(void* h, const double* p_foot, const double* J, const double* qdot, const double* contact) {
    set_leg((EstOracle*)h, p_foot, J, qdot, contact);
}
void orc_est_set_vo(void* h, double t_pre, double t_now, const double* dp) {
    EstOracle* e = (EstOracle*)h;
    e->store.vo_new = true;
    e->store.vo_time_pre = t_pre;
    e->store.vo_time_now = t_now;
    e->store.vo_p.assign(dp, dp + 3);
}
void orc_est_initialize(void* h) { ((EstOracle*)h)->initialize(); }
void orc_est_update(void* h, int T) { ((EstOracle*)h)->update(T); }
void orc_est_get(void* h, double* x, double* v_b, double* p_vo) {
    EstOracle* e = (EstOracle*)h;
    const Vec& xs = e->est_type == 0 ? e->x_MHE : e->x_KF;
    const Vec& vb = e->est_type == 0 ? e->v_MHE_b : e->v_KF_b;
    if (x && !xs.empty()) std::memcpy(x, xs.data(), xs.size() * sizeof(double));
    if (v_b) std::memcpy(v_b, vb.data(), 3 * sizeof(double));
    if (p_vo) std::memcpy(p_vo, e->p_vo_accumulate.data(), 3 * sizeof(double));
}
void orc_est_qp_dims(void* h, int* n, int* m) {
    EstOracle* e = (EstOracle*)h;
    *n = e->qp.H.r;
    *m = e->qp.A.r;
}
void orc_est_qp_copy(void* h, double* H, double* g, double* A, double* l, double* u) {
    EstOracle* e = (EstOracle*)h;
    if (H) std::memcpy(H, e->qp.H.a.data(), e->qp.H.a.size() * sizeof(double));
    if (g) std::memcpy(g, e->qp.g.data(), e->qp.g.size() * sizeof(double));
    if (A) std::memcpy(A, e->qp.A.a.data(), e->qp.A.a.size() * sizeof(double));
    if (l) std::memcpy(l, e->qp.lb.data(), e->qp.lb.size() * sizeof(double));
    if (u) std::memcpy(u, e->qp.ub.data(), e->qp.ub.size() * sizeof(double));
}
void orc_est_solution(void* h, double* x_full) {
    EstOracle* e = (EstOracle*)h;
    std::memcpy(x_full, e->qp.solution.data(), e->qp.solution.size() * sizeof(double));
}
void orc_est_solver_info(void* h, int* iters, int* status, int* rho_updates, int* factorizations,
                         double* pri_res, double* dua_res, int* nnzL, double* rho) {
    EstOracle* e = (EstOracle*)h;
    const OsqpRestate& o = e->qp.osqp;
    if (iters) *iters = o.iters;
    if (status) *status = o.status;
    if (rho_updates) *rho_updates = o.rho_updates;
    if (factorizations) *factorizations = o.factorizations;
    if (pri_res) *pri_res = o.pri_res;
    if (dua_res) *dua_res = o.dua_res;
    if (nnzL) *nnzL = o.ldl.nnzL();
    if (rho) *rho = o.s.rho;
}
void orc_est_polish_info(void* h, int* status, double* pri_res, double* dua_res) {
    const OsqpRestate& o = ((EstOracle*)h)->qp.osqp;
    if (status) *status = o.polish_status;
    if (pri_res) *pri_res = o.pol_pri_res;
    if (dua_res) *dua_res = o.pol_dua_res;
}
void orc_est_scaling(void* h, double* D, double* E, double* c) {
    const OsqpRestate& o = ((EstOracle*)h)->qp.osqp;
    if (D) std::memcpy(D, o.D.data(), o.D.size() * sizeof(double));
    if (E) std::memcpy(E, o.E.data(), o.E.size() * sizeof(double));
    if (c) *c = o.c;
}
void orc_est_arrival(void* h, double* M, double* n) {
    EstOracle* e = (EstOracle*)h;
    if (M && e->qp.M_p.r) std::memcpy(M, e->qp.M_p.a.data(), e->qp.M_p.a.size() * sizeof(double));
    if (n && !e->qp.n_p.empty()) std::memcpy(n, e->qp.n_p.data(), e->qp.n_p.size() * sizeof(double));
}
void orc_est_kf_cov(void* h, double* C) {
    EstOracle* e = (EstOracle*)h;
    std::memcpy(C, e->C_KF.a.data(), e->C_KF.a.size() * sizeof(double));
}
void orc_est_rotation(void* h, double* R) { std::memcpy(R, ((EstOracle*)h)->R_sb.a.data(), 9 * sizeof(double)); }

// ---------------------------------------------------------------- EKF -> estimator pipeline
void* orc_pipe_create(const dekf_params* p) { return new Pipe(*p); }
void orc_pipe_destroy(void* h) { delete (Pipe*)h; }
void orc_pipe_set_imu(void* h, double t, const double* accel, const double* gyro) {
    Pipe* p = (Pipe*)h;
    p->ekf.set_imu(t, accel, gyro);
    orc_est_set_imu(&p->est, t, accel, gyro);
}
void orc_pipe_set_leg(void* h, const double* p_foot, const double* J, const double* qdot, const double* contact) {
    set_leg(&((Pipe*)h)->est, p_foot, J, qdot, contact);
}
void orc_pipe_set_vo(void* h, double t_pre, double t_now, const double* dp, double t_pose, const double* q_vo) {
    Pipe* p = (Pipe*)h;
    orc_est_set_vo(&p->est, t_pre, t_now, dp);
    p->ekf.set_vo(t_pose, q_vo);
}
void orc_pipe_step(void* h, int T) { ((Pipe*)h)->step(T); }
void* orc_pipe_est(void* h) { return &((Pipe*)h)->est; }
void* orc_pipe_ekf(void* h) { return &((Pipe*)h)->ekf; }

// Run whole sensor logs for `B` instances over steps T = 0..nsteps-1 on `nthreads` host
// threads (instances partitioned contiguously).  Logs are [nsteps][B][...] row-major.
// Outputs x_out[nsteps][B][ns], vb_out[nsteps][B][3], quat_out[nsteps][B][4],
// iters_out[nsteps][B] (any may be NULL).  Returns wall seconds of the stepping loop.
double orc_pipe_run(const dekf_params* prm, int B, int nsteps, int nthreads, const double* imu_t,
                    const double* accel, const double* gyro, const double* p_foot, const double* J,
                    const double* qdot, const double* contact, const int* vo_mask, const double* vo_t_pre,
                    const double* vo_t_now, const double* vo_dp, const double* vo_t_pose,
                    const double* vo_q, double* x_out, double* vb_out, double* quat_out, int* iters_out) {
    int L = prm->num_legs, nj = prm->joints_per_leg;
    int ns = 9 + 3 * prm->leg_odom_type * L;
    // the dense H / A copies are MB-sized: keep them on the per-thread malloc arenas instead of
    // one mmap/munmap per copy, which serialises all threads on the process's mm lock
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > B) nthreads = B;
    auto worker = [&](int b0, int b1) {
        for (int b = b0; b < b1; ++b) {
            Pipe pipe(*prm);
            for (int T = 0; T < nsteps; ++T) {
                size_t o = (size_t)T * B + b;
                orc_pipe_set_imu(&pipe, imu_t[o], accel + 3 * o, gyro + 3 * o);
                set_leg(&pipe.est, p_foot + 3 * L * o, J + 3 * L * nj * o, qdot + L * nj * o, contact + L * o);
                if (vo_mask && vo_mask[o])
                    orc_pipe_set_vo(&pipe, vo_t_pre[o], vo_t_now[o], vo_dp + 3 * o, vo_t_pose[o], vo_q + 4 * o);
                pipe.step(T);
                if (x_out) {
                    double* xo = x_out + (size_t)ns * o;
                    const Vec& xs = pipe.est.est_type == 0 ? pipe.est.x_MHE : pipe.est.x_KF;
                    for (int i = 0; i < ns; ++i) xo[i] = xs.empty() ? 0.0 : xs[i];
                }
                if (vb_out) {
                    const Vec& vb = pipe.est.est_type == 0 ? pipe.est.v_MHE_b : pipe.est.v_KF_b;
                    std::memcpy(vb_out + 3 * o, vb.data(), 3 * sizeof(double));
                }
                if (quat_out) std::memcpy(quat_out + 4 * o, pipe.ekf.quat.data(), 4 * sizeof(double));
                if (iters_out) iters_out[o] = (T > 0 && pipe.est.est_type == 0) ? pipe.est.qp.osqp.iters : 0;
            }
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (nthreads == 1) worker(0, B);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) {
            int b0 = (int)((long)B * t / nthreads), b1 = (int)((long)B * (t + 1) / nthreads);
            th.emplace_back(worker, b0, b1);
        }
        for (auto& t : th) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
