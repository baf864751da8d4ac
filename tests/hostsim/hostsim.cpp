// TEST INFRASTRUCTURE ONLY.  Lane-sequential g++ build (-DDEKF_HOSTSIM) of the exact device
// cores in decentralized_ekf_mhe_amd/csrc/*_core.h, so their arithmetic and indexing can be
// checked against the oracle (and run under sanitizers) in a container without a GPU.
// It is not linked into libdekf.so, not reachable from include/dekf.h, and not a fallback:
// the product fails with DEKF_ERR_NO_DEVICE when there is no GPU.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../decentralized_ekf_mhe_amd/csrc/cfg.h"
#include "../../decentralized_ekf_mhe_amd/csrc/ekf_core.h"
#include "../../decentralized_ekf_mhe_amd/csrc/host_common.h"
#include "../../decentralized_ekf_mhe_amd/csrc/kf_core.h"
#include "../../decentralized_ekf_mhe_amd/csrc/mhe_assemble_core.h"
#include "../../decentralized_ekf_mhe_amd/csrc/mhe_solve_core.h"

using namespace dekf;

struct Sim {
    DevCfg c;
    DevState s;
    std::vector<void*> blocks;
    std::vector<double> lds;
    int ekf_count = 0, pushes = 0;
    int gws_len = 0;
};

extern "C" {

void* hs_create(const dekf_params* p, int B) {
    Sim* h = new Sim();
    if (fill_cfg(*p, B, h->c)) { delete h; return nullptr; }
    alloc_state(h->c, h->s, 1, [&](size_t bytes) { void* q = std::calloc(1, bytes ? bytes : 8); h->blocks.push_back(q); return q; });
    Gws g; g.init(h->c.N, h->c.L, h->c.ft);
    h->gws_len = g.total;
    SolveLayout lay; lay.init(h->c.N, h->c.L, h->c.ft);
    int n = (int)(lay.lds_bytes() / 8);
    int a = AsmScratch::len(h->c.L, h->c.ft), k = KfScratch::len(h->c.L, h->c.ft);
    h->lds.assign((size_t)std::max(n, std::max(a, k)), 0.0);
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < 4; ++i) { h->s.ekf_q[(size_t)i * B + b] = h->c.ekf_q0[i]; h->s.quat[4 * b + i] = h->c.ekf_q0[i]; }
        for (int i = 0; i < 4; ++i) h->s.ekf_P[(size_t)(5 * i) * B + b] = h->c.ekf_P0[i];
    }
    return h;
}
void hs_destroy(void* hv) {
    Sim* h = (Sim*)hv;
    for (void* q : h->blocks) std::free(q);
    delete h;
}
void hs_push_imu(void* hv, const double* t, const double* accel, const double* gyro) {
    Sim* h = (Sim*)hv; size_t B = h->c.B;
    std::memcpy(h->s.imu_t, t, B * 8); std::memcpy(h->s.accel, accel, 3 * B * 8); std::memcpy(h->s.gyro, gyro, 3 * B * 8);
}
void hs_push_leg(void* hv, const double* p_foot, const double* J, const double* qdot, const double* contact) {
    Sim* h = (Sim*)hv; size_t B = h->c.B, L = h->c.L, nj = h->c.nj;
    std::memcpy(h->s.p_foot, p_foot, 3 * L * B * 8); std::memcpy(h->s.J, J, 3 * L * nj * B * 8);
    std::memcpy(h->s.qdot, qdot, L * nj * B * 8); std::memcpy(h->s.contact, contact, L * B * 8);
}
void hs_push_vo(void* hv, const int* mask, const double* t_pre, const double* t_now, const double* dp,
                const double* t_pose, const double* q_vo) {
    Sim* h = (Sim*)hv;
    for (int b = 0; b < h->c.B; ++b) {
        if (!mask[b]) continue;
        h->s.vo_flag[b] = 1; h->s.vo_tpre[b] = t_pre[b]; h->s.vo_tnow[b] = t_now[b];
        for (int i = 0; i < 3; ++i) h->s.vo_dp[3 * b + i] = dp[3 * b + i];
        if (q_vo) {
            h->s.ekf_vo_flag[b] = 1; h->s.ekf_vo_t[b] = t_pose[b];
            for (int i = 0; i < 4; ++i) h->s.ekf_vo_q[4 * b + i] = q_vo[4 * b + i];
        }
    }
}
void hs_push_quat(void* hv, const double* q) { Sim* h = (Sim*)hv; std::memcpy(h->s.quat, q, 4 * (size_t)h->c.B * 8); }
void hs_ekf_step(void* hv) {
    Sim* h = (Sim*)hv;
    for (int b = 0; b < h->c.B; ++b) ekf_tick(h->c, h->s, b, h->ekf_count);
    h->ekf_count++;
}
void hs_initialize(void* hv) {
    Sim* h = (Sim*)hv;
    for (int b = 0; b < h->c.B; ++b) {
        if (h->c.est_type == 0) assemble_initialize(h->c, h->s, b, h->lds.data());
        else kf_initialize(h->c, h->s, b, h->lds.data());
    }
    h->pushes = h->c.est_type == 0 ? 1 : 2;
}
void hs_update(void* hv, int T) {
    Sim* h = (Sim*)hv;
    for (int b = 0; b < h->c.B; ++b) {
        if (h->c.est_type == 0) {
            assemble_update(h->c, h->s, b, T, h->pushes, h->lds.data());
            int kstart = T - h->c.N + 1 > 0 ? T - h->c.N + 1 : 0;
            SolveLayout lay; lay.init(h->c.N, h->c.L, h->c.ft);
            int K = T - kstart + 1;
#define HS_SOLVE(LEGS)                                                                                      \
    if (lay.pa_in_lds()) solve_window_t<true, LEGS, true, true>(h->c, h->s, b, kstart, K, h->lds.data(), h->s.gws); \
    else if (lay.factor_in_lds()) solve_window_t<true, LEGS, true, false>(h->c, h->s, b, kstart, K, h->lds.data(), h->s.gws); \
    else solve_window_t<true, LEGS, false, false>(h->c, h->s, b, kstart, K, h->lds.data(), h->s.gws);
#define HS_SOLVE_FOOT(LEGS)                                                                                              \
    if (lay.factor_in_lds()) solve_window_t<true, LEGS, true, false, 0, 1>(h->c, h->s, b, kstart, K, h->lds.data(), h->s.gws);     \
    else solve_window_t<true, LEGS, false, false, 0, 1>(h->c, h->s, b, kstart, K, h->lds.data(), h->s.gws);
            if (h->c.ft) {
                switch (h->c.L) {
                    case 1: HS_SOLVE_FOOT(1) break;
                    case 2: HS_SOLVE_FOOT(2) break;
                    case 3: HS_SOLVE_FOOT(3) break;
                    default: HS_SOLVE_FOOT(4) break;
                }
            } else
            switch (h->c.L) {
                case 1: HS_SOLVE(1) break;
                case 2: HS_SOLVE(2) break;
                case 3: HS_SOLVE(3) break;
                default: HS_SOLVE(4) break;
            }
#undef HS_SOLVE
#undef HS_SOLVE_FOOT
        } else {
            kf_update(h->c, h->s, b, h->pushes, h->lds.data());
        }
    }
    h->pushes++;
}
void hs_get(void* hv, double* x, double* vb, double* quat, double* p_vo, int* status, int* iters, int* rho_updates) {
    Sim* h = (Sim*)hv; size_t B = h->c.B;
    if (x) std::memcpy(x, h->s.x_mhe, (size_t)h->c.ns * B * 8);
    if (vb) std::memcpy(vb, h->s.v_b, 3 * B * 8);
    if (quat) std::memcpy(quat, h->s.quat, 4 * B * 8);
    if (p_vo) std::memcpy(p_vo, h->s.p_vo, 3 * B * 8);
    if (status) std::memcpy(status, h->s.status, B * 4);
    if (iters) std::memcpy(iters, h->s.iters, B * 4);
    if (rho_updates) std::memcpy(rho_updates, h->s.rho_updates, B * 4);
}
void hs_get_polish_status(void* hv, int* st) {
    Sim* h = (Sim*)hv;
    std::memcpy(st, h->s.polish_status, (size_t)h->c.B * 4);
}
void hs_get_residuals(void* hv, double* pri, double* dua) {
    Sim* h = (Sim*)hv;
    std::memcpy(pri, h->s.pri_res, (size_t)h->c.B * 8);
    std::memcpy(dua, h->s.dua_res, (size_t)h->c.B * 8);
}
void hs_get_ekf_cov(void* hv, double* P) {
    Sim* h = (Sim*)hv; size_t B = h->c.B;
    for (size_t b = 0; b < B; ++b) for (int i = 0; i < 16; ++i) P[16 * b + i] = h->s.ekf_P[(size_t)i * B + b];
}
void hs_get_arrival(void* hv, double* M, double* n) {
    Sim* h = (Sim*)hv; size_t B = h->c.B;
    size_t ns = h->c.ns;
    std::memcpy(M, h->s.Mp, ns * ns * B * 8); std::memcpy(n, h->s.np_, ns * B * 8);
}
// scaling vectors of the LAST instance solved (scratch slab 0): D[n], E[m]
void hs_get_scaling(void* hv, int n, int m, double* D, double* E) {
    Sim* h = (Sim*)hv;
    Gws g; g.init(h->c.N, h->c.L, h->c.ft);
    SolveLayout lay; lay.init(h->c.N, h->c.L, h->c.ft);
    const double* Dp = h->s.gws + g.D;
    const double* Ep = h->s.gws + g.E;
    if (lay.factor_in_lds()) {  // same carve order as solve_window
        Dp = h->lds.data() + 2 * lay.n_pad + 4 * lay.m_pad + 2 * lay.ns * h->c.N + solve_tmp_len(lay.ns);
        Ep = Dp + lay.n_pad;
    }
    std::memcpy(D, Dp, (size_t)n * 8); std::memcpy(E, Ep, (size_t)m * 8);
}
}  // extern "C"

#include "../../decentralized_ekf_mhe_amd/csrc/go1_kin.h"
extern "C" {
// Go1 leg odometry front-end (go1_kin.h) for n joint vectors jp[n][12] -> p[n][12] (with p_ib), J[n][36]
void hs_go1_fk(int n, const double* jp, const double* p_ib, double* p_out, double* J_out) {
    for (int i = 0; i < n; ++i)
        for (int leg = 0; leg < 4; ++leg) {
            double p[3], J[9];
            go1_leg_fk(leg, jp + 12 * i + 3 * leg, p, J);
            for (int a = 0; a < 3; ++a) p_out[12 * i + 3 * leg + a] = p[a] + p_ib[a];
            for (int a = 0; a < 9; ++a) J_out[36 * i + 9 * leg + a] = J[a];
        }
}
void hs_push_go1_joints(void* hv, const double* jp, const double* jv, const double* force, double thr, const double* p_ib) {
    Sim* h = (Sim*)hv;
    for (int b = 0; b < h->c.B; ++b) go1_leg_odometry(h->s, b, jp, jv, force, thr, p_ib);
}
}
