// cfg.h — constants of one estimator handle, passed by value to every kernel, and the
// layout of the per-instance state in HBM.
//
// HBM layout (all fp64 unless noted), B = instances on this GPU:
//   sensor latch   [B][...] instance-major, exactly what dekf_push_* receives
//   EKF state      [field][B] field-major (one thread per instance -> coalesced)
//   EKF history    [slot][27][B]   ring of (gyro3, accel3, t, q4, P16)
//   MHE stack      [B][ring]        imu time, R (9) per pushed sample; ring = 4N+1
//                                   (the reference trims its stacks to 4N+1, DecentralEst.cpp:963)
//   MHE window     [B][N+1][REC]    one record per window step (see Rec below)
//   arrival cost   [B][ns^2 + ns]   M_p, n_p (MheSrb.cpp:594-598), ns = 9 (21 for Go1 with foot-position states)
//   solver scratch [B][GWS]         factor + scaled data of the ADMM solve (streamed from L2)
// One wavefront owns one instance in the MHE kernels, so instance-major keeps every
// access of a wave inside one contiguous slab.
#pragma once
#include "../../include/dekf.h"
#include "wave.h"

namespace dekf {

#ifndef DEKF_SOLVE_THREADS
#define DEKF_SOLVE_THREADS 256  // lanes of the workgroup that solves one instance (4 wavefronts)
#endif

#ifndef DEKF_QUEUE_MODE
#define DEKF_QUEUE_MODE 2  // the solve kernels' instance queue (kernels.hip: DEKF_QUEUE_LOOP); 0: static grid-stride, A/B builds only
#endif
#define DEKF_R4_THREADS 192  // the four-per-CU kernels' workgroup: the solve wavefront + two workers (mhe_admm_core.h: admm_chunk_r4)

constexpr int DEKF_PROF_SLOTS = 32;  // section stamps per instance, diagnostic build only (DevState::prof)
constexpr double OSQP_INFTY = 1e30;
constexpr double RHO_MIN = 1e-6, RHO_MAX = 1e6, RHO_EQ_OVER_RHO_INEQ = 1e3, RHO_TOL = 1e-4;
constexpr double MIN_SCALING = 1e-4, MAX_SCALING = 1e4;

struct DevCfg {
    int B, L, nj, N, nm;  // nm = 3L
    int ft;               // leg_odom_type: 0 foot-velocity measurements (9 states), 1 foot positions as states
    int ns;               // dim_state = 9 + 3 L ft (DecentralEst.cpp:20)
    int SV, SC;           // per-step variable / row block: 2 ns + nm + 3, nm + ns + 3
    int ring;             // 4N+1 stack entries
    int wcap;             // window records: N + 1 (N + 2 for a pipelined handle)
    int rec;              // doubles per window record
    int snap_len;         // doubles per instance of the solve's input snapshot: M_p (ns^2) | n_p (ns) | VO flag + bound of every ring slot (4 wcap)
    int est_type;
    int marg_info;        // leg_odom_type 1: fold a step into the arrival cost in information form (dekf_params.arrival_cost_form)
    double dt;
    // Uniform constants the solve kernels would otherwise COMPUTE (f64 arithmetic is vector-only: the result sits in a VGPR pair,
    // gets hoisted to the top of the kernel and stays live across the whole solve — spilled at 168 VGPRs) or could not encode as
    // an instruction literal; as kernel arguments they arrive in SGPRs: dt^2 / 2, the clamped initial rho, OSQP's "infinite bound"
    // threshold OSQP_INFTY * MIN_SCALING (rho_of)
    double hdt2, rho0c, inf_thr;
    int gws_wt;  // the solver slabs carry the transposed W blocks (Gws::WT): 9-state shapes whose factor streams from the slab
    // estimator constants (DecentralEst.cpp:39-51, 236-253)
    double C_p[3], C_accel[3], C_accel_bias[3], C_gyro[3];
    double C_enc_pos[DEKF_MAX_JOINTS], C_enc_vel[DEKF_MAX_JOINTS];
    double C_swing[3], Q_swing[3], Q_vo[3], Q_bias_dt2[3];
    double C_slide[3], Q_slide[3];                 // foot_slide_std (leg_odom_type 1: process noise of a stance foot)
    double Q_prior[9], C_prior[9];
    double Q_foot_init[3], C_foot_init[3];         // prior of the foot-position states (DecentralEst.cpp:310-325)
    // OSQP settings (DecentralEst.cpp:204-217 + defaults)
    double rho0, sigma, alpha, eps_abs, eps_rel;
    int max_iter, scaling, check_termination, adaptive_rho, adaptive_rho_interval;
    double adaptive_rho_tolerance;
    int polish, polish_refine_iter;  // osqp.polish (DecentralEst.cpp:207), OSQP's polish_refine_iter (default 3)
    int polish_accept_osqp;          // 1: polish.c's acceptance test verbatim (dekf_params.polish_accept_osqp); 0: third clause symmetric
    double delta;                    // osqp.delta (DecentralEst.cpp:211)
    // EKF (orien_ekf.cpp:13-31)
    double ekf_dt, ekf_Cgyro[3], ekf_Caccel[3], ekf_Cvo[4], ekf_P0[4], ekf_q0[4];
    int ekf_hist;
};

// argument block of k_latch4: up to four double arrays copied by one launch; end[i] = running element count
struct LatchCopy4 {
    double* dst[4];
    const double* src[4];
    size_t end[4];
};

// window record of step k (doubles)
struct Rec {
    // R_sb (9) | a_s (3) | gyro (3) | Qd 6x6 sym packed (21) | Qc 3x3 sym packed (6) |
    // vo flag (1) | vo bound (3) | bm (nm) | Qm L x sym packed (6L) |
    // leg_odom_type 1 only: Qf L x sym packed (6L), the process gain 1/dt^2 R Q_{slide|swing} R' of every foot
    // position (DecentralEst.cpp:432-451; in KF mode the covariance dt^2 R C R', :753)
    static constexpr int R = 0, AS = 9, GY = 12, QD = 15, QC = 36, VOF = 42, VOB = 43, BM = 46;
    DEKF_HD static int qm(int nm) { return BM + nm; }
    DEKF_HD static int qf(int nm) { return BM + nm + 2 * nm; }
    DEKF_HD static int len(int L, int ft = 0) { return BM + 3 * L + 6 * L + (ft ? 6 * L : 0); }
};

// packed symmetric index, i <= j, n x n
DEKF_FN int symidx(int i, int j, int n) { return i * n - (i * (i - 1)) / 2 + (j - i); }
// branch-free for either order of (i, j): lo * (2n - 1 - lo) / 2 + hi
template <class P>  // P: const double* or the checked pointer of the -DDEKF_BOUNDS build
DEKF_FN double symget(P s, int i, int j, int n) {
    int lo = i < j ? i : j, hi = i < j ? j : i;
    return s[(lo * (2 * n - 1 - lo)) / 2 + hi];
}

// variable / row indices of the QP in the reference's own order
// (x_k v_k w_k c_k per step; Meas_k Dyn_k VO_k per step — SURVEY.md Appendix A)
struct Idx {
    int nm, SV, SC;
    DEKF_FN int x(int k, int j) const { return k * SV + j; }
    DEKF_FN int v(int k, int r) const { return k * SV + 9 + r; }
    DEKF_FN int w(int k, int r) const { return k * SV + 9 + nm + r; }
    DEKF_FN int c(int k, int a) const { return k * SV + 18 + nm + a; }
    DEKF_FN int rm(int k, int r) const { return k * SC + r; }
    DEKF_FN int rd(int k, int r) const { return k * SC + nm + r; }
    DEKF_FN int rv(int k, int a) const { return k * SC + nm + 9 + a; }
};

// global-memory scratch of one solve (doubles), K_max = N steps, NS = 9 + 3 L ft states per step
struct Gws {
    // D (n) | E (m) | lo (m) | hi (m) | rho (m) | Sv (K*6L) | Sw (K*25) | Sc (K*6) | Sf (K*6L ft) |
    // Wm (K*6L) | Wd (K*24) | Wc (K*6) | Wf (K*6L ft) | PA (K*NS^2) | Sinv (K*NS^2) | Wk (K*NS^2) |
    // x (n) | z (m) | y (m) | zt (m) | cf (m): the row-phase state between chunks of iterations, three-workgroup placement only |
    // pol (2 K NS + 5 m): osqp.polish — the polished point and the bounds while a refinement step solves for a correction, and
    // the KKT residual vectors that step starts from (mhe_solve_core.h: polish_swap_in / polish_accumulate) |
    // WT (K*81, LAST and only with `wt`: 9-state shapes whose factor streams from here keep the W blocks once more, transposed, for the
    // outward legs.  Behind everything else, so that every other offset — and, for the LDS-resident shapes, the slab itself — is what it was
    // without it: inserted behind Wk it moved the regions after it by 13.6 KB and the dominant Go1 kernel's fetch traffic up by 12 %;
    // padding the END of the slab does nothing of the kind, profiles/r04_slab_pad_sweep.txt)
    int n_pad, m_pad, K;
    int D, E, lo, hi, rho, Sv, Sw, Sc, Sf, Wm, Wd, Wc, Wf, PA, Sinv, Wk, WT, x, z, y, zt, cf, pol, total;
    DEKF_HD void init(int N, int L, int ft = 0, int wt = 0) {
        K = N;
        const int nm = 3 * L, ns = 9 + (ft ? nm : 0), b2 = ns * ns;
        n_pad = N * (2 * ns + nm + 3);
        m_pad = N * (nm + ns + 3);
        int o = 0;
        D = o; o += n_pad;
        E = o; o += m_pad;
        lo = o; o += m_pad;
        hi = o; o += m_pad;
        rho = o; o += m_pad;
        Sv = o; o += K * 6 * L;
        Sw = o; o += K * 25;  // SWS (mhe_solve_core.h)
        Sc = o; o += K * 6;
        Sf = o; o += ft ? K * 6 * L : 0;
        Wm = o; o += K * 6 * L;
        Wd = o; o += K * 24;
        Wc = o; o += K * 6;
        Wf = o; o += ft ? K * 6 * L : 0;
        PA = o; o += K * b2;
        Sinv = o; o += K * b2;
        Wk = o; o += K * b2;
        x = o; o += n_pad;
        z = o; o += m_pad;
        y = o; o += m_pad;
        zt = o; o += m_pad;
        cf = o; o += m_pad;
        pol = o; o += 2 * K * ns + 5 * m_pad;
        WT = o; o += wt && !ft ? K * b2 : 0;
        // slack behind the last array, only where the factor streams from here (`wt`): the run-time chain of the slab-resident factor
        // loads its operands unconditionally, up to a ring of blocks beyond a leg's end (mhe_admm_core.h: sweeps_one_wave_rt) — never
        // used, but it must be the workgroup's own memory.  (Every other shape keeps the slab size it had: the size of a slab decides
        // which L2 sets its arrays share, profiles/r04_slab_pad_sweep.txt.)
        o += wt && !ft ? 8 * 81 : 0;
        total = o;
    }
};

// pointers to the persistent per-instance state (device memory)
struct DevState {
    // sensor latch
    double *imu_t, *accel, *gyro, *p_foot, *J, *qdot, *contact, *quat;
    int* vo_flag;
    double *vo_tpre, *vo_tnow, *vo_dp;
    int* ekf_vo_flag;
    double *ekf_vo_t, *ekf_vo_q;
    // EKF
    double *ekf_q, *ekf_P, *ekf_hist;
    // MHE
    double *st_time, *st_R;
    int* st_dtime;
    double *rec, *Mp, *np_;
    // The arrival cost of the NEXT step, computed ahead of time (k_mhe_marginalize_early, on a second stream under the tail of the
    // current solve): folding window step T - N into (M_p, n_p) needs nothing of step T unless a vision pose arriving at step T
    // rewrites that record's bound (GetMeasurement comes first in the reference's update).  marg_tag[b] = the step the pair was
    // computed for; the assemble of that step takes it if no vision interval arrived, and runs the marginalisation itself otherwise.
    double *Mp_next, *np_next;
    int* marg_tag;
    // What a solve reads of the state the NEXT step's assemble overwrites (arrival cost, VO flags / bounds of the window), copied
    // by k_mhe_assemble at its end: [B][snap_len].  A pipelined handle has three copies (by T mod 3; outputs and solver scratch: two,
    // by parity of T), so that the EKF tick and the assemble of step T + 1 — and of step T + 2 — can run while step T's solve is
    // still waiting for, or holding, the machine (dekf_capi.hip: dekf_update).
    double* snap;
    double *wp, *wpt;
    int* wp_count;
    double* p_vo;
    int *vo_ins_idx, *vo_ins_dtime;
    double* gws;
    int* queue;  // the solve kernels' instance queue: [0] next instance, [1] workgroups that have left (kernels.hip: solve_queue_leave); one pair per output set
    // KF
    double *kf_x, *kf_C;
    // outputs
    double *x_mhe, *v_b;
    int *status, *iters, *rho_updates;
    int* polish_status;  // 0 polishing off / not reached, 1 polished point accepted, -1 rejected (OSQP's status_polish)
    double *pri_res, *dua_res;
    double* prof;  // [B][DEKF_PROF_SLOTS] section cycles, written by the diagnostic (-DDEKF_PROFILE) build only
};

}  // namespace dekf
