// mhe_assemble_core.h — per-instance (one wavefront) construction of everything
// DecentralizedEstimation::update(T) does BEFORE the QP solve:
//   UpdateMHE(T)            dynamics / camera terms of step T-1, measurement terms of step T
//                           (src/decentral_legged_est/src/DecentralEst.cpp:353-585)
//   GetMeasurement(T)       sample intake, VO synchronisation, Bezier interpolation  (:864-985,
//                           src/Spline/Bezier_simple.cpp:12-82)
//   UpdateVOConstraints(T)  equality bounds on the VO rows                            (:987-1009)
//   marginalizeQP(T-N)      Schur arrival cost                      (src/MheSrb.cpp:475-713)
// Instead of the reference's string-keyed registries and growing sparse H/A, every window
// step owns one fixed record (cfg.h: Rec) in HBM; H, g, A, l, u are never materialised.
#pragma once
#include "cfg.h"
#include "smallmat.h"

namespace dekf {

struct AsmScratch {
    // LDS doubles needed by assemble_update for (L, leg_odom_type): the generic marginalisation's buffers
    DEKF_HD static int len(int L, int ft = 0) {
        const int ns = 9 + (ft ? 3 * L : 0), na = ns + 3, dim = na + 3 * L;
        return ns * ns /*Minv*/ + 2 * na * ns /*Am, AmMi*/ + dim * dim /*S*/ + 2 * dim /*pivot row, column*/ +
               dim /*u*/ + dim /*Yu*/ + ns /*M^-1 n*/ + 3 * L * ns /*H M^-1*/ + 16 +
               (ft && !DEKF_DEVICE_BUILD ? dim * dim + dim : 0) /*augmented half of the pivoted inverse in LDS: foot-position states, lane-sequential build only — the device inverts in registers*/;
    }
};

// A_dyn,k[r][j] for R (row-major) and dt (DecentralEst.cpp:395-398)
// (rows / columns >= 9 are the foot-position states of leg_odom_type 1: identity, :395-398)
template <class P>  // P: const double* or the checked pointer of the -DDEKF_BOUNDS build
DEKF_FN double adyn_entry(P R, double dt, int r, int j) {
    if (r == j) return 1.0;
    if (r < 3) {
        if (j == r + 3) return dt;
        if (j >= 6 && j < 9) return -0.5 * dt * dt * R[3 * r + (j - 6)];
        return 0.0;
    }
    if (r < 6 && j >= 6 && j < 9) return -dt * R[3 * (r - 3) + (j - 6)];
    return 0.0;
}

// A_meas row q = (leg, a) applied to a state-sized vector: leg_odom_type 0 picks the velocity ([0 I 0],
// DecentralEst.cpp:95-98), type 1 the foot position relative to the base ([-I 0 0 .. I ..], :106-110)
DEKF_FN double ameas_dot(int ft, const double* v, int q) {
    const int a = q % 3;
    return ft ? v[9 + q] - v[a] : v[3 + a];
}
DEKF_FN double ameas_dot_strided(int ft, const double* v, int stride, int q) {  // v[t * stride]
    const int a = q % 3;
    return ft ? v[(9 + q) * stride] - v[a * stride] : v[(3 + a) * stride];
}

// gains that depend only on R_sb of a step: Q_dyn's 6x6 block and Q_cam (one lane)
DEKF_FN void step_gains(const DevCfg& c, const double* R, double* qd21, double* qc6) {
    double dt = c.dt;
    double RCp[9], RCa[9];  // R C R'
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double sp = 0, sa = 0;
            for (int t = 0; t < 3; ++t) { sp += R[3 * i + t] * c.C_p[t] * R[3 * j + t]; sa += R[3 * i + t] * c.C_accel[t] * R[3 * j + t]; }
            RCp[3 * i + j] = sp; RCa[3 * i + j] = sa;
        }
    double G[36];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            G[6 * i + j] = dt * dt * RCp[3 * i + j] + 0.25 * dt * dt * dt * dt * RCa[3 * i + j];
            G[6 * i + 3 + j] = 0.5 * dt * dt * dt * RCa[3 * i + j];
            G[6 * (3 + i) + j] = 0.5 * dt * dt * dt * RCa[3 * j + i];
            G[6 * (3 + i) + 3 + j] = dt * dt * RCa[3 * i + j];
        }
    inv_spd_unrolled<6>(G);  // G C G' is SPD; static indices keep G in registers (no scratch memory)
    int p = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) qd21[p++] = G[6 * i + j];
    p = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            double s = 0;
            for (int t = 0; t < 3; ++t) s += R[3 * i + t] * c.Q_vo[t] * R[3 * j + t];
            qc6[p++] = s;
        }
}

// b_meas and the 3x3 gain (as_gain) or covariance of leg `leg` at the latched sample (one lane)
// (DecentralEst.cpp:513-547 / :807-837)
template <int NJ>
DEKF_FN void leg_terms_nj(const DevCfg& c, const double* R, const double* gyro, const double* p_foot, const double* J,
                       const double* qdot, double contact, bool as_gain, double* bm3, double* w6) {
    const int nj = NJ > 0 ? NJ : c.nj;  // NJ > 0: every loop below unrolls and the local arrays stay in registers
    double Jq[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < nj; ++j) Jq[i] += J[i * nj + j] * qdot[j];
    double wxp[3], t1[3], t2[3];
    cross3(gyro, p_foot, wxp);
    mv3(R, Jq, t1);
    mv3(R, wxp, t2);
    for (int i = 0; i < 3; ++i) bm3[i] = -t1[i] - t2[i];
    if (contact == 0.0) {
        const double* d = as_gain ? c.Q_swing : c.C_swing;
        w6[0] = d[0]; w6[1] = 0; w6[2] = 0; w6[3] = d[1]; w6[4] = 0; w6[5] = d[2];
        return;
    }
    // G C G' with G = [-J, -w^x J, p^x], C = diag(C_enc_vel, C_enc_pos, C_gyro)
    double WJ[3 * (NJ > 0 ? NJ : DEKF_MAX_JOINTS)];
#pragma unroll
    for (int j = 0; j < nj; ++j) {
        double col[3] = {J[0 * nj + j], J[1 * nj + j], J[2 * nj + j]}, o[3];
        cross3(gyro, col, o);
        WJ[0 * nj + j] = o[0]; WJ[1 * nj + j] = o[1]; WJ[2 * nj + j] = o[2];
    }
    double Px[9] = {0, -p_foot[2], p_foot[1], p_foot[2], 0, -p_foot[0], -p_foot[1], p_foot[0], 0};
    double Cb[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double s = 0;
#pragma unroll
            for (int j = 0; j < nj; ++j) s += J[i * nj + j] * c.C_enc_vel[j] * J[k * nj + j] + WJ[i * nj + j] * c.C_enc_pos[j] * WJ[k * nj + j];
            for (int t = 0; t < 3; ++t) s += Px[3 * i + t] * c.C_gyro[t] * Px[3 * k + t];
            Cb[3 * i + k] = s;
        }
    double RC[9], Cw[9];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            double s = 0;
            for (int t = 0; t < 3; ++t) s += R[3 * i + t] * Cb[3 * t + k];
            RC[3 * i + k] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            double s = 0;
            for (int t = 0; t < 3; ++t) s += RC[3 * i + t] * R[3 * k + t];
            Cw[3 * i + k] = s;
        }
    const double cs[6] = {Cw[0], 0.5 * (Cw[1] + Cw[3]), 0.5 * (Cw[2] + Cw[6]), Cw[4], 0.5 * (Cw[5] + Cw[7]), Cw[8]};
    if (as_gain) { inv3_sym(cs, w6); return; }
    w6[0] = Cw[0]; w6[1] = Cw[1]; w6[2] = Cw[2]; w6[3] = Cw[4]; w6[4] = Cw[5]; w6[5] = Cw[8];
}
// leg_odom_type 1 (DecentralEst.cpp:310-325, 550-563, 753, 803): b_meas = R p_foot; measurement weight
// R (J C_enc_pos J')^-1 R' (as_gain) or the covariance R J C_enc_pos J' R'; process weight of the foot-position state
// 1/dt^2 R Q R' with Q = Q_foot_slide in contact, Q_foot_swing otherwise (:432-451), or its covariance dt^2 R C R'
DEKF_FN void leg_terms_position(const DevCfg& c, const double* R, const double* p_foot, const double* J, double contact,
                                bool as_gain, double* bm3, double* w6, double* f6) {
    const int nj = c.nj;
    mv3(R, p_foot, bm3);
    double JCJ[9];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            double s = 0;
            for (int j = 0; j < nj; ++j) s += J[i * nj + j] * c.C_enc_pos[j] * J[k * nj + j];
            JCJ[3 * i + k] = s;
        }
    double Mi[6];
    const double cs[6] = {JCJ[0], 0.5 * (JCJ[1] + JCJ[3]), 0.5 * (JCJ[2] + JCJ[6]), JCJ[4], 0.5 * (JCJ[5] + JCJ[7]), JCJ[8]};
    if (as_gain) inv3_sym(cs, Mi);
    else for (int t = 0; t < 6; ++t) Mi[t] = cs[t];
    // R Mi R' and R diag(d) R', packed upper triangles
    const double* d = contact != 0.0 ? (as_gain ? c.Q_slide : c.C_slide) : (as_gain ? c.Q_swing : c.C_swing);
    const double sc = as_gain ? 1.0 / (c.dt * c.dt) : c.dt * c.dt;
    int pk = 0;
    for (int i = 0; i < 3; ++i)
        for (int k = i; k < 3; ++k) {
            double sm = 0, sf = 0;
            for (int t = 0; t < 3; ++t) {
                double rm = 0;
                for (int u = 0; u < 3; ++u) rm += R[3 * i + u] * symget(Mi, u, t, 3);
                sm += rm * R[3 * k + t];
                sf += R[3 * i + t] * d[t] * R[3 * k + t];
            }
            w6[pk] = sm;
            f6[pk] = sc * sf;
            ++pk;
        }
}
DEKF_FN void leg_terms(const DevCfg& c, const double* R, const double* gyro, const double* p_foot, const double* J,
                       const double* qdot, double contact, bool as_gain, double* bm3, double* w6) {
    if (c.nj == 3) leg_terms_nj<3>(c, R, gyro, p_foot, J, qdot, contact, as_gain, bm3, w6);       // Go1, PogoX
    else if (c.nj == 5) leg_terms_nj<5>(c, R, gyro, p_foot, J, qdot, contact, as_gain, bm3, w6);  // Cassie
    else leg_terms_nj<0>(c, R, gyro, p_foot, J, qdot, contact, as_gain, bm3, w6);
}

// GetMeasurement(T): returns through LDS/HBM; `pushes` = samples already on the stack.
// sm: LDS scratch (>= 16 doubles used here)
// Returns whether a vision interval rewrote VO flags / bounds of window records at this step (wave-uniform): the one thing of step T
// that the marginalisation of step T depends on (marginalize_early).
DEKF_FN bool get_measurement(const DevCfg& c, const DevState& s, int b, int T, int pushes, double* sm) {
    bool vo_rewrite = false;
    const int ring = c.ring;
    double* st_time = s.st_time + (size_t)b * ring;
    double* st_R = s.st_R + (size_t)b * ring * 9;
    double* rec_base = s.rec + (size_t)b * c.wcap * c.rec;
    const double* quat = s.quat + 4 * (size_t)b;
    double R[9];
    quat_to_rot(quat, R);
    int size = pushes < ring ? pushes : ring;  // entries held before this push
    int base = pushes - size;                  // logical index of the oldest one
    int flag = s.vo_flag[b];
    DEKF_SYNC();
    if (flag && size > 0) {  // wave-uniform
        double t_pre = s.vo_tpre[b], t_now = s.vo_tnow[b];
        // upper_bound on a sorted stack == count of entries <= t
        double cnt_pre = wred_sum(size, [&](int i) { return st_time[(base + i) % ring] <= t_pre ? 1.0 : 0.0; });
        double cnt_now = wred_sum(size, [&](int i) { return st_time[(base + i) % ring] <= t_now ? 1.0 : 0.0; });
        if (cnt_pre > 0.5) {
            int idx_pre = base + (int)(cnt_pre + 0.5) - 1;
            int idx_now = base + (int)(cnt_now + 0.5) - 1;
            const double* Rp = st_R + (size_t)(idx_pre % ring) * 9;
            const double* dp = s.vo_dp + 3 * (size_t)b;
            double* pv = s.p_vo + 3 * (size_t)b;
            double* wp = s.wp + 12 * (size_t)b;
            double* wpt = s.wpt + 4 * (size_t)b;
            int wcnt = s.wp_count[b];
            double acc[3];
            for (int i = 0; i < 3; ++i) acc[i] = pv[i] + Rp[3 * i] * dp[0] + Rp[3 * i + 1] * dp[1] + Rp[3 * i + 2] * dp[2];
            int win_start = pushes - (c.N < T ? c.N : T);
            int interp_start = win_start > idx_pre ? win_start : idx_pre;
            double t_interp = st_time[interp_start % ring];
            // Bezier::add_way_point: keep the last four
            double P[4][3], tw[4];
            int nw = wcnt < 4 ? wcnt + 1 : 4;
            int shift = wcnt < 4 ? 0 : 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // static indices: P and tw stay in registers
                const int src = i + shift < 3 ? i + shift : 3;
#pragma unroll
                for (int a = 0; a < 3; ++a) P[i][a] = i < nw - 1 ? wp[3 * src + a] : (i == nw - 1 ? acc[a] : 0.0);
                tw[i] = i < nw - 1 ? wpt[src] : (i == nw - 1 ? t_now : 0.0);
            }
            DEKF_SYNC();
            if (DEKF_LANE() == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nw) {
#pragma unroll
                        for (int a = 0; a < 3; ++a) wp[3 * i + a] = P[i][a];
                        wpt[i] = tw[i];
                    }
                s.wp_count[b] = nw;
                for (int a = 0; a < 3; ++a) pv[a] = acc[a];
            }
            if (idx_now > win_start && nw >= 4) {
                vo_rewrite = true;
                double t_interval = tw[3] - tw[0];
                double u0 = (t_interp - tw[0]) / t_interval;
                double uinc = c.dt / t_interval;
                int num = idx_now - interp_start + 1;
                // node_i, i = 0..num-1; bound of step interp_start+i is -(node_{i+1} - node_i)
                wfor(num - 1, [&](int i) {
                    double nd[2][3];
                    for (int q = 0; q < 2; ++q) {
                        double u = u0 + uinc * (double)(i + q);
                        for (int a = 0; a < 3; ++a) {
                            double v = u * u * u * ((-1) * P[0][a] + 3 * P[1][a] - 3 * P[2][a] + P[3][a]);
                            v += u * u * (3 * P[0][a] - 6 * P[1][a] + 3 * P[2][a]);
                            v += u * ((-3) * P[0][a] + 3 * P[1][a]);
                            v += P[0][a];
                            nd[q][a] = v;
                        }
                    }
                    double* r = rec_base + (size_t)((interp_start + i) % c.wcap) * c.rec;
                    r[Rec::VOF] = 1.0;
                    for (int a = 0; a < 3; ++a) r[Rec::VOB + a] = -(nd[1][a] - nd[0][a]);
                });
                if (DEKF_LANE() == 0) { s.vo_ins_idx[b] = interp_start - win_start; s.vo_ins_dtime[b] = interp_start; }
            }
        }
        if (DEKF_LANE() == 0) s.vo_flag[b] = 0;
    }
    // push
    if (DEKF_LANE() == 0) {
        st_time[pushes % ring] = s.imu_t[b];
        s.st_dtime[(size_t)b * ring + pushes % ring] = T;
        for (int i = 0; i < 9; ++i) st_R[(size_t)(pushes % ring) * 9 + i] = R[i];
    }
    (void)sm;
    DEKF_SYNC();
    return vo_rewrite;
}

// measurement-side part of the record of step T from the latched sample (after get_measurement);
// as_gain = false stores covariances instead of gains (KF mode)
DEKF_FN void write_measurement_record(const DevCfg& c, const DevState& s, int b, int T, bool as_gain = true) {
    double* r = s.rec + ((size_t)b * c.wcap + (T % c.wcap)) * c.rec;
    const double* quat = s.quat + 4 * (size_t)b;
    const double* accel = s.accel + 3 * (size_t)b;
    const double* gyro = s.gyro + 3 * (size_t)b;
    double R[9];
    quat_to_rot(quat, R);
    int L = c.L, nj = c.nj;
    wfor(L + 1, [&](int i) {
        if (i < L) {
            double bm[3], w6[6];
            if (c.ft) {
                double f6[6];
                leg_terms_position(c, R, s.p_foot + ((size_t)b * L + i) * 3, s.J + ((size_t)b * L + i) * 3 * nj,
                                   s.contact[(size_t)b * L + i], as_gain, bm, w6, f6);
                for (int a = 0; a < 6; ++a) r[Rec::qf(c.nm) + 6 * i + a] = f6[a];
            } else {
                leg_terms(c, R, gyro, s.p_foot + ((size_t)b * L + i) * 3, s.J + ((size_t)b * L + i) * 3 * nj,
                          s.qdot + ((size_t)b * L + i) * nj, s.contact[(size_t)b * L + i], as_gain, bm, w6);
            }
            for (int a = 0; a < 3; ++a) r[Rec::BM + 3 * i + a] = bm[a];
            for (int a = 0; a < 6; ++a) r[Rec::qm(c.nm) + 6 * i + a] = w6[a];
        } else {
            double as[3];
            mv3(R, accel, as);
            for (int a = 0; a < 9; ++a) r[Rec::R + a] = R[a];
            r[Rec::AS + 0] = as[0]; r[Rec::AS + 1] = as[1]; r[Rec::AS + 2] = as[2] - 9.81;
            for (int a = 0; a < 3; ++a) r[Rec::GY + a] = gyro[a];
            r[Rec::VOF] = 0.0;
            r[Rec::VOB] = r[Rec::VOB + 1] = r[Rec::VOB + 2] = 0.0;
        }
    });
}

#if DEKF_DEVICE_BUILD
// ---- register-resident forms for the device: one matrix column per lane, no LDS round trips ---------
// In-place inverse of a definite N x N matrix: lane j < N holds column j in a[0..N).  Same pivot order
// and update as winverse_definite (natural order, no search).  The loop over pivots is a real loop: after
// each pivot the register array is rotated by one row, so that the pivot row is always a[0] (static
// register indices) — after N pivots the rows are back in place.  The pivot column comes from lane p
// through v_readlane (SGPR operands of the FMAs).
template <int N>
DEKF_FN bool gj_columns(double (&a)[N], int lane) {
    bool ok = true;
    for (int p = 0; p < N; ++p) {
        const double piv = readlane_f64(a[0], p);
        if (!(fabs(piv) > 0.0) || !(fabs(piv) < 1e300)) { ok = false; break; }  // wave-uniform
        const double d = rcp_fast(piv);  // v_rcp + two Newton steps: 5 instructions instead of the 12 of a division
        const bool is_p = lane == p;
        const double rd = (is_p ? 1.0 : a[0]) * d;  // new pivot-row entry of this column (d itself in the pivot column)
#pragma unroll
        for (int i = 1; i < N; ++i) {
            const double ci = readlane_f64(a[i], p);
            a[i - 1] = fma(-ci, rd, is_p ? 0.0 : a[i]);  // pivot column: -c d; elsewhere a - c (r d)
        }
        a[N - 1] = rd;
    }
    return ok;
}

// The same sweep WITHOUT the in-place handling of the pivot column: every lane runs the plain elimination a_i -= s_ip (a_p / s_pp).
// For a caller that lets the columns of the identity ride along in other lanes (they end as the columns of the inverse; the lanes
// that hold the matrix itself end as junk): two selects less per row and pivot.
template <int N>
DEKF_FN bool gj_columns_plain(double (&a)[N], int lane) {
    (void)lane;
    bool ok = true;
    for (int p = 0; p < N; ++p) {
        const double piv = readlane_f64(a[0], p);
        if (!(fabs(piv) > 0.0) || !(fabs(piv) < 1e300)) { ok = false; break; }  // wave-uniform
        const double rd = a[0] * rcp_fast(piv);
#pragma unroll
        for (int i = 1; i < N; ++i) {
            const double ci = readlane_f64(a[i], p);
            a[i - 1] = fma(-ci, rd, a[i]);
        }
        a[N - 1] = rd;
    }
    return ok;
}

// M^-1 for the generic marginalisation, in registers: lane j < N takes column j of the arrival information from HBM (through the
// LOWER triangle, see marginalize_generic), the wavefront runs the Gauss-Jordan sweep of winverse_definite — natural pivot order,
// d = 1 / pivot, pivot row r_j d, pivot column -c_i d, elsewhere a_ij - c_i r_j d, the very expressions — and leaves M^-1 in LDS.
// Bit-identical to the LDS form (every entry sees the same operations in the same order); 21 loads in flight per lane instead of
// seven dependent rounds of them, no LDS round trip per pivot.
template <int N>
DEKF_FN bool inverse_definite_regs(const double* Mp, double* Minv, int lane) {
    static_assert(N <= WAVE, "one column per lane");
    const int j = lane < N ? lane : N - 1;
    double a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = i >= j ? Mp[N * i + j] : Mp[N * j + i];
    bool ok = true;
    for (int p = 0; p < N; ++p) {
        const double piv = readlane_f64(a[0], p);
        if (!(fabs(piv) > 0.0) || !(fabs(piv) < 1e300)) { ok = false; break; }  // wave-uniform
        const double d = 1.0 / piv;
        const bool is_p = lane == p;
        const double rp = a[0];
        const double newrow = is_p ? d : rp * d;
#pragma unroll
        for (int i = 1; i < N; ++i) {
            const double ci = readlane_f64(a[i], p);
            const double pc = -ci * d;
            const double el = a[i] - ci * rp * d;
            a[i - 1] = is_p ? pc : el;
        }
        a[N - 1] = newrow;
    }
    if (lane < N) {
#pragma unroll
        for (int i = 0; i < N; ++i) Minv[N * i + lane] = a[i];
    }
    wave_sync();
    return ok;
}

// In-place inverse WITH ROW PIVOTING of the N x N matrix S (row-major in LDS, leading dimension N, N <= 64): lane j < N takes
// column j into registers, the wavefront runs Gauss-Jordan with the pivot rule and the arithmetic of winverse(.., pivoting = true)
// — first largest |entry| of the pivot column at or below the diagonal, row swap, d = 1 / pivot (IEEE), row p scaled by d, every other
// row i minus s_ip times the scaled row — and writes S^-1 back.  Bit-identical to the augmented [S | I] form in LDS: the in-place
// scheme keeps the one column of the growing inverse that is no longer a unit vector where the eliminated column of S was (all other
// entries of [S | I] it leaves out are exact 0 / 1 / never read again), and a row swap moves a register in EVERY lane, i.e. in the
// columns of S and of the inverse alike.  What the swaps leave behind is a column permutation of the result, B = (P S)^-1 = S^-1 P':
// lane j carries the original index of the row that sits at position j (`lab`, exchanged between the two lanes of a swap), which is
// the column of S^-1 its registers hold at the end — the write-back applies it for free.
// The register array rotates by one row per pivot (gj_columns): the candidate rows p .. N-1 are a[0 .. N-1-p], the pivot row is a[0]
// after the swap, all register indices are static.  Replaces 41 rounds of LDS read-modify-write per pivot (with two run-time integer
// divisions per element) for the reference form of the foot-state arrival cost (MheSrb.cpp:588,640: Eigen inverse() = LU with
// partial pivoting).
template <int N>
DEKF_FN bool inverse_pivoted_regs(double* S, int lane) {
    static_assert(N <= WAVE, "one column per lane");
    const int j = lane < N ? lane : N - 1;  // lanes beyond the matrix mirror the last column (never broadcast, never stored)
    double a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = S[N * i + j];
    int lab = lane;
    bool ok = true;
    for (int p = 0; p < N; ++p) {
        // pivot search in column p (lane p) over rows p .. N-1
        double best = fabs(a[0]);
        int bi = 0;
#pragma unroll
        for (int i = 1; i < N; ++i) {
            const double v = fabs(a[i]);
            const bool take = i < N - p && v > best;
            best = take ? v : best;
            bi = take ? i : bi;
        }
        bi = __builtin_amdgcn_readlane(bi, p);
        if (readlane_f64(best, p) == 0.0) { ok = false; break; }  // wave-uniform
        // rows at positions p and p + bi change places (registers 0 and bi of every lane; their labels between the two lanes)
        if (bi != 0) {  // wave-uniform
            const double r0 = a[0];
            double pr = r0;
#pragma unroll
            for (int i = 1; i < N; ++i) {
                const bool hit = i == bi;
                pr = hit ? a[i] : pr;
                a[i] = hit ? r0 : a[i];
            }
            a[0] = pr;
            const int lp = __builtin_amdgcn_readlane(lab, p), lq = __builtin_amdgcn_readlane(lab, p + bi);
            lab = lane == p ? lq : (lane == p + bi ? lp : lab);
        }
        const double d = 1.0 / readlane_f64(a[0], p);
        const bool is_p = lane == p;
        const double rd = (is_p ? 1.0 : a[0]) * d;
#pragma unroll
        for (int i = 1; i < N; ++i) {
            const double ci = readlane_f64(a[i], p);
            a[i - 1] = fma(-ci, rd, is_p ? 0.0 : a[i]);
        }
        a[N - 1] = rd;
    }
    wave_sync();  // every lane has read its column before any lane overwrites S
    if (ok && lane < N) {
#pragma unroll
        for (int i = 0; i < N; ++i) S[N * i + lab] = a[i];
    }
    wave_sync();
    return ok;
}

// column j (0..5) of the 6x6 process covariance G C G' of a step with rotation R (the matrix step_gains inverts)
DEKF_FN void cov6_column(const DevCfg& c, const double* R, int j, double (&col)[6]) {
    const double dt = c.dt;
    const int jj = j < 3 ? j : j - 3;
    const double* Rj = R + 3 * jj;
    double cp[3], ca[3], ra[3];  // (R C_p R')[i][jj], (R C_a R')[i][jj], (R C_a R')[jj][i]
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double sp = 0, sa = 0, sr = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            sp += R[3 * i + t] * c.C_p[t] * Rj[t];
            sa += R[3 * i + t] * c.C_accel[t] * Rj[t];
            sr += Rj[t] * c.C_accel[t] * R[3 * i + t];
        }
        cp[i] = sp; ca[i] = sa; ra[i] = sr;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        col[i] = j < 3 ? dt * dt * cp[i] + 0.25 * dt * dt * dt * dt * ca[i] : 0.5 * dt * dt * dt * ca[i];
        col[3 + i] = j < 3 ? 0.5 * dt * dt * dt * ra[i] : dt * dt * ca[i];
    }
}

// entry (a, d) of a packed symmetric 3x3 (6 values) for a compile-time a and a lane-dependent d, by selects
template <int A>
DEKF_FN double sym3_row(const double (&q)[6], int d) {
    constexpr int i0 = A == 0 ? 0 : (A == 1 ? 1 : 2), i1 = A == 0 ? 1 : (A == 1 ? 3 : 4), i2 = A == 0 ? 2 : (A == 1 ? 4 : 5);
    return d == 0 ? q[i0] : (d == 1 ? q[i1] : q[i2]);
}
DEKF_FN double sym3_at(const double (&q)[6], int a_static, int d) {
    return a_static == 0 ? sym3_row<0>(q, d) : (a_static == 1 ? sym3_row<1>(q, d) : sym3_row<2>(q, d));
}

// marginalize_step with the 9x9 and the (21|24)-dim inverses in registers.  sm: Minv 81 | AmMi 108 | u 24 | Min 9
template <int L, bool VO>
DEKF_FN bool marginalize_regs(const DevCfg& c, const DevState& s, int b, const double* r, double* sm) {
    constexpr int NM = 3 * L, NA = VO ? 12 : 9, DIM = NA + NM;
    const int lane = DEKF_LANE();
    double* Mp = s.Mp + 81 * (size_t)b;
    double* np = s.np_ + 9 * (size_t)b;
    double* Minv = sm;
    double* AmMi = Minv + 81;
    double* u = AmMi + 108;
    double* Min = u + 24;
    const double* R = r + Rec::R;
    const double dt = c.dt;
    const int j = lane < DIM ? lane : DIM - 1;  // lanes beyond the matrix mirror the last column (no stores)
    const int j9 = lane < 9 ? lane : 8;

    // M^-1: lane j holds column j
    // (both sweeps below carry the columns of the identity in other lanes — lanes 16..24 here, 32..32 + NA - 1 for S — which end as
    // the columns of the inverse: gj_columns_plain, no in-place handling of the pivot column, two selects less per row and pivot)
    double m9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m9[i] = (lane >= 16 && lane < 25) ? (i == lane - 16 ? 1.0 : 0.0) : Mp[9 * i + j9];
    bool ok = gj_columns_plain<9>(m9, lane);
    if (lane >= 16 && lane < 25) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Minv[9 * i + lane - 16] = m9[i];
    }
    DEKF_SYNC();
    // row `ia` of [A_dyn; A_cam] and of Am M^-1 (lanes < NA), M^-1 n (lanes < 9)
    const int ia = lane < NA ? lane : NA - 1;
    double am[9], ami[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) am[t] = ia < 9 ? adyn_entry(R, dt, ia, t) : (ia - 9 == t ? 1.0 : 0.0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        double acc = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) acc += am[q] * Minv[9 * q + t];
        ami[t] = acc;
    }
    double min_own = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) min_own += Minv[9 * j9 + t] * np[t];
    if (lane < NA) {
#pragma unroll
        for (int t = 0; t < 9; ++t) AmMi[9 * lane + t] = ami[t];
    }
    if (lane < 9) Min[lane] = min_own;
    DEKF_SYNC();

    // column j of S = -([A; H] M^-1 [A; H]' + blkdiag(Q_dyn^-1, Q_cam^-1, Q_meas^-1))
    double a[DIM];
    const bool dyn_col = j < NA;
    const int cq = 3 + (dyn_col ? 0 : (j - NA) % 3);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        double acc = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += AmMi[9 * i + t] * am[t];
        a[i] = dyn_col ? -acc : -AmMi[9 * i + cq];
    }
#pragma unroll
    for (int q = 0; q < NM; ++q) a[NA + q] = dyn_col ? -ami[3 + q % 3] : -Minv[9 * (3 + q % 3) + cq];
    {   // diagonal blocks
        double col6[6];
        cov6_column(c, R, j < 6 ? j : 5, col6);
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i] -= j < 6 ? col6[i] : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) a[6 + i] -= j == 6 + i ? 1.0 / c.Q_bias_dt2[i] : 0.0;
        if (VO) {
            double qi[6];
            inv3_sym(r + Rec::QC, qi);
            const int d = j >= 9 && j < 12 ? j - 9 : 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) a[9 + i] -= (j >= 9 && j < 12) ? sym3_at(qi, i, d) : 0.0;
        }
        const int leg = dyn_col ? 0 : (j - NA) / 3, d = dyn_col ? 0 : (j - NA) % 3;
        double qi[6];
        inv3_sym(r + Rec::qm(NM) + 6 * leg, qi);
#pragma unroll
        for (int q = 0; q < NM; ++q) a[NA + q] -= (!dyn_col && q / 3 == leg) ? sym3_at(qi, q % 3, d) : 0.0;
    }
    // u = [ b_dyn ; vo bound ] + Am M^-1 n   |   b_meas + H M^-1 n
    if (lane < DIM) {
        double v;
        if (lane < NA) {
            double acc = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc += am[t] * Min[t];
            double rhs;
            if (lane < 3) rhs = -0.5 * dt * dt * r[Rec::AS + lane];
            else if (lane < 6) rhs = -dt * r[Rec::AS + lane - 3];
            else if (lane < 9) rhs = 0.0;
            else rhs = r[Rec::VOB + lane - 9];
            v = rhs + acc;
        } else {
            v = r[Rec::BM + lane - NA] + Min[3 + (lane - NA) % 3];
        }
        u[lane] = v;
    }
    DEKF_SYNC();
    // only the columns 0 .. NA - 1 of S^-1 are used below: their identity columns ride along in lanes 32 .. 32 + NA - 1
    const int jl = lane - 32;  // the column of S^-1 this lane ends up with (0 <= jl < NA)
    if (jl >= 0 && jl < NA) {
#pragma unroll
        for (int i = 0; i < DIM; ++i) a[i] = i == jl ? 1.0 : 0.0;
    }
    ok = gj_columns_plain<DIM>(a, lane) && ok;
    // (S^-1 u)_j through column j (S^-1 is symmetric)
    double yu = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) yu += a[i] * u[i];
    // M+ = -B' S^-1 B, n+ = B' S^-1 u with B = [-I9; -[I3 0] (VO); 0]: rows/columns 9..11 fold onto 0..2
    double out[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) out[i] = a[i];
    if (VO) {
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] += a[9 + i];
        double fold[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) fold[i] = __shfl_down(a[i], 9);
#pragma unroll
        for (int i = 0; i < 3; ++i) fold[i] += __shfl_down(a[9 + i], 9);
        const double yf = __shfl_down(yu, 9);
        if (jl >= 0 && jl < 3) {
#pragma unroll
            for (int i = 0; i < 9; ++i) out[i] += fold[i];
            yu += yf;
        }
    }
    if (jl >= 0 && jl < 9) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Mp[9 * i + jl] = -out[i];
        np[jl] = -yu;
    }
    DEKF_SYNC();
    return ok;
}
#endif  // DEKF_DEVICE_BUILD

// marginalizeQP in INFORMATION form, used when the foot positions are states (leg_odom_type 1).
// The reference (MheSrb.cpp:527-651) builds the saddle matrix S = -([A;H] M^-1 [A;H]' + blkdiag(Q^-1, R^-1)) out of
// COVARIANCES and inverts it.  With foot-position states that is numerically fragile: a swinging foot has process
// covariance dt^2 * 1e14 and, after a few swing steps, an arrival information of 1e-10, so M^-1 and Q^-1 carry entries of
// 1e9..1e10 next to measurement covariances of 1e-4 — the information of the measurement survives the additions with
// three or four digits.  (Measured on the oracle, which follows the reference's formulas: its states drift 1e-5 away from
// the never-marginalised problem once a foot has been through a swing phase, tests/test_oracle_mhe.py.)  Eliminating
// x_k from   1/2 x'Mx + n'x + 1/2 |A x - x+ - b|_Q^2 + 1/2 |H x - y|_R^2 [+ 1/2 |E x - E x+ - b_c|_Qc^2]
// directly gives the same arrival cost (matrix inversion lemma) from gains only:
//     Lambda = M + A'QA + H'RH [+ E'Qc E],   G = QA [+ E'Qc E],   r0 = -n + A'Qb + H'Ry [+ E'Qc b_c]
//     M+ = Q [+ E'Qc E] - G Lambda^-1 G',      n+ = Qb [+ E'Qc b_c] - G Lambda^-1 r0
// sm: QA/G ns^2 | Lambda ns^2 | T1 ns^2 | Qb ns | r0 ns | scratch 2 ns
DEKF_FN bool marginalize_info(const DevCfg& c, const DevState& s, int b, const double* r, double* sm) {
    const int nm = c.nm, L = c.L, ns = c.ns, ft = c.ft, n2 = ns * ns;
    double* Mp = s.Mp + (size_t)n2 * b;
    double* np = s.np_ + (size_t)ns * b;
    const bool vo = r[Rec::VOF] != 0.0;
    double* QA = sm;
    double* Lam = QA + n2;
    double* T1 = Lam + n2;
    double* Qb = T1 + n2;
    double* r0 = Qb + ns;
    double* wsc = r0 + ns;
    const double* R = r + Rec::R;
    const double dt = c.dt;
    // Q(i, t): process gain, block diagonal [6x6 | diag 3 | 3x3 per foot]
    auto qdyn = [&](int i, int t) -> double {
        if (i < 6) return t < 6 ? symget(r + Rec::QD, i, t, 6) : 0.0;
        if (i < 9) return t == i ? c.Q_bias_dt2[i - 6] : 0.0;
        const int leg = (i - 9) / 3;
        return (t >= 9 + 3 * leg && t < 12 + 3 * leg) ? symget(r + Rec::qf(nm) + 6 * leg, i - 9 - 3 * leg, t - 9 - 3 * leg, 3) : 0.0;
    };
    auto bdyn = [&](int i) -> double { return i < 3 ? -0.5 * dt * dt * r[Rec::AS + i] : (i < 6 ? -dt * r[Rec::AS + i - 3] : 0.0); };
    // A_meas' W A_meas, entry (i, j), W = blkdiag of the packed 3x3 measurement gains
    auto hrh = [&](int i, int j) -> double {
        double v = 0.0;
        for (int leg = 0; leg < L; ++leg) {
            // column i of A_meas restricted to this leg's rows: coefficient and local row index
            double ci, cj;
            int ai, aj;
            if (ft) {
                ci = i < 3 ? -1.0 : ((i >= 9 + 3 * leg && i < 12 + 3 * leg) ? 1.0 : 0.0); ai = i < 3 ? i : i - 9 - 3 * leg;
                cj = j < 3 ? -1.0 : ((j >= 9 + 3 * leg && j < 12 + 3 * leg) ? 1.0 : 0.0); aj = j < 3 ? j : j - 9 - 3 * leg;
            } else {
                ci = (i >= 3 && i < 6) ? 1.0 : 0.0; ai = i - 3;
                cj = (j >= 3 && j < 6) ? 1.0 : 0.0; aj = j - 3;
            }
            if (ci != 0.0 && cj != 0.0) v += ci * cj * symget(r + Rec::qm(nm) + 6 * leg, ai, aj, 3);
        }
        return v;
    };
    wfor(n2 + ns, [&](int e) {
        if (e < n2) {
            const int i = e / ns, j = e - ns * i;
            double acc = 0.0;
            for (int t = 0; t < ns; ++t) {
                const double qv = qdyn(i, t);
                if (qv != 0.0) acc += qv * adyn_entry(R, dt, t, j);
            }
            QA[e] = acc;
        } else {
            const int i = e - n2;
            double acc = 0.0;
            for (int t = 0; t < ns; ++t) acc += qdyn(i, t) * bdyn(t);
            Qb[i] = acc;
        }
    });
    wfor(n2 + ns, [&](int e) {
        if (e < n2) {
            const int i = e / ns, j = e - ns * i;
            double acc = Mp[e] + hrh(i, j);
            for (int t = 0; t < ns; ++t) {
                const double at = adyn_entry(R, dt, t, i);
                if (at != 0.0) acc += at * QA[ns * t + j];
            }
            if (vo && i < 3 && j < 3) acc += symget(r + Rec::QC, i, j, 3);
            Lam[e] = acc;
        } else {
            const int i = e - n2;
            double acc = -np[i];
            for (int t = 0; t < ns; ++t) acc += adyn_entry(R, dt, t, i) * Qb[t];
            for (int q = 0; q < nm; ++q) {  // (H' R y)_i
                const int leg = q / 3, a = q - 3 * leg;
                const double hi = ft ? (i < 3 ? (i == a ? -1.0 : 0.0) : (i == 9 + q ? 1.0 : 0.0)) : (i == 3 + a ? 1.0 : 0.0);
                if (hi != 0.0) {
                    double ry = 0.0;
                    for (int d = 0; d < 3; ++d) ry += symget(r + Rec::qm(nm) + 6 * leg, a, d, 3) * r[Rec::BM + 3 * leg + d];
                    acc += hi * ry;
                }
            }
            if (vo && i < 3)
                for (int d = 0; d < 3; ++d) acc += symget(r + Rec::QC, i, d, 3) * r[Rec::VOB + d];
            r0[i] = acc;
        }
    });
    if (vo) {  // G = QA + E'Qc E, and the same term on the right-hand side Qb
        wfor(12, [&](int e) {
            if (e < 9) QA[ns * (e / 3) + e % 3] += symget(r + Rec::QC, e / 3, e % 3, 3);
            else {
                const int i = e - 9;
                double acc = 0.0;
                for (int d = 0; d < 3; ++d) acc += symget(r + Rec::QC, i, d, 3) * r[Rec::VOB + d];
                Qb[i] += acc;
            }
        });
    }
    const bool ok = winverse_definite(Lam, ns, wsc);
    wmatmul<false, false>(T1, ns, QA, ns, Lam, ns, ns, ns, ns);  // G Lambda^-1
    wfor(n2 + ns, [&](int e) {
        if (e < n2) {
            const int i = e / ns, j = e - ns * i;
            double acc = qdyn(i, j);
            if (vo && i < 3 && j < 3) acc += symget(r + Rec::QC, i, j, 3);
            double sub = 0.0;
            for (int t = 0; t < ns; ++t) sub += T1[ns * i + t] * QA[ns * j + t];
            Mp[e] = acc - sub;
        } else {
            const int i = e - n2;
            double sub = 0.0;
            for (int t = 0; t < ns; ++t) sub += T1[ns * i + t] * r0[t];
            np[i] = Qb[i] - sub;
        }
    });
    return ok;
}

// marginalizeQP(step) in the generic form (any state dimension): rows [Dyn (ns) | VO (3, when flagged) | Meas (nm)]   (MheSrb.cpp:475-713)
// LT / VT: 0 / -1 = leg count and VO flag read at run time (the lane-sequential build, type 0 beyond four legs); LT = 1..4, VT = 0 / 1:
// the foot-state shapes of the device build with every dimension a compile-time constant — the same statements in the same order
// (bit-identical results), but the element -> (row, column) divisions become multiplications and the dot products unroll.
template <int LT, int VT>
DEKF_FN bool marginalize_generic(const DevCfg& c, const DevState& s, int b, const double* r, double* sm) {
    const int L = LT > 0 ? LT : c.L, ft = LT > 0 ? 1 : c.ft;
    const int nm = 3 * L, ns = 9 + (ft ? 3 * L : 0);
    double* Mp = s.Mp + (size_t)ns * ns * b;
    double* np = s.np_ + (size_t)ns * b;
    const bool vo = VT >= 0 ? VT != 0 : r[Rec::VOF] != 0.0;
    const int na = vo ? ns + 3 : ns;
    const int dim = na + nm;
    double* Minv = sm;                      // ns x ns
    double* Am = Minv + ns * ns;            // na x ns   ([A_dyn; A_cam])
    double* AmMi = Am + (ns + 3) * ns;      // na x ns
    double* S = AmMi + (ns + 3) * ns;       // dim x dim
    double* wsc = S + (ns + 3 + nm) * (ns + 3 + nm);  // 2 dim
    double* u = wsc + 2 * (ns + 3 + nm);    // dim
    double* Yu = u + (ns + 3 + nm);         // dim
    double* Min = Yu + (ns + 3 + nm);       // ns: M^-1 n
    double* HMi = Min + ns;                 // nm x ns: A_meas M^-1
    const double* R = r + Rec::R;
    const double dt = c.dt;
#if DEKF_DEVICE_BUILD
    constexpr bool REGS = LT > 0;  // M^-1 in registers (same arithmetic: inverse_definite_regs)
#else
    constexpr bool REGS = false;
#endif
    // M enters through its LOWER triangle, as in the reference: M^-1 comes from Eigen's SimplicialLLT (default UpLo = Lower,
    // MheSrb.cpp:524-525), while the QP takes the upper one (OSQP reads triu(P)).  M+ = -B' S^-1 B is symmetric only up to
    // rounding, and with foot-position states that rounding is amplified at every swing phase (measured: 1e-13 -> 1e-1
    // over three phases when both triangles feed the inverse)
    bool ok;
    if constexpr (REGS) {
#if DEKF_DEVICE_BUILD
        wfor(na * ns, [&](int q) {
            const int i = q / ns, j = q - ns * i;
            Am[q] = i < ns ? adyn_entry(R, dt, i, j) : ((i - ns) == j ? 1.0 : 0.0);
        });
        ok = inverse_definite_regs<9 + 3 * (LT > 0 ? LT : 1)>(Mp, Minv, DEKF_LANE());
#endif
    } else {
        wfor(ns * ns + na * ns, [&](int e) {
            if (e < ns * ns) { const int i = e / ns, j = e - ns * i; Minv[e] = i >= j ? Mp[e] : Mp[ns * j + i]; }
            else {
                int q = e - ns * ns, i = q / ns, j = q - ns * i;
                Am[q] = i < ns ? adyn_entry(R, dt, i, j) : ((i - ns) == j ? 1.0 : 0.0);
            }
        });
        ok = winverse_definite(Minv, ns, wsc);
    }
    wmatmul<false, false>(AmMi, ns, Am, ns, Minv, ns, na, ns, ns);
    wfor(nm * ns + ns, [&](int e) {
        if (e >= nm * ns) {
            int i = e - nm * ns;
            double sacc = 0;
            for (int t = 0; t < ns; ++t) sacc += Minv[ns * i + t] * np[t];
            Min[i] = sacc;
            return;
        }
        int q = e / ns, t = e - q * ns;
        HMi[e] = ameas_dot_strided(ft, Minv + t, ns, q);   // (A_meas M^-1)[q][t], M^-1 symmetric up to rounding
    });
    wfor(dim * dim, [&](int e) {
        int i = e / dim, j = e - i * dim;
        double v;
        if (i < na && j < na) {
            double sacc = 0;
            for (int t = 0; t < ns; ++t) sacc += AmMi[ns * i + t] * Am[ns * j + t];
            v = -sacc;
        } else if (i >= na && j >= na) {
            v = -ameas_dot(ft, HMi + (i - na) * ns, j - na);
        } else {
            int a = i < na ? i : j, q = i < na ? j : i;
            v = -ameas_dot(ft, AmMi + ns * a, q - na);
        }
        S[e] = v;
    });
    // minus the inverse gains on the diagonal blocks: one lane per block
    // (measured and not kept: the 6 x 6 process gain spread over six lanes with the operations of inv_spd_unrolled — the compiler
    // contracts the fully unrolled serial form differently somewhere, outputs move by 1e-8 against the lane-per-block form below;
    // the block is 12 k of the 250 k cycles of a marginalisation)
    wfor(2 * L + 2, [&](int blk) {
        if (blk < L) {
            double qi[6];
            inv3_sym(r + Rec::qm(nm) + 6 * blk, qi);
            for (int a = 0; a < 3; ++a)
                for (int d = 0; d < 3; ++d) S[(na + 3 * blk + a) * dim + na + 3 * blk + d] -= symget(qi, a, d, 3);
        } else if (blk == L) {
            double Q[36];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) Q[6 * i + j] = symget(r + Rec::QD, i, j, 6);
            inv_spd_unrolled<6>(Q);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) S[i * dim + j] -= Q[6 * i + j];
            for (int i = 0; i < 3; ++i) S[(6 + i) * dim + 6 + i] -= 1.0 / c.Q_bias_dt2[i];
        } else if (blk == L + 1) {
            if (vo) {
                double qi[6];
                inv3_sym(r + Rec::QC, qi);
                for (int a = 0; a < 3; ++a)
                    for (int d = 0; d < 3; ++d) S[(ns + a) * dim + ns + d] -= symget(qi, a, d, 3);
            }
        } else if (ft) {  // process gain of a foot-position state
            const int leg = blk - L - 2;
            double qi[6];
            inv3_sym(r + Rec::qf(nm) + 6 * leg, qi);
            for (int a = 0; a < 3; ++a)
                for (int d = 0; d < 3; ++d) S[(9 + 3 * leg + a) * dim + 9 + 3 * leg + d] -= symget(qi, a, d, 3);
        }
    });
    // u = [ b_dyn ; vo bound ] + Am M^-1 n   |   b_meas + H M^-1 n
    wfor(dim, [&](int i) {
        double v;
        if (i < na) {
            double sacc = 0;
            for (int t = 0; t < ns; ++t) sacc += Am[ns * i + t] * Min[t];
            double rhs;
            if (i < 3) rhs = -0.5 * dt * dt * r[Rec::AS + i];
            else if (i < 6) rhs = -dt * r[Rec::AS + i - 3];
            else if (i < ns) rhs = 0.0;
            else rhs = r[Rec::VOB + i - ns];
            v = rhs + sacc;
        } else {
            v = r[Rec::BM + i - na] + ameas_dot(ft, Min, i - na);
        }
        u[i] = v;
    });
    // S = -([A; H] M^-1 [A; H]' + blkdiag(Q^-1, R^-1)) is negative definite: no pivoting needed — as long as its entries are of
    // comparable size.  With foot-position states they are not: a swinging foot puts dt^2 1e14 on the diagonal next to
    // measurement covariances of 1e-4, and elimination in natural order (small, huge, medium) loses the small entries
    // (measured: an instance turns indefinite after 130 ticks of a 5 Hz gait).  The reference inverts this matrix with
    // Eigen's inverse() = LU with partial pivoting (MheSrb.cpp:588,640); Gauss-Jordan with the same row-pivot rule visits
    // the same pivots.
    if (ft) {
#if DEKF_DEVICE_BUILD
        // (device: the same elimination with one column per lane in registers, bit-identical — inverse_pivoted_regs; dim = 9 + 6 L [+ 3])
        if constexpr (LT > 0) ok = inverse_pivoted_regs<9 + 6 * LT + 3 * (VT > 0 ? 1 : 0)>(S, DEKF_LANE()) && ok;
        else ok = false;
#else
        ok = winverse(S, dim, HMi + nm * ns, true) && ok;
#endif
    } else {
        ok = winverse_definite(S, dim, wsc) && ok;
    }
    wfor(dim, [&](int i) {
        double sacc = 0;
        for (int t = 0; t < dim; ++t) sacc += S[i * dim + t] * u[t];
        Yu[i] = sacc;
    });
    // M+ = -B' S^-1 B, n+ = B' S^-1 u with B = [-I; -[I3 0] (VO); 0]
    wfor(ns * ns + ns, [&](int e) {
        if (e < ns * ns) {
            int i = e / ns, j = e - ns * i;
            double v = S[i * dim + j];
            if (vo) {
                if (i < 3) v += S[(ns + i) * dim + j];
                if (j < 3) v += S[i * dim + ns + j];
                if (i < 3 && j < 3) v += S[(ns + i) * dim + ns + j];
            }
            Mp[e] = -v;
        } else {
            int i = e - ns * ns;
            double v = Yu[i];
            if (vo && i < 3) v += Yu[ns + i];
            np[i] = -v;
        }
    });
    return ok;
}

// marginalizeQP(step): fold window step `step` into (Mp, np)   (MheSrb.cpp:475-713)
DEKF_FN bool marginalize_step(const DevCfg& c, const DevState& s, int b, int step, double* sm) {
    const double* r = s.rec + ((size_t)b * c.wcap + (step % c.wcap)) * c.rec;
    const int L = c.L, ft = c.ft;
    const bool vo = r[Rec::VOF] != 0.0;
#if DEKF_DEVICE_BUILD
    if (!ft) {
        switch (L) {  // wave-uniform
            case 1: return vo ? marginalize_regs<1, true>(c, s, b, r, sm) : marginalize_regs<1, false>(c, s, b, r, sm);
            case 2: return vo ? marginalize_regs<2, true>(c, s, b, r, sm) : marginalize_regs<2, false>(c, s, b, r, sm);
            case 3: return vo ? marginalize_regs<3, true>(c, s, b, r, sm) : marginalize_regs<3, false>(c, s, b, r, sm);
            case 4: return vo ? marginalize_regs<4, true>(c, s, b, r, sm) : marginalize_regs<4, false>(c, s, b, r, sm);
            default: break;
        }
    }
#endif
    if (ft && c.marg_info) return marginalize_info(c, s, b, r, sm);
#if DEKF_DEVICE_BUILD
    if (ft) {
        switch (L) {  // wave-uniform
            case 1: return vo ? marginalize_generic<1, 1>(c, s, b, r, sm) : marginalize_generic<1, 0>(c, s, b, r, sm);
            case 2: return vo ? marginalize_generic<2, 1>(c, s, b, r, sm) : marginalize_generic<2, 0>(c, s, b, r, sm);
            case 3: return vo ? marginalize_generic<3, 1>(c, s, b, r, sm) : marginalize_generic<3, 0>(c, s, b, r, sm);
            case 4: return vo ? marginalize_generic<4, 1>(c, s, b, r, sm) : marginalize_generic<4, 0>(c, s, b, r, sm);
            default: return false;  // (dekf_create refuses more than four legs)
        }
    }
#endif
    (void)vo;
    return marginalize_generic<0, -1>(c, s, b, r, sm);
}

// UpdateMHE part 1: gains of step T - 1 (Q_dyn^-1, R Q_vo R') from its rotation, into its window record.  Depends on nothing
// of step T: assemble_update runs it unless marginalize_early has (same code, same bits).
DEKF_FN void gains_of_previous_step(const DevCfg& c, const DevState& s, int b, int T) {
    double* rprev = s.rec + ((size_t)b * c.wcap + ((T - 1) % c.wcap)) * c.rec;
#if DEKF_DEVICE_BUILD
    {   // step_gains with the 6x6 inverse spread over six lanes (column j of G C G' in lane j)
        const int lane = DEKF_LANE();
        const double* R = rprev + Rec::R;
        double g[6];
        cov6_column(c, R, lane < 6 ? lane : 5, g);
        gj_columns<6>(g, lane);
        if (lane < 6) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (i <= lane) rprev[Rec::QD + (i * (11 - i)) / 2 + lane] = g[i];  // packed upper triangle, entry (i, lane)
        } else if (lane < 12) {
            // Q_cam = R Q_vo R': packed entry e = lane - 6 -> (i, j)
            const int e = lane - 6;
            const int i = e < 3 ? 0 : (e < 5 ? 1 : 2), j = e < 3 ? e : (e < 5 ? e - 2 : 2);
            double acc = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) acc += R[3 * i + t] * c.Q_vo[t] * R[3 * j + t];
            rprev[Rec::QC + e] = acc;
        }
    }
#else
    if (DEKF_LANE() == 0) {
        double qd[21], qc[6];
        step_gains(c, rprev + Rec::R, qd, qc);
        for (int i = 0; i < 21; ++i) rprev[Rec::QD + i] = qd[i];
        for (int i = 0; i < 6; ++i) rprev[Rec::QC + i] = qc[i];
    }
#endif
}

// marginalizeQP(T - N) ahead of time, into (Mp_next, np_next): see DevState::Mp_next.  Same functions, same operands, same order of
// operations as the in-place call of assemble_update — only earlier and into a copy.
DEKF_FN void marginalize_early(const DevCfg& c, const DevState& s, int b, int T, double* sm) {
    const int ns = c.ns, ns2 = c.ns * c.ns;
    gains_of_previous_step(c, s, b, T);
    DEKF_SYNC();
    if (T >= c.N) {
        const double* Mp = s.Mp + (size_t)ns2 * b;
        const double* np = s.np_ + (size_t)ns * b;
        double* Mn = s.Mp_next + (size_t)ns2 * b;
        double* nn = s.np_next + (size_t)ns * b;
        wfor(ns2 + ns, [&](int e) {
            if (e < ns2) Mn[e] = Mp[e];
            else nn[e - ns2] = np[e - ns2];
        });
    }
    DEKF_SYNC();
    if (T >= c.N) {
        DevState s2 = s;
        s2.Mp = s.Mp_next;
        s2.np_ = s.np_next;
        marginalize_step(c, s2, b, T - c.N, sm);
    }
    DEKF_SYNC();
    if (DEKF_LANE() == 0) s.marg_tag[b] = T;
}

// everything update(T) does before initQP/solveQP; T >= 1
DEKF_FN void assemble_update(const DevCfg& c, const DevState& s, int b, int T, int pushes, double* sm) {
    const bool early = s.marg_tag[b] == T;  // wave-uniform: marginalize_early ran for this step
    if (!early) gains_of_previous_step(c, s, b, T);
    DEKF_SYNC();
    const bool vo_rewrite = get_measurement(c, s, b, T, pushes, sm);
    write_measurement_record(c, s, b, T);
    if (T >= c.N) {
        if (!vo_rewrite && early) {  // wave-uniform: the pair computed ahead of time is the pair this step would compute
            const int ns = c.ns, ns2 = c.ns * c.ns;
            double* Mp = s.Mp + (size_t)ns2 * b;
            double* np = s.np_ + (size_t)ns * b;
            const double* Mn = s.Mp_next + (size_t)ns2 * b;
            const double* nn = s.np_next + (size_t)ns * b;
            wfor(ns2 + ns, [&](int e) {
                if (e < ns2) Mp[e] = Mn[e];
                else np[e - ns2] = nn[e - ns2];
            });
        } else {
            marginalize_step(c, s, b, T - c.N, sm);
        }
    }
    // the solve's input snapshot (cfg.h: DevState::snap): the arrival cost as this step leaves it and the VO flag / bound of every
    // ring slot — the two things update(T + 1) rewrites in place while the solve of step T may still be reading them
    DEKF_SYNC();
    {
        const int ns2 = c.ns * c.ns, ns = c.ns;
        const double* Mp = s.Mp + (size_t)ns2 * b;
        const double* np = s.np_ + (size_t)ns * b;
        const double* recb = s.rec + (size_t)b * c.wcap * c.rec;
        double* sn = s.snap + (size_t)c.snap_len * b;
        wfor(c.snap_len, [&](int e) {
            if (e < ns2) sn[e] = Mp[e];
            else if (e < ns2 + ns) sn[e] = np[e - ns2];
            else { const int q = e - ns2 - ns; sn[e] = recb[(size_t)(q >> 2) * c.rec + Rec::VOF + (q & 3)]; }
        });
    }
}

// InitializeMHE (DecentralEst.cpp:200-351): first sample, prior as arrival cost.  With foot-position states the
// prior mean of every foot is its first measurement R p_foot (:310-325), so n = -Q_prior x_prior is not zero.
DEKF_FN void assemble_initialize(const DevCfg& c, const DevState& s, int b, double* sm) {
    get_measurement(c, s, b, 0, 0, sm);
    write_measurement_record(c, s, b, 0);
    DEKF_SYNC();
    const int ns = c.ns;
    double* Mp = s.Mp + (size_t)ns * ns * b;
    double* np = s.np_ + (size_t)ns * b;
    const double* r0 = s.rec + ((size_t)b * c.wcap + 0) * c.rec;
    wfor(ns * ns + ns, [&](int e) {
        if (e < ns * ns) {
            const int i = e / ns;
            Mp[e] = (i == e % ns) ? (i < 9 ? c.Q_prior[i] : c.Q_foot_init[(i - 9) % 3]) : 0.0;
        } else {
            const int i = e - ns * ns;
            np[i] = i < 9 ? 0.0 : -c.Q_foot_init[(i - 9) % 3] * r0[Rec::BM + i - 9];
        }
    });
}

}  // namespace dekf
