// kf_core.h — the Kalman-filter alternative (est_type 1), one wavefront per instance.
// Replaces DecentralizedEstimation::InitializeKF / UpdateKF
//   (src/decentral_legged_est/src/DecentralEst.cpp:592-700, :702-861), including the reference's
//   initialize() running InitializeKF() AND UpdateKF() on the first sample (:140-141).
#pragma once
#include "cfg.h"
#include "mhe_assemble_core.h"
#include "smallmat.h"
#include <type_traits>

namespace dekf {

struct KfScratch {
    DEKF_HD static int len(int L, int ft = 0) {
        const int nm = 3 * L, ns = 9 + (ft ? nm : 0);
        return ns * ns * 3 + nm * nm * 2 + nm + ns * nm * 2 + ns + 32;
    }
};

// x <- A x - b ; C <- A C A' + G C_in G'   with the sample stored in record `r`
// (foot-position states: identity dynamics, process noise dt^2 R C_{slide|swing} R' per foot, DecentralEst.cpp:753)
DEKF_FN void kf_predict(const DevCfg& c, const DevState& s, int b, const double* r, double* sm) {
    const int ns = c.ns, n2 = ns * ns;
    double* x = s.kf_x + (size_t)ns * b;
    double* C = s.kf_C + (size_t)n2 * b;
    double* A = sm;         // ns x ns
    double* AC = A + n2;    // ns x ns
    double* xn = AC + n2;   // ns
    const double* R = r + Rec::R;
    const double dt = c.dt;
    // G C_in G' (DecentralEst.cpp:742-751, 784), entry (i, j)
    auto gcg = [&](int i, int j) -> double {
        const int bi = i / 3, bj = j / 3, a = i % 3, d = j % 3;
        auto rcr = [&](const double* cv) {
            double v = 0;
            for (int t = 0; t < 3; ++t) v += R[3 * a + t] * cv[t] * R[3 * d + t];
            return v;
        };
        if (bi == 0 && bj == 0) return dt * dt * rcr(c.C_p) + 0.25 * dt * dt * dt * dt * rcr(c.C_accel);
        if ((bi == 0 && bj == 1) || (bi == 1 && bj == 0)) return 0.5 * dt * dt * dt * rcr(c.C_accel);
        if (bi == 1 && bj == 1) return dt * dt * rcr(c.C_accel);
        if (bi == 2 && bj == 2 && a == d) return dt * dt * c.C_accel_bias[a];
        if (bi >= 3 && bi == bj) return symget(r + Rec::qf(c.nm) + 6 * (bi - 3), a, d, 3);  // stored as dt^2 R C R'
        return 0.0;
    };
#if DEKF_DEVICE_BUILD
    if (ns == 9 && DEKF_NLANES() == 64) {
        // The dense covariance contraction C <- A C A' on the matrix core (the one place of the path where it pays:
        // tools/probes/cov_mfma_probe.hip, profiles/r02_cov_mfma_probe.txt — 0.70x the time of the LDS-staged form, 0.33x when
        // ALU-bound; for the EKF's 4x4 the matrix core is 3-7x SLOWER than one lane per instance).  One v_mfma_f64_16x16x4_f64 tile
        // holds the 9x9 blocks zero-padded; K = 9 -> three K chunks.  Operand maps: A-operand lane l = M[i = l & 15][k = l >> 4],
        // B-operand lane l = M[k = l >> 4][j = l & 15], result lane l reg r = D[(l >> 4) + 4 r][l & 15]:
        //   T = C A'   A-operand C[i][k],  B-operand A'[k][j] = A[j][k]      chunk c of T as a B-operand IS result register c
        //   C' = A T   A-operand A[i][k],  B-operand T                       (the accumulator re-enters without a move)
        typedef double v4d __attribute__((ext_vector_type(4)));
        const int l = DEKF_LANE(), lo = l & 15, hi = l >> 4;
        double a_op[3], c_op[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int k = 4 * ch + hi;
            const bool in = lo < 9 && k < 9;
            a_op[ch] = in ? adyn_entry(R, dt, lo, k) : 0.0;
            c_op[ch] = in ? C[9 * (in ? lo : 0) + (in ? k : 0)] : 0.0;
        }
        double xnew = 0.0;
        if (l < 9) {
            double acc = 0;
            for (int t = 0; t < 9; ++t) acc += adyn_entry(R, dt, l, t) * x[t];
            const double bd = l < 3 ? -0.5 * dt * dt * r[Rec::AS + l] : (l < 6 ? -dt * r[Rec::AS + l - 3] : 0.0);
            xnew = acc - bd;
        }
        v4d tt = {0.0, 0.0, 0.0, 0.0}, o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(c_op[ch], a_op[ch], tt, 0, 0, 0);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[ch], tt[ch], o, 0, 0, 0);
        // every lane's operand loads completed before the first MFMA issued, so C and x can be overwritten in place
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int row = hi + 4 * rr;
            if (lo < 9 && row < 9) C[9 * row + lo] = o[rr] + gcg(row, lo);
        }
        if (l < 9) x[l] = xnew;
        DEKF_SYNC();
        return;
    }
#endif
    wfor(n2 + ns, [&](int e) {
        if (e < n2) A[e] = adyn_entry(R, dt, e / ns, e % ns);
        else {
            int i = e - n2;
            double acc = 0;
            for (int t = 0; t < ns; ++t) acc += adyn_entry(R, dt, i, t) * x[t];
            double bd = i < 3 ? -0.5 * dt * dt * r[Rec::AS + i] : (i < 6 ? -dt * r[Rec::AS + i - 3] : 0.0);
            xn[i] = acc - bd;
        }
    });
    wmatmul<false, false>(AC, ns, A, ns, C, ns, ns, ns, ns);
    wfor(n2 + ns, [&](int e) {
        if (e >= n2) { x[e - n2] = xn[e - n2]; return; }
        int i = e / ns, j = e - ns * i;
        double acc = 0;
        for (int t = 0; t < ns; ++t) acc += AC[ns * i + t] * A[ns * j + t];
        C[e] = acc + gcg(i, j);
    });
}

// K = C H'(H C H' + C_meas)^-1 ; x += K (b_meas - H x) ; C = (I - K H) C, H = A_meas
DEKF_FN void kf_correct(const DevCfg& c, const DevState& s, int b, const double* r, double* sm) {
    const int nm = c.nm, ns = c.ns, n2 = ns * ns, ft = c.ft;
    double* x = s.kf_x + (size_t)ns * b;
    double* C = s.kf_C + (size_t)n2 * b;
    double* S = sm;                   // nm x nm
    double* wsc = S + nm * nm;        // nm*nm + nm
    double* Kg = wsc + nm * nm + nm;  // ns x nm
    double* HC = Kg + ns * nm;        // nm x ns: A_meas C
    double* Cn = HC + ns * nm;        // ns x ns
    double* xn = Cn + n2;             // ns
    wfor(nm * ns, [&](int e) {
        int q = e / ns, t = e - q * ns;
        HC[e] = ameas_dot_strided(ft, C + t, ns, q);
    });
    wfor(nm * nm, [&](int e) {
        int i = e / nm, j = e - nm * i;
        double v = ameas_dot(ft, HC + i * ns, j);
        if (i / 3 == j / 3) v += symget(r + Rec::qm(nm) + 6 * (i / 3), i % 3, j % 3, 3);
        S[e] = v;
    });
#if DEKF_DEVICE_BUILD
    // H C H' + C_meas is symmetric positive definite: the register-resident Gauss-Jordan of the assemble kernel (one
    // column per lane, no pivot search) instead of the LDS sweep with pivot search
    {
        const int lane = DEKF_LANE();
        auto invert = [&](auto tag) {
            constexpr int NMC = decltype(tag)::value;
            const int j = lane < NMC ? lane : NMC - 1;
            double a[NMC];
#pragma unroll
            for (int i = 0; i < NMC; ++i) a[i] = S[i * NMC + j];
            gj_columns<NMC>(a, lane);
            DEKF_SYNC();
            if (lane < NMC) {
#pragma unroll
                for (int i = 0; i < NMC; ++i) S[i * NMC + lane] = a[i];
            }
            DEKF_SYNC();
        };
        switch (c.L) {  // wave-uniform
            case 1: invert(std::integral_constant<int, 3>()); break;
            case 2: invert(std::integral_constant<int, 6>()); break;
            case 3: invert(std::integral_constant<int, 9>()); break;
            case 4: invert(std::integral_constant<int, 12>()); break;
            default: winverse(S, nm, wsc, true);
        }
    }
#else
    winverse(S, nm, wsc, true);
#endif
    wfor(ns * nm, [&](int e) {  // K = (H C)' S^-1  (C symmetric)
        int i = e / nm, j = e - nm * i;
        double acc = 0;
        for (int t = 0; t < nm; ++t) acc += ameas_dot_strided(ft, C + ns * i, 1, t) * S[t * nm + j];
        Kg[e] = acc;
    });
    wfor(n2 + ns, [&](int e) {
        if (e < n2) {
            int i = e / ns, j = e - ns * i;
            double acc = C[e];
            for (int t = 0; t < nm; ++t) acc -= Kg[i * nm + t] * HC[t * ns + j];
            Cn[e] = acc;
        } else {
            int i = e - n2;
            double acc = x[i];
            for (int t = 0; t < nm; ++t) acc += Kg[i * nm + t] * (r[Rec::BM + t] - ameas_dot(ft, x, t));
            xn[i] = acc;
        }
    });
    wfor(n2 + ns, [&](int e) { if (e < n2) C[e] = Cn[e]; else x[e - n2] = xn[e - n2]; });
}

// v_KF_b_ is only written by update(), not by initialize() (DecentralEst.cpp:189-196)
DEKF_FN void kf_output(const DevCfg& c, const DevState& s, int b, const double* r, bool write_vb) {
    if (DEKF_LANE() == 0) {
        const int ns = c.ns;
        const double* x = s.kf_x + (size_t)ns * b;
        const double p_opti[3] = {0.016041, 0.089061, 0.0579875};
        double wxp[3], t[3], vb[3];
        cross3(r + Rec::GY, p_opti, wxp);
        for (int a = 0; a < 3; ++a) t[a] = x[3 + a] + wxp[a];
        mv3(r + Rec::R, t, vb);
        for (int j = 0; j < ns; ++j) s.x_mhe[(size_t)ns * b + j] = x[j];
        if (write_vb)
            for (int a = 0; a < 3; ++a) s.v_b[3 * (size_t)b + a] = vb[a];
        s.status[b] = DEKF_SOLVE_NONE;
    }
    DEKF_SYNC();
}

// UpdateKF: predict with stack.back(), GetMeasurement(0), correct.  `pushes` samples so far.
DEKF_FN void kf_update(const DevCfg& c, const DevState& s, int b, int pushes, double* sm, bool write_vb = true) {
    const double* rprev = s.rec + ((size_t)b * c.wcap + ((pushes - 1) % c.wcap)) * c.rec;
    kf_predict(c, s, b, rprev, sm);
    get_measurement(c, s, b, 0, pushes, sm);
    write_measurement_record(c, s, b, pushes, false);
    const double* r = s.rec + ((size_t)b * c.wcap + (pushes % c.wcap)) * c.rec;
    kf_correct(c, s, b, r, sm);
    kf_output(c, s, b, r, write_vb);
}

// InitializeKF + UpdateKF, as DecentralizedEstimation::initialize does for est_type 1
DEKF_FN void kf_initialize(const DevCfg& c, const DevState& s, int b, double* sm) {
    get_measurement(c, s, b, 0, 0, sm);
    write_measurement_record(c, s, b, 0, false);
    DEKF_SYNC();
    const int ns = c.ns, n2 = ns * ns;
    double* x = s.kf_x + (size_t)ns * b;
    double* C = s.kf_C + (size_t)n2 * b;
    const double* r0 = s.rec + ((size_t)b * c.wcap + 0) * c.rec;
    wfor(n2 + ns, [&](int e) {
        if (e < n2) { const int i = e / ns; C[e] = (i == e % ns) ? (i < 9 ? c.C_prior[i] : c.C_foot_init[(i - 9) % 3]) : 0.0; }
        else { const int i = e - n2; x[i] = i < 9 ? 0.0 : r0[Rec::BM + i - 9]; }  // prior mean of a foot: its first measurement (:640)
    });
    kf_correct(c, s, b, r0, sm);
    kf_update(c, s, b, 1, sm, false);
}

}  // namespace dekf
