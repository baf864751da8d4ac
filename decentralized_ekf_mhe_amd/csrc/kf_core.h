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
    DEKF_HD static int len(int L) {
        int nm = 3 * L;
        return 81 * 3 + nm * nm * 2 + nm + 9 * nm * 2 + 32;
    }
};

// x <- A x - b ; C <- A C A' + G C_in G'   with the sample stored in record `r`
DEKF_FN void kf_predict(const DevCfg& c, const DevState& s, int b, const double* r, double* sm) {
    double* x = s.kf_x + 9 * (size_t)b;
    double* C = s.kf_C + 81 * (size_t)b;
    double* A = sm;         // 81
    double* AC = A + 81;    // 81
    double* xn = AC + 81;   // 9
    const double* R = r + Rec::R;
    const double dt = c.dt;
    wfor(81 + 9, [&](int e) {
        if (e < 81) A[e] = adyn_entry(R, dt, e / 9, e % 9);
        else {
            int i = e - 81;
            double acc = 0;
            for (int t = 0; t < 9; ++t) acc += adyn_entry(R, dt, i, t) * x[t];
            double bd = i < 3 ? -0.5 * dt * dt * r[Rec::AS + i] : (i < 6 ? -dt * r[Rec::AS + i - 3] : 0.0);
            xn[i] = acc - bd;
        }
    });
    wmatmul<false, false>(AC, 9, A, 9, C, 9, 9, 9, 9);
    wfor(81 + 9, [&](int e) {
        if (e >= 81) { x[e - 81] = xn[e - 81]; return; }
        int i = e / 9, j = e - 9 * i;
        double acc = 0;
        for (int t = 0; t < 9; ++t) acc += AC[9 * i + t] * A[9 * j + t];
        // G C_in G' (DecentralEst.cpp:742-751, 784)
        int bi = i / 3, bj = j / 3, a = i % 3, d = j % 3;
        auto rcr = [&](const double* cv) {
            double v = 0;
            for (int t = 0; t < 3; ++t) v += R[3 * a + t] * cv[t] * R[3 * d + t];
            return v;
        };
        if (bi == 0 && bj == 0) acc += dt * dt * rcr(c.C_p) + 0.25 * dt * dt * dt * dt * rcr(c.C_accel);
        else if ((bi == 0 && bj == 1) || (bi == 1 && bj == 0)) acc += 0.5 * dt * dt * dt * rcr(c.C_accel);
        else if (bi == 1 && bj == 1) acc += dt * dt * rcr(c.C_accel);
        else if (bi == 2 && bj == 2 && a == d) acc += dt * dt * c.C_accel_bias[a];
        C[e] = acc;
    });
}

// K = C H'(H C H' + C_meas)^-1 ; x += K (b_meas - H x) ; C = (I - K H) C, H = A_meas (type 0)
DEKF_FN void kf_correct(const DevCfg& c, const DevState& s, int b, const double* r, double* sm) {
    const int nm = c.nm;
    double* x = s.kf_x + 9 * (size_t)b;
    double* C = s.kf_C + 81 * (size_t)b;
    double* S = sm;                   // nm x nm
    double* wsc = S + nm * nm;        // nm*nm + nm
    double* Kg = wsc + nm * nm + nm;  // 9 x nm
    double* Cn = Kg + 9 * nm;         // 81
    double* xn = Cn + 81;             // 9
    wfor(nm * nm, [&](int e) {
        int i = e / nm, j = e - nm * i;
        double v = C[9 * (3 + i % 3) + 3 + j % 3];
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
    wfor(9 * nm, [&](int e) {
        int i = e / nm, j = e - nm * i;
        double acc = 0;
        for (int t = 0; t < nm; ++t) acc += C[9 * i + 3 + t % 3] * S[t * nm + j];
        Kg[e] = acc;
    });
    wfor(81 + 9, [&](int e) {
        if (e < 81) {
            int i = e / 9, j = e - 9 * i;
            double acc = C[e];
            for (int t = 0; t < nm; ++t) acc -= Kg[i * nm + t] * C[9 * (3 + t % 3) + j];
            Cn[e] = acc;
        } else {
            int i = e - 81;
            double acc = x[i];
            for (int t = 0; t < nm; ++t) acc += Kg[i * nm + t] * (r[Rec::BM + t] - x[3 + t % 3]);
            xn[i] = acc;
        }
    });
    wfor(90, [&](int e) { if (e < 81) C[e] = Cn[e]; else x[e - 81] = xn[e - 81]; });
}

// v_KF_b_ is only written by update(), not by initialize() (DecentralEst.cpp:189-196)
DEKF_FN void kf_output(const DevCfg& c, const DevState& s, int b, const double* r, bool write_vb) {
    if (DEKF_LANE() == 0) {
        const double* x = s.kf_x + 9 * (size_t)b;
        const double p_opti[3] = {0.016041, 0.089061, 0.0579875};
        double wxp[3], t[3], vb[3];
        cross3(r + Rec::GY, p_opti, wxp);
        for (int a = 0; a < 3; ++a) t[a] = x[3 + a] + wxp[a];
        mv3(r + Rec::R, t, vb);
        for (int j = 0; j < 9; ++j) s.x_mhe[9 * (size_t)b + j] = x[j];
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
    double* x = s.kf_x + 9 * (size_t)b;
    double* C = s.kf_C + 81 * (size_t)b;
    wfor(90, [&](int e) {
        if (e < 81) C[e] = (e / 9 == e % 9) ? c.C_prior[e / 9] : 0.0;
        else x[e - 81] = 0.0;
    });
    const double* r0 = s.rec + ((size_t)b * c.wcap + 0) * c.rec;
    kf_correct(c, s, b, r0, sm);
    kf_update(c, s, b, 1, sm, false);
}

}  // namespace dekf
