// RECORD of a measured-and-rejected alternative (round 1, commit 77e7bfc): one-sided block-tridiagonal
// sweeps on v_mfma_f64_16x16x4_f64.  Not compiled into the product (the product factorisation is
// two-sided now; this fragment assumes the one-sided W_k = C_k S_k^-1 convention for all k).
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

