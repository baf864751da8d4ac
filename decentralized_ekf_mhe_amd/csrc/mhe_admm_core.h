// mhe_admm_core.h — the phases of one ADMM iteration of the window QP (included by
// mhe_solve_core.h after SolveCtx; same namespace, same execution model: wave.h).
//
// OSQP's iteration (osqp_solve -> update_xz_tilde / update_x / update_z / update_y; called by the
// reference through MHEproblem::solveQP, src/decentral_legged_est/src/MheSrb.cpp:340-349) is
//     solve  [P + sigma I, A'; A, -1/rho] [xt; nu] = [sigma x - q; z - y/rho]
//     x <- alpha xt + (1-alpha) x ;  z <- clip(alpha zt + (1-alpha) z + y/rho) ;  y <- y + rho (...)
// Here the slack variables v_k, w_k, c_k (each lives in exactly one row, with a block-diagonal P)
// are eliminated analytically, which leaves a block-tridiagonal SPD system in the x_k (9x9 blocks,
// factor: solve_factor) — so one iteration is four phases separated by workgroup barriers:
//
//   X  phase_xcols   x columns : xs = sigma x - q + D .* A_x' w             (w: row vector left by R)
//   S1 phase_sweeps  two wavefronts eliminate towards the middle block     (forward legs)
//   S2               meeting block on wavefront 0 | g_k = S_k^-1 f_k on the other three
//   S3               two wavefronts substitute outwards; x <- alpha u + (1-alpha) x, xd = D .* u
//   R  phase_rows    rows, ONE LANE PER 3-ROW (or 6-row) BLOCK, everything of a row block in registers:
//                    a = A_x xd ; slack solution s = t + S^-1(rho E D a) ; zt = a - E D s ;
//                    x_s, z, y updates ; next slack right-hand side ; t = S^-1 rhs ; w = E (u + rho E D t)
//
// Row-block kinds (Meas leg blocks, Dyn position+velocity 6-blocks, Dyn bias, VO) are mapped to
// wavefront-sized tiles (wtiles): a wavefront executes one kind's straight-line body, and a row
// block never leaves its lane, so phase R needs no LDS exchange and no barrier inside.
#pragma once
// ONE RESULT PER ROBOT, WHATEVER THE BATCH.  The iteration phases exist in several code shapes (row blocks in registers for a chunk of
// iterations: three-workgroup and rows-in-registers kernels; row tiles prefetched per iteration: two-workgroup kernels; plain tile
// loops: the generic kernels), and which shape runs depends on the batch (dekf_create).  Left to -ffp-contract=fast the compiler
// decides per shape which product of an a b + c d it fuses, and the shapes drifted apart by an ulp per iteration (1e-8 in the states
// after 75 iterations with VO weights of 4.4e9).  So this header is compiled with contraction OFF: a product and a sum are rounded
// separately unless the source says fma(), and the hot expressions of the row and x-column phases say it, in the same form in every
// shape (dot3 / fma chains below).  Measured (Go1, B = 4096, A/B on one box): contraction off without the explicit forms +5.3 % solve
// time, with them see EXPERIMENTS.md round 5; r3 == ll bit for bit (tools/r3_identity_check.py, tests/test_gpu_configs.py).
#pragma clang fp contract(off)
// (no namespace block of its own: the including header is inside namespace dekf)

// sum over the rows that touch x_k[a] / x_k[3+a] / x_k[6+a] of A(row, col) * w(row), w(row) already
// carrying the row scaling E[row]
// Branch-free: the neighbour step is clamped to a valid one and its contribution selected away, so
// every load of a gather is issued up front (a branch per neighbour costs an LDS round trip each).
template <class Q, class WF>
DEKF_FN double gather_pcol(const Q& q, int k, int a, WF w) {
    const bool hn = k < q.K - 1, hp = k > 0;
    const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;
    const double n0 = w(q.ix.rd(kn, a)), n1 = w(q.ix.rv(kn, a)), p0 = w(q.ix.rd(kp, a)), p1 = w(q.ix.rv(kp, a));
    double g = (hn ? n0 + n1 : 0.0) - (hp ? p0 + p1 : 0.0);
    if constexpr (Q::FOOT) {  // A_meas = [-I 0 0 .. I ..] (DecentralEst.cpp:106-110)
#pragma unroll
        for (int leg = 0; leg < Q::LEGS; ++leg) g -= w(q.ix.rm(k, 3 * leg + a));
    }
    return g;
}
// foot-position column (leg, a) of step k: its Meas row, the Dyn row of this step (+I) and of the previous one (-I)
template <class Q, class WF>
DEKF_FN double gather_fcol(const Q& q, int k, int la, WF w) {
    const bool hn = k < q.K - 1, hp = k > 0;
    const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;
    const double m0 = w(q.ix.rm(k, la)), n0 = w(q.ix.rd(kn, 9 + la)), p0 = w(q.ix.rd(kp, 9 + la));
    return m0 + (hn ? n0 : 0.0) - (hp ? p0 : 0.0);
}
template <class Q, class WF>
DEKF_FN double gather_vcol(const Q& q, int k, int a, WF w) {
    constexpr int L = Q::LEGS;
    const bool hn = k < q.K - 1, hp = k > 0;
    const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;
    const double n0 = w(q.ix.rd(kn, 3 + a)), n1 = w(q.ix.rd(kn, a)), p0 = w(q.ix.rd(kp, 3 + a));
    double g = 0.0;
    if constexpr (!Q::FOOT) {
#pragma unroll
        for (int leg = 0; leg < L; ++leg) g += w(q.ix.rm(k, 3 * leg + a));
    }
    return g + (hn ? fma(q.c.dt, n1, n0) : 0.0) - (hp ? p0 : 0.0);
}
template <class Q, class WF>
DEKF_FN double gather_bcol(const Q& q, int k, int a, WF w) {
    const bool hn = k < q.K - 1, hp = k > 0;
    const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;
    const double dt = q.c.dt, hdt2 = q.c.hdt2;
    cdptr R = q.R + 9 * kn;
    const int rn = q.ix.rd(kn, 0);
    const double p0 = w(q.ix.rd(kp, 6 + a));
    double g = w(rn + 6 + a);
#pragma unroll
    for (int r = 0; r < 3; ++r) g -= R[3 * r + a] * (hdt2 * w(rn + r) + dt * w(rn + 3 + r));
    return (hn ? g : 0.0) - (hp ? p0 : 0.0);
}

// ---------------------------------------------------------------- X: reduced right-hand side
// xs_k = sigma x_k - q_k + D_k .* (A_x' w)_k, one lane per x entry, one tile kind per column
// kind (position / velocity / bias) because their gathers differ.
template <class Q>
DEKF_FN void phase_xcols(Q& q, double sigma) {
    constexpr int NS = Q::NS, NM = 3 * Q::LEGS, SV = 2 * NS + 3 + NM;
    const int K = q.K, n3 = 3 * K, nt = (n3 + 63) >> 6;
    const int ntf = Q::FOOT ? (NM * K + 63) >> 6 : 0;
    cdptr qsl = q.tmp + TmpMap<NS>::QSL;
    cdptr at = q.at;
    auto w = [&](int r) { return at[r]; };
#if defined(DEKF_PROFILE_TL) && defined(DEKF_PROFILE_TLX)  // -DDEKF_PROFILE_TLX: slots 4.. and 8.. describe phase X instead of the row phase
    const long long tx0 = clock64();
#endif
    wtiles(3 * nt + ntf, [&](int tile, int lane) {
        if (Q::FOOT && tile >= 3 * nt) {  // foot-position columns
            const int e = (tile - 3 * nt) * 64 + lane;
            if (e >= NM * K) return;
            const int k = e / NM, la = e - NM * k, j = 9 + la, i = k * SV + j;
            const double xv = q.x[i], dv = q.D[i], qv = qsl[j];
            const double g = gather_fcol(q, k, la, w);
            q.xs[NS * k + j] = fma(dv, g, fma(sigma, xv, -(k == 0 ? qv : 0.0)));
            return;
        }
        const int kind = tile < nt ? 0 : (tile < 2 * nt ? 1 : 2);
        const int e = (tile - kind * nt) * 64 + lane;
        if (e >= n3) return;
        const int k = e / 3, a = e - 3 * k, j = 3 * kind + a, i = k * SV + j;
        const double xv = q.x[i], dv = q.D[i], qv = qsl[j];
        double g;
        if (kind == 0) g = gather_pcol(q, k, a, w);
        else if (kind == 1) g = gather_vcol(q, k, a, w);
        else {  // gather_bcol with R' (dt^2/2 w_p + dt w_v) already formed by the row phase (q.gb)
            const bool hn = k < K - 1, hp = k > 0;
            const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;
            const double n0 = w(q.ix.rd(kn, 6 + a)), n1 = q.gb[3 * kn + a], p0 = w(q.ix.rd(kp, 6 + a));
            g = (hn ? n0 - n1 : 0.0) - (hp ? p0 : 0.0);
        }
        q.xs[NS * k + j] = fma(dv, g, fma(sigma, xv, -(k == 0 ? qv : 0.0)));
    });
#if defined(DEKF_PROFILE_TL) && defined(DEKF_PROFILE_TLX)
    __builtin_amdgcn_s_waitcnt(0);
    const long long tx1 = clock64();
    DEKF_TL_ADD(q, 4 + (DEKF_LANE() >> 6), tx0, tx1);
#endif
    DEKF_SYNC();
#if defined(DEKF_PROFILE_TL) && defined(DEKF_PROFILE_TLX)
    DEKF_TL_ADD(q, 8 + (DEKF_LANE() >> 6), tx1, clock64());
#endif
}

// ---------------------------------------------------------------- S: block-tridiagonal solve

#if DEKF_DEVICE_BUILD
// rhs - W v for the chain: lane i < 9 holds row i of W in w[0..8] and component i of v; component t of v
// is taken straight from lane t of the 16-lane row by the FMA itself (v_fmac_f64_dpp ... row_newbcast:t,
// gfx950): 9 instructions instead of 18 v_readlane + 9 FMA.  Alone on a CU the step is 169 against 204
// cycles (tools/probes/dpp_chain_probe.hip); in the kernel, where the legs share their SIMDs with the
// other resident workgroup, halving the instruction count of the step is worth 10 % of the whole solve.
// Inline assembly gets no hazard handling from the compiler: a VALU write of v followed by a DPP read of
// it needs two wait states, hence the s_nop inside the first statement (which has v as an input, so the
// nop cannot be scheduled before the instruction that produces v).
DEKF_FN double chain_matvec_dpp(double v, cdptr w, double rhs) {
    // TWO accumulators in alternation: a dependent DPP FMA two slots (32 issue cycles) after its predecessor does not
    // stall, and only one addition is left behind the last FMA (three accumulators and two dependent additions were
    // 1.4 % of the whole solve slower; three FMAs in a row on one accumulator stall)
    double a0 = rhs, a1 = 0.0;
#define DEKF_DPP_FMAC(pre, acc, wt, T) \
    asm volatile(pre "v_fmac_f64_dpp %0, %1, -%2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(wt))
    DEKF_DPP_FMAC("s_nop 1\n\t", a0, w[0], 0); DEKF_DPP_FMAC("", a1, w[1], 1); DEKF_DPP_FMAC("", a0, w[2], 2);
    DEKF_DPP_FMAC("", a1, w[3], 3); DEKF_DPP_FMAC("", a0, w[4], 4); DEKF_DPP_FMAC("", a1, w[5], 5);
    DEKF_DPP_FMAC("", a0, w[6], 6); DEKF_DPP_FMAC("", a1, w[7], 7); DEKF_DPP_FMAC("", a0, w[8], 8);
#undef DEKF_DPP_FMAC
    return a0 + a1;
}
#endif

// One leg of the two-sided block-tridiagonal solve: a chain of `steps` dependent 9x9 mat-vecs
//     v_new = rhs[k_new] - M v_prev,   k_new = k_prev + dk,   M = Wk[k_new + wofs] (TR: transposed)
// starting from the vector stored in xs at block k0.  Forward legs (BWD = false) read rhs from xs
// and overwrite it with f; outward legs (BWD = true) read rhs = g from xd, and leave
// x <- alpha u + (1-alpha) x (the ADMM relaxation of the x blocks) and xd = D .* u.
// Device: the running vector sits in registers of lanes 0..8 of the calling wavefront and is broadcast
// inside the FMAs (chain_matvec_dpp), so the dependent chain never touches LDS or a barrier; all 64 lanes
// execute it (lanes >= 9 mirror lane 8 and never store).  Host build: plain loops.
#ifndef DEKF_GG_RING
#define DEKF_GG_RING 6
#endif
#ifndef DEKF_RT_ROUNDS
#define DEKF_RT_ROUNDS 1  // rounds of the operand ring per loop iteration of the run-time-horizon solve with the factor in the slab (sweeps_one_wave_rt)
#endif
template <bool TR, bool BWD, int STEPS = 0, class Q>
DEKF_FN void sweep_chain(Q& q, int k0, int dk, int steps, int wofs, double alpha) {
    constexpr int SV = 21 + 3 * Q::LEGS;
    dptr xs = q.xs, xd = q.xd, x = q.x;
#if DEKF_DEVICE_BUILD
    const int lane = DEKF_LANE() & 63;
    const int i = lane < 9 ? lane : 8;
    const bool act = lane < 9;
    // Software-pipelined: the operands of step s+1 (a 9-vector of W, rhs, D, x) are requested from
    // LDS before the dependent arithmetic of step s, so the chain sees register operands only.  (The
    // compiler cannot do this itself: it may not move the loads above the step's own LDS store.)
    struct Ops { double w[9], rhs, dsc, xo; };
    auto load = [&](int kn, Ops& o) {
        cdptr W = q.Wk + (kn + wofs) * 81;
#pragma unroll
        for (int t = 0; t < 9; ++t) o.w[t] = TR ? W[9 * t + i] : W[9 * i + t];
        o.rhs = BWD ? xd[9 * kn + i] : xs[9 * kn + i];
        o.dsc = BWD ? q.D[kn * SV + i] : 0.0;
        o.xo = BWD ? x[kn * SV + i] : 0.0;
    };
    double v = xs[9 * k0 + i];
    auto step = [&](const Ops& c, Ops& n, int s, int ahead = 1) {
        const int kn = k0 + s * dk;
        if (s + ahead <= steps) load(kn + ahead * dk, n);
        v = chain_matvec_dpp(v, c.w, c.rhs);  // 5 dependent f64 operations per step
        if (act) {
            if (BWD) {
                xd[9 * kn + i] = c.dsc * v;
                x[kn * SV + i] = relax(alpha, v, c.xo);
            } else {
                xs[9 * kn + i] = v;
            }
        }
    };
    if constexpr (STEPS > 0) {
        // compile-time step count: straight-line code, so every s_waitcnt the compiler inserts counts
        // exactly; across a loop back-edge it waits for lgkmcnt(0), i.e. for the prefetch just issued
        // (tools/probes/chain_probe.hip: 198 against 261 cycles per step, 331 without the pipeline)
        Ops o[2];
        load(k0 + dk, o[0]);
#pragma unroll
        for (int s = 1; s <= STEPS; ++s) {
            const int kn = k0 + s * dk;
            const Ops& c = o[(s - 1) & 1];
            if (s < STEPS) load(kn + dk, o[s & 1]);
            v = chain_matvec_dpp(v, c.w, c.rhs);
            if (act) {
                if (BWD) {
                    xd[9 * kn + i] = c.dsc * v;
                    x[kn * SV + i] = relax(alpha, v, c.xo);
                } else {
                    xs[9 * kn + i] = v;
                }
            }
        }
    } else {
        // run-time step count: a ring of RING operand sets, RING - 1 steps of prefetch in flight.  The compiler waits
        // for every outstanding load at the loop back-edge, so one latency is exposed per RING steps (it was per
        // two with two sets).  Four sets when W sits in LDS; DEKF_GG_RING when it streams from the HBM slab
        // (N = 100: 50-step legs, an L2/HBM round trip per back-edge).
        constexpr int RING = Q::FACTOR_LDS ? 4 : DEKF_GG_RING;
        Ops r[RING];
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
            if (steps >= u + 1) load(k0 + (u + 1) * dk, r[u]);
        int s = 1;
        for (; s + RING - 1 <= steps; s += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) step(r[u], r[(u + RING - 1) % RING], s + u, RING - 1);
        }
        // here r[0 .. RING-2] hold the operands of steps s .. s+RING-2 (as far as they exist)
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
            if (s + u <= steps) step(r[u], r[RING - 1], s + u, RING);
    }
#else
    double v[9], nv[9];
    for (int i = 0; i < 9; ++i) v[i] = xs[9 * k0 + i];
    for (int s = 1; s <= steps; ++s) {
        const int kn = k0 + s * dk;
        cdptr W = q.Wk + (kn + wofs) * 81;
        for (int i = 0; i < 9; ++i) {
            double w[9];
            for (int t = 0; t < 9; ++t) w[t] = TR ? W[9 * t + i] : W[9 * i + t];
            double a0 = (BWD ? xd[9 * kn + i] : xs[9 * kn + i]) - w[0] * v[0];
            double a1 = w[1] * v[1], a2 = w[2] * v[2];
            a0 -= w[3] * v[3]; a1 += w[4] * v[4]; a2 += w[5] * v[5];
            a0 -= w[6] * v[6]; a1 += w[7] * v[7]; a2 += w[8] * v[8];
            nv[i] = a0 - (a1 + a2);
        }
        for (int i = 0; i < 9; ++i) {
            v[i] = nv[i];
            if (BWD) {
                xd[9 * kn + i] = q.D[kn * SV + i] * v[i];
                x[kn * SV + i] = relax(alpha, v[i], x[kn * SV + i]);
            } else {
                xs[9 * kn + i] = v[i];
            }
        }
    }
#endif
}

// The meeting block: f_m -= W^_m f^_{m+1} (the top leg has already folded in W_{m-1} f_{m-1}), then
// u_m = S_m^-1 f_m.  Leaves xs_m = u_m (start vector of both outward legs), xd_m, x_m.
// Device: called for all 64 lanes of one wavefront.  Host build: lane 0 does the whole block.
template <class Q>
DEKF_FN void sweep_mid_block(Q& q, int lane, double alpha) {
    constexpr int SV = 21 + 3 * Q::LEGS;
    const int K = q.K, mid = mid_block(K);
    dptr xs = q.xs, xd = q.xd, x = q.x;
    cdptr Si = q.Sinv + mid * 81;
#if DEKF_DEVICE_BUILD
    const int i = lane < 9 ? lane : 8;
    double f = xs[9 * mid + i];
    if (mid < K - 1) {
        cdptr W = q.Wk + mid * 81 + 9 * i;
        cdptr fh = xs + 9 * (mid + 1);
        double a0 = W[0] * fh[0] + W[3] * fh[3] + W[6] * fh[6];
        double a1 = W[1] * fh[1] + W[4] * fh[4] + W[7] * fh[7];
        double a2 = W[2] * fh[2] + W[5] * fh[5] + W[8] * fh[8];
        f -= a0 + a1 + a2;
    }
    double s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = Si[9 * i + t];
    const double u = -chain_matvec_dpp(f, s, 0.0);  // S_m^-1 f_m, f_m broadcast inside the FMAs
    if (lane < 9) {
        const int xi = mid * SV + i;
        xs[9 * mid + i] = u;
        xd[9 * mid + i] = q.D[xi] * u;
        x[xi] = relax(alpha, u, x[xi]);
    }
#else
    if (lane != 0) return;
    double f[9];
    for (int i = 0; i < 9; ++i) {
        f[i] = xs[9 * mid + i];
        if (mid < K - 1) {
            cdptr W = q.Wk + mid * 81 + 9 * i;
            cdptr fh = xs + 9 * (mid + 1);
            double a0 = W[0] * fh[0] + W[3] * fh[3] + W[6] * fh[6];
            double a1 = W[1] * fh[1] + W[4] * fh[4] + W[7] * fh[7];
            double a2 = W[2] * fh[2] + W[5] * fh[5] + W[8] * fh[8];
            f[i] -= a0 + a1 + a2;
        }
    }
    for (int i = 0; i < 9; ++i) {
        double a0 = Si[9 * i + 0] * f[0] + Si[9 * i + 3] * f[3] + Si[9 * i + 6] * f[6];
        double a1 = Si[9 * i + 1] * f[1] + Si[9 * i + 4] * f[4] + Si[9 * i + 7] * f[7];
        double a2 = Si[9 * i + 2] * f[2] + Si[9 * i + 5] * f[5] + Si[9 * i + 8] * f[8];
        const double u = a0 + a1 + a2;
        const int xi = mid * SV + i;
        xs[9 * mid + i] = u;
        xd[9 * mid + i] = q.D[xi] * u;
        x[xi] = relax(alpha, u, x[xi]);
    }
#endif
}

#if DEKF_DEVICE_BUILD
// ---- cross-row moves inside one wavefront (gfx950 v_permlane16_swap / v_permlane32_swap); row = 16 lanes
DEKF_FN double rows23_from_rows01(double v) {  // [r0 r1 r2 r3] -> [r0 r1 r0 r1]
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
DEKF_FN double row1_from_row0(double v) {  // [r0 r1 r2 r3] -> [r0 r0 r2 r2]
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
DEKF_FN double all_rows_from_row1(double v) {  // [r0 r1 r2 r3] -> [r1 r1 r1 r1]
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // second result: [r1 r1 r3 r3]
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    auto c = __builtin_amdgcn_permlane32_swap(a[1], a[1], false, false);
    auto d = __builtin_amdgcn_permlane32_swap(b[1], b[1], false, false);
    return __hiloint2double((int)d[0], (int)c[0]);
}

DEKF_FN double all_rows_from_row0(double v) {  // [r0 r1 r2 r3] -> [r0 r0 r0 r0]
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // first result: [r0 r0 r2 r2]
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    auto c = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false);
    auto d = __builtin_amdgcn_permlane32_swap(b[0], b[0], false, false);
    return __hiloint2double((int)d[0], (int)c[0]);
}
DEKF_FN double rows01_from_rows23(double v) {  // [r0 r1 r2 r3] -> [r2 r3 r2 r3]
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}

// The whole two-sided block-tridiagonal solve on ONE wavefront, its four 16-lane rows in lock step, with no
// workgroup barrier inside (the two-wavefront form needs three: legs | meeting block + g | legs):
//   row 0  top leg        f_k = b_k - W_{k-1} f_{k-1}          row 2  g_{k-1} = S_{k-1}^-1 f_{k-1} (top half)
//   row 1  bottom leg     f^_k = b_k - W^_k f^_{k+1}           row 3  g of the bottom half
// One DPP-broadcast mat-vec (9 v_fmac_f64_dpp row_newbcast) per step serves all four rows: the rows differ
// only in the operand matrix a lane has loaded (a row of W or of S^-1).  Rows 2 and 3 take their broadcast
// source from rows 0 and 1 through v_permlane32_swap.  Then the meeting block in row 0 (f^_{m+1} comes over
// from row 1 by v_permlane16_swap), u_m is copied to row 1, and rows 0 and 1 substitute outwards.
// Needs an even compile-time horizon (both forward legs equally long) and the full window.
// lane_in >= 0: the caller's lane id, read ONCE outside its iteration loop (the three-workgroup kernels' solve loop): the per-lane
// pointers and strides below are then loop-invariant for the compiler, which may keep them in registers across the iterations —
// this wavefront's loop holds nothing else.  (Everywhere else the lane id stays opaque at every use, see DEKF_LANE.)
template <int NF, class Q>
DEKF_FN void sweeps_one_wave(Q& q, double alpha, int lane_in = -1) {
    constexpr int SV = 21 + 3 * Q::LEGS, K = NF, M = mid_block(NF);
    static_assert(NF % 2 == 0 && K - 2 - M == M, "equal forward legs");
    const int lane = lane_in >= 0 ? lane_in : (DEKF_LANE() & 63), row = lane >> 4, li = lane & 15;
    const int i = li < 9 ? li : 8;
    const bool act = li < 9, leg = row < 2, top = (row & 1) == 0;
    // the x blocks of x and D: inside the full variable vector (stride SV per step), or compact (R3: [K][9] in LDS)
    constexpr int XST = Q::R3 ? 9 : SV;
    dptr xs = q.xs, xd = q.xd, x = Q::R3 ? q.xb : q.x;
    cdptr Dx = Q::R4 ? q.Dxb : q.D;  // inside the full scaling vector (stride SV); R4: the compact LDS copy of the x blocks' entries (D is in the slab)
    constexpr int DST = Q::R4 ? 9 : SV;
    struct Ops { double w[9], rhs; };
    // ---------------- forward: step s = 1..M
    cdptr fm = (leg ? q.Wk : q.Sinv) + ((top ? 0 : (leg ? K - 2 : K - 1)) * 81 + 9 * i);  // block of step 1
    const int fstep = top ? 81 : -81;
    q.tmp[260 + lane] = 0.0;  // a zero the rows without a right-hand side can load (tmp[257..338) is free on the device)
    const int rstep = top ? 9 : -9;
    cdptr fr = leg ? xs + (top ? 9 : 9 * (K - 2)) + i : q.tmp + 260 + lane;  // rhs of step 1 (legs) | 0
    const int frstep = leg ? rstep : 0;
    // Stores go through per-lane pointers and strides, lanes with nothing to store aim at a private dummy slot
    // (tmp[176 + lane], free between factorisations): no EXEC juggling and no select inside a step.
    dptr dummy = q.tmp + 176 + lane;
    const bool gown = !leg && act;
    dptr gst = gown ? xd + (top ? 0 : 9 * (K - 1)) + i : dummy;  // where rows 2, 3 put -g of step 1
    const int gstep = gown ? rstep : 0;
    auto fload = [&](int s, Ops& o) {
        cdptr W = fm + (s - 1) * fstep;
#pragma unroll
        for (int t = 0; t < 9; ++t) o.w[t] = W[t];
        o.rhs = fr[(s - 1) * frstep];
    };
    double v = xs[(top ? 0 : 9 * (K - 1)) + i];  // f_0 = b_0 / f^_{K-1} = b_{K-1}
    {
        Ops o[2];
        fload(1, o[0]);
#pragma unroll
        for (int s = 1; s <= M; ++s) {
            const Ops& c = o[(s - 1) & 1];
            if (s < M) fload(s + 1, o[s & 1]);
            // (letting rows 2, 3 lag one step, so that this cross-row move leaves the dependent path, was
            // measured 1.4 % slower: one more live register and a select per step)
            const double src = rows23_from_rows01(v);
            const double r = chain_matvec_dpp(src, c.w, c.rhs);
            v = r;                        // rows 2, 3 never read their own v
            gst[(s - 1) * gstep] = r;     // -g_{s-1} (row 2) / -g_{K-s} (row 3)
        }
    }
    // ---------------- joint middle: blocks M and M + 1 together (solve_factor left P12 in W[M] and P22 in S^-1[M + 1])
    //   row 0: P11 f_M        row 1: P22 f^_{M+1}        row 2: P12 f^_{M+1}        row 3: P12' f_M
    //   u_M = row 0 + row 2,  u_{M+1} = row 1 + row 3
    {
        double w[9];
        cdptr W = row == 0 ? q.Sinv + M * 81 + 9 * i : (row == 1 ? q.Sinv + (M + 1) * 81 + 9 * i : q.Wk + M * 81 + (row == 2 ? 9 * i : i));
        const int wst = row == 3 ? 9 : 1;  // row 3 reads P12 transposed
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = W[t * wst];
        const int blk = top ? M : M + 1;                          // rows 0 / 1 own blocks M / M + 1
        const double dm = Dx[blk * DST + i], xm = x[blk * XST + i];
        const double f0 = all_rows_from_row0(v), f1 = all_rows_from_row1(v);
        const double src = (row == 0 || row == 3) ? f0 : f1;
        const double part = -chain_matvec_dpp(src, w, 0.0);       // (chain_matvec_dpp returns rhs - W v)
        const double um = part + rows01_from_rows23(part);        // rows 0, 1: u_M, u_{M+1}
        if (leg && act) {
            xd[9 * blk + i] = dm * um;
            x[blk * XST + i] = relax(alpha, um, xm);
        }
        v = um;
    }
    // ---------------- outward: step s = 1..M, row 0 block M - s, row 1 block M + 1 + s
    {
        struct Bops { double w[9], ng, dsc, xo; };
        static_assert(K - 2 - M == M, "both outward legs M steps");
        const bool own = leg && act;
        const int b0 = top ? M : M + 1;
        dptr xdp = own ? xd + 9 * b0 + i : dummy;       // the leg's start block; step s is at +- s blocks
        dptr xp = own ? x + b0 * XST + i : dummy;
        cdptr Dp = own ? Dx + b0 * DST + i : dummy;
        const int xdstep = own ? (top ? -9 : 9) : 0, xstep = own ? (top ? -XST : XST) : 0, dstep = own ? (top ? -DST : DST) : 0;
        cdptr Wp = q.Wk + M * 81 + i;              // row 0: W_{M-s}', row 1: W^_{M+s}'
        const int wstep = top ? -81 : 81;
        auto bload = [&](int s, Bops& o) {
            cdptr W = Wp + s * wstep;
#pragma unroll
            for (int t = 0; t < 9; ++t) o.w[t] = W[9 * t];
            o.ng = xdp[s * xdstep];
            o.dsc = Dp[s * dstep];
            o.xo = xp[s * xstep];
        };
        Bops o[2];
        bload(1, o[0]);
#pragma unroll
        for (int s = 1; s <= M; ++s) {
            const Bops& c = o[(s - 1) & 1];
            if (s < M) bload(s + 1, o[s & 1]);
            const double r = chain_matvec_dpp(v, c.w, -c.ng);
            v = r;
            xdp[s * xdstep] = c.dsc * r;
            xp[s * xstep] = relax(alpha, r, c.xo);
        }
    }
}

// The same one-wavefront, four-row solve for a RUN-TIME (even) window length: the steps are loops, the operands of the next
// RING - 1 steps are in flight in a ring of register sets (the compiler waits for everything outstanding at a loop back-edge, so
// one load latency is exposed per RING steps).  Used by the generic instantiations — PogoX's 100-step window, whose factor
// streams from the HBM slab (DEKF_GG_RING sets), other horizons, the window-fill ticks — instead of the three-phase two-wavefront
// form (legs | g + meeting block | legs): no barrier inside, the g streams ride along for free.
template <class Q>
DEKF_FN void sweeps_one_wave_rt(Q& q, double alpha) {
    constexpr int SV = 21 + 3 * Q::LEGS;
    // (RR — rows in registers, two workgroups per CU — has the registers for eight sets and the latency to hide: 171 k -> 177 k
    // steps/s on PogoX against six; ten spill)
    constexpr int RING = Q::FACTOR_LDS ? 4 : (Q::R3 ? 8 : DEKF_GG_RING);
    const int K = q.K, M = mid_block(K), NOUT = K - 1 - M;  // K even: K - 2 - M == M, both forward legs M steps
    // (the lane id stays opaque here: taken from q.lane0, i.e. loop-invariant, the run-time ring of operand sets and the hoisted
    // addresses together spill 63 VGPRs of the 256 and PogoX drops from 102 k to 80 k steps/s)
    const int lane = DEKF_LANE() & 63, row = lane >> 4, li = lane & 15;
    const int i = li < 9 ? li : 8;
    const bool act = li < 9, leg = row < 2, top = (row & 1) == 0;
    // the x blocks: inside the full variable vector (stride SV per step), or compact (rows in registers: [K][9] in LDS); D is
    // always indexed inside the full scaling vector
    constexpr int XST = Q::R3 ? 9 : SV;
    dptr xs = q.xs, xd = q.xd, x = Q::R3 ? q.xb : q.x;
    struct Ops { double w[9], rhs; };
    // ---------------- forward: step s = 1..M
    cdptr fm = (leg ? q.Wk : q.Sinv) + ((top ? 0 : (leg ? K - 2 : K - 1)) * 81 + 9 * i);  // block of step 1
    const int fstep = top ? 81 : -81;
    q.tmp[260 + lane] = 0.0;  // a zero the rows without a right-hand side can load
    const int rstep = top ? 9 : -9;
    cdptr fr = leg ? xs + (top ? 9 : 9 * (K - 2)) + i : q.tmp + 260 + lane;
    const int frstep = leg ? rstep : 0;
    dptr dummy = q.tmp + 176 + lane;
    const bool gown = !leg && act;
    dptr gst = gown ? xd + (top ? 0 : 9 * (K - 1)) + i : dummy;  // where rows 2, 3 put -g of step 1
    const int gstep = gown ? rstep : 0;
    // (the operands of step s sit s - 1 lane-dependent strides behind those of step 1, and the loads are issued in the order of the
    // steps: running pointers, one addition per array and step — as s * stride every address cost a quarter-rate 32-bit multiply
    // on the wavefront whose issue slots are the iteration's critical path, nine of them per pair of steps)
    cdptr fmr = fm, frr = fr;
    auto fload = [&](int, Ops& o) {
#pragma unroll
        for (int t = 0; t < 9; ++t) o.w[t] = fmr[t];
        o.rhs = frr[0];
        fmr = fmr + fstep;
        frr = frr + frstep;
    };
    double v = xs[(top ? 0 : 9 * (K - 1)) + i];  // f_0 = b_0 / f^_{K-1} = b_{K-1}
    // UNCONDITIONAL operand loads (round 6).  Every step loads the operands of the step RING - 1 ahead of it — also in the last RING - 1
    // steps of a leg, where those lie beyond the leg's end: the two legs meet in the middle of arrays that span the whole window, so
    // the reads stay inside the arrays (forward) or inside the workgroup's slab / LDS (outward; Gws::total carries RING blocks of slack
    // behind the last array) and their values are never used.  A load behind a branch — even a scalar one that is never taken — makes
    // the number of loads in flight path-dependent for the compiler's wait-count pass, which then drains the whole pipeline of operand
    // sets (s_waitcnt vmcnt(0)) in front of every round of the ring; and it drains at every loop header anyway, so a loop iteration
    // runs ROUNDS rounds (RING * ROUNDS steps).  PogoX: 14 drains of the slab latency per ADMM iteration before, 4-5 now.
    // (-DDEKF_BOUNDS keeps the guarded loads: the checked build counts every out-of-range read)
    // (the LDS-resident factor keeps them too: its last array ends where the workgroup's LDS ends)
#ifdef DEKF_BOUNDS
    constexpr bool GUARD = true;
#else
    constexpr bool GUARD = Q::FACTOR_LDS;
#endif
    constexpr int ROUNDS = Q::FACTOR_LDS ? 1 : DEKF_RT_ROUNDS;
    {
        Ops r[RING];
        dptr gsr = gst;
        auto fcompute = [&](const Ops& c) {
            const double src = rows23_from_rows01(v);
            const double res = chain_matvec_dpp(src, c.w, c.rhs);
            v = res;
            gsr[0] = res;
            gsr = gsr + gstep;
        };
        auto fstepf = [&](const Ops& c, Ops& n, int s, int ahead) {
            if (!GUARD || s + ahead <= M) fload(s + ahead, n);
            fcompute(c);
        };
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
            if (!GUARD || M >= u + 1) fload(u + 1, r[u]);
        int s = 1;
        if constexpr (ROUNDS > 1) {
            for (; s + RING * ROUNDS - 1 <= M; s += RING * ROUNDS) {
#pragma unroll
                for (int rep = 0; rep < ROUNDS; ++rep)
#pragma unroll
                    for (int u = 0; u < RING; ++u) fstepf(r[u], r[(u + RING - 1) % RING], s + rep * RING + u, RING - 1);
            }
        }
        for (; s + RING - 1 <= M; s += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) fstepf(r[u], r[(u + RING - 1) % RING], s + u, RING - 1);
        }
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
            if (s + u <= M) fcompute(r[u]);   // (the operands of the last RING - 1 steps are in flight already)
    }
    // ---------------- meeting block (row 0) and the last g of the bottom half (row 3)
    {
        double w[9];
        cdptr W = row == 0 ? q.Wk + M * 81 + 9 * i : q.Sinv + (M + 1) * 81 + 9 * i;
        const double z = (row == 0 || row == 3) ? 1.0 : 0.0;
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = z * W[t];
        double s9[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) s9[t] = q.Sinv[M * 81 + 9 * i + t];
        const double dm = q.D[M * SV + i], xm = x[M * XST + i];
        const double src = all_rows_from_row1(v);                 // f^_{M+1} everywhere
        const double res = chain_matvec_dpp(src, w, row == 0 ? v : 0.0);
        if (row == 3 && act) xd[9 * (M + 1) + i] = res;            // -g_{M+1}
        const double um = -chain_matvec_dpp(res, s9, 0.0);         // row 0: S_M^-1 (f_M - W^_M f^_{M+1})
        if (row == 0 && act) {
            xd[9 * M + i] = dm * um;
            x[M * XST + i] = relax(alpha, um, xm);
        }
        v = row1_from_row0(um);                                    // rows 0 and 1 start from u_M
    }
    // ---------------- outward: step s = 1..NOUT, row 0 block M - s (s <= M), row 1 block M + s
    {
        struct Bops { double w[9], ng, dsc, xo; };
        const bool own = leg && act;
        dptr xdp = own ? xd + 9 * M + i : dummy;
        dptr xp = own ? x + M * XST + i : dummy;
        cdptr Dp = q.D + (own ? M * SV + i : 0);     // (D may sit in HBM: no LDS dummy here, idle lanes re-read D[0])
        const int xdstep = own ? (top ? -9 : 9) : 0, xstep = own ? (top ? -XST : XST) : 0, dstep = own ? (top ? -SV : SV) : 0;
        // (with the factor in the slab the column comes as a ROW of the transposed copy the factorisation leaves next to W: 72
        // contiguous bytes per lane, five loads instead of nine on the wavefront whose issue slots are the critical path)
        constexpr bool WTB = !Q::FACTOR_LDS;
        cdptr Wp = (WTB ? q.WT : q.Wk) + (top ? M : M - 1) * 81 + (WTB ? 9 * i : i);  // row 0: W_{M-s}', row 1: W^_{M+s-1}'
        const int wstep = top ? -81 : 81;
        cdptr Wr = Wp, Dr = Dp;
        dptr xdr = xdp, xr = xp;       // operands of the step being loaded
        dptr xdw = xdp, xw = xp;       // results of the step being computed
        auto bload = [&](int s, Bops& o) {
            const bool adv = s <= M || !top;  // row 0 has one step less: its last load repeats block 0
            Wr = Wr + (adv ? wstep : 0);
            xdr = xdr + (adv ? xdstep : 0);
            Dr = Dr + (adv ? dstep : 0);
            xr = xr + (adv ? xstep : 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) o.w[t] = WTB ? Wr[t] : Wr[9 * t];
            o.ng = xdr[0];
            o.dsc = Dr[0];
            o.xo = xr[0];
        };
        Bops r[RING];
        auto bcompute = [&](const Bops& c, int s) {
            const double res = chain_matvec_dpp(v, c.w, -c.ng);
            v = res;
            xdw = xdw + xdstep;
            xw = xw + xstep;
            // every lane stores in every step, the ones with nothing to store into their dummy word (lanes that own no component
            // point there anyway; the top leg's lanes in the one step the bottom leg is longer): an address select instead of
            // two guarded copies of the stores with their exec-mask bookkeeping
            const bool real = s <= M || !top;
            dptr dxd = real ? xdw : dummy, dx = real ? xw : dummy;
            dxd[0] = c.dsc * res;
            dx[0] = relax(alpha, res, c.xo);
        };
        auto bstepf = [&](const Bops& c, Bops& n, int s, int ahead) {
            if (!GUARD || s + ahead <= NOUT) bload(s + ahead, n);
            bcompute(c, s);
        };
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
            if (!GUARD || NOUT >= u + 1) bload(u + 1, r[u]);
        int s = 1;
        if constexpr (ROUNDS > 1) {
            for (; s + RING * ROUNDS - 1 <= NOUT; s += RING * ROUNDS) {
#pragma unroll
                for (int rep = 0; rep < ROUNDS; ++rep)
#pragma unroll
                    for (int u = 0; u < RING; ++u) bstepf(r[u], r[(u + RING - 1) % RING], s + rep * RING + u, RING - 1);
            }
        }
        for (; s + RING - 1 <= NOUT; s += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) bstepf(r[u], r[(u + RING - 1) % RING], s + u, RING - 1);
        }
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
            if (s + u <= NOUT) bcompute(r[u], s + u);
    }
}
#endif

// ---------------------------------------------------------------- S for state blocks wider than a DPP row
// With foot-position states a block has NS = 9 + 3 L entries (21 for Go1): its running vector no longer fits the 16
// lanes a DPP row broadcast reaches.  TWO ROWS PER BLOCK: row 2s holds components 0..15 of side s, row 2s + 1 components
// 16..NS-1; every lane keeps BOTH halves of the running vector of its side (va: lane li holds v[li], vb: lane li holds
// v[16 + li]), so a lane's NS products are NS DPP-broadcast FMAs on its own row (16 on va, NS - 16 on vb), and ONE
// v_permlane16_swap per 32-bit half redistributes the new components (even rows -> va of the pair, odd rows -> vb).
// Rows 0/1 run the top leg and rows 2/3 the bottom leg of the two-sided solve in lock step: both legs of a phase on one
// wavefront, no workgroup barrier inside a phase.  Host build: plain loops.
#if DEKF_DEVICE_BUILD
struct RowPairVec { double va, vb; };
DEKF_FN RowPairVec pair_redistribute(double own) {  // rows [r0 r1 r2 r3] -> va = [r0 r0 r2 r2], vb = [r1 r1 r3 r3]
    unsigned lo = (unsigned)__double2loint(own), hi = (unsigned)__double2hiint(own);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    RowPairVec r;
    r.va = __hiloint2double((int)b[0], (int)a[0]);
    r.vb = __hiloint2double((int)b[1], (int)a[1]);
    return r;
}
template <int NS>
DEKF_FN double pair_matvec_dpp(double va, double vb, cdptr w, double rhs) {
    static_assert(NS > 9 && NS <= 32, "two rows per block");
    double a0 = rhs, a1 = 0.0;
#define DEKF_PFMAC(pre, acc, src, T, IDX)                                                                          \
    if constexpr (NS > IDX)                                                                                        \
        asm volatile(pre "v_fmac_f64_dpp %0, %1, -%2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(w[IDX]))
    DEKF_PFMAC("s_nop 1\n\t", a0, va, 0, 0); DEKF_PFMAC("", a1, va, 1, 1); DEKF_PFMAC("", a0, va, 2, 2); DEKF_PFMAC("", a1, va, 3, 3);
    DEKF_PFMAC("", a0, va, 4, 4); DEKF_PFMAC("", a1, va, 5, 5); DEKF_PFMAC("", a0, va, 6, 6); DEKF_PFMAC("", a1, va, 7, 7);
    DEKF_PFMAC("", a0, va, 8, 8); DEKF_PFMAC("", a1, va, 9, 9); DEKF_PFMAC("", a0, va, 10, 10); DEKF_PFMAC("", a1, va, 11, 11);
    DEKF_PFMAC("", a0, va, 12, 12); DEKF_PFMAC("", a1, va, 13, 13); DEKF_PFMAC("", a0, va, 14, 14); DEKF_PFMAC("", a1, va, 15, 15);
    DEKF_PFMAC("s_nop 1\n\t", a0, vb, 0, 16); DEKF_PFMAC("", a1, vb, 1, 17); DEKF_PFMAC("", a0, vb, 2, 18); DEKF_PFMAC("", a1, vb, 3, 19);
    DEKF_PFMAC("", a0, vb, 4, 20); DEKF_PFMAC("", a1, vb, 5, 21); DEKF_PFMAC("", a0, vb, 6, 22); DEKF_PFMAC("", a1, vb, 7, 23);
    DEKF_PFMAC("", a0, vb, 8, 24); DEKF_PFMAC("", a1, vb, 9, 25); DEKF_PFMAC("", a0, vb, 10, 26); DEKF_PFMAC("", a1, vb, 11, 27);
    DEKF_PFMAC("", a0, vb, 12, 28); DEKF_PFMAC("", a1, vb, 13, 29); DEKF_PFMAC("", a0, vb, 14, 30); DEKF_PFMAC("", a1, vb, 15, 31);
#undef DEKF_PFMAC
    return a0 + a1;
}
#endif

// both legs of one phase of the two-sided solve.  Forward (BWD = false): f_k = b_k - W_{k-1} f_{k-1} downwards from block 0
// and f^_k = b_k - W^_k f^_{k+1} upwards from block K-1, in xs.  Outward (BWD = true): u_k = g_k - W_k' u_{k+1} from the
// meeting block to block 0 and u_k = g_k - W^_{k-1}' u_{k-1} to block K-1; leaves xd = D .* u and the relaxed x.
template <bool BWD, class Q>
DEKF_FN void sweep_legs_generic(Q& q, double alpha) {
    constexpr int NS = Q::NS, NS2 = Q::NS2, SV = 2 * NS + 3 + 3 * Q::LEGS;
    const int K = q.K, mid = mid_block(K);
    dptr xs = q.xs, xd = q.xd, x = q.x;
    // side 0 (top): blocks k0 + s dk; the matrix of step s is Wk[kn + wofs], transposed on the way out
    const int k0s[2] = {BWD ? mid : 0, BWD ? mid : K - 1};
    const int dks[2] = {BWD ? -1 : 1, BWD ? 1 : -1};
    const int nst[2] = {mid, BWD ? K - 1 - mid : K - 2 - mid};
    const int wof[2] = {BWD ? 0 : -1, BWD ? -1 : 0};
#if DEKF_DEVICE_BUILD
    const int lane = q.lane0, row = lane >> 4, li = lane & 15, side = row >> 1, half = row & 1;
    const int ic = half * 16 + li;
    const bool act = ic < NS;
    const int i = act ? ic : NS - 1;
    const int k0 = k0s[side], dk = dks[side], steps = nst[side], wofs = wof[side];
    const int smax = nst[0] > nst[1] ? nst[0] : nst[1];
    struct Ops { double w[NS], rhs, dsc, xo; };
    auto load = [&](int s, Ops& o) {
        const int sc = s <= steps ? s : (steps > 0 ? steps : 1);  // a finished (or empty) side re-reads a valid block
        const int kn = steps > 0 ? k0 + sc * dk : k0;
        const int kw = steps > 0 ? kn + wofs : (k0 + wofs >= 0 && k0 + wofs < K - 1 ? k0 + wofs : 0);
        cdptr W = q.Wk + kw * NS2;
#pragma unroll
        for (int t = 0; t < NS; ++t) o.w[t] = BWD ? W[NS * t + i] : W[NS * i + t];
        o.rhs = BWD ? xd[NS * kn + i] : xs[NS * kn + i];
        o.dsc = BWD ? q.D[kn * SV + i] : 0.0;
        o.xo = BWD ? x[kn * SV + i] : 0.0;
    };
    // (lanes beyond the block's last component mirror it: with NS < 16 a plain xs[NS k0 + li] read past the last block of the array,
    // found by the -DDEKF_BOUNDS build; the mirrored values are never broadcast)
    double va = xs[NS * k0 + (li < NS ? li : NS - 1)], vb = xs[NS * k0 + (16 + li < NS ? 16 + li : NS - 1)];
    auto step = [&](int s, const Ops& c) {
        const double own = pair_matvec_dpp<NS>(va, vb, c.w, c.rhs);
        const RowPairVec nv = pair_redistribute(own);
        va = nv.va;
        vb = nv.vb;
        if (act && s <= steps) {
            const int kn = k0 + s * dk;
            if (BWD) {
                xd[NS * kn + i] = c.dsc * own;
                x[kn * SV + i] = relax(alpha, own, c.xo);
            } else {
                xs[NS * kn + i] = own;
            }
        }
    };
    if (smax > 0) {
        // ring of operand sets, RING - 1 steps of prefetch in flight: the factor of these shapes streams from the HBM slab
        // (two sets, one step ahead, left a memory round trip exposed in every step: 1.1-1.3 k cycles per step measured)
        // (round 4: four sets — with machine LICM off the kernel has the registers, 207 -> 255 VGPRs: 275 k -> 279 k steps/s on Go1)
        constexpr int RING = 4;
        Ops r[RING];
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
            if (smax >= u + 1) load(u + 1, r[u]);
        int s = 1;
        for (; s + RING - 1 <= smax; s += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                // (unconditional, clamped loads as in sweeps_one_wave_rt were measured here too, round 6: 290.7 k -> 283.7 k steps/s on
                // Go1 with foot states — this shape runs against the slab stream itself, every extra block read costs)
                if (s + u + RING - 1 <= smax) load(s + u + RING - 1, r[(u + RING - 1) % RING]);
                step(s + u, r[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
            if (s + u <= smax) step(s + u, r[u]);
    }
#else
    for (int side = 0; side < 2; ++side) {
        const int k0 = k0s[side], dk = dks[side], steps = nst[side], wofs = wof[side];
        double v[NS], nv[NS];
        for (int i = 0; i < NS; ++i) v[i] = xs[NS * k0 + i];
        for (int s = 1; s <= steps; ++s) {
            const int kn = k0 + s * dk;
            cdptr W = q.Wk + (kn + wofs) * NS2;
            for (int i = 0; i < NS; ++i) {
                double a0 = BWD ? xd[NS * kn + i] : xs[NS * kn + i], a1 = 0.0;
                for (int t = 0; t < NS; ++t) {
                    const double wt = BWD ? W[NS * t + i] : W[NS * i + t];
                    if (t & 1) a1 -= wt * v[t]; else a0 -= wt * v[t];
                }
                nv[i] = a0 + a1;
            }
            for (int i = 0; i < NS; ++i) {
                v[i] = nv[i];
                if (BWD) {
                    xd[NS * kn + i] = q.D[kn * SV + i] * v[i];
                    x[kn * SV + i] = relax(alpha, v[i], x[kn * SV + i]);
                } else {
                    xs[NS * kn + i] = v[i];
                }
            }
        }
    }
#endif
}

// the meeting block for any NS: f_m -= W^_m f^_{m+1}, u_m = S_m^-1 f_m (f_m passes through LDS between the two products)
template <class Q>
DEKF_FN void sweep_mid_block_generic(Q& q, int lane, double alpha) {
    constexpr int NS = Q::NS, NS2 = Q::NS2, SV = 2 * NS + 3 + 3 * Q::LEGS;
    const int K = q.K, mid = mid_block(K);
    dptr xs = q.xs, xd = q.xd, x = q.x;
    cdptr Si = q.Sinv + mid * NS2;
    dptr ft = q.tmp + TmpMap<NS>::SIDE0;  // factor-time scratch, free during the iterations
#if DEKF_DEVICE_BUILD
    const int i = lane < NS ? lane : NS - 1;
    // row i of S_m^-1 is requested FIRST: it does not depend on the product in front of it, and with the factor in the HBM slab the
    // two dependent mat-vecs otherwise pay two memory round trips (the fence between them keeps the compiler from hoisting the loads)
    double si[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) si[t] = Si[NS * i + t];
    double f = xs[NS * mid + i];
    if (mid < K - 1) {
        cdptr W = q.Wk + mid * NS2 + NS * i;
        cdptr fh = xs + NS * (mid + 1);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int t = 0; t + 2 < NS; t += 3) { a0 += W[t] * fh[t]; a1 += W[t + 1] * fh[t + 1]; a2 += W[t + 2] * fh[t + 2]; }
        f -= a0 + (a1 + a2);
    }
    if (lane < NS) ft[lane] = f;
    wave_sync();
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int t = 0; t + 2 < NS; t += 3) { a0 += si[t] * ft[t]; a1 += si[t + 1] * ft[t + 1]; a2 += si[t + 2] * ft[t + 2]; }
    const double u = a0 + (a1 + a2);
    if (lane < NS) {
        const int xi = mid * SV + i;
        xs[NS * mid + i] = u;
        xd[NS * mid + i] = q.D[xi] * u;
        x[xi] = relax(alpha, u, x[xi]);
    }
#else
    if (lane != 0) return;
    for (int i = 0; i < NS; ++i) {
        double f = xs[NS * mid + i];
        if (mid < K - 1) {
            cdptr W = q.Wk + mid * NS2 + NS * i;
            cdptr fh = xs + NS * (mid + 1);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int t = 0; t + 2 < NS; t += 3) { a0 += W[t] * fh[t]; a1 += W[t + 1] * fh[t + 1]; a2 += W[t + 2] * fh[t + 2]; }
            f -= a0 + (a1 + a2);
        }
        ft[i] = f;
    }
    for (int i = 0; i < NS; ++i) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int t = 0; t + 2 < NS; t += 3) { a0 += Si[NS * i + t] * ft[t]; a1 += Si[NS * i + t + 1] * ft[t + 1]; a2 += Si[NS * i + t + 2] * ft[t + 2]; }
        const double u = a0 + (a1 + a2);
        const int xi = mid * SV + i;
        xs[NS * mid + i] = u;
        xd[NS * mid + i] = q.D[xi] * u;
        x[xi] = relax(alpha, u, x[xi]);
    }
#endif
}

template <class Q>
DEKF_FN void phase_sweeps_generic(Q& q, double alpha) {
    constexpr int NS = Q::NS, NS2 = Q::NS2;
    const int K = q.K, mid = mid_block(K);
#if DEKF_DEVICE_BUILD
    const bool w0 = __builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6) == 0;
    if (w0) {
        __builtin_amdgcn_s_setprio(3);
        sweep_legs_generic<false>(q, alpha);
        __builtin_amdgcn_s_setprio(0);
    }
#else
    sweep_legs_generic<false>(q, alpha);
#endif
    // (Measured and not kept, round 4: the three waiting wavefronts requesting their rows of S_k^-1 — and the meeting block's operands —
    // while wavefront 0 works through the forward legs, so that this phase reads LDS only: 272 k against 279 k steps/s on Go1 with
    // foot states.  The requests compete with the legs' own operand stream, which is the critical path.)
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 3);
    // g_k = S_k^-1 f_k is outside both recursions: one entry per lane, next to the meeting block
    const int ngt = (K * NS + 63) >> 6;
    wtiles(1 + ngt, [&](int tile, int lane) {
        if (tile == 0) { sweep_mid_block_generic(q, lane, alpha); return; }
        const int e = (tile - 1) * 64 + lane;
        if (e >= K * NS) return;
        const int k = e / NS, i = e - NS * k;
        if (k == mid) return;
        cdptr Si = q.Sinv + k * NS2 + NS * i;
        cdptr f = q.xs + NS * k;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int t = 0; t + 2 < NS; t += 3) { a0 += Si[t] * f[t]; a1 += Si[t + 1] * f[t + 1]; a2 += Si[t + 2] * f[t + 2]; }
        q.xd[e] = a0 + (a1 + a2);
    });
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 4);
#if DEKF_DEVICE_BUILD
    if (w0) {
        __builtin_amdgcn_s_setprio(3);
        sweep_legs_generic<true>(q, alpha);
        __builtin_amdgcn_s_setprio(0);
    }
#else
    sweep_legs_generic<true>(q, alpha);
#endif
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 5);
}

// In: xs = reduced right-hand side.  Out: x blocks relaxed, xd = D .* (solution).  The
// factorisation is two-sided (solve_factor 3d): W_k = C_k S_k^-1 in Wk[k] for k < mid,
// W^_k = C_k' S^_{k+1}^-1 in Wk[k] for k >= mid.
template <class Q>
DEKF_FN void phase_sweeps(Q& q, double alpha) {
    if constexpr (Q::NS != 9) {
        phase_sweeps_generic(q, alpha);
        return;
    } else {
    const int K = q.K, mid = mid_block(K);
    constexpr int NF = Q::NFIXED, FM = mid_block(NF);  // full window (steady state) of a compile-time horizon
    const bool fixed = NF >= 4 && K == NF;
#if DEKF_DEVICE_BUILD
    if constexpr (NF >= 4 && NF % 2 == 0) {
        if (fixed) {  // one wavefront, four rows, no barrier inside
            if (__builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6) == 0) {
                __builtin_amdgcn_s_setprio(3);
                sweeps_one_wave<NF>(q, alpha);
                __builtin_amdgcn_s_setprio(0);
            }
            DEKF_SYNC();
            DEKF_PROF_MARK(q, 5);
            return;
        }
    }
    if (K >= 4 && (K & 1) == 0) {  // even run-time window: one wavefront, four rows, no barrier inside
        if (__builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6) == 0) {
            __builtin_amdgcn_s_setprio(3);
            sweeps_one_wave_rt(q, alpha);
            __builtin_amdgcn_s_setprio(0);
        }
        DEKF_SYNC();
        DEKF_PROF_MARK(q, 5);
        return;
    }
    __builtin_amdgcn_s_setprio(3);  // the legs are the critical path and share their SIMDs
#endif
    if (fixed)
        two_waves([&] { sweep_chain<false, false, (NF >= 4 ? FM : 1)>(q, 0, 1, FM, -1, alpha); },
                  [&] { sweep_chain<false, false, (NF >= 4 ? NF - 2 - FM : 1)>(q, NF - 1, -1, NF - 2 - FM, 0, alpha); });
    else
        two_waves([&] { sweep_chain<false, false>(q, 0, 1, mid, -1, alpha); },
                  [&] { sweep_chain<false, false>(q, K - 1, -1, K - 2 - mid, 0, alpha); });
#if DEKF_DEVICE_BUILD
    __builtin_amdgcn_s_setprio(0);
#endif
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 3);
    // g_k = S_k^-1 f_k is outside both recursions: 7 blocks (63 lanes) per tile, next to the meeting block
    const int ngt = (K + 6) / 7;
    wtiles(1 + ngt, [&](int tile, int lane) {
        if (tile == 0) { sweep_mid_block(q, lane, alpha); return; }
        const int blk = lane / 9, i = lane - 9 * blk, k = (tile - 1) * 7 + blk;
        if (blk >= 7 || k >= K || k == mid) return;
        cdptr Si = q.Sinv + k * 81 + 9 * i;
        cdptr f = q.xs + 9 * k;
        double sv[9], fv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) { sv[t] = Si[t]; fv[t] = f[t]; }
        double a0 = sv[0] * fv[0] + sv[3] * fv[3] + sv[6] * fv[6];
        double a1 = sv[1] * fv[1] + sv[4] * fv[4] + sv[7] * fv[7];
        double a2 = sv[2] * fv[2] + sv[5] * fv[5] + sv[8] * fv[8];
        q.xd[9 * k + i] = a0 + a1 + a2;
    });
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 4);
#if DEKF_DEVICE_BUILD
    __builtin_amdgcn_s_setprio(3);
#endif
    if (fixed)
        two_waves([&] { sweep_chain<true, true, (NF >= 4 ? FM : 1)>(q, FM, -1, FM, 0, alpha); },
                  [&] { sweep_chain<true, true, (NF >= 4 ? NF - 1 - FM : 1)>(q, FM, 1, NF - 1 - FM, -1, alpha); });
    else
        two_waves([&] { sweep_chain<true, true>(q, mid, -1, mid, 0, alpha); },
                  [&] { sweep_chain<true, true>(q, mid, 1, K - 1 - mid, -1, alpha); });
#if DEKF_DEVICE_BUILD
    __builtin_amdgcn_s_setprio(0);
#endif
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 5);
    }  // NS == 9
}

// ---------------------------------------------------------------- R: rows
// Everything a block of NR rows (one slack block: S is NR x NR) does in an iteration, in registers.
// ar[j] = (A_x xd)(row r0+j), already scaled by E.  sapply(in, out): out = S^-1 in (S = P_s + sigma I +
// rho (E D)^2 of the block, inverted by solve_factor 3a).  EQ: the rows are equalities by
// construction (Meas, Dyn: l == u), so rho = rho_eq and the projection returns the bound itself.
// State kept between iterations: zt[r] = t (slack forward-elimination result), at[r] = w (what the
// next phase X gathers), q.cf[r] = rho E D (refreshed by rows_restart after every factorisation).
// Every LDS load of the block is issued before the first store (the compiler cannot move a load
// across a store through another pointer, so interleaving them serialises one LDS round trip per row).
template <int NR>
struct RowPre {  // what a row block needs that does not depend on the solve of this iteration
    double e[NR], cf[NR], d[NR], t0[NR], x0[NR], z0[NR], y0[NR], lo[NR], hi[NR];
};
template <int NR, bool EQ, class Q>
DEKF_FN void row_block_load(const Q& q, int r0, int sv0, bool has_hi, RowPre<NR>& p) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + j, sv = sv0 + j;
        p.e[j] = q.E[r];
        p.cf[j] = q.cf[r];
        p.d[j] = q.D[sv];
        p.t0[j] = q.zt[r];
        p.x0[j] = q.x[sv];
        p.z0[j] = q.z[r];
        p.y0[j] = q.y[r];
        p.lo[j] = q.lo[r];
        p.hi[j] = EQ ? 0.0 : q.hi[has_hi ? r - q.ix.rvb : 0];  // !has_hi: an equality block on the generic path (hi = lo)
    }
}
template <int NR, bool EQ, class Q, class SM>
DEKF_FN void row_block_compute(Q& q, int r0, int sv0, cdptr ar, const SM& S, const RowPre<NR>& p, double alpha, double sigma,
                               bool has_hi = true, dptr wout = nullptr) {
    const double rho_eq = uni(RHO_EQ_OVER_RHO_INEQ * q.rho);
    double c2[NR], hi[NR], v[NR], sl[NR], xn[NR], zn[NR], yn[NR], un[NR], rhs[NR], t[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        hi[j] = (!EQ && !has_hi) ? p.lo[j] : p.hi[j];
        c2[j] = p.d[j] * p.e[j];
        v[j] = p.cf[j] * ar[j];
    }
    S.apply(v, sl);
    // the rows of a VO block switch between the +-inf box and an equality TOGETHER (DecentralEst.cpp:995-1005:
    // the three bounds of a step are written at once), so rho is decided once per block, from its first row
    const double rv_blk = EQ ? rho_eq : q.rho_of(p.lo[0], hi[0]);
    const double rinv_blk = EQ ? 0.0 : rcp_fast(rv_blk);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double sj = p.t0[j] + sl[j];       // slack solution
        const double ztn = fma(-c2[j], sj, ar[j]);   // (A xt)(r)
        xn[j] = relax(alpha, sj, p.x0[j]);
        const double zh = relax(alpha, ztn, p.z0[j]);
        double rv;
        if (EQ) {
            rv = rho_eq;
            zn[j] = p.lo[j];
        } else {
            rv = rv_blk;
            zn[j] = dmin(dmax(fma(rinv_blk, p.y0[j], zh), p.lo[j]), hi[j]);
        }
        yn[j] = fma(rv, zh - zn[j], p.y0[j]);
        un[j] = fma(rv, zn[j], -yn[j]);
        rhs[j] = lin2(sigma, xn[j], -c2[j], un[j]);
    }
    S.apply(rhs, t);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + j, sv = sv0 + j;
        q.x[sv] = xn[j];
        q.y[r] = yn[j];
        q.z[r] = zn[j];
        q.zt[r] = t[j];
        const double wj = p.e[j] * fma(p.cf[j], t[j], un[j]);
        q.at[r] = wj;
        if (wout) wout[j] = wj;
    }
}
template <int NR, bool EQ, class Q, class SM>
DEKF_FN void row_block_update(Q& q, int r0, int sv0, cdptr ar, const SM& S, double alpha, double sigma, bool has_hi = true,
                              dptr wout = nullptr) {
    RowPre<NR> p;
    row_block_load<NR, EQ>(q, r0, sv0, has_hi, p);
    row_block_compute<NR, EQ>(q, r0, sv0, ar, S, p, alpha, sigma, has_hi, wout);
}
// the same block from (x, z, y) alone: after a (re)factorisation, and for the cold start
template <int NR, bool EQ, class Q, class SM>
DEKF_FN void row_block_restart(Q& q, int r0, int sv0, const SM& S, double sigma, dptr wout = nullptr) {
    const double rho_eq = uni(RHO_EQ_OVER_RHO_INEQ * q.rho);
    double e[NR], cf[NR], un[NR], rhs[NR], t[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + j, sv = sv0 + j;
        const double rv = EQ ? rho_eq : q.rho_at(r);
        const double d = q.D[sv];
        e[j] = q.E[r];
        cf[j] = rv * e[j] * d;
        un[j] = fma(rv, q.z[r], -q.y[r]);
        rhs[j] = lin2(sigma, q.x[sv], -(e[j] * d), un[j]);
    }
    S.apply(rhs, t);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        q.cf[r0 + j] = cf[j];
        q.zt[r0 + j] = t[j];
        const double wj = e[j] * fma(cf[j], t[j], un[j]);
        q.at[r0 + j] = wj;
        if (wout) wout[j] = wj;
    }
}

template <int N>
struct SymMat {  // symmetric N x N, packed upper triangle, held in registers
    double p[N * (N + 1) / 2];
    DEKF_FN explicit SymMat(cdptr s) {
#pragma unroll
        for (int i = 0; i < N * (N + 1) / 2; ++i) p[i] = s[i];
    }
    DEKF_FN void apply(cdptr in, dptr out) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) a = fma(p[i < j ? symidx(i, j, N) : symidx(j, i, N)], in[j], a);
            out[i] = a;
        }
    }
};
// 3x3 slack-block inverse of a VO block (packed symmetric) or of a Dyn bias block (diagonal) through ONE code
// path, so that both kinds share a tile: the diagonal goes into the packed slots 0, 3, 5, the rest is zero
struct VoOrBiasMat {
    double p[6];
    DEKF_FN VoOrBiasMat(cdptr sc6, cdptr diag3, bool vo, int st = 1) {  // st: element stride of both arrays
        cdptr sp = vo ? sc6 : diag3;
        // (a diagonal has three entries: for it the loads of s3..s5 repeat s0..s2 instead of running past the array)
        const int o3 = vo ? 3 * st : 0, o4 = vo ? 4 * st : st, o5 = vo ? 5 * st : 2 * st;
        const double s0 = sp[0], s1 = sp[st], s2 = sp[2 * st], s3 = sp[o3], s4 = sp[o4], s5 = sp[o5];
        p[0] = s0; p[1] = vo ? s1 : 0.0; p[2] = vo ? s2 : 0.0;
        p[3] = vo ? s3 : s1; p[4] = vo ? s4 : 0.0; p[5] = vo ? s5 : s2;
    }
    DEKF_FN void apply(cdptr in, dptr out) const {
        out[0] = dot3(p[0], in[0], p[1], in[1], p[2], in[2]);
        out[1] = dot3(p[1], in[0], p[3], in[1], p[4], in[2]);
        out[2] = dot3(p[2], in[0], p[4], in[1], p[5], in[2]);
    }
};

#if DEKF_DEVICE_BUILD
// The Dyn 6x6 slack block on a PAIR of adjacent lanes (even lane: position rows, odd lane: velocity
// rows): each lane keeps its 3 rows of S (own 3x3 symmetric part + 3x3 coupling part) and gets the
// partner's 3-vector through DPP quad_perm [1,0,3,2] — no LDS, no barrier.  One lane per 6-block made
// this tile the critical path of the row phase (2.3x the instructions of a Meas lane on 19 lanes).
// (pair_swap: wave.h)
struct DynPairMat {
    double a[6];  // own rows x own columns (symmetric, packed)
    double b[9];  // own rows x partner columns
    DEKF_FN DynPairMat(cdptr s, bool vel, int st = 1) {  // st: element stride of s
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) a[symidx(i, j, 3)] = s[(vel ? symidx(3 + i, 3 + j, 6) : symidx(i, j, 6)) * st];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) b[3 * i + j] = s[(vel ? symidx(j, 3 + i, 6) : symidx(i, 3 + j, 6)) * st];
    }
    DEKF_FN void apply(cdptr in, dptr out) const {
        double pin[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) pin[j] = pair_swap(in[j]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = fma(a[i < j ? symidx(i, j, 3) : symidx(j, i, 3)], in[j], acc);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = fma(b[3 * i + j], pin[j], acc);
            out[i] = acc;
        }
    }
};
#endif

// RESTART = false: one iteration's row work (needs xd from phase_sweeps).  RESTART = true: rebuild
// cf, t, w from (x, z, y).  Tiles: [Meas leg blocks][Dyn p+v 6-blocks][VO and Dyn bias blocks].
template <bool RESTART, class Q>
DEKF_FN void phase_rows(Q& q, double alpha, double sigma) {
    constexpr int L = Q::LEGS, NM = 3 * L, FT = Q::FOOT, NS = Q::NS;
    const int K = q.K, K1 = K - 1, nmeas = K * L;
    const int ntm = (nmeas + 63) >> 6;
#if DEKF_DEVICE_BUILD
    const int ntp = (2 * K1 + 63) >> 6;  // Dyn p+v blocks: a lane pair per block
#else
    const int ntp = (K1 + 63) >> 6;      // host build: one (sequential) lane per 6-block
#endif
    const double dt = q.c.dt, hdt2 = q.c.hdt2;
    cdptr xd = q.xd, E = q.E;
    // Tile order [Meas | Dyn | VO + bias | foot-position Dyn].  VO blocks and Dyn bias blocks share tiles and ONE code
    // path (3 rows, a 3x3 symmetric slack-block inverse that is diagonal for the bias rows, the generic projection):
    // as separate kinds they were two tile bodies run one after the other by some wavefront.
    const int nvb = (2 * K1 + 63) >> 6;
    const int ntf = FT ? (K1 * L + 63) >> 6 : 0;
    wtiles(ntm + ntp + nvb + ntf, [&](int tile, int lane) {
        if (tile < ntm) {  // Meas: leg block (k, leg), A_meas = [0 I 0]  (foot-position states: [-I 0 0 .. I ..])
            const int e = tile * 64 + lane;
            if (e >= nmeas) return;
            const int k = e / L, leg = e - k * L;
            const int r0 = q.ix.rm(k, 3 * leg), sv0 = q.ix.v(k, 3 * leg);
            const SymMat<3> S(q.Sv + e * 6);
            if (RESTART) { row_block_restart<3, true>(q, r0, sv0, S, sigma); return; }
            double ar[3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                ar[a] = FT ? E[r0 + a] * (xd[NS * k + 9 + 3 * leg + a] - xd[NS * k + a]) : E[r0 + a] * xd[NS * k + 3 + a];
            row_block_update<3, true>(q, r0, sv0, ar, S, alpha, sigma);
            return;
        }
        const int td = tile - ntm;
#if DEKF_DEVICE_BUILD
        if (td < ntp) {  // Dyn position + velocity rows: lane pair per 6x6 slack block
            const int pl = td * 64 + lane, k = pl >> 1;
            const bool vel = pl & 1;
            if (k >= K1) return;
            const int r0 = q.ix.rd(k, vel ? 3 : 0), sv0 = q.ix.w(k, vel ? 3 : 0);
            const DynPairMat S(q.Sw + k * SWS, vel);
            cdptr R = q.R + 9 * k;
            double Rk[9], wo[3];
#pragma unroll
            for (int t = 0; t < 9; ++t) Rk[t] = R[t];
            if (RESTART) row_block_restart<3, true>(q, r0, sv0, S, sigma, wo);
            else {
                cdptr xk = xd + NS * k;
                const double c1 = vel ? 0.0 : dt, c2 = vel ? dt : hdt2;
                const int o = vel ? 3 : 0;
                double ar[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double rb = dot3(Rk[3 * a], xk[6], Rk[3 * a + 1], xk[7], Rk[3 * a + 2], xk[8]);
                    ar[a] = E[r0 + a] * (fma(-c2, rb, fma(c1, xk[3 + a], xk[o + a])) - xk[NS + o + a]);
                }
                row_block_update<3, true>(q, r0, sv0, ar, S, alpha, sigma, true, wo);
            }
            // what the bias columns of x_k gather from these six rows: R' (dt^2/2 w_p + dt w_v), formed here where
            // w_p and w_v are in registers (the x-column phase then reads 3 values instead of 6 w and 9 R)
            double u[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double pw = pair_swap(wo[r]);
                u[r] = vel ? fma(hdt2, pw, dt * wo[r]) : fma(hdt2, wo[r], dt * pw);
            }
            if (!vel) {
#pragma unroll
                for (int a = 0; a < 3; ++a) q.gb[3 * k + a] = dot3(Rk[a], u[0], Rk[3 + a], u[1], Rk[6 + a], u[2]);
            }
            return;
        }
#endif
        if (td < ntp) {  // Dyn position + velocity rows: 6x6 slack block on one lane (host build)
            const int k = td * 64 + lane;
            if (k >= K1) return;
            cdptr xk = xd + NS * k;
            const int r0 = q.ix.rd(k, 0), sv0 = q.ix.w(k, 0);
            const SymMat<6> S(q.Sw + k * SWS);
            cdptr R = q.R + 9 * k;
            double wo[6];
            if (RESTART) row_block_restart<6, true>(q, r0, sv0, S, sigma, wo);
            else {
                double ar[6];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double rb = dot3(R[3 * a], xk[6], R[3 * a + 1], xk[7], R[3 * a + 2], xk[8]);
                    ar[a] = E[r0 + a] * (fma(-hdt2, rb, fma(dt, xk[3 + a], xk[a])) - xk[NS + a]);
                    ar[3 + a] = E[r0 + 3 + a] * (fma(-dt, rb, xk[3 + a]) - xk[NS + 3 + a]);
                }
                row_block_update<6, true>(q, r0, sv0, ar, S, alpha, sigma, true, wo);
            }
            for (int a = 0; a < 3; ++a) {
                // (the lane pair's forms: u_r = fma(hdt2, w_p, dt w_v), then the three products in order)
                const double u0 = fma(hdt2, wo[0], dt * wo[3]), u1 = fma(hdt2, wo[1], dt * wo[4]), u2 = fma(hdt2, wo[2], dt * wo[5]);
                q.gb[3 * k + a] = dot3(R[a], u0, R[3 + a], u1, R[6 + a], u2);
            }
            return;
        }
        if (td < ntp + nvb) {   // VO rows (+-inf box or equality, per-row rho) and Dyn bias rows (equalities, diagonal slack block)
            const int idx = (td - ntp) * 64 + lane;
            if (idx >= 2 * K1) return;
            const bool vo = idx < K1;
            const int k = vo ? idx : idx - K1, o = vo ? 0 : 6;
            const int r0 = vo ? q.ix.rv(k, 0) : q.ix.rd(k, 6), sv0 = vo ? q.ix.c(k, 0) : q.ix.w(k, 6);
            const VoOrBiasMat S(q.Sc + k * 6, q.Sw + k * SWS + 21, vo);
            if (RESTART) { row_block_restart<3, false>(q, r0, sv0, S, sigma); return; }
            cdptr xk = xd + NS * k;
            double ar[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) ar[a] = E[r0 + a] * (xk[o + a] - xk[NS + o + a]);
            row_block_update<3, false>(q, r0, sv0, ar, S, alpha, sigma, vo);
            return;
        }
        if constexpr (FT) {  // Dyn rows of a foot-position state: f_k - f_{k+1} - w = 0 (identity dynamics, DecentralEst.cpp:395-398)
            const int e = (td - ntp - nvb) * 64 + lane;
            if (e >= K1 * L) return;
            const int k = e / L, leg = e - k * L;
            const int r0 = q.ix.rd(k, 9 + 3 * leg), sv0 = q.ix.w(k, 9 + 3 * leg);
            const SymMat<3> S(q.Sf + e * 6);
            if (RESTART) { row_block_restart<3, true>(q, r0, sv0, S, sigma); return; }
            cdptr xk = xd + NS * k + 9 + 3 * leg;
            double ar[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) ar[a] = E[r0 + a] * (xk[a] - xk[NS + a]);
            row_block_update<3, true>(q, r0, sv0, ar, S, alpha, sigma);
        }
    });
    DEKF_SYNC();
}

#if DEKF_DEVICE_BUILD
// ---------------------------------------------------------------- S + R with the row operands prefetched
// When every row tile has its own wavefront (Go1: Meas | Meas | Dyn pairs | VO + bias on four wavefronts) the
// wavefronts that do not run the solve use that time to bring their row blocks' operands into registers
// (scaling, bounds, slack state, block inverse: everything but xd), so after the barrier the row phase starts
// with its arithmetic instead of an LDS round trip of ~45 loads.
struct RowTile {
    int kind;  // 0 Meas leg block, 1 Dyn position / velocity half (lane pair), 2 VO or bias block; -1 nothing
    int k, r0, sv0;
    bool vel, vo;  // kind 1: velocity half; kind 2: VO block (else bias)
    RowPre<3> pre;
    double a[6], b[9];  // slack-block inverse: symmetric own part; coupling to the partner lane (kind 1)
    DEKF_FN void apply(cdptr in, dptr out) const {
        double pin[3] = {0.0, 0.0, 0.0};
        if (kind == 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) pin[j] = pair_swap(in[j]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = fma(a[i < j ? symidx(i, j, 3) : symidx(j, i, 3)], in[j], acc);
            if (kind == 1) {
#pragma unroll
                for (int j = 0; j < 3; ++j) acc = fma(b[3 * i + j], pin[j], acc);
            }
            out[i] = acc;
        }
    }
};
template <class Q>
DEKF_FN void row_tile_load(const Q& q, int tile, int lane, RowTile& t) {
    constexpr int L = Q::LEGS, NM = 3 * L, SV = 21 + NM;
    const int K = q.K, K1 = K - 1, nmeas = K * L;
    const int ntm = (nmeas + 63) >> 6, ntp = (2 * K1 + 63) >> 6;
    t.kind = -1; t.k = 0; t.r0 = 0; t.sv0 = 0; t.vel = false; t.vo = false;
    if (tile < ntm) {
        const int e = tile * 64 + lane;
        if (e >= nmeas) return;
        const int k = e / L, leg = e - k * L;
        t.kind = 0; t.k = k; t.r0 = q.ix.rm(k, 3 * leg); t.sv0 = k * SV + 9 + 3 * leg;
        cdptr sp = q.Sv + e * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) t.a[i] = sp[i];
        row_block_load<3, true>(q, t.r0, t.sv0, true, t.pre);
    } else if (tile < ntm + ntp) {
        const int pl = (tile - ntm) * 64 + lane, k = pl >> 1;
        if (k >= K1) return;
        const bool vel = pl & 1;
        t.kind = 1; t.k = k; t.vel = vel; t.r0 = q.ix.rd(k, vel ? 3 : 0); t.sv0 = k * SV + 9 + NM + (vel ? 3 : 0);
        const DynPairMat S(q.Sw + k * SWS, vel);
#pragma unroll
        for (int i = 0; i < 6; ++i) t.a[i] = S.a[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) t.b[i] = S.b[i];
        row_block_load<3, true>(q, t.r0, t.sv0, true, t.pre);
    } else {
        const int idx = (tile - ntm - ntp) * 64 + lane;
        if (idx >= 2 * K1) return;
        const bool vo = idx < K1;
        const int k = vo ? idx : idx - K1;
        t.kind = 2; t.k = k; t.vo = vo; t.r0 = vo ? q.ix.rv(k, 0) : q.ix.rd(k, 6); t.sv0 = k * SV + (vo ? 18 + NM : 9 + NM + 6);
        const VoOrBiasMat S(q.Sc + k * 6, q.Sw + k * SWS + 21, vo);
#pragma unroll
        for (int i = 0; i < 6; ++i) t.a[i] = S.p[i];
        row_block_load<3, false>(q, t.r0, t.sv0, vo, t.pre);
    }
}
template <class Q>
DEKF_FN void row_tile_finish(Q& q, const RowTile& t, double alpha, double sigma) {
    if (t.kind < 0) return;
    const double dt = q.c.dt, hdt2 = q.c.hdt2;
    cdptr xk = q.xd + 9 * t.k;
    cdptr E = q.E;
    double ar[3];
    if (t.kind == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) ar[a] = E[t.r0 + a] * xk[3 + a];
        row_block_compute<3, true>(q, t.r0, t.sv0, ar, t, t.pre, alpha, sigma);
    } else if (t.kind == 1) {
        cdptr R = q.R + 9 * t.k;
        double Rk[9], wo[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rk[i] = R[i];
        const bool vel = t.vel;
        const double c1 = vel ? 0.0 : dt, c2 = vel ? dt : hdt2;
        const int o = vel ? 3 : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double rb = dot3(Rk[3 * a], xk[6], Rk[3 * a + 1], xk[7], Rk[3 * a + 2], xk[8]);
            ar[a] = E[t.r0 + a] * (fma(-c2, rb, fma(c1, xk[3 + a], xk[o + a])) - xk[9 + o + a]);
        }
        row_block_compute<3, true>(q, t.r0, t.sv0, ar, t, t.pre, alpha, sigma, true, wo);
        double u[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double pw = pair_swap(wo[r]);
            u[r] = vel ? fma(hdt2, pw, dt * wo[r]) : fma(hdt2, wo[r], dt * pw);
        }
        if (!vel) {
#pragma unroll
            for (int a = 0; a < 3; ++a) q.gb[3 * t.k + a] = dot3(Rk[a], u[0], Rk[3 + a], u[1], Rk[6 + a], u[2]);
        }
    } else {
        const int o = t.vo ? 0 : 6;
#pragma unroll
        for (int a = 0; a < 3; ++a) ar[a] = E[t.r0 + a] * (xk[o + a] - xk[9 + o + a]);
        row_block_compute<3, false>(q, t.r0, t.sv0, ar, t, t.pre, alpha, sigma, t.vo);
    }
}
#endif

#if DEKF_DEVICE_BUILD
// ---------------------------------------------------------------- three workgroups per CU (R3)
// What limits the solve kernel's throughput is how many instances a CU holds (the solve is a dependent chain that leaves
// three of four wavefronts waiting), and what limits THAT is LDS: 79.7 KiB per instance allow two.  Of those, 36 KiB are
// lane-private: a row block's state (slack x, z, y, t) and its constants (E, rho E D, D, bounds, slack-block inverse) are only
// ever touched by the lane that owns the block.  Here they live in that lane's REGISTERS for a whole chunk of iterations (all
// iterations up to the next termination check / rho adaptation), and in the workgroup's HBM slab in between, where the rare
// phases (residuals, refactorisation) find the state.  LDS keeps what crosses lanes — x blocks, xs, xd, w, gb, the factor — and
// the scaling vectors D, E that every rare phase reads: 52.9 KiB per instance, three workgroups per CU (decided by an
// aliased-layout build before this was written: +33 %; 81 doubles more and the hardware places only two, see r3_fits).
//
// WAVE-SPECIALISED LOOPS.  Wavefront 0 runs nothing but the block-tridiagonal solve; wavefronts 1-3 own the row tiles and the
// x-column tiles.  Each side has its own loop with the same sequence of workgroup barriers (s_barrier counts arrivals, it does
// not care where a wavefront's program counter is), so the 78 state registers are not live in the solve's code and the solve's
// operand sets are not live in the row code: both fit the 168 VGPRs that three wavefronts per SIMD leave each.
//   barrier B1: xs complete   (workers: x columns)      ->  wavefront 0: solve
//   barrier B2: xd complete   (wavefront 0: solve)      ->  workers: rows from registers, w and gb to LDS
//   barrier B3: w, gb complete (workers)                ->  workers: x columns of the next iteration
// One row block's state and constants, by KIND of block — a compile-time parameter, because each worker wavefront runs ONE kind
// in its own loop (admm_chunk_r3) and must not carry the registers of the others: 0 Meas leg block (equality; 27 doubles),
// 1 Dyn position / velocity half on a lane pair (equality, 6x6 slack block: + 9 coupling entries), 2 VO / Dyn bias / Meas block on
// the generic projection path (+ upper bounds and z).  An equality block keeps no z: it is the bound itself from the first
// iteration on (0 before it, on a cold start).
template <int KIND>
struct RowRegsT {
    static constexpr int kind = KIND;
    bool valid;      // the lane owns a block
    int k, r0, sv0;
    bool vel, meas;  // kind 1: velocity half; kind 2: a Meas block folded into the VO / bias tile
    bool vo;         // kind 2: a VO block (the only kind whose z is state)
    int xo;          // kind 2: offset of the block's x entries inside a step (0 VO, 6 bias, 3 Meas)
    double e[3], c2[3], cf[3], lo[3];
    double t[3], xs[3], y[3];
    double z[KIND == 2 ? 3 : 1];
    double a[6];                                  // slack-block inverse: own 3x3 (symmetric, packed)
    double b[KIND == 1 ? 9 : (KIND == 2 ? 3 : 1)];  // kind 1: coupling to the partner lane; kind 2: upper bounds
    double rk[KIND == 1 ? 9 : 1];                 // kind 1: the step's rotation R_k (nine LDS reads per iteration otherwise)
    DEKF_FN void apply(cdptr in, dptr out) const {
        double pin[3] = {0.0, 0.0, 0.0};
        if constexpr (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) pin[j] = pair_swap(in[j]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = fma(a[i < j ? symidx(i, j, 3) : symidx(j, i, 3)], in[j], acc);
            if constexpr (KIND == 1) {
#pragma unroll
                for (int j = 0; j < 3; ++j) acc = fma(b[3 * i + j], pin[j], acc);
            }
            out[i] = acc;
        }
    }
};
// Tiles: wavefront 1 the first 64 Meas leg blocks, wavefront 2 the Dyn lane pairs, wavefront 3 the VO and bias blocks plus the
// Meas blocks beyond 64 (Go1: 16) on the generic projection path — for an equality block that path gives bit-identical results
// (hi = lo: the clamp returns lo, rho_of returns rho_eq).
// Loading a block IS the restart after a (re)factorisation (row_block_restart: rho E D, t = S^-1 (sigma x_s - E D u),
// w = E (u + rho E D t) from x_s, z, y and the block's new inverse) — the same expressions, so t and w also come out as the last
// iteration left them when a chunk merely continues after a termination check.  Neither t nor rho E D ever go to the slab, and
// the three-workgroup kernels run no separate restart phase.  Writes w (and gb) to LDS: a workgroup barrier follows.
template <int KIND, class Q>
DEKF_FN void row_regs_fill(Q& q, double sigma, RowRegsT<KIND>& t, cdptr sp);
template <int KIND, class Q>
DEKF_FN void row_regs_load(Q& q, int lane, double sigma, RowRegsT<KIND>& t) {
    constexpr int L = Q::LEGS, NM = 3 * L, SV = 21 + NM;
    const int K = q.K, K1 = K - 1, nmeas = K * L;
    t.valid = false; t.k = 0; t.r0 = 0; t.sv0 = 0; t.vel = false; t.meas = false; t.vo = false; t.xo = 0;
    cdptr sp = q.Sv;  // (the slack-block inverses of the three-workgroup kernels are stored entry-major in the slab: solve_factor 3a)
    if constexpr (KIND == 0) {
        if (lane >= nmeas || lane >= 64) return;
        const int k = lane / L, leg = lane - k * L;
        t.k = k; t.r0 = q.ix.rm(k, 3 * leg); t.sv0 = k * SV + 9 + 3 * leg; t.xo = 3;
        sp = q.Sv + lane;
    } else if constexpr (KIND == 1) {
        const int k = lane >> 1;
        if (k >= K1) return;
        t.k = k; t.vel = lane & 1; t.r0 = q.ix.rd(k, t.vel ? 3 : 0); t.sv0 = k * SV + 9 + NM + (t.vel ? 3 : 0);
    } else {
        if (lane < 2 * K1) {
            t.vo = lane < K1;
            const int k = t.vo ? lane : lane - K1;
            t.k = k; t.xo = t.vo ? 0 : 6; t.r0 = t.vo ? q.ix.rv(k, 0) : q.ix.rd(k, 6); t.sv0 = k * SV + (t.vo ? 18 + NM : 9 + NM + 6);
        } else {
            const int e = 64 + lane - 2 * K1;
            if (e >= nmeas) return;
            const int k = e / L, leg = e - k * L;
            t.meas = true; t.k = k; t.xo = 3; t.r0 = q.ix.rm(k, 3 * leg); t.sv0 = k * SV + 9 + 3 * leg;
            sp = q.Sv + e;
        }
    }
    t.valid = true;
    row_regs_fill<KIND>(q, sigma, t, sp);
}
// the block is chosen (t.valid, k, r0, sv0, vel, meas, vo, xo; sp: its slack-block inverse when it is a Meas block): constants, state
// and the restart
template <int KIND, class Q>
DEKF_FN void row_regs_fill(Q& q, double sigma, RowRegsT<KIND>& t, cdptr sp) {
    constexpr int L = Q::LEGS;
    const int K = q.K, nmeas = K * L;
    const bool vo = t.vo;
    // slack-block inverses: entry-major in the slab of the three-workgroup kernels, block-major for RR (solve_factor 3a)
    constexpr bool EM = Q::FACTOR_LDS;
    if constexpr (KIND == 1) {
        const DynPairMat S(EM ? q.Sw + t.k : q.Sw + t.k * SWS, t.vel, EM ? K : 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) t.a[i] = S.a[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) t.b[i] = S.b[i];
    } else if (KIND == 2 && !t.meas) {
        const VoOrBiasMat S(EM ? q.Sc + t.k : q.Sc + 6 * t.k, EM ? q.Sw + t.k + 21 * K : q.Sw + t.k * SWS + 21, vo, EM ? K : 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) t.a[i] = S.p[i];
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) t.a[i] = sp[i * (EM ? nmeas : 1)];
    }
    // State: zero before the first chunk (the cold start); afterwards the slack x and y from where the previous chunk left them
    // in LDS; z of an equality row IS its bound after one iteration (the projection returns it), only the VO rows keep theirs.
    const bool cold = q.cold;
    double dd[3], zz[3], hi[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = t.r0 + j, sv = t.sv0 + j;
        const double d = q.D[sv];
        t.e[j] = q.E[r];
        t.c2[j] = d * t.e[j];
        t.lo[j] = q.lo[r];
        hi[j] = t.lo[j];
        if constexpr (KIND == 2) { hi[j] = vo ? q.hi[r - q.ix.rvb] : t.lo[j]; t.b[j] = hi[j]; }  // an equality block on the generic path: hi = lo
        const double sxv = q.sx[r], syv = q.sy[r], szv = q.sz[vo ? r - q.ix.rvb : 0];
        t.xs[j] = cold ? 0.0 : sxv;
        t.y[j] = cold ? 0.0 : syv;
        // (polishing restarts with z ON the bound of every equality row; a VO row without bounds starts at 0 like everything else)
        const bool z_on_bound = q.polishing() && !(KIND == 2 && q.rho_of(t.lo[j], hi[j]) == RHO_MIN);
        zz[j] = cold ? (z_on_bound ? t.lo[j] : 0.0) : (vo ? szv : t.lo[j]);
        if constexpr (KIND == 2) t.z[j] = zz[j];
        dd[j] = d;
    }
    const double rho_eq = uni(RHO_EQ_OVER_RHO_INEQ * q.rho);
    double un[3], rhs[3], tn[3], wo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double rv = KIND != 2 ? rho_eq : q.rho_of(t.lo[j], hi[j]);
        t.cf[j] = rv * t.e[j] * dd[j];
        un[j] = fma(rv, zz[j], -t.y[j]);
        rhs[j] = lin2(sigma, t.xs[j], -(t.e[j] * dd[j]), un[j]);
    }
    t.apply(rhs, tn);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t.t[j] = tn[j];
        wo[j] = t.e[j] * fma(t.cf[j], tn[j], un[j]);
        q.at[t.r0 + j] = wo[j];
    }
    if constexpr (KIND == 1) {
        const double dt = q.c.dt, hdt2 = q.c.hdt2;
        cdptr R = q.R + 9 * t.k;
#pragma unroll
        for (int i = 0; i < 9; ++i) t.rk[i] = R[i];
        double u[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double pw = pair_swap(wo[r]);
            u[r] = t.vel ? fma(hdt2, pw, dt * wo[r]) : fma(hdt2, wo[r], dt * pw);
        }
        if (!t.vel) {
#pragma unroll
            for (int a = 0; a < 3; ++a) q.gb[3 * t.k + a] = dot3(R[a], u[0], R[3 + a], u[1], R[6 + a], u[2]);
        }
    }
}
template <int KIND, class Q>
DEKF_FN void row_regs_store(Q& q, const RowRegsT<KIND>& t) {
    if (!t.valid) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = t.r0 + j;
        q.sx[r] = t.xs[j];   // (aliases the w vector, which nobody reads after the last iteration of a chunk)
        q.sy[r] = t.y[j];
        if constexpr (KIND == 2) { if (t.vo) q.sz[r - q.ix.rvb] = t.z[j]; }
    }
}
// one iteration of a row block: the arithmetic of row_block_compute, operand for operand, with the state in registers.
// first_cold: the first iteration after a cold start (z of an equality row is still 0, not yet its bound)
template <int KIND, class Q>
DEKF_FN void row_regs_iter(Q& q, RowRegsT<KIND>& t, double alpha, double sigma, bool first_cold) {
    if (!t.valid) return;
    const double dt = q.c.dt, hdt2 = q.c.hdt2;
    cdptr xk = q.xd + 9 * t.k;
    double ar[3], Rk[9];
    if constexpr (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Rk[i] = t.rk[i];
        const bool vel = t.vel;
        const double c1 = vel ? 0.0 : dt, c2 = vel ? dt : hdt2;
        const int o = vel ? 3 : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double rb = dot3(Rk[3 * a], xk[6], Rk[3 * a + 1], xk[7], Rk[3 * a + 2], xk[8]);
            ar[a] = t.e[a] * (fma(-c2, rb, fma(c1, xk[3 + a], xk[o + a])) - xk[9 + o + a]);
        }
    } else if (KIND == 0 || t.meas) {
#pragma unroll
        for (int a = 0; a < 3; ++a) ar[a] = t.e[a] * xk[3 + a];
    } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) ar[a] = t.e[a] * (xk[t.xo + a] - xk[9 + t.xo + a]);
    }
    constexpr bool eq = KIND != 2;
    const double rho_eq = uni(RHO_EQ_OVER_RHO_INEQ * q.rho);
    double v[3], sl[3], un[3], rhs[3], tn[3], wo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = t.cf[j] * ar[j];
    t.apply(v, sl);
    double rv_blk = rho_eq, rinv_blk = 0.0;
    if constexpr (!eq) { rv_blk = q.rho_of(t.lo[0], t.b[0]); rinv_blk = rcp_fast(rv_blk); }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double zprev;
        if constexpr (eq) zprev = first_cold ? 0.0 : t.lo[j];
        else zprev = t.z[j];
        const double sj = t.t[j] + sl[j];
        const double ztn = fma(-t.c2[j], sj, ar[j]);
        const double xn = relax(alpha, sj, t.xs[j]);
        const double zh = relax(alpha, ztn, zprev);
        double zn;
        if constexpr (eq) zn = t.lo[j];
        else zn = dmin(dmax(fma(rinv_blk, t.y[j], zh), t.lo[j]), t.b[j]);
        const double yn = fma(rv_blk, zh - zn, t.y[j]);
        un[j] = fma(rv_blk, zn, -yn);
        rhs[j] = lin2(sigma, xn, -t.c2[j], un[j]);
        t.xs[j] = xn;
        if constexpr (!eq) t.z[j] = zn;
        t.y[j] = yn;
    }
    t.apply(rhs, tn);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t.t[j] = tn[j];
        wo[j] = t.e[j] * fma(t.cf[j], tn[j], un[j]);
        q.at[t.r0 + j] = wo[j];
    }
    if constexpr (KIND == 1) {
        double u[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double pw = pair_swap(wo[r]);
            u[r] = t.vel ? fma(hdt2, pw, dt * wo[r]) : fma(hdt2, wo[r], dt * pw);
        }
        if (!t.vel) {
#pragma unroll
            for (int a = 0; a < 3; ++a) q.gb[3 * t.k + a] = dot3(Rk[a], u[0], Rk[3 + a], u[1], Rk[6 + a], u[2]);
        }
    }
}
// x-column tile `kind` (0 position, 1 velocity, 2 bias) on the compact x blocks
template <class Q>
DEKF_FN void xcols_tile_r3(Q& q, int kind, int lane, double sigma) {
    const int K = q.K;
    if (lane >= 3 * K) return;
    cdptr qsl = q.tmp + TmpMap<9>::QSL;
    cdptr at = q.at;
    auto w = [&](int r) { return at[r]; };
    const int k = lane / 3, a = lane - 3 * k, j = 3 * kind + a, i = 9 * k + j;
    const double xv = q.xb[i], dv = q.D[k * (21 + 3 * Q::LEGS) + j], qv = qsl[j];
    double g;
    if (kind == 0) g = gather_pcol(q, k, a, w);
    else if (kind == 1) g = gather_vcol(q, k, a, w);
    else {
        const bool hn = k < K - 1, hp = k > 0;
        const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;
        const double n0 = w(q.ix.rd(kn, 6 + a)), n1 = q.gb[3 * kn + a], p0 = w(q.ix.rd(kp, 6 + a));
        g = (hn ? n0 - n1 : 0.0) - (hp ? p0 : 0.0);
    }
    q.xs[i] = fma(dv, g, fma(sigma, xv, -(k == 0 ? qv : 0.0)));
}
// The same tile with everything an iteration does not change in registers for a whole chunk (XKIND is a compile-time constant of
// the worker's loop): the entry's index, its scaling D, the linear cost of x_0 and the row indices of the gather — 2 LDS reads and
// the index arithmetic (a division by 3, the clamps at the window's ends) less per iteration, in the phase that the solve waits for.
template <int XKIND>
struct XcolRegs {
    bool valid, hn, hp;
    int i;        // entry of xb / xs
    int r[4];     // XKIND 0: rd(kn, a), rv(kn, a), rd(kp, a), rv(kp, a); 1: rm(k, a), rd(kn, 3 + a), rd(kn, a), rd(kp, 3 + a); 2: rd(kn, 6 + a), 3 kn + a (gb), rd(kp, 6 + a)
    double dv, q0;
};
template <int XKIND, class Q>
DEKF_FN void xcols_regs_load(const Q& q, int lane, XcolRegs<XKIND>& x) {
    const int K = q.K;
    x.valid = lane < 3 * K;
    const int ln = x.valid ? lane : 0;
    const int k = ln / 3, a = ln - 3 * k, j = 3 * XKIND + a;
    x.i = 9 * k + j;
    x.hn = k < K - 1; x.hp = k > 0;
    const int kn = x.hn ? k : 0, kp = x.hp ? k - 1 : 0;
    if constexpr (XKIND == 0) { x.r[0] = q.ix.rd(kn, a); x.r[1] = q.ix.rv(kn, a); x.r[2] = q.ix.rd(kp, a); x.r[3] = q.ix.rv(kp, a); }
    else if constexpr (XKIND == 1) { x.r[0] = q.ix.rm(k, a); x.r[1] = q.ix.rd(kn, 3 + a); x.r[2] = q.ix.rd(kn, a); x.r[3] = q.ix.rd(kp, 3 + a); }
    else { x.r[0] = q.ix.rd(kn, 6 + a); x.r[1] = 3 * kn + a; x.r[2] = q.ix.rd(kp, 6 + a); x.r[3] = 0; }
    x.dv = q.D[k * (21 + 3 * Q::LEGS) + j];
    const double qv = (q.tmp + TmpMap<9>::QSL)[j];
    x.q0 = k == 0 ? qv : 0.0;
}
template <int XKIND, class Q>
DEKF_FN void xcols_regs_tile(Q& q, const XcolRegs<XKIND>& x, double sigma) {
    if (!x.valid) return;
    cdptr at = q.at;
    const double xv = q.xb[x.i];
    double g;
    if constexpr (XKIND == 0) {  // (gather_pcol)
        const double n0 = at[x.r[0]], n1 = at[x.r[1]], p0 = at[x.r[2]], p1 = at[x.r[3]];
        g = (x.hn ? n0 + n1 : 0.0) - (x.hp ? p0 + p1 : 0.0);
    } else if constexpr (XKIND == 1) {  // (gather_vcol)
        const double n0 = at[x.r[1]], n1 = at[x.r[2]], p0 = at[x.r[3]];
        double gm = 0.0;
#pragma unroll
        for (int leg = 0; leg < Q::LEGS; ++leg) gm += at[x.r[0] + 3 * leg];
        g = gm + (x.hn ? fma(q.c.dt, n1, n0) : 0.0) - (x.hp ? p0 : 0.0);
    } else {
        const double n0 = at[x.r[0]], n1 = q.gb[x.r[1]], p0 = at[x.r[2]];
        g = (x.hn ? n0 - n1 : 0.0) - (x.hp ? p0 : 0.0);
    }
    q.xs[x.i] = fma(x.dv, g, fma(sigma, xv, -x.q0));
}
template <int NF, class Q>
DEKF_FN void admm_chunk_r3(Q& q, int iters, double alpha, double sigma) {
    constexpr int L = Q::LEGS, SV = 21 + 3 * L;
    static_assert(3 * NF <= 64 && 2 * (NF - 1) <= 64, "one tile per kind");
    static_assert(NF * L <= 64 || (NF * L - 64) + 2 * (NF - 1) <= 64, "the Meas blocks beyond 64 fit the VO / bias wavefront");
    const int w = __builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6), lane = DEKF_LANE() & 63;
    // (the x blocks stay in LDS, q.xb, for the whole solve)
    // -DDEKF_PROFILE -DDEKF_PROFILE_TL: per-wavefront intervals, summed over the iterations (tools/profile_sections.py, DEKF_TIMELINE=1)
    //   prof[w]: w0 the solve, workers the x-column tile | prof[4 + w]: w0 its wait from B2 to B1, workers the row tile |
    //   prof[8 + w]: workers' wait for the solve (B1 to B2)
#if defined(DEKF_PROFILE_TL)
#define DEKF_R3_T(var) const long long var = clock64()
#else
#define DEKF_R3_T(var) ((void)0)
#endif
    if (w == 0) {
        DEKF_SYNC();  // B0: w, gb of the (re)start complete
        DEKF_PROF_MARK(q, 14);
        for (int it = 0; it < iters; ++it) {
            DEKF_R3_T(t0);
            DEKF_SYNC();  // B1
            DEKF_R3_T(t1);
            __builtin_amdgcn_s_setprio(3);
            sweeps_one_wave<NF>(q, alpha, lane);
            __builtin_amdgcn_s_setprio(0);
#if defined(DEKF_PROFILE_TL)
            __builtin_amdgcn_s_waitcnt(0);
#endif
            DEKF_R3_T(t2);
            DEKF_SYNC();  // B2
            DEKF_SYNC();  // B3
            DEKF_R3_T(t3);
            DEKF_TL_ADD(q, 0, t1, t2);
            DEKF_TL_ADD(q, 4, t2, t3);
            DEKF_TL_ADD(q, 8, t0, t1);
        }
    } else {
        // one loop per kind of row block (wave-uniform branch): each is compiled with the registers of ITS block kind only
        auto worker = [&](auto tag) {
            constexpr int KIND = decltype(tag)::value;
            RowRegsT<KIND> t;
            row_regs_load<KIND>(q, lane, sigma, t);
            DEKF_SYNC();  // B0
            constexpr int xkind = KIND == 0 ? 1 : (KIND == 1 ? 2 : 0);  // velocity columns (the longest gather) next to the shortest row tile
            const bool cold = q.cold;
            XcolRegs<xkind> xc;
            xcols_regs_load<xkind>(q, lane, xc);
            for (int it = 0; it < iters; ++it) {
                DEKF_R3_T(t0);
                xcols_regs_tile<xkind>(q, xc, sigma);
#if defined(DEKF_PROFILE_TL)
                __builtin_amdgcn_s_waitcnt(0);
#endif
                DEKF_R3_T(t1);
                DEKF_SYNC();  // B1
                DEKF_SYNC();  // B2
                DEKF_R3_T(t2);
                if constexpr (KIND == 1) __builtin_amdgcn_s_setprio(2);
                row_regs_iter<KIND>(q, t, alpha, sigma, cold && it == 0 && !q.polishing());
                if constexpr (KIND == 1) __builtin_amdgcn_s_setprio(0);
#if defined(DEKF_PROFILE_TL)
                __builtin_amdgcn_s_waitcnt(0);
#endif
                DEKF_R3_T(t3);
                DEKF_SYNC();  // B3
                DEKF_TL_ADD(q, w, t0, t1);
                DEKF_TL_ADD(q, 4 + w, t2, t3);
                DEKF_TL_ADD(q, 8 + w, t1, t2);
            }
            row_regs_store<KIND>(q, t);
        };
        if (w == 1) worker(std::integral_constant<int, 0>{});
        else if (w == 2) worker(std::integral_constant<int, 1>{});
        else worker(std::integral_constant<int, 2>{});
    }
#undef DEKF_R3_T
    DEKF_PROF_MARK(q, 9);
    DEKF_SYNC();
    q.cold = false;
    DEKF_PROF_MARK(q, 15);
}

// ---------------------------------------------------------------- four workgroups of THREE wavefronts per CU (R4, round 6)
// The three-workgroup kernel's workers wait three quarters of an iteration for the solve wavefront; a CU holds 12 wavefronts at 168
// VGPRs either way.  Here a workgroup is the solve wavefront plus TWO workers that carry the three row tiles and the three x-column
// tiles between them, so that a CU holds FOUR solves (one solve wavefront per SIMD) instead of three:
//   wavefront 1   the Dyn lane pairs (the long pole of the row phase)             + the velocity columns (the longest gather)
//   wavefront 2   the first 64 Meas leg blocks, then VO / bias / remaining Meas   + the position and the bias columns
// Same barriers, same per-block arithmetic and the same owners' stores as admm_chunk_r3: bit-identical iterates.  LDS holds only what
// an iteration shares (SolveLayout::r4_*: 38.5 KiB); D, E, the stash of y / z and the factor-time product live in the slab.
template <int NF, class Q>
DEKF_FN void admm_chunk_r4(Q& q, int iters, double alpha, double sigma) {
    constexpr int L = Q::LEGS;
    static_assert(3 * NF <= 64 && 2 * (NF - 1) <= 64, "one tile per kind");
    static_assert(NF * L <= 64 || (NF * L - 64) + 2 * (NF - 1) <= 64, "the Meas blocks beyond 64 fit the VO / bias tile");
    const int w = __builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6), lane = DEKF_LANE() & 63;
#if defined(DEKF_PROFILE_TL)
#define DEKF_R4_T(var) const long long var = clock64()
#else
#define DEKF_R4_T(var) ((void)0)
#endif
    if (w == 0) {
        DEKF_SYNC();  // B0: w, gb of the (re)start complete
        DEKF_PROF_MARK(q, 14);
        for (int it = 0; it < iters; ++it) {
            DEKF_R4_T(t0);
            DEKF_SYNC();  // B1
            DEKF_R4_T(t1);
            __builtin_amdgcn_s_setprio(3);
            sweeps_one_wave<NF>(q, alpha, lane);
            __builtin_amdgcn_s_setprio(0);
#if defined(DEKF_PROFILE_TL)
            __builtin_amdgcn_s_waitcnt(0);
#endif
            DEKF_R4_T(t2);
            DEKF_SYNC();  // B2
            DEKF_SYNC();  // B3
            DEKF_R4_T(t3);
            DEKF_TL_ADD(q, 0, t1, t2);
            DEKF_TL_ADD(q, 4, t2, t3);
            DEKF_TL_ADD(q, 8, t0, t1);
        }
    } else if (w == 1) {
        RowRegsT<1> t;
        row_regs_load<1>(q, lane, sigma, t);
        DEKF_SYNC();  // B0
        const bool cold = q.cold;
        XcolRegs<1> xc;
        xcols_regs_load<1>(q, lane, xc);
        for (int it = 0; it < iters; ++it) {
            DEKF_R4_T(t0);
            xcols_regs_tile<1>(q, xc, sigma);
#if defined(DEKF_PROFILE_TL)
            __builtin_amdgcn_s_waitcnt(0);
#endif
            DEKF_R4_T(t1);
            DEKF_SYNC();  // B1
            DEKF_SYNC();  // B2
            DEKF_R4_T(t2);
            row_regs_iter<1>(q, t, alpha, sigma, cold && it == 0 && !q.polishing());
#if defined(DEKF_PROFILE_TL)
            __builtin_amdgcn_s_waitcnt(0);
#endif
            DEKF_R4_T(t3);
            DEKF_SYNC();  // B3
            DEKF_TL_ADD(q, w, t0, t1);
            DEKF_TL_ADD(q, 4 + w, t2, t3);
            DEKF_TL_ADD(q, 8 + w, t1, t2);
        }
        row_regs_store<1>(q, t);
    } else {
        RowRegsT<0> ta;
        RowRegsT<2> tb;
        row_regs_load<0>(q, lane, sigma, ta);
        row_regs_load<2>(q, lane, sigma, tb);
        DEKF_SYNC();  // B0
        const bool cold = q.cold;
        XcolRegs<0> xa;
        XcolRegs<2> xb;
        xcols_regs_load<0>(q, lane, xa);
        xcols_regs_load<2>(q, lane, xb);
        for (int it = 0; it < iters; ++it) {
            DEKF_R4_T(t0);
            xcols_regs_tile<0>(q, xa, sigma);
            xcols_regs_tile<2>(q, xb, sigma);
#if defined(DEKF_PROFILE_TL)
            __builtin_amdgcn_s_waitcnt(0);
#endif
            DEKF_R4_T(t1);
            DEKF_SYNC();  // B1
            DEKF_SYNC();  // B2
            DEKF_R4_T(t2);
            const bool fc = cold && it == 0 && !q.polishing();
            row_regs_iter<0>(q, ta, alpha, sigma, fc);
            row_regs_iter<2>(q, tb, alpha, sigma, fc);
#if defined(DEKF_PROFILE_TL)
            __builtin_amdgcn_s_waitcnt(0);
#endif
            DEKF_R4_T(t3);
            DEKF_SYNC();  // B3
            DEKF_TL_ADD(q, w, t0, t1);
            DEKF_TL_ADD(q, 4 + w, t2, t3);
            DEKF_TL_ADD(q, 8 + w, t1, t2);
        }
        row_regs_store<0>(q, ta);
        row_regs_store<2>(q, tb);
    }
#undef DEKF_R4_T
    DEKF_PROF_MARK(q, 9);
    DEKF_SYNC();
    q.cold = false;
    DEKF_PROF_MARK(q, 15);
}
#endif

#if DEKF_DEVICE_BUILD
// ---------------------------------------------------------------- rows in registers at a RUN-TIME horizon, two workgroups per CU (RR)
// PogoX (1 leg, N = 100) keeps 103 KB of iterates in LDS in the generic placement: ONE workgroup per CU, whose solve wavefront works
// through 101 dependent steps per iteration while three SIMDs idle.  Measured with emulation builds before this was written
// (profiles/r04_pogox_factor_in_lds_emulation.txt): the factor stream from the slab costs 15 %, a second resident workgroup is worth
// 1.75 x.  So the three-workgroup kernels' split — a row block's state in the registers of the lane that owns it for a whole chunk
// of iterations, LDS only for what crosses lanes, wave-specialised loops with matched barriers — at a run-time horizon:
//   LDS    R | D | E | x blocks | xd | w (= stash of the slack x between chunks) | xs | gb | scratch       77 KB at N = 100: two per CU
//   slab   factor (S^-1, W), slack-block inverses (BLOCK-major here: one base address and wide loads per block), scaled bounds, stash of y and of the VO rows' z between chunks
//   regs   per owned block: t, slack x, y, scaled bound [, z]  (12 / 15 doubles); a lane owns up to three blocks
// What a block needs beyond that in an iteration is re-read or rebuilt: its slack-block inverse from the slab (L2), E, D (and with
// them rho E D and E D, the same products as at the load: same bits) and R_k from LDS.
// Tiles of 64 blocks by kind (0 the first 64 Meas leg blocks, 1 Dyn position / velocity halves on lane pairs, 2 VO blocks, bias
// blocks and the Meas blocks beyond 64 on the generic projection path); the three worker wavefronts take them as (kind, tile)
// slots fixed at compile time, three each — valid for at most 1 + 4 + 4 tiles (SolveLayout::rr_fits):
//   worker 1: (0,0) (1,1) (2,0)     worker 2: (1,2) (2,1) (2,3)     worker 3: (1,0) (1,3) (2,2)
// The arithmetic is row_regs_iter's, called on a RowRegsT assembled from the kept state and the re-read constants.
// (Requesting the first block's inverse while the solve wavefront still works — before barrier B2 — was measured: 175.8 k against
// 177.3 k steps/s; not kept.)
template <int KIND>
struct RowStT {
    bool valid, vel, vo, meas;
    int k, r0, sv0;
    double t[3], xs[3], y[3], lo[3];
    double z[KIND == 2 ? 3 : 1];
};
template <int KIND, class Q>
DEKF_FN void rr_load(Q& q, int j, int lane, double sigma, RowStT<KIND>& st) {
    constexpr int L = Q::LEGS, NM = 3 * L, SV = 21 + NM;
    const int K = q.K, K1 = K - 1, nmeas = K * L, e = 64 * j + lane;
    RowRegsT<KIND> t;
    t.valid = false; t.k = 0; t.r0 = 0; t.sv0 = 0; t.vel = false; t.meas = false; t.vo = false; t.xo = 0;
    cdptr sp = q.Sv;
    if constexpr (KIND == 0) {
        if (e < nmeas && e < 64) {
            const int k = e / L, leg = e - k * L;
            t.valid = true; t.k = k; t.r0 = q.ix.rm(k, 3 * leg); t.sv0 = k * SV + 9 + 3 * leg; t.xo = 3;
            sp = q.Sv + 6 * e;
        }
    } else if constexpr (KIND == 1) {
        const int k = e >> 1;
        if (k < K1) { t.valid = true; t.k = k; t.vel = e & 1; t.r0 = q.ix.rd(k, t.vel ? 3 : 0); t.sv0 = k * SV + 9 + NM + (t.vel ? 3 : 0); }
    } else {
        if (e < 2 * K1) {
            t.valid = true; t.vo = e < K1;
            const int k = t.vo ? e : e - K1;
            t.k = k; t.xo = t.vo ? 0 : 6; t.r0 = t.vo ? q.ix.rv(k, 0) : q.ix.rd(k, 6); t.sv0 = k * SV + (t.vo ? 18 + NM : 9 + NM + 6);
        } else {  // the Meas blocks beyond the first 64, on the generic projection path (bit-identical for an equality block)
            const int m = 64 + e - 2 * K1;
            if (m < nmeas) {
                const int k = m / L, leg = m - k * L;
                t.valid = true; t.meas = true; t.k = k; t.xo = 3; t.r0 = q.ix.rm(k, 3 * leg); t.sv0 = k * SV + 9 + 3 * leg;
                sp = q.Sv + 6 * m;
            }
        }
    }
    st.valid = t.valid; st.vel = t.vel; st.vo = t.vo; st.meas = t.meas; st.k = t.k; st.r0 = t.r0; st.sv0 = t.sv0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { st.t[a] = 0.0; st.xs[a] = 0.0; st.y[a] = 0.0; st.lo[a] = 0.0; }
    st.z[0] = 0.0;
    if (!t.valid) return;
    row_regs_fill<KIND>(q, sigma, t, sp);
#pragma unroll
    for (int a = 0; a < 3; ++a) { st.t[a] = t.t[a]; st.xs[a] = t.xs[a]; st.y[a] = t.y[a]; st.lo[a] = t.lo[a]; }
    if constexpr (KIND == 2) {
#pragma unroll
        for (int a = 0; a < 3; ++a) st.z[a] = t.z[a];
    }
}
// the slack-block inverse of the block, from the slab (block-major for RR: solve_factor 3a), and what else row_regs_iter wants in registers
template <int KIND, class Q>
DEKF_FN void rr_assemble(const Q& q, const RowStT<KIND>& st, RowRegsT<KIND>& t) {
    constexpr int L = Q::LEGS;
    const int K = q.K, nmeas = K * L;
    // (block indices opaque per iteration: hoisted out of the iteration loop, the address arithmetic of up to four blocks — a dozen
    // pointers each — stays live across the x-column tiles and the barriers and pushes the kept state into scratch)
    int k_ = st.k, r0_ = st.r0, sv0_ = st.sv0;
    asm volatile("" : "+v"(k_), "+v"(r0_), "+v"(sv0_));
    t.valid = st.valid; t.k = k_; t.r0 = r0_; t.sv0 = sv0_; t.vel = st.vel; t.vo = st.vo; t.meas = st.meas;
    t.xo = (KIND == 0 || st.meas) ? 3 : (st.vo ? 0 : 6);
    if constexpr (KIND == 1) {
        const DynPairMat S(q.Sw + k_ * SWS, st.vel);
#pragma unroll
        for (int i = 0; i < 6; ++i) t.a[i] = S.a[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) t.b[i] = S.b[i];
        cdptr R = q.R + 9 * k_;
#pragma unroll
        for (int i = 0; i < 9; ++i) t.rk[i] = R[i];
    } else if constexpr (KIND == 2) {
        // (one code path for the three kinds of block on this tile: the six entries come from Sc, from the diagonal tail of Sw, or from Sv)
        if (st.meas) {
            cdptr sp = q.Sv + 2 * r0_;  // Meas block (k, leg): block index = its first row / 3, six entries each
#pragma unroll
            for (int i = 0; i < 6; ++i) t.a[i] = sp[i];
        } else {
            const VoOrBiasMat S(q.Sc + 6 * k_, q.Sw + k_ * SWS + 21, st.vo);
#pragma unroll
            for (int i = 0; i < 6; ++i) t.a[i] = S.p[i];
        }
        // upper bounds: a bias block is an equality (hi = lo); a VO block is an equality or the box -+1e30 E, whose upper end is
        // exactly -lo — so the array of upper bounds need not be read back (solve_window_t writes hi = ub E next to lo = lb E)
#pragma unroll
        for (int a = 0; a < 3; ++a) t.b[a] = st.lo[a] < -1e20 ? -st.lo[a] : st.lo[a];
    } else {
        cdptr sp = q.Sv + 2 * r0_;  // Meas block (k, leg): block index k L + leg = its first row / 3, six entries each
#pragma unroll
        for (int i = 0; i < 6; ++i) t.a[i] = sp[i];
    }
    const double rho_eq = uni(RHO_EQ_OVER_RHO_INEQ * q.rho);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double d = q.D[sv0_ + a];
        t.e[a] = q.E[r0_ + a];
        t.lo[a] = st.lo[a];
        // (row_regs_fill's expressions: rho E D = (rv E) D, E D = D E)
        double rv = rho_eq;
        if constexpr (KIND == 2) rv = q.rho_of(st.lo[a], t.b[a]);
        t.cf[a] = rv * t.e[a] * d;
        t.c2[a] = d * t.e[a];
        t.t[a] = st.t[a]; t.xs[a] = st.xs[a]; t.y[a] = st.y[a];
    }
    if constexpr (KIND == 2) {
#pragma unroll
        for (int a = 0; a < 3; ++a) t.z[a] = st.z[a];
    }
}
template <int KIND, class Q>
DEKF_FN void rr_iter(Q& q, RowStT<KIND>& st, double alpha, double sigma, bool first_cold) {
    if (!st.valid) return;
    RowRegsT<KIND> t;
    rr_assemble<KIND>(q, st, t);
    row_regs_iter<KIND>(q, t, alpha, sigma, first_cold);
#pragma unroll
    for (int a = 0; a < 3; ++a) { st.t[a] = t.t[a]; st.xs[a] = t.xs[a]; st.y[a] = t.y[a]; }
    if constexpr (KIND == 2) {
#pragma unroll
        for (int a = 0; a < 3; ++a) st.z[a] = t.z[a];
    }
}
template <int KIND, class Q>
DEKF_FN void rr_store(Q& q, const RowStT<KIND>& st) {
    if (!st.valid) return;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int r = st.r0 + a;
        q.sx[r] = st.xs[a];
        q.sy[r] = st.y[a];
        if constexpr (KIND == 2) { if (st.vo) q.sz[r - q.ix.rvb] = st.z[a]; }
    }
}
template <class Q>
DEKF_FN void admm_chunk_rr(Q& q, int iters, double alpha, double sigma) {
    const int w = __builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6), lane = DEKF_LANE() & 63;
    const int K = q.K, ntx = (3 * K + 63) >> 6;
    if (w == 0) {
        DEKF_SYNC();  // B0: w, gb of the (re)start complete
        DEKF_PROF_MARK(q, 14);
        for (int it = 0; it < iters; ++it) {
            DEKF_SYNC();  // B1: xs complete
            __builtin_amdgcn_s_setprio(3);
            sweeps_one_wave_rt(q, alpha);
            __builtin_amdgcn_s_setprio(0);
            DEKF_SYNC();  // B2: xd complete
            DEKF_SYNC();  // B3: w, gb complete
        }
    } else {
        const bool cold = q.cold, pol = q.polishing();
        // this worker's x-column tiles: flat tile f = kind * ntx + j, every third one
        auto xcols = [&] {
            for (int f = w - 1; f < 3 * ntx; f += 3) {
                const int kind = f / ntx, j = f - kind * ntx;
                xcols_tile_r3(q, kind, 64 * j + lane, sigma);
            }
        };
#define DEKF_RR_LOOP(ITERS_BODY, STORE_BODY)                      \
        DEKF_SYNC(); /* B0 */                                      \
        for (int it = 0; it < iters; ++it) {                       \
            xcols();                                               \
            DEKF_SYNC(); /* B1 */                                  \
            DEKF_SYNC(); /* B2 */                                  \
            const bool fc = cold && it == 0 && !pol;               \
            ITERS_BODY                                             \
            DEKF_SYNC(); /* B3 */                                  \
        }                                                          \
        STORE_BODY
        if (w == 1) {
            RowStT<0> s0; RowStT<1> s1; RowStT<2> s2;
            rr_load<0>(q, 0, lane, sigma, s0); rr_load<1>(q, 1, lane, sigma, s1); rr_load<2>(q, 0, lane, sigma, s2);
            DEKF_RR_LOOP(rr_iter<1>(q, s1, alpha, sigma, fc); rr_iter<2>(q, s2, alpha, sigma, fc); rr_iter<0>(q, s0, alpha, sigma, fc);,
                         rr_store<0>(q, s0); rr_store<1>(q, s1); rr_store<2>(q, s2);)
        } else if (w == 2) {
            RowStT<1> s0; RowStT<2> s1, s2;
            rr_load<1>(q, 2, lane, sigma, s0); rr_load<2>(q, 1, lane, sigma, s1); rr_load<2>(q, 3, lane, sigma, s2);
            DEKF_RR_LOOP(rr_iter<1>(q, s0, alpha, sigma, fc); rr_iter<2>(q, s1, alpha, sigma, fc); rr_iter<2>(q, s2, alpha, sigma, fc);,
                         rr_store<1>(q, s0); rr_store<2>(q, s1); rr_store<2>(q, s2);)
        } else {
            RowStT<1> s0, s1; RowStT<2> s2;
            rr_load<1>(q, 0, lane, sigma, s0); rr_load<1>(q, 3, lane, sigma, s1); rr_load<2>(q, 2, lane, sigma, s2);
            DEKF_RR_LOOP(rr_iter<1>(q, s0, alpha, sigma, fc); rr_iter<1>(q, s1, alpha, sigma, fc); rr_iter<2>(q, s2, alpha, sigma, fc);,
                         rr_store<1>(q, s0); rr_store<1>(q, s1); rr_store<2>(q, s2);)
        }
#undef DEKF_RR_LOOP
    }
    DEKF_PROF_MARK(q, 9);
    DEKF_SYNC();
    q.cold = false;
    DEKF_PROF_MARK(q, 15);
}
#endif

// one ADMM iteration after phase X: the block-tridiagonal solve, then the rows
template <class Q>
DEKF_FN void phase_sweeps_rows(Q& q, double alpha, double sigma) {
#if DEKF_DEVICE_BUILD
    constexpr int NF = Q::NFIXED, L = Q::LEGS;
    if constexpr (NF >= 4 && NF % 2 == 0 && Q::NS == 9) {
        const int K = q.K;
        const int ntiles = ((K * L + 63) >> 6) + 2 * ((2 * (K - 1) + 63) >> 6);
        if (K == NF && ntiles <= wave_count()) {
            const int w = __builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6), lane = DEKF_LANE() & 63;
            RowTile t;
            t.kind = -1;
#if defined(DEKF_PROFILE_TL)
            const long long tl0 = clock64();
#endif
            // With at most three tiles (one Meas tile: 2 legs) the three waiting wavefronts take them all and
            // prefetch while the solve runs; with four (Go1) the solve wavefront does the first Meas tile behind
            // its solve, unprefetched (folding the fourth tile into another wavefront was slower: EXPERIMENTS.md II §7).
            constexpr bool spare = ((NF * L + 63) >> 6) + 2 * ((2 * (NF - 1) + 63) >> 6) <= 3;
            const int tile = spare ? w - 1 : w;
            if (w == 0) {
                __builtin_amdgcn_s_setprio(3);
                sweeps_one_wave<NF>(q, alpha);
                __builtin_amdgcn_s_setprio(0);
            } else if (tile < ntiles) {
                row_tile_load(q, tile, lane, t);
            }
#if defined(DEKF_PROFILE_TL)
            const long long tl1 = clock64();
            DEKF_TL_ADD(q, w, tl0, tl1);
#endif
            DEKF_SYNC();
            DEKF_PROF_MARK(q, 5);
#if defined(DEKF_PROFILE_TL)
            const long long tl2 = clock64();
#endif
            if (!spare && w == 0) row_tile_load(q, 0, lane, t);
            // the Dyn lane-pair tile is the long pole of the row phase (profiles/r01_final_timeline.txt): its wavefront
            // gets priority on the SIMD it shares with a wavefront of the other resident workgroup
            const bool long_pole = tile == ((NF * L + 63) >> 6);
            if (long_pole) __builtin_amdgcn_s_setprio(2);
            row_tile_finish(q, t, alpha, sigma);
            if (long_pole) __builtin_amdgcn_s_setprio(0);
#if defined(DEKF_PROFILE_TL)
            __builtin_amdgcn_s_waitcnt(0);  // the tile's stores have left the wavefront
            const long long tl3 = clock64();
#if !defined(DEKF_PROFILE_TLX)
            DEKF_TL_ADD(q, 4 + w, tl2, tl3);
#endif
#endif
            DEKF_SYNC();
#if defined(DEKF_PROFILE_TL) && !defined(DEKF_PROFILE_TLX)
            DEKF_TL_ADD(q, 8 + w, tl3, clock64());
#endif
            return;
        }
    }
#endif
    phase_sweeps(q, alpha);
    phase_rows<false>(q, alpha, sigma);
}

// ---------------------------------------------------------------- residual norms
// What OSQP's compute_pri_res / compute_dua_res / compute_pri_tol / compute_dua_tol / compute_rho_estimate
// need, in one pass: ra = {|Ax-z|/E, |z|/E, |Ax|/E, |Ax-z|, |z|, |Ax|} (inf-norms over the rows) and
// va = {|Px+q+A'y|/D, |q|/D, |A'y|/D, |Px|/D, and the same four scaled} (over the variables).
// Same lane mapping as the row phase: a lane owns a 3-row block AND its slack block (the slack of a row
// lives in that row only), so Ax, P_s x_s and A'y of the block are lane-local; the x columns go through
// the branch-free gathers.  The P blocks come from the window records in HBM (requested first, they
// arrive while the LDS part is being computed).  Replaces two wfor sweeps over rows / variables with
// per-item kind decoding (50 k cycles per check on Go1, 3 checks per solve).
// rv != nullptr: additionally the scaled KKT residual VECTORS of the current point, for the refinement steps of OSQP's polishing
// (polish.c: rhs - K sol):  rxb[k NS + j] = -(q + P x + A'y) on the x blocks, rxs[r] the same on the slack variable of row r,
// ry[r] = z_r - (A x)_r, i.e. b - A x on a row held at its bound
struct ResidVec { dptr rxb, rxs, ry; };
template <class Q>
DEKF_FN void residual_norms(Q& q, dptr ra, dptr va, const ResidVec* rv = nullptr) {
    constexpr int L = Q::LEGS, NM = 3 * L, FT = Q::FOOT, NS = Q::NS, SV = 2 * NS + 3 + NM;
    const int K = q.K, K1 = K - 1, nmeas = K * L;
    const double dt = q.c.dt, hdt2 = q.c.hdt2, cc = q.cc;
    dptr x = q.x, z = q.z, y = q.y, xd = q.xd;
    cdptr D = q.D, E = q.E;
    // Where the iterates are.  Three-workgroup kernels (R3): x blocks compact in LDS (xb); the slack x and y by ROW in sx / sy, z of
    // the VO rows in sz (what the last chunk of iterations left, admm_chunk_r3); z of an equality row is its scaled bound (the
    // projection returns it from the first iteration on).  Everywhere else: the full vectors x, z, y.
    auto XB = [&](int k, int j) -> double { if constexpr (Q::R3) return q.xb[NS * k + j]; else return x[k * SV + j]; };
    auto XS = [&](int sv, int r) -> double { if constexpr (Q::R3) { (void)sv; return q.sx[r]; } else { (void)r; return x[sv]; } };
    auto YR = [&](int r) -> double { if constexpr (Q::R3) return q.sy[r]; else return y[r]; };
    auto ZR = [&](int r, bool vo) -> double {
        if constexpr (Q::R3) return vo ? q.sz[r - q.ix.rvb] : q.lo[r];
        else { (void)vo; return z[r]; }
    };
    wfor(K * NS, [&](int e) {
        int k = e / NS, j = e - NS * k;
        xd[e] = D[k * SV + j] * XB(k, j);
    });
    DEKF_PROF_MARK(q, 20);
    double acc[14];
#pragma unroll
    for (int r = 0; r < 14; ++r) acc[r] = 0.0;
    // rows r0.. and slack variables sv0.. of one 3-block: ar = E .* (A_x D x), ps = unscaled P_s (D_s x_s); vo: a VO block
    auto block = [&](int r0, int sv0, cdptr ar, cdptr ps, bool vo = false) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int r = r0 + j, sv = sv0 + j;
            const double e = E[r], d = D[sv];
            const double Ax = ar[j] - e * d * XS(sv, r);
            const double zr = ZR(r, vo);
            const double pr = Ax - zr, ei = rcp_fast(e);
            acc[0] = dmax(acc[0], fabs(pr) * ei);
            acc[1] = dmax(acc[1], fabs(zr) * ei);
            acc[2] = dmax(acc[2], fabs(Ax) * ei);
            acc[3] = dmax(acc[3], fabs(pr));
            acc[4] = dmax(acc[4], fabs(zr));
            acc[5] = dmax(acc[5], fabs(Ax));
            const double Px = cc * d * ps[j], Aty = -e * d * YR(r);
            const double dr = Px + Aty, di = rcp_fast(d);
            if (rv) { rv->ry[r] = -pr; rv->rxs[r] = -dr; }
            acc[6] = dmax(acc[6], fabs(dr) * di);
            acc[8] = dmax(acc[8], fabs(Aty) * di);
            acc[9] = dmax(acc[9], fabs(Px) * di);
            acc[10] = dmax(acc[10], fabs(dr));
            acc[12] = dmax(acc[12], fabs(Aty));
            acc[13] = dmax(acc[13], fabs(Px));
        }
    };
    // a 3-block whose P block is a packed symmetric 3x3 in the window record
    auto block_sym3 = [&](int r0, int sv0, cdptr ar, cdptr q6, bool vo = false) {
        double p6[6], dx[3], ps[3];
#pragma unroll
        for (int t = 0; t < 6; ++t) p6[t] = ld_stream_resid(q6, t);
#pragma unroll
        for (int a = 0; a < 3; ++a) dx[a] = D[sv0 + a] * XS(sv0 + a, r0 + a);
#pragma unroll
        for (int a = 0; a < 3; ++a)
            ps[a] = p6[symidx(0, a, 3)] * dx[0] + p6[1 < a ? symidx(1, a, 3) : symidx(a, 1, 3)] * dx[1] + p6[symidx(a, 2, 3)] * dx[2];
        block(r0, sv0, ar, ps, vo);
    };
    const int ntm = (nmeas + 63) >> 6, ntp = (2 * K1 + 63) >> 6, ntd = (K1 + 63) >> 6, ntx = (3 * K + 63) >> 6;
    const int ntf = FT ? (K1 * L + 63) >> 6 : 0, ntxf = FT ? (NM * K + 63) >> 6 : 0;
    cdptr qsl = q.tmp + TmpMap<NS>::QSL;
    auto wy = [&](int r) { return E[r] * YR(r); };
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_RESID2) && DEKF_DEVICE_BUILD  // per-wavefront: tile phase (slots 24..27), its share of the wait + reduction (28..31)
    const long long trs0 = clock64();
#endif
    wtiles(ntm + ntp + 2 * ntd + 3 * ntx + ntf + ntxf, [&](int tile, int lane) {
        if (tile < ntm) {  // Meas leg blocks
            const int e = tile * 64 + lane;
            if (e >= nmeas) return;
            const int k = e / L, leg = e - k * L;
            const int r0 = q.ix.rm(k, 3 * leg), sv0 = q.ix.v(k, 3 * leg);
            double ar[3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                ar[a] = FT ? E[r0 + a] * (xd[NS * k + 9 + 3 * leg + a] - xd[NS * k + a]) : E[r0 + a] * xd[NS * k + 3 + a];
            block_sym3(r0, sv0, ar, q.rec(k) + Rec::qm(NM) + 6 * leg);
            return;
        }
        int td = tile - ntm;
        if (td < ntp) {  // Dyn position / velocity rows: lane pair per step (no exchange needed here)
            const int pl = td * 64 + lane, k = pl >> 1;
            if (k >= K1) return;
            const bool vel = pl & 1;
            const int o = vel ? 3 : 0;
            const int r0 = q.ix.rd(k, o), sv0 = q.ix.w(k, o), w0 = q.ix.w(k, 0);
            cdptr q21 = q.rec(k) + Rec::QD;
            double p21[21], dx[6], ps[3], ar[3];
#pragma unroll
            for (int t = 0; t < 21; ++t) p21[t] = ld_stream_resid(q21, t);
#pragma unroll
            for (int t = 0; t < 6; ++t) dx[t] = D[w0 + t] * XS(w0 + t, q.ix.rd(k, t));
            cdptr xk = xd + NS * k;
            cdptr R = q.R + 9 * k;
            const double c1 = vel ? 0.0 : dt, c2 = vel ? dt : hdt2;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double rb = R[3 * a] * xk[6] + R[3 * a + 1] * xk[7] + R[3 * a + 2] * xk[8];
                ar[a] = E[r0 + a] * (xk[o + a] + c1 * xk[3 + a] - c2 * rb - xk[NS + o + a]);
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    s0 += p21[a < t ? symidx(a, t, 6) : symidx(t, a, 6)] * dx[t];
                    s1 += p21[3 + a < t ? symidx(3 + a, t, 6) : symidx(t, 3 + a, 6)] * dx[t];
                }
                ps[a] = vel ? s1 : s0;
            }
            block(r0, sv0, ar, ps);
            return;
        }
        td -= ntp;
        if (td < 2 * ntd) {  // VO rows, then bias rows: the four kinds that fetch P from HBM come first, one per wavefront
            const bool vo = td < ntd;
            const int k = (td - (vo ? 0 : ntd)) * 64 + lane;
            if (k >= K1) return;
            cdptr xk = xd + NS * k;
            double ar[3], ps[3];
            if (!vo) {
                const int r0 = q.ix.rd(k, 6), sv0 = q.ix.w(k, 6);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    ar[a] = E[r0 + a] * (xk[6 + a] - xk[NS + 6 + a]);
                    ps[a] = q.c.Q_bias_dt2[a] * D[sv0 + a] * XS(sv0 + a, r0 + a);
                }
                block(r0, sv0, ar, ps);
            } else {
                const int r0 = q.ix.rv(k, 0), sv0 = q.ix.c(k, 0);
#pragma unroll
                for (int a = 0; a < 3; ++a) ar[a] = E[r0 + a] * (xk[a] - xk[NS + a]);
                block_sym3(r0, sv0, ar, q.rec(k) + Rec::QC, true);
            }
            return;
        }
        td -= 2 * ntd;
        if (td < 3 * ntx) {  // x columns: position / velocity / bias tiles
            const int kind = td < ntx ? 0 : (td < 2 * ntx ? 1 : 2);
            const int e = (td - kind * ntx) * 64 + lane;
            if (e >= 3 * K) return;
            const int k = e / 3, a = e - 3 * k, j = 3 * kind + a, i = k * SV + j;
            const double d = D[i], qv = k == 0 ? qsl[j] : 0.0;
            double g;
            if (kind == 0) g = gather_pcol(q, k, a, wy);
            else if (kind == 1) g = gather_vcol(q, k, a, wy);
            else g = gather_bcol(q, k, a, wy);
            double Px = 0.0;
            if (k == 0) {  // the arrival cost is the only Hessian on an x block
                for (int t = 0; t < NS; ++t) Px += (j <= t ? q.Mp[NS * j + t] : q.Mp[NS * t + j]) * xd[t];
                Px *= cc * d;
            }
            const double Aty = d * g, dr = qv + Px + Aty, di = rcp_fast(d);
            if (rv) rv->rxb[NS * k + j] = -dr;
            acc[6] = dmax(acc[6], fabs(dr) * di);
            acc[7] = dmax(acc[7], fabs(qv) * di);
            acc[8] = dmax(acc[8], fabs(Aty) * di);
            acc[9] = dmax(acc[9], fabs(Px) * di);
            acc[10] = dmax(acc[10], fabs(dr));
            acc[11] = dmax(acc[11], fabs(qv));
            acc[12] = dmax(acc[12], fabs(Aty));
            acc[13] = dmax(acc[13], fabs(Px));
            return;
        }
        td -= 3 * ntx;
        if constexpr (FT) {
            if (td < ntf) {  // Dyn rows of the foot-position states
                const int e = td * 64 + lane;
                if (e >= K1 * L) return;
                const int k = e / L, leg = e - k * L;
                const int r0 = q.ix.rd(k, 9 + 3 * leg), sv0 = q.ix.w(k, 9 + 3 * leg);
                cdptr xk = xd + NS * k + 9 + 3 * leg;
                double ar[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) ar[a] = E[r0 + a] * (xk[a] - xk[NS + a]);
                block_sym3(r0, sv0, ar, q.rec(k) + Rec::qf(NM) + 6 * leg);
                return;
            }
            td -= ntf;
            const int e = td * 64 + lane;  // foot-position columns
            if (e >= NM * K) return;
            const int k = e / NM, la = e - NM * k, j = 9 + la, i = k * SV + j;
            const double d = D[i], qv = k == 0 ? qsl[j] : 0.0;
            const double g = gather_fcol(q, k, la, wy);
            double Px = 0.0;
            if (k == 0) {
                for (int t = 0; t < NS; ++t) Px += (j <= t ? q.Mp[NS * j + t] : q.Mp[NS * t + j]) * xd[t];
                Px *= cc * d;
            }
            const double Aty = d * g, dr = qv + Px + Aty, di = rcp_fast(d);
            if (rv) rv->rxb[NS * k + j] = -dr;
            acc[6] = dmax(acc[6], fabs(dr) * di);
            acc[7] = dmax(acc[7], fabs(qv) * di);
            acc[8] = dmax(acc[8], fabs(Aty) * di);
            acc[9] = dmax(acc[9], fabs(Px) * di);
            acc[10] = dmax(acc[10], fabs(dr));
            acc[11] = dmax(acc[11], fabs(qv));
            acc[12] = dmax(acc[12], fabs(Aty));
            acc[13] = dmax(acc[13], fabs(Px));
        }
    });
    DEKF_PROF_MARK(q, 21);
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_RESID2) && DEKF_DEVICE_BUILD
    __builtin_amdgcn_s_waitcnt(0);
    const long long trs1 = clock64();
#endif
    wave_max_n<14>(acc);
    group_combine<14, false>(acc);
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_RESID2) && DEKF_DEVICE_BUILD
    {
        const long long trs2 = clock64();
        const int w_ = DEKF_LANE() >> 6;
        if ((DEKF_LANE() & 63) == 0) { q.prof[24 + w_] += (double)(trs1 - trs0); q.prof[28 + w_] += (double)(trs2 - trs1); }
    }
#endif
#pragma unroll
    for (int r = 0; r < 6; ++r) ra[r] = acc[r];
#pragma unroll
    for (int r = 0; r < 8; ++r) va[r] = acc[6 + r];
}
#pragma clang fp contract(fast)  // (what follows the include in mhe_solve_core.h is shared by every kernel shape)
