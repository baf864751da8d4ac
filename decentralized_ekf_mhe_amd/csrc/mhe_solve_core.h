// mhe_solve_core.h — per-instance (one workgroup of four wavefronts) OSQP-style ADMM solve of the MHE
// window QP: Ruiz scaling, factorisation and the solve driver; the phases of an ADMM iteration and the
// residual norms are in mhe_admm_core.h.
//
// Replaces  MHEproblem::initQP + solveQP + getsolution  (src/decentral_legged_est/src/MheSrb.cpp:
// 272-349, 715-723), i.e. osqp_setup + osqp_solve on
//     min 1/2 x'Hx + g'x   s.t.  l <= Ax <= u
// with H, g, A, l, u as MHEproblem::updateQP / marginalizeQP leave them (:351-447, :475-713),
// and the tail of DecentralizedEstimation::update (DecentralEst.cpp:179-185).
//
// Same algorithm as OSQP (Ruiz equilibration, per-row rho, sigma/alpha-relaxed ADMM,
// termination every check_termination iterations on unscaled residuals, adaptive rho with
// refactorisation, cold start), different linear algebra: the QP is never assembled.
// Per window step k the variables are [x_k(9) v_k(3L) w_k(9) c_k(3)] and the rows
// [Meas_k(3L) Dyn_k(9) VO_k(3)], every slack v/w/c appears in exactly
// one row with coefficient -1, and H is block diagonal.  The ADMM linear system
//     (P + sigma I + A' diag(rho) A) xt = sigma x - q + A'(diag(rho) z - y),   zt = A xt
// (OSQP's KKT system with nu eliminated) therefore reduces, after eliminating the slack blocks
// (3x3 per leg, 6x6+3 for w, 3x3 for c), to a block-tridiagonal SPD system in the x_k with
// 9x9 blocks — a fixed-interval smoother — solved by a TWO-SIDED block LDL' (blocks eliminated from both
// ends of the window towards block (K-1)/2): S_k^-1 (full 9x9) and W_k = C_k S_k^-1 / W^_k = C_k' S_{k+1}^-1
// are the "banded KKT factor", refreshed whenever rho changes.
//
// Memory (SolveMem): when it fits 2 workgroups per CU (Go1/Cassie at N = 20: <= 80 KiB) the
// ADMM iterates, the scaling vectors, the bounds, the slack-block inverses, the block
// tridiagonal factor and the rotations all live in LDS, so an ADMM iteration touches HBM only
// at the termination checks (window records, every 25 iterations).  Longer windows (PogoX,
// N = 100) keep the iterates in LDS and stream the factor from the workgroup's HBM scratch slab.
#pragma once
#include "cfg.h"
#include "mhe_assemble_core.h"
#include "smallmat.h"

namespace dekf {

// Section stamps for the DIAGNOSTIC build only (-DDEKF_PROFILE -> libdekf_prof.so); the product
// kernel executes none of this.  Stamp values go to DevState::prof and nowhere else.
// -DDEKF_PROFILE_TL (with -DDEKF_PROFILE): instead of the section stamps, every wavefront of an instance adds up
// three intervals of the fixed-horizon iteration (tools/profile_sections.py --timeline):
//   prof[w]      solve (w = 0) / row-tile prefetch (w > 0) until the wavefront reaches the barrier behind the solve
//   prof[4 + w]  its row-tile work behind that barrier
//   prof[8 + w]  its wait at the barrier that ends the row phase
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_TL) && DEKF_DEVICE_BUILD
#define DEKF_PROF_MARK(q, sec) ((void)0)
#define DEKF_TL_ADD(q, slot, t0, t1)                                            \
    do {                                                                        \
        if ((DEKF_LANE() & 63) == 0) (q).prof[slot] += (double)((t1) - (t0));   \
    } while (0)
#elif defined(DEKF_PROFILE) && DEKF_DEVICE_BUILD
#define DEKF_TL_ADD(q, slot, t0, t1) ((void)0)
#define DEKF_PROF_MARK(q, sec)                                                  \
    do {                                                                        \
        if (DEKF_LANE() == 0) {                                                 \
            long long now_ = clock64();                                         \
            (q).prof[sec] += (double)(now_ - (q).prof_last);                    \
            (q).prof_last = now_;                                               \
        }                                                                       \
    } while (0)
#else
#define DEKF_PROF_MARK(q, sec) ((void)0)
#define DEKF_TL_ADD(q, slot, t0, t1) ((void)0)
#endif

// The meeting block of the two-sided block-tridiagonal factorisation / solve.  (K - 1) / 2 makes the
// two forward legs equally long for an even window (9 + 9 steps at K = 20; K / 2 gave 10 + 8) and the
// critical path forward + outward 9 + 10 steps instead of 10 + 10.
DEKF_HD constexpr int mid_block(int K) { return (K - 1) / 2; }
constexpr int SWS = 25;  // doubles per step of Sw (21 packed 6x6 + 3 bias diagonal + 1 pad: an odd stride keeps one-lane-per-step reads off the same banks)
constexpr int SOLVE_TMP = 344;  // [162,171) scaled q; at factor time [0,162) and [176,338): two Gauss-Jordan ping-pong pairs
// Where things sit inside q.tmp.  NS = 9 keeps the round-1 map (above); with foot-position states (NS = 9 + 3 L)
// the scaled q comes first, then one factorisation scratch per side of the two-sided block LDL' (S, then S^-1:
// NS^2, plus a pivot row / column for the lane-sequential build).
template <int NS>
struct TmpMap {
    static constexpr int QSL = NS == 9 ? 162 : 0;
    static constexpr int FAIL = NS == 9 ? 172 : NS;
    static constexpr int SIDE0 = NS == 9 ? 0 : NS + 8;
    static constexpr int SIDE1 = NS == 9 ? 176 : SIDE0 + NS * NS + 2 * NS;
    static constexpr int LEN = NS == 9 ? SOLVE_TMP : SIDE1 + NS * NS + 2 * NS + 8;
};
DEKF_HD constexpr int solve_tmp_len(int ns) { return ns == 9 ? SOLVE_TMP : 2 * (ns * ns + 2 * ns) + ns + 16; }

// how many doubles of LDS a solve needs in each placement mode
struct SolveLayout {
    int n_pad, m_pad, K, ns;
    int vec;        // iterates: x z y xt zt at xs xd tmp
    int resident;   // D E lo hi(VO rows) Sv Sw Sc [Sf] Sinv(full NS x NS) Wk R
    DEKF_HD void init(int N, int L, int ft = 0) {
        int nm = 3 * L;
        K = N;
        ns = 9 + (ft ? nm : 0);
        n_pad = N * (2 * ns + nm + 3);
        m_pad = N * (nm + ns + 3);
        vec = 2 * n_pad + 4 * m_pad + 2 * ns * N + solve_tmp_len(ns);
        resident = n_pad + 2 * m_pad + 3 * N + N * (6 * L + SWS + 6 + (ft ? 6 * L : 0)) + N * 2 * ns * ns + 9 * N;
    }
    // LDS-resident factor only if two workgroups still fit in one CU's 160 KiB
    DEKF_HD bool factor_in_lds() const { return (size_t)(vec + resident) * 8 <= 80 * 1024; }
    // the factor-time product P A_dyn can alias the (then dead) xt|zt|at vectors when they are big enough
    DEKF_HD bool pa_in_lds() const { return factor_in_lds() && (n_pad + 2 * m_pad >= (K - 1) * ns * ns); }
    // Three-workgroup placement (R3, fixed-horizon kernels with the full window only): the row phase keeps its state and its
    // constants in REGISTERS for a whole chunk of iterations, so LDS holds only what crosses lanes:
    //   R | D E (the scaling vectors: every phase reads them) | xb (x blocks, compact) | [xd at xs gb sy sz + pad] = factor-time PA |
    //   tmp | Sinv | Wk
    // Between chunks of iterations the row state (slack x, y, z of the VO rows) rests in LDS too: sx aliases `at` entry for entry
    // (a lane overwrites only what it has just read), sy | sz sit in the part of the PA region the iterations leave unused.
#ifndef DEKF_R3_WAVES
#define DEKF_R3_WAVES 3     // resident workgroups per CU the R3 kernels are compiled for (launch bound: wavefronts per SIMD)
#endif
#ifndef DEKF_R3_PA_LDS
#define DEKF_R3_PA_LDS 1    // 0: the factor-time product P A in the HBM slab (40 KiB of LDS per instance: a fourth workgroup fits)
#endif
    // xd | at (= the stash of the slack x between chunks) | xs | gb | stash of y | stash of z (VO rows): the factor-time PA aliases all of it
    DEKF_HD int r3_pa_region() const { int a = 2 * m_pad + 2 * ns * K + 3 * K + 3 * K, b = DEKF_R3_PA_LDS ? (K - 1) * ns * ns : 0; return a > b ? a : b; }
    DEKF_HD int r3_doubles() const { return 9 * K + n_pad + m_pad + ns * K + r3_pa_region() + solve_tmp_len(ns) + 2 * K * ns * ns; }
    DEKF_HD size_t r3_lds_bytes() const { return (size_t)r3_doubles() * 8; }
    // the Ruiz passes' temporaries (pc | En | Dn) sit behind the staged P blocks inside Sinv | Wk, which are not live yet
    DEKF_HD bool r3_fits(int L, size_t static_lds = 512) const {
        const int ps = K * (6 * L + 27) + ns * (ns + 1) / 2;
        // LDS is granted in coarse units: measured on gfx950, 53 464 B per workgroup (this layout + 480 B static) run three per
        // CU, 54 112 B run two although the occupancy query still answers three — consistent with a 1 280 or 1 536 B granule
        const size_t granule = 1536, need = (r3_lds_bytes() + static_lds + granule - 1) / granule * granule;
        return ps + 2 * n_pad + m_pad <= 2 * K * ns * ns && DEKF_R3_WAVES * need <= 160 * 1024;
    }
    // Four-workgroup placement (R4, round 6; mhe_admm_core.h: admm_chunk_r4): workgroups of THREE wavefronts (the solve + two
    // workers) at four per CU = 12 wavefronts per CU = the 168 VGPRs of the three-workgroup kernels.  LDS holds only what an
    // ITERATION shares between lanes; what the rare phases read (D, E, the factor-time product, the stash of y and of the VO rows'
    // z between chunks) lives in the workgroup's slab:
    //   R | xb | xd | at (= the stash of the slack x between chunks) | xs | gb | pad (-> 4928 doubles) | tmp | Sinv | Wk
    // The compact copy of the x blocks' scaling that the solve wavefront reads in every step (Dxb, [K][ns]) ends at tmp + 96: the
    // pad and the first Gauss-Jordan scratch, free between factorisations (rebuilt after each).  During the Ruiz passes D | E sit in
    // xb .. pad (dead until the cold start) and go to the slab afterwards.
    static constexpr int R4_DOUBLES = 4928;  // (160 KiB / 4 in 1536 B granules, less 512 B of static LDS)
    static constexpr int R4_DXB_IN_TMP = 96;
    DEKF_HD int r4_iter_doubles() const { return 9 * K + ns * K + (2 * ns * K + m_pad + 3 * K) + solve_tmp_len(ns) + 2 * K * ns * ns; }
    DEKF_HD int r4_pad() const { return R4_DOUBLES - r4_iter_doubles(); }
    DEKF_HD size_t r4_lds_bytes() const { return (size_t)R4_DOUBLES * 8; }
    DEKF_HD bool r4_fits(int L, size_t static_lds = 512) const {
        const int ps = K * (6 * L + 27) + ns * (ns + 1) / 2;
        const size_t granule = 1536, need = (r4_lds_bytes() + static_lds + granule - 1) / granule * granule;
        return ns == 9 && r4_pad() >= 0 && r4_pad() + R4_DXB_IN_TMP >= ns * K &&                  // Dxb
               n_pad + m_pad <= ns * K + 2 * ns * K + m_pad + 3 * K + r4_pad() &&                   // D | E during the Ruiz passes
               ps + 2 * n_pad + m_pad <= 2 * K * ns * ns && 4 * need <= 160 * 1024;
    }
    // Rows in registers at a run-time horizon with the factor in the slab (RR, mhe_admm_core.h: admm_chunk_rr):
    //   R | D | E | xb | xd | at (= the stash of the slack x between chunks) | xs | gb | tmp
    // The stash of y and of the VO rows' z between chunks, the scaled bounds, the slack-block inverses and the factor live in the slab;
    // the Ruiz passes' temporaries Dn | En overlay xb .. gb (dead during the scaling), the P column norms go to the slab.
    DEKF_HD int rr_doubles() const { return 9 * K + n_pad + m_pad + ns * K + (2 * ns * K + m_pad + 3 * K) + solve_tmp_len(ns); }
    DEKF_HD size_t rr_lds_bytes() const { return (size_t)rr_doubles() * 8; }
    DEKF_HD bool rr_fits(int L, size_t static_lds = 512) const {
        const size_t granule = 1536, need = (rr_lds_bytes() + static_lds + granule - 1) / granule * granule;
        // row tiles (admm_chunk_rr): the first 64 Meas blocks | up to four tiles of Dyn lane pairs | up to four of VO + bias blocks
        // + the remaining Meas blocks
        const int over = K * L > 64 ? K * L - 64 : 0;
        return ns == 9 && K >= 4 && (K & 1) == 0 && 2 * (K - 1) <= 256 && 2 * (K - 1) + over <= 256 &&
               n_pad + m_pad <= 3 * ns * K + m_pad + 3 * K && 2 * need <= 160 * 1024;
    }
    // Factor in the HBM slab (one workgroup per CU anyway): whatever LDS the iterates leave free takes the per-row constants
    // that every phase reads (D, E, scaled bounds, R) — they are what made a PogoX iteration fetch 53 KB more from beyond L2.
    DEKF_HD int gg_consts() const { return n_pad + 2 * m_pad + 3 * K + 9 * K; }
    DEKF_HD static int wgs_per_cu(size_t bytes) { const size_t g = 1536, need = (bytes + 512 + g - 1) / g * g; return (int)((160 * 1024) / need); }
    DEKF_HD bool gg_consts_in_lds() const {  // only where it costs no resident workgroup
        if (factor_in_lds()) return false;
        const int with = wgs_per_cu((size_t)(vec + gg_consts()) * 8);
        return with >= 1 && with == (wgs_per_cu((size_t)vec * 8) < 2 ? wgs_per_cu((size_t)vec * 8) : 2);
    }
    DEKF_HD size_t lds_bytes() const { return (size_t)(factor_in_lds() ? vec + resident : (gg_consts_in_lds() ? vec + gg_consts() : vec)) * 8; }
};

// variable / row indices with the leg count known at compile time: every / and % by the
// per-step block sizes becomes a multiply-shift instead of a ~40-instruction software divide
template <int L, int FT = 0>
struct IdxT {
    static constexpr int nm = 3 * L, NS = 9 + 3 * L * FT, SV = 2 * NS + 3 + 3 * L, SC = NS + 3 + 3 * L;
    DEKF_FN static int x(int k, int j) { return k * SV + j; }
    DEKF_FN static int v(int k, int r) { return k * SV + NS + r; }
    DEKF_FN static int w(int k, int r) { return k * SV + NS + nm + r; }
    DEKF_FN static int c(int k, int a) { return k * SV + 2 * NS + nm + a; }
    // Rows are stored KIND-MAJOR in LDS: [Meas rows of all steps | Dyn rows of all steps | VO rows of all
    // steps], not step-major as in the reference's QP (k * SC + ...).  A wavefront works on one kind of
    // row with one lane per step, so consecutive lanes are 3 / 9 / nm doubles apart (conflict-free);
    // step-major they were SC = 24 doubles = 48 banks apart, which mod 32 banks leaves two distinct
    // banks for 16 lanes: every ds_read2_b64 of the row phase ran 8-way conflicted.
    int rdb, rvb;  // first Dyn row, first VO row
    DEKF_FN explicit IdxT(int K) : rdb(K * nm), rvb(K * nm + NS * (K - 1)) {}
    DEKF_FN int rm(int k, int r) const { return k * nm + r; }
    DEKF_FN int rd(int k, int r) const { return rdb + NS * k + r; }
    DEKF_FN int rv(int k, int a) const { return rvb + 3 * k + a; }
};

// a * b rounded on its own: the compiler must not contract it into a following addition (the kernels that keep scaled bounds in an
// array and the ones that rebuild them from the window record have to see the same bits)
DEKF_FN double mul_rounded(double a, double b) {
    double p = a * b;
#if DEKF_DEVICE_BUILD
    asm volatile("" : "+v"(p));
#endif
    return p;
}

template <int L, int NF = 0, bool FLDS = true, int FT = 0, bool R3_ = false, bool POLISH_ = false, bool R4_ = false>
struct SolveCtx {
    static constexpr bool R4 = R4_;  // four workgroups of three wavefronts per CU (round 6): R3's split with D, E and the factor-time product in the slab
    static constexpr int WAVES = R4_ ? 3 : 4;  // wavefronts of the workgroup
    static constexpr bool POLISH = POLISH_;  // the kernel carries OSQP's polishing step (its own instantiations, k_mhe_solve_*_pol:
                                             // the code behind it costs the others 87 spilled VGPRs when it is merely compiled in)
    static constexpr bool R3 = R3_;  // three workgroups per CU: row state in registers, x blocks compact in LDS (xb)
    static constexpr int LEGS = L;
    static constexpr int NFIXED = NF;  // != 0: horizon known at compile time (sweeps fully unrolled when K == NF)
    static constexpr bool FACTOR_LDS = FLDS;  // false: W_k streams from the HBM slab (deeper operand prefetch in the sweeps)
    static constexpr int FOOT = FT;           // leg_odom_type 1: the foot positions are states (NS = 9 + 3 L)
    static constexpr int NS = 9 + 3 * L * FT, NS2 = NS * NS;
    const DevCfg& c;
    const DevState& s;
    int b, K, kstart, n, m;
    IdxT<L, FT> ix;
    // LDS always
    dptr x, z, y, xt, zt, at, xs, xd, tmp;
    dptr xb;       // R3: the x blocks of x, [K][NS], the only part of x an iteration shares between lanes
    dptr Dxb;      // R4: the scaling of the x blocks, compact [K][NS] in LDS (D itself is in the slab); rebuilt after every factorisation
    dptr sx, sy, sz;  // R3: where the row state rests between chunks, by row: slack x (= at), y, z of the VO rows (from the first VO row)
    bool cold;        // R3: no chunk has run yet (x = z = y = 0)
    dptr cf;  // per row rho E D (aliases xt, which only the factorisation uses otherwise)
    dptr gb;  // per step R' (dt^2/2 w_p + dt w_v): the Dyn rows' contribution to the bias columns (tail of xt)
    // LDS or HBM scratch
    dptr D, E, lo, hi, Sv, Sw, Sc, Sinv, Wk, R;
    dptr Sf;  // FOOT: inverses of the slack blocks of the foot-position Dyn rows, [K][L][6]
    // factor-time temporaries
    dptr Wm, Wd, Wc, PA;
    dptr WT;   // 9-state kernels with the factor in the slab: the W blocks once more, transposed (what the outward legs read); else null
    dptr Wf;  // FOOT: effective row weights of the foot-position Dyn rows
    cdptr Mp, np;
    cdptr vo;  // [wcap][4] VO flag + bound per ring slot, from the solve's input snapshot
    double cc;   // cost scaling c
    double rho;  // current scalar rho
    int lane0;   // this lane's index inside its wavefront, read ONCE per solve: the solve phases that run on the critical wavefront take
                 // it from here, so that their per-lane pointers and strides are loop-invariant across the iterations (DEKF_LANE() is
                 // opaque at every use, on purpose, everywhere else)
    double sigma_pol;  // POLISH: the regularisation of the x block while polishing (delta)
    bool zlo;          // POLISH, while polishing: the cold start puts z on the bound of every equality row (x = y = 0 as usual)
    // the regularisation of the x block in the linear system: settings sigma; delta while polishing
    DEKF_FN double sigma() const { if constexpr (POLISH) return zlo ? sigma_pol : c.sigma; else return c.sigma; }
    DEKF_FN bool polishing() const { if constexpr (POLISH) return zlo; else return false; }
    dptr pol;  // POLISH: slab scratch of the refinement steps (Gws::pol): saved point | saved bounds | KKT residual vectors
    dptr Pst;  // unscaled P blocks staged by solve_scale (aliases Sinv | Wk until the first factorisation)
    bool staged;  // Pst valid
    dptr prof;         // diagnostic build only
    long long prof_last;  // diagnostic build only

    DEKF_FN cdptr rec(int k) const {
        return DEKF_CSPAN(s.rec + ((size_t)b * c.wcap + ((kstart + k) % c.wcap)) * c.rec, c.rec);
    }
    DEKF_FN double adyn(int k, int r, int j) const { return adyn_entry(R + 9 * k, c.dt, r, j); }
    // per-row rho from the scaled bounds (OSQP set_rho_vec / osqp_update_rho)
    DEKF_FN double rho_of(double lb, double ub) const {
        if (lb < -c.inf_thr && ub > c.inf_thr) return RHO_MIN;
        return (ub - lb < RHO_TOL) ? RHO_EQ_OVER_RHO_INEQ * rho : rho;
    }
    // Meas and Dyn rows are equalities by construction (l == u): only the VO rows keep an upper bound
    // (hi is indexed from the first VO row; the full-length vector was 3.3 KiB of LDS for nothing)
    DEKF_FN double rho_at(int r) const { return r < ix.rvb ? RHO_EQ_OVER_RHO_INEQ * rho : rho_of(lo[r], hi[r - ix.rvb]); }
    // unscaled bound of row (k, kind, o): kind 0 Meas, 1 Dyn, 2 VO
    DEKF_FN void bounds(int k, int kind, int o, double& lb, double& ub) const { bounds_at((kstart + k) % c.wcap, kind, o, lb, ub); }
    // the same for a ring slot computed once (the modulo by a run-time ring size is a ~40-instruction software division)
    DEKF_FN void bounds_at(int slot, int kind, int o, double& lb, double& ub) const {
        cdptr r = DEKF_CSPAN(s.rec + ((size_t)b * c.wcap + slot) * c.rec, c.rec);
        if (kind == 0) lb = ub = ld_stream(r, Rec::BM + o);
        else if (kind == 1) lb = ub = (o < 3 ? -c.hdt2 * ld_stream(r, Rec::AS + o) : (o < 6 ? -c.dt * ld_stream(r, Rec::AS + o - 3) : 0.0));
        else {  // VO flag and bound come from the step's snapshot, not from the record (update(T + 1) may be rewriting them)
            cdptr v = vo + 4 * slot;
            if (v[0] != 0.0) lb = ub = v[1 + o];
            else { lb = -OSQP_INFTY; ub = OSQP_INFTY; }
        }
    }
    DEKF_FN void dec_row(int r, int& k, int& kind, int& o) const {
        if (r < ix.rdb) { kind = 0; k = r / ix.nm; o = r - k * ix.nm; }
        else if (r < ix.rvb) { kind = 1; int t = r - ix.rdb; k = t / NS; o = t - NS * k; }
        else { kind = 2; int t = r - ix.rvb; k = t / 3; o = t - 3 * k; }
    }
    // slack variable of row (k, kind, o)
    DEKF_FN int row_slack(int k, int kind, int o) const { return kind == 0 ? ix.v(k, o) : (kind == 1 ? ix.w(k, o) : ix.c(k, o)); }

};

DEKF_FN double limit_scaling(double v) {
    v = v < MIN_SCALING ? 1.0 : v;
    v = v > MAX_SCALING ? MAX_SCALING : v;
    return v;
}

// Ruiz equilibration + cost scaling (OSQP scale_data), on the structured QP
//
// The unscaled P blocks (Q_meas, Q_dyn, Q_cam per step, M on x_0) are staged once into `q.Pst`
// (it aliases the S^-1 / W arrays, which are dead until the factorisation), the P column norms are
// cached in `q.x` (dead until the cold start) and every norm loop is kind-homogeneous, so the ten
// passes touch LDS only.  Pst stays valid for step 3a of the FIRST factorisation.
// stage the unscaled P blocks from the HBM window records into q.Pst:
// per step [Qm 6L | Qd 21 | Qc 6], then M (upper triangle, packed 45)
template <class Q>
DEKF_FN void stage_p(Q& q) {
    constexpr int L = Q::LEGS, NM = 3 * L, FT = Q::FOOT, NS = Q::NS, PS = 6 * L + 27 + 6 * L * FT, MP = NS * (NS + 1) / 2;
    const int K = q.K;
    dptr Pst = q.Pst;
    wfor(K * PS + MP, [&](int e) {
        if (e < K * PS) {
            int k = e / PS, o = e - k * PS;
            cdptr r = q.rec(k);
            // INVARIANT: Qd | Qc of the NEWEST record (k = K - 1) are not read.  They are the gains of a step whose successor does not
            // exist yet — no consumer uses them (every one guards k < K - 1) — and k_mhe_marginalize_early of the NEXT step writes them
            // on another stream while this solve runs (mhe_assemble_core.h: gains_of_previous_step).  Zeros in their place.
            if (k == K - 1 && o >= 6 * L && o < 6 * L + 27) { Pst[e] = 0.0; return; }
            // [Qm 6L | Qd 21 | Qc 6 | Qf 6L (foot states)]: the last block is contiguous with Qm in the record
            Pst[e] = ld_stream(r, o < 6 * L ? Rec::qm(NM) + o
                                             : (o < 6 * L + 21 ? Rec::QD + o - 6 * L : (o < 6 * L + 27 ? Rec::QC + o - 6 * L - 21 : Rec::qf(NM) + o - 6 * L - 27)));
        } else {
            int p = e - K * PS, i = 0;
            while (p >= NS - i) { p -= NS - i; ++i; }
            Pst[e] = ld_stream(q.Mp, NS * i + i + p);
        }
    });
    q.staged = true;
}

template <class Q>
DEKF_FN void solve_scale(Q& q) {
    constexpr int L = Q::LEGS, NM = 3 * L, FT = Q::FOOT, NS = Q::NS, SV = 2 * NS + 3 + NM, PS = 6 * L + 27 + 6 * L * FT;
    const int n = q.n, m = q.m, K = q.K, K1 = q.K - 1, nmeas = K * L;
    const double dt = q.c.dt, hdt2 = q.c.hdt2;
    dptr D = q.D, E = q.E, Pst = q.Pst, pc = q.x, Dn = q.xt, En = q.zt;  // (R3: the caller points all of these at LDS)
    cdptr g = q.np;
    const auto& ix = q.ix;
    stage_p(q);
    wfor(n + m, [&](int e) { if (e < n) { D[e] = 1.0; Dn[e] = 1.0; } else { E[e - n] = 1.0; En[e - n] = 1.0; } });
    q.cc = 1.0;
    cdptr Mst = Pst + K * PS;
    // Lane ownership is the row phase's: a lane owns one 3-row block and its 3 slack variables (Meas leg
    // block, Dyn position or velocity rows as a lane pair, Dyn bias, VO, [foot-position Dyn block]), or one x
    // entry (tiles by column kind).  One Ruiz pass is two tile phases:
    //   equil  new E of the owned rows, new D of the owned variables from the OLD D, E -> En, Dn
    //   adopt  D = Dn, E = En on the owned entries; inf-norms of the owned columns of D P D (read from Dn, so
    //          no lane reads an entry another lane is overwriting) -> pc; their sum for the cost scaling
    // i.e. 3 workgroup barriers per pass (one after equil, two in the sum) where the item-per-lane sweeps
    // with per-item kind decoding needed 6 and about 17 k cycles.
    const int ntm = (nmeas + 63) >> 6, ntp = (2 * K1 + 63) >> 6, ntd = (K1 + 63) >> 6, ntx = (3 * K + 63) >> 6;
    const int ntf = FT ? (K1 * L + 63) >> 6 : 0, ntxf = FT ? (3 * L * K + 63) >> 6 : 0;  // foot Dyn blocks, foot columns
    const int ntiles = ntm + ntp + 2 * ntd + 3 * ntx + ntf + ntxf;
    // decode a tile/lane into an owned block: kind 0 Meas, 1 Dyn p/v, 2 bias, 3 VO, 4..6 x columns,
    // 7 foot-position Dyn block (k, leg = sub), 8 foot-position column (k, sub = 3 leg + a)
    auto decode = [&](int tile, int lane, int& kind, int& k, int& sub) -> bool {
        if (tile < ntm) { int e = tile * 64 + lane; kind = 0; k = e / L; sub = e - k * L; return e < nmeas; }
        int td = tile - ntm;
        if (td < ntp) { int pl = td * 64 + lane; kind = 1; k = pl >> 1; sub = pl & 1; return k < K1; }
        td -= ntp;
        if (td < 2 * ntd) { bool vo = td >= ntd; kind = vo ? 3 : 2; k = (td - (vo ? ntd : 0)) * 64 + lane; sub = 0; return k < K1; }
        td -= 2 * ntd;
        if (td < 3 * ntx) {
            // column tiles in the order velocity, position, bias: with one tile per kind (K <= 21) the wavefront that owns the Dyn
            // lane pairs — the longest row tile — then also gets the position columns, the shortest column tile
            int pos = td < ntx ? 0 : (td < 2 * ntx ? 1 : 2), ck = pos == 0 ? 1 : (pos == 1 ? 0 : 2);
            int e = (td - pos * ntx) * 64 + lane;
            kind = 4 + ck; k = e / 3; sub = e - 3 * k;
            return e < 3 * K;
        }
        td -= 3 * ntx;
        if (td < ntf) { int e = td * 64 + lane; kind = 7; k = e / L; sub = e - k * L; return e < K1 * L; }
        td -= ntf;
        int e = td * 64 + lane;
        kind = 8; k = e / NM; sub = e - k * NM;
        return e < NM * K;
    };
    auto row_base = [&](int kind, int k, int sub, int& r0, int& sv0) {
        if (kind == 0) { r0 = ix.rm(k, 3 * sub); sv0 = ix.v(k, 3 * sub); }
        else if (kind == 1) { r0 = ix.rd(k, 3 * sub); sv0 = ix.w(k, 3 * sub); }
        else if (kind == 2) { r0 = ix.rd(k, 6); sv0 = ix.w(k, 6); }
        else if (kind == 3) { r0 = ix.rv(k, 0); sv0 = ix.c(k, 0); }
        else { r0 = ix.rd(k, 9 + 3 * sub); sv0 = ix.w(k, 9 + 3 * sub); }
    };
    // Both lambdas read everything they need BEFORE their first store (a store through one pointer pins
    // every later load behind it: interleaved, a block cost one LDS round trip per row).
    // inf-norms of the owned columns of D P D (cc excluded) from the new scaling Dn; returns their sum
    auto adopt = [&](int tile, int lane) -> double {
        int kind, k, sub;
        if (!decode(tile, lane, kind, k, sub)) return 0.0;
        if ((kind >= 4 && kind <= 6) || kind == 8) {
            const int j = kind == 8 ? 9 + sub : 3 * (kind - 4) + sub, i = k * SV + j;
            const double dj = Dn[i];
            double v = 0.0;
            if (k == 0) {
                if constexpr (FT) {
                    for (int t = 0; t < NS; ++t) v = dmax(v, fabs(dj * symget(Mst, j, t, NS) * Dn[t]));
                } else {
                    double mj[9], dt9[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) { mj[t] = symget(Mst, j, t, 9); dt9[t] = Dn[t]; }
#pragma unroll
                    for (int t = 0; t < 9; ++t) v = dmax(v, fabs(dj * mj[t] * dt9[t]));
                }
            }
            D[i] = dj;
            pc[i] = v;
            return v;
        }
        int r0, sv0;
        row_base(kind, k, sub, r0, sv0);
        double en[3], dn[3], v[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { en[a] = En[r0 + a]; dn[a] = Dn[sv0 + a]; }
        if (kind == 1) {
            cdptr q21 = Pst + k * PS + 6 * L;
            cdptr d = Dn + ix.w(k, 0);
            double d6[6], p[3][6];
#pragma unroll
            for (int t = 0; t < 6; ++t) d6[t] = d[t];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const double p0 = q21[a < t ? symidx(a, t, 6) : symidx(t, a, 6)];
                    const double p1 = q21[3 + a < t ? symidx(3 + a, t, 6) : symidx(t, 3 + a, 6)];
                    p[a][t] = sub ? p1 : p0;
                }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double m = 0.0;
#pragma unroll
                for (int t = 0; t < 6; ++t) m = dmax(m, fabs(p[a][t] * d6[t]));
                v[a] = m * dn[a];
            }
        } else if (kind == 2) {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = dn[a] * q.c.Q_bias_dt2[a] * dn[a];
        } else {
            cdptr q6 = Pst + k * PS + (kind == 0 ? 6 * sub : (kind == 3 ? 6 * L + 21 : 6 * L + 27 + 6 * sub));
            double p6[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) p6[t] = q6[t];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                v[a] = dmax(fabs(p6[symidx(0, a, 3)] * dn[0]), dmax(fabs(p6[1 < a ? symidx(1, a, 3) : symidx(a, 1, 3)] * dn[1]), fabs(p6[symidx(a, 2, 3)] * dn[2]))) * dn[a];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { E[r0 + a] = en[a]; D[sv0 + a] = dn[a]; pc[sv0 + a] = v[a]; }
        return v[0] + v[1] + v[2];
    };
    auto equil = [&](int tile, int lane, double cc) {
        int kind, k, sub;
        if (!decode(tile, lane, kind, k, sub)) return;
        if ((kind >= 4 && kind <= 6) || kind == 8) {  // x column: inf-norm over the rows that touch it
            const int a = kind == 8 ? sub % 3 : sub, i = k * SV + (kind == 8 ? 9 + sub : 3 * (kind - 4) + a);
            const bool hn = k < K1, hp = k > 0;
            const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;
            const double di = D[i], pci = pc[i];
            double an = 0.0;
            if (kind == 4) {
                const double n0 = E[ix.rd(kn, a)], n1 = E[ix.rv(kn, a)], p0 = E[ix.rd(kp, a)], p1 = E[ix.rv(kp, a)];
                if (hn) an = dmax(n0, n1);
                if (hp) an = dmax(an, dmax(p0, p1));
                if constexpr (FT) {  // A_meas = [-I 0 0 .. I ..]: every leg's Meas rows touch the position
#pragma unroll
                    for (int leg = 0; leg < L; ++leg) an = dmax(an, E[ix.rm(k, 3 * leg + a)]);
                }
            } else if (kind == 5) {
                const double n0 = E[ix.rd(kn, 3 + a)], n1 = E[ix.rd(kn, a)], p0 = E[ix.rd(kp, 3 + a)];
                if constexpr (!FT) {
#pragma unroll
                    for (int leg = 0; leg < L; ++leg) an = dmax(an, E[ix.rm(k, 3 * leg + a)]);
                }
                if (hn) an = dmax(an, dmax(n0, dt * n1));
                if (hp) an = dmax(an, p0);
            } else if (kind == 6) {
                cdptr R = q.R + 9 * kn;
                double bn = E[ix.rd(kn, 6 + a)];
                const double p0 = E[ix.rd(kp, 6 + a)];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double ra = fabs(R[3 * r + a]);
                    bn = dmax(bn, dmax(hdt2 * ra * E[ix.rd(kn, r)], dt * ra * E[ix.rd(kn, 3 + r)]));
                }
                if (hn) an = bn;
                if (hp) an = dmax(an, p0);
            } else {  // foot-position column: its leg's Meas row, the Dyn rows of this and the previous step
                const double m0 = E[ix.rm(k, sub)], n0 = E[ix.rd(kn, 9 + sub)], p0 = E[ix.rd(kp, 9 + sub)];
                an = m0;
                if (hn) an = dmax(an, n0);
                if (hp) an = dmax(an, p0);
            }
            Dn[i] = di * rsqrt_fast(limit_scaling(dmax(cc * pci, an * di)));
            return;
        }
        int r0, sv0;
        row_base(kind, k, sub, r0, sv0);
        cdptr d = D + k * SV;
        double e3[3], d3[3], p3[3], v[3];  // own rows / slacks; v = inf-norm of the row of A D (before E)
#pragma unroll
        for (int a = 0; a < 3; ++a) { e3[a] = E[r0 + a]; d3[a] = D[sv0 + a]; p3[a] = pc[sv0 + a]; }
        if (kind == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = FT ? dmax(d3[a], dmax(d[a], d[9 + 3 * sub + a])) : dmax(d3[a], d[3 + a]);
        } else if (kind == 1) {
            cdptr R = q.R + 9 * k;
            double dk[9], dnx[6], Ra[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) { dk[t] = d[t]; Ra[t] = fabs(R[t]); }
#pragma unroll
            for (int t = 0; t < 6; ++t) dnx[t] = d[SV + t];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double vp = dmax(dmax(d3[a], dk[a]), dmax(dt * dk[3 + a], dnx[a]));
                double vv = dmax(dmax(d3[a], dk[3 + a]), dnx[3 + a]);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    vp = dmax(vp, hdt2 * Ra[3 * a + j] * dk[6 + j]);
                    vv = dmax(vv, dt * Ra[3 * a + j] * dk[6 + j]);
                }
                v[a] = sub ? vv : vp;
            }
        } else if (kind == 2) {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = dmax(d3[a], dmax(d[6 + a], d[SV + 6 + a]));
        } else if (kind == 3) {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = dmax(d3[a], dmax(d[a], d[SV + a]));
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = dmax(d3[a], dmax(d[9 + 3 * sub + a], d[SV + 9 + 3 * sub + a]));
        }
        double eo[3], dout[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            eo[a] = e3[a] * rsqrt_fast(limit_scaling(e3[a] * v[a]));
            dout[a] = d3[a] * rsqrt_fast(limit_scaling(dmax(cc * p3[a], e3[a] * d3[a])));  // slack column: one entry, in its row
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { En[r0 + a] = eo[a]; Dn[sv0 + a] = dout[a]; }
    };
#if DEKF_DEVICE_BUILD
    auto fused = [&](int tile, int lane, double cc, cdptr Dr, cdptr Er, dptr Dw, dptr Ew) -> double {
        int kind, k, sub;
        if (!decode(tile, lane, kind, k, sub)) return 0.0;
        if ((kind >= 4 && kind <= 6) || kind == 8) {  // x column: inf-norm over the rows that touch it
            const int a = kind == 8 ? sub % 3 : sub, i = k * SV + (kind == 8 ? 9 + sub : 3 * (kind - 4) + a);
            const bool hn = k < K1, hp = k > 0;
            const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;
            const double di = Dr[i], pci = pc[i];
            double an = 0.0;
            if (kind == 4) {
                const double n0 = Er[ix.rd(kn, a)], n1 = Er[ix.rv(kn, a)], p0 = Er[ix.rd(kp, a)], p1 = Er[ix.rv(kp, a)];
                if (hn) an = dmax(n0, n1);
                if (hp) an = dmax(an, dmax(p0, p1));
                if constexpr (FT) {  // A_meas = [-I 0 0 .. I ..]: every leg's Meas rows touch the position
#pragma unroll
                    for (int leg = 0; leg < L; ++leg) an = dmax(an, Er[ix.rm(k, 3 * leg + a)]);
                }
            } else if (kind == 5) {
                const double n0 = Er[ix.rd(kn, 3 + a)], n1 = Er[ix.rd(kn, a)], p0 = Er[ix.rd(kp, 3 + a)];
                if constexpr (!FT) {
#pragma unroll
                    for (int leg = 0; leg < L; ++leg) an = dmax(an, Er[ix.rm(k, 3 * leg + a)]);
                }
                if (hn) an = dmax(an, dmax(n0, dt * n1));
                if (hp) an = dmax(an, p0);
            } else if (kind == 6) {
                cdptr R = q.R + 9 * kn;
                double bn = Er[ix.rd(kn, 6 + a)];
                const double p0 = Er[ix.rd(kp, 6 + a)];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double ra = fabs(R[3 * r + a]);
                    bn = dmax(bn, dmax(hdt2 * ra * Er[ix.rd(kn, r)], dt * ra * Er[ix.rd(kn, 3 + r)]));
                }
                if (hn) an = bn;
                if (hp) an = dmax(an, p0);
            } else {  // foot-position column: its leg's Meas row, the Dyn rows of this and the previous step
                const double m0 = Er[ix.rm(k, sub)], n0 = Er[ix.rd(kn, 9 + sub)], p0 = Er[ix.rd(kp, 9 + sub)];
                an = m0;
                if (hn) an = dmax(an, n0);
                if (hp) an = dmax(an, p0);
            }
            Dw[i] = di * rsqrt_fast(limit_scaling(dmax(cc * pci, an * di)));
            return 0.0;  // pc of an x column: 0 for k > 0 (stays 0), the arrival-cost block of x_0 after the barrier (fused_x0)
        }
        int r0, sv0;
        row_base(kind, k, sub, r0, sv0);
        cdptr d = Dr + k * SV;
        double e3[3], d3[3], p3[3], v[3];  // own rows / slacks; v = inf-norm of the row of A D (before E)
#pragma unroll
        for (int a = 0; a < 3; ++a) { e3[a] = Er[r0 + a]; d3[a] = Dr[sv0 + a]; p3[a] = pc[sv0 + a]; }
        if (kind == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = FT ? dmax(d3[a], dmax(d[a], d[9 + 3 * sub + a])) : dmax(d3[a], d[3 + a]);
        } else if (kind == 1) {
            cdptr R = q.R + 9 * k;
            double dk[9], dnx[6], Ra[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) { dk[t] = d[t]; Ra[t] = fabs(R[t]); }
#pragma unroll
            for (int t = 0; t < 6; ++t) dnx[t] = d[SV + t];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double vp = dmax(dmax(d3[a], dk[a]), dmax(dt * dk[3 + a], dnx[a]));
                double vv = dmax(dmax(d3[a], dk[3 + a]), dnx[3 + a]);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    vp = dmax(vp, hdt2 * Ra[3 * a + j] * dk[6 + j]);
                    vv = dmax(vv, dt * Ra[3 * a + j] * dk[6 + j]);
                }
                v[a] = sub ? vv : vp;
            }
        } else if (kind == 2) {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = dmax(d3[a], dmax(d[6 + a], d[SV + 6 + a]));
        } else if (kind == 3) {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = dmax(d3[a], dmax(d[a], d[SV + a]));
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = dmax(d3[a], dmax(d[9 + 3 * sub + a], d[SV + 9 + 3 * sub + a]));
        }
        double eo[3], dout[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            eo[a] = e3[a] * rsqrt_fast(limit_scaling(e3[a] * v[a]));
            dout[a] = d3[a] * rsqrt_fast(limit_scaling(dmax(cc * p3[a], e3[a] * d3[a])));  // slack column: one entry, in its row
        }
        // ---- adopt, fused: inf-norms of the owned slack columns of D P D from the scaling just computed (the partner half of
        //      a Dyn lane pair comes over by DPP); every load above and below is issued before the first store
        double vv3[3];
        if (kind == 1) {
            cdptr q21 = Pst + k * PS + 6 * L;
            double d6[6], pq[3][6];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const double other = pair_swap(dout[t]);
                d6[t] = sub ? other : dout[t];
                d6[3 + t] = sub ? dout[t] : other;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    // ONE load per entry: the lane's half (position / velocity rows) selects the INDEX (a 32-bit select of two
                    // constants), not one of two loaded values — 18 LDS reads instead of 36 in the longest tile of a pass
                    const int i0 = a < t ? symidx(a, t, 6) : symidx(t, a, 6);
                    const int i1 = 3 + a < t ? symidx(3 + a, t, 6) : symidx(t, 3 + a, 6);
                    pq[a][t] = q21[sub ? i1 : i0];
                }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double m = 0.0;
#pragma unroll
                for (int t = 0; t < 6; ++t) m = dmax(m, fabs(pq[a][t] * d6[t]));
                vv3[a] = m * dout[a];
            }
        } else if (kind == 2) {
#pragma unroll
            for (int a = 0; a < 3; ++a) vv3[a] = dout[a] * q.c.Q_bias_dt2[a] * dout[a];
        } else {
            cdptr q6 = Pst + k * PS + (kind == 0 ? 6 * sub : (kind == 3 ? 6 * L + 21 : 6 * L + 27 + 6 * sub));
            double p6[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) p6[t] = q6[t];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                vv3[a] = dmax(fabs(p6[symidx(0, a, 3)] * dout[0]), dmax(fabs(p6[1 < a ? symidx(1, a, 3) : symidx(a, 1, 3)] * dout[1]), fabs(p6[symidx(a, 2, 3)] * dout[2]))) * dout[a];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { Ew[r0 + a] = eo[a]; Dw[sv0 + a] = dout[a]; pc[sv0 + a] = vv3[a]; }
        return vv3[0] + vv3[1] + vv3[2];
    };
    // the arrival-cost block (x_0 columns): needs the new scaling of all of x_0, i.e. the barrier behind `fused`
    auto fused_x0 = [&](int tile, int lane, cdptr Dw) -> double {
        int kind, k, sub;
        if (tile < ntm + ntp + 2 * ntd || tile >= ntm + ntp + 2 * ntd + 3 * ntx) return 0.0;
        if (!decode(tile, lane, kind, k, sub) || k != 0) return 0.0;
        const int j = 3 * (kind - 4) + sub;
        const double dj = Dw[j];
        double mj[9], dt9[9], v = 0.0;
#pragma unroll
        for (int t = 0; t < 9; ++t) { mj[t] = symget(Mst, j, t, 9); dt9[t] = Dw[t]; }
#pragma unroll
        for (int t = 0; t < 9; ++t) v = dmax(v, fabs(dj * mj[t] * dt9[t]));
        pc[j] = v;
        return v;
    };
#endif
    wtiles(ntiles, [&](int tile, int lane) { (void)adopt(tile, lane); });  // column norms of P for D = 1
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 16);
    double gq[NS];  // the linear cost sits in HBM: read it once, not once per pass (a global round trip each)
#pragma unroll
    for (int j = 0; j < NS; ++j) gq[j] = g[j];
#if DEKF_DEVICE_BUILD
    {
        // One Ruiz pass = ONE tile phase (equilibrate + the new column norms of the owned slack blocks, ping-pong between the two
        // pairs of scaling vectors instead of a copy), a barrier, the nine x_0 lanes' arrival-cost norms, the sum.  Same
        // arithmetic, same lanes, same order of the sum as the two-phase form below (which the lane-sequential build and the
        // foot-state shapes keep): bit-identical D, E, c.
        dptr Dr = D, Er = E, Dw = Dn, Ew = En;
        // (lane and wavefront are read ONCE in front of the passes: what a tile's lane derives from them — its block, the indices,
        // the constants it loads — is then loop-invariant for the compiler)
        const int rz_lane = DEKF_LANE() & 63, rz_wave = __builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6), rz_nw = DEKF_NLANES() > WAVE ? DEKF_NLANES() >> 6 : 1;
        (void)rz_lane; (void)rz_wave; (void)rz_nw;
        for (int it = 0; it < q.c.scaling; ++it) {
            const double cc = q.cc;
            double psum = 0.0;
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_RUIZ)  // per-wavefront: tile phase | wait at its barrier | x_0 norms + sum (slots 24.., 28.., 20..)
            const long long tr0 = clock64();
#endif
            // Three-wavefront workgroups (R4): the sum of the column norms keeps the association of the four-wavefront kernels — tile t
            // belongs to "virtual wavefront" t & 3, whose lanes add up their tiles and are then summed by DPP — so that the cost scaling
            // c, and with it every later bit, does not depend on the workgroup's shape.  A physical wavefront hosts whole virtual ones
            // (RZ_PW: the virtual wavefront with the Dyn lane-pair tile, the longest, alone); the x-column tiles contribute exactly 0.0
            // to the sum and go wherever the load is lightest.
            constexpr bool RZ3 = Q::WAVES == 3;
            constexpr int RZ_PW[4] = {0, Q::LEGS == 4 ? 1 : 2, Q::LEGS == 4 ? 2 : 1, Q::LEGS == 4 ? 1 : 0};
            const int rz_nrow = ntm + ntp + 2 * ntd + ntf;
            auto rz_phys = [&](int t) { return t < rz_nrow ? RZ_PW[t & 3] : 2 - (t - rz_nrow) % 3; };
            double pv[4] = {0.0, 0.0, 0.0, 0.0};
            if constexpr (RZ3) {
                for (int t = 0; t < ntiles; ++t) {
                    if (rz_phys(t) != rz_wave) continue;  // (scalar)
                    const double f = fused(t, rz_lane, cc, Dr, Er, Dw, Ew);
                    const int v = t & 3;
                    pv[0] += v == 0 ? f : 0.0; pv[1] += v == 1 ? f : 0.0; pv[2] += v == 2 ? f : 0.0; pv[3] += v == 3 ? f : 0.0;
                }
            } else {
                for (int t = rz_wave; t < ntiles; t += rz_nw) psum += fused(t, rz_lane, cc, Dr, Er, Dw, Ew);
            }
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_RUIZ)
            long long tr1 = 0, tr2 = 0;
#endif
            // ONE barrier per pass: every wavefront leaves the sum of its tiles in LDS before it (two sets of slots, by parity of
            // the pass: a wavefront that is already in the next pass must not overwrite what a slower one still reads), and behind
            // it every wavefront adds up the four partials in the same order and computes the nine arrival-cost norms itself
            // (lanes 0..8; their owner lanes store them for their own next pass).  Was: barrier | x_0 norms | sum through
            // group_combine with two more barriers.
            {
                const int wv = __builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6), ln = DEKF_LANE() & 63;
                dptr part = q.tmp + 176 + 4 * (it & 1);  // (the Gauss-Jordan scratch, dead during the scaling)
                if constexpr (RZ3) {
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (RZ_PW[v] == wv) { const double sv_ = wave_sum_dpp(pv[v]); if (ln == 0) part[v] = sv_; }
                } else {
                    psum = wave_sum_dpp(psum);
                    if (ln == 0) part[wv] = psum;
                }
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_RUIZ)
                __builtin_amdgcn_s_waitcnt(0);
                tr1 = clock64();
#endif
                DEKF_SYNC();
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_RUIZ)
                tr2 = clock64();
#endif
                const int j = ln < NS ? ln : NS - 1;
                const double dj = Dw[j];
                double v = 0.0;
                if constexpr (FT) {
                    for (int t = 0; t < NS; ++t) v = dmax(v, fabs(dj * symget(Mst, j, t, NS) * Dw[t]));
                } else {
                    double mj[9], dt9[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) { mj[t] = symget(Mst, j, t, 9); dt9[t] = Dw[t]; }
#pragma unroll
                    for (int t = 0; t < 9; ++t) v = dmax(v, fabs(dj * mj[t] * dt9[t]));
                }
                // pc of column j of x_0 is read by the wavefront that owns the column's tile (lane j % 3 of the first tile of
                // kind j / 3) in ITS next pass: that wavefront stores it (LDS serves a wavefront's accesses in order)
                const int cpos = j < 3 ? 1 : (j < 6 ? 0 : 2);  // (the order of the column tiles: velocity, position, bias)
                // (a foot-position column of x_0 sits in lane j - 9 of the first foot-column tile)
                const int otile = j < 9 ? ntm + ntp + 2 * ntd + cpos * ntx : ntm + ntp + 2 * ntd + 3 * ntx + ntf;
                if (ln < NS && (RZ3 ? rz_phys(otile) : (otile & (DEKF_NLANES() / WAVE - 1))) == wv) pc[j] = v;
                const double x0 = wave_sum_dpp(ln < NS ? v : 0.0);
                const int nw = RZ3 ? 4 : DEKF_NLANES() >> 6;
                double tot = part[0];
                for (int i = 1; i < nw; ++i) tot += part[i];
                psum = tot + x0;
            }
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_RUIZ)
            {
                const long long tr3 = clock64();
                const int w_ = DEKF_LANE() >> 6;
                if ((DEKF_LANE() & 63) == 0) { q.prof[24 + w_] += (double)(tr1 - tr0); q.prof[28 + w_] += (double)(tr2 - tr1); q.prof[17 + (w_ & 1)] += 0.5 * (double)(tr3 - tr2); }
            }
#endif
            psum *= cc;
            double qn = 0.0;
#pragma unroll
            for (int j = 0; j < NS; ++j) qn = dmax(qn, fabs(cc * Dw[j] * gq[j]));
            double ct = 1.0 / limit_scaling(dmax(psum / uni((double)n), limit_scaling(qn)));
            q.cc = uni_pin(cc * ct);
            dptr t0 = Dr; Dr = Dw; Dw = t0;
            dptr t1 = Er; Er = Ew; Ew = t1;
        }
        if (Dr != D) {  // odd number of passes: the result sits in the other pair
            DEKF_SYNC();
            cdptr Ds = Dr, Es = Er;
            wfor(n + m, [&](int e) { if (e < n) D[e] = Ds[e]; else E[e - n] = Es[e - n]; });
        }
        DEKF_SYNC();
        return;
    }
#endif
    for (int it = 0; it < q.c.scaling; ++it) {
        const double cc = q.cc;
        wtiles(ntiles, [&](int tile, int lane) { equil(tile, lane, cc); });
        DEKF_SYNC();
        double psum = 0.0;
        wtiles(ntiles, [&](int tile, int lane) { psum += adopt(tile, lane); });
        psum = wave_sum(psum);
        group_combine<1, true>(&psum);  // two barriers: D, E, pc of this pass are visible afterwards
        psum *= cc;
        // ---- cost normalisation: mean column norm of the re-scaled P against |q|_inf
        double qn = 0.0;
#pragma unroll
        for (int j = 0; j < NS; ++j) qn = dmax(qn, fabs(cc * D[j] * gq[j]));
        double ct = 1.0 / limit_scaling(dmax(psum / uni((double)n), limit_scaling(qn)));
        q.cc = uni_pin(cc * ct);
    }
    DEKF_SYNC();
}

// Two-sided block LDL' for a state dimension other than 9 (foot-position states: NS = 9 + 3 L): same recurrences and
// storage as step 3d of solve_factor, one wavefront per side, wave-level fences only.  The NS x NS inverse is the
// register-resident Gauss-Jordan of the assemble kernel (gj_columns: lane j holds column j, the pivot column comes
// over by v_readlane) — a 21-wide column no longer fits the 16-lane DPP row the 9 x 9 form broadcasts in.
template <class Q>
DEKF_FN bool factor_blocks_generic(Q& q) {
    constexpr int NS = Q::NS, NS2 = Q::NS2;
    using TM = TmpMap<NS>;
    const int K = q.K, mid = mid_block(K);
    dptr fail = q.tmp + TM::FAIL;
    if (DEKF_LANE() == 0) *fail = 0.0;
    DEKF_SYNC();
    // Operand staging in LDS.  The factor lives in the workgroup's HBM slab (S^-1 and W alone are 141 KB for Go1), and a
    // block needs W and C of its predecessor for the Schur update, then its own C for W = C S^-1: three 21 x 21 x 21
    // products whose operands, read from the slab element by element, cost a memory round trip per handful of loads
    // (92 k cycles per block measured).  xt | zt | at are dead during a factorisation (phase_rows<RESTART> rebuilds them),
    // so each side keeps two NS x NS panels there: `wa` = W of the block just finished, `ca` = its C — exactly what the next
    // block of that side subtracts.  Per block the slab is then touched four times, coalesced: T_kk in, C in, S^-1 out, W out.
    // (only when the factor is NOT in LDS already, and only if the dead vectors are long enough: short windows are not)
    dptr stage[2] = {q.xt, q.xt + 2 * NS2 + 8};
    const bool staged = !Q::FACTOR_LDS && 2 * (2 * NS2 + 8) <= (int)(q.xs - q.xt);
#if DEKF_DEVICE_BUILD
    const int lane = DEKF_LANE() & 63, l0 = lane, st = WAVE;
#else
    const int l0 = 0, st = 1;
#endif
    auto copy_panel = [&](dptr dst, cdptr src) {
        for (int p = l0; p < NS2; p += st) dst[p] = src[p];
    };
    // wmode 0: none (meeting block), 1: W_k = C_k S^-1 -> Wk[k], 2: What_{k-1} = C_{k-1}' S^-1 -> Wk[k-1]
    // have_prev: wa / ca of this side already hold the predecessor's W and C
    auto factor_block = [&](int k, bool use_top, bool use_bot, int wmode, dptr ts, int side, bool have_prev) -> bool {
        dptr wa = stage[side];
        dptr ca = stage[side] + NS2;
        if (staged && !have_prev) {  // (not taken today: every caller arrives with its predecessor's panels staged)
            if (use_top) { copy_panel(stage[0], q.Wk + (k - 1) * NS2); copy_panel(stage[0] + NS2, q.PA + (k - 1) * NS2); }
            if (use_bot) { copy_panel(stage[1], q.Wk + k * NS2); copy_panel(stage[1] + NS2, q.PA + k * NS2); }
        }
        cdptr Wt = staged ? stage[0] : q.Wk + (use_top ? k - 1 : 0) * NS2;
        cdptr Ct = staged ? stage[0] + NS2 : q.PA + (use_top ? k - 1 : 0) * NS2;
        cdptr Wb = staged ? stage[1] : q.Wk + k * NS2;
        cdptr Cb = staged ? stage[1] + NS2 : q.PA + k * NS2;
        for (int p = l0; p < NS2; p += st) ts[p] = q.Sinv[k * NS2 + p];
        wave_sync();
        auto schur = [&](cdptr W, cdptr C, bool top) {
            for (int p = l0; p < NS2; p += st) {
                const int i = p / NS, j = p - NS * i;
                cdptr wr = W + NS * i;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                // C couples the base states among themselves (9 x 9) and every foot with itself (3 x 3): row / column j of C is
                // exactly zero outside [t0, t1) (solve_factor 3b-3c writes those zeros), so only that range is summed — in the
                // same three accumulators, i.e. bit-identical to the full sum: 5.6 products per entry instead of 21
                // (branch-free: the first group of three for every entry, the second and third — base columns only — computed from clamped
                // indices and selected; with run-time loop bounds every group waited for its own LDS reads, 7.4 k cycles per block)
                const bool base = j < 9;
                const int t0 = base ? 0 : 9 + 3 * ((j - 9) / 3), tb = base ? 3 : t0, tc = base ? 6 : t0;
                double w[9], cv[9];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int t = (u == 0 ? t0 : (u == 1 ? tb : tc));
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        w[3 * u + v] = wr[t + v];
                        cv[3 * u + v] = top ? C[NS * j + t + v] : C[NS * (t + v) + j];
                    }
                }
                s0 += w[0] * cv[0]; s1 += w[1] * cv[1]; s2 += w[2] * cv[2];
#pragma unroll
                for (int u = 1; u < 3; ++u) {
                    const double n0 = s0 + w[3 * u] * cv[3 * u], n1 = s1 + w[3 * u + 1] * cv[3 * u + 1], n2 = s2 + w[3 * u + 2] * cv[3 * u + 2];
                    s0 = base ? n0 : s0; s1 = base ? n1 : s1; s2 = base ? n2 : s2;
                }
                ts[p] -= s0 + (s1 + s2);
            }
        };
        // the top side's panels are in stage[0], the bottom side's in stage[1] (the meeting block reads both)
        if (use_top) schur(Wt, Ct, true);
        if (use_top && use_bot) wave_sync();
        if (use_bot) schur(Wb, Cb, false);
        wave_sync();
        bool good;
#if DEKF_DEVICE_BUILD
        // AUGMENTED sweep (as in the 9-state form, solve_factor 3d): lanes NS .. 2 NS - 1 carry the columns of C' (W_k = C_k S^-1: row c
        // of C) or of C (W^_{k-1} = C_{k-1}' S^-1: column c); the sweep turns a column v into S^-1 v = row c of the W block with the
        // instructions every lane executes anyway (the pivot column travels by v_readlane, i.e. to all 64 lanes) — no product.
        constexpr bool AUG = 2 * NS <= WAVE;
        {
            const int j = lane < NS ? lane : NS - 1;
            const int kw_ = wmode == 1 ? k : k - 1;
            const bool ccol = AUG && wmode != 0 && lane >= NS && lane < 2 * NS;
            const int cidx = ccol ? lane - NS : 0;
            // ... and, when they fit too, the columns of the identity (lanes 2 NS .. 3 NS - 1), which end as the columns of S^-1: the
            // sweep then needs no in-place handling of the pivot column (gj_columns_plain)
            constexpr bool IDC = AUG && 3 * NS <= WAVE;
            const bool icol = IDC && lane >= 2 * NS && lane < 3 * NS;
            const int ic = icol ? lane - 2 * NS : 0;
            double a[NS];
            if (ccol) {
                cdptr src = q.PA + kw_ * NS2 + (wmode == 1 ? NS * cidx : cidx);
                const int cst = wmode == 1 ? 1 : NS;
#pragma unroll
                for (int i = 0; i < NS; ++i) a[i] = src[cst * i];
            } else {
#pragma unroll
                for (int i = 0; i < NS; ++i) a[i] = ts[NS * i + j];
            }
            if (icol) {
#pragma unroll
                for (int i = 0; i < NS; ++i) a[i] = i == ic ? 1.0 : 0.0;
            }
            if constexpr (IDC) good = gj_columns_plain<NS>(a, lane);
            else good = gj_columns<NS>(a, lane);
            wave_sync();  // every lane has read its column of S
            if (IDC ? icol : lane < NS) {
                const int jc = IDC ? ic : lane;
#pragma unroll
                for (int i = 0; i < NS; ++i) ts[NS * i + jc] = a[i];
            }
            if (ccol) {
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    if (staged) wa[NS * cidx + t] = a[t];
                    q.Wk[kw_ * NS2 + NS * cidx + t] = a[t];
                }
            }
        }
#else
        constexpr bool AUG = false;
        good = winverse_definite(ts, NS, ts + NS2);
#endif
        if (staged && wmode != 0) copy_panel(ca, q.PA + (wmode == 1 ? k : k - 1) * NS2);
        wave_sync();
        copy_panel(q.Sinv + k * NS2, ts);
        if (wmode != 0 && !AUG) {
            const int kw = wmode == 1 ? k : k - 1;
            cdptr cs0 = staged ? ca : q.PA + kw * NS2;
            for (int p = l0; p < NS2; p += st) {
                const int i = p / NS, jj = p - NS * i;
                cdptr cr = wmode == 1 ? cs0 + NS * i : cs0 + i;  // row i of C, or column i (C')
                const int cs = wmode == 1 ? 1 : NS;
                cdptr tc = ts + jj;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                for (int t = 0; t + 2 < NS; t += 3) {
                    s0 += cr[t * cs] * tc[NS * t];
                    s1 += cr[(t + 1) * cs] * tc[NS * (t + 1)];
                    s2 += cr[(t + 2) * cs] * tc[NS * (t + 2)];
                }
                const double wv = s0 + (s1 + s2);
                if (staged) wa[p] = wv;
                q.Wk[kw * NS2 + p] = wv;
            }
        }
        wave_sync();
        return good;
    };
    static_assert(NS % 3 == 0, "state dimension is a multiple of 3");
    two_waves(
        [&] {
            bool g = true;
            for (int k = 0; k < mid; ++k) g = factor_block(k, k > 0, false, 1, q.tmp + TM::SIDE0, 0, k > 0) && g;
            if (!g && (DEKF_LANE() & 63) == 0) *fail = 1.0;
        },
        [&] {
            bool g = true;
            for (int k = K - 1; k > mid; --k) g = factor_block(k, false, k < K - 1, 2, q.tmp + TM::SIDE1, 1, k < K - 1) && g;
            if (!g && (DEKF_LANE() & 63) == 0) *fail = 1.0;
        });
    DEKF_SYNC();
    two_waves(
        [&] {
            // the panels of both sides are still staged when the side ran at least one block
            bool g = factor_block(mid, mid > 0, mid < K - 1, 0, q.tmp + TM::SIDE0, 0, true);
            if (!g && (DEKF_LANE() & 63) == 0) *fail = 1.0;
        },
        [&] {});
    DEKF_SYNC();
    const bool ok = *fail == 0.0;
    DEKF_SYNC();
    return ok;
}

// numeric factorisation for the current rho: slack-block inverses, effective row weights,
// block-tridiagonal LDL' (S_k^-1 packed symmetric, W_k)
template <class Q>
DEKF_FN bool solve_factor(Q& q) {
    const DevCfg& c = q.c;
    constexpr int L = Q::LEGS, FT = Q::FOOT, NS = Q::NS, NS2 = Q::NS2;
    const int K = q.K;
    const auto& ix = q.ix;
    const double sigma = q.sigma(), cc = q.cc;
    // 3a. slack blocks: one lane per block, P blocks from the staged copy (Sinv | Wk are dead here:
    //     the previous factor is being replaced)
    constexpr int PS = 6 * L + 27 + 6 * L * FT, NB = L + 2 + L * FT;  // blocks per step: Meas legs, Dyn p+v+bias, VO, [foot Dyn]
    if (!q.staged) stage_p(q);
    q.staged = false;  // 3c overwrites the staging area
    // a 3x3 slack block with a packed symmetric P block q6: rows r0.., slack variables sv0.. -> inverse Si, row weights Wt
    // Three-workgroup kernels keep the slack-block inverses in the HBM slab: there they are stored ENTRY-major ([entry][block]) so
    // that the lanes of a wavefront, which own consecutive blocks, write and read consecutive doubles (block-major, a wave's store
    // of one entry touched up to 64 different 32-byte sectors: 288 such stores per solve were most of the kernel's write traffic)
    // (RR — rows in registers with the factor in the slab — keeps them block-major: a lane re-reads its block's inverse in every
    // iteration, and a contiguous block is one base address with immediate offsets and wide loads instead of 6 to 15 separate
    // 64-bit addresses)
    constexpr bool ST = Q::R3 && Q::FACTOR_LDS;
    const int stv = ST ? K * L : 1, stw = ST ? K : 1;
    auto block3 = [&](cdptr q6, int r0, int sv0, dptr Si_out, dptr W_out, int sst) {
        double gv[3], rr[3], S6[6], Si[6];
        for (int a = 0; a < 3; ++a) {
            rr[a] = q.rho_at(r0 + a);
            gv[a] = rr[a] * q.E[r0 + a] * q.D[sv0 + a];
        }
        for (int a = 0; a < 3; ++a)
            for (int d = a; d < 3; ++d)
                S6[symidx(a, d, 3)] = cc * q.D[sv0 + a] * q6[symidx(a, d, 3)] * q.D[sv0 + d];
        for (int a = 0; a < 3; ++a) S6[symidx(a, a, 3)] += sigma + gv[a] * q.E[r0 + a] * q.D[sv0 + a];
        inv3_sym(S6, Si);
        for (int t = 0; t < 6; ++t) Si_out[t * sst] = Si[t];
        for (int a = 0; a < 3; ++a)
            for (int d = a; d < 3; ++d)
                W_out[symidx(a, d, 3)] = (a == d ? rr[a] : 0.0) - gv[a] * Si[symidx(a, d, 3)] * gv[d];
    };
    // One kind of block per wavefront-sized tile (Meas leg blocks | Dyn 6x6 + bias | VO | foot Dyn), so that a wavefront runs ONE
    // of the four bodies: as items of mixed kind spread over all lanes, every wavefront ran all of them one after the other
    // (Go1: 14 k cycles per factorisation, most of it the 6x6 inverse that only 19 of 256 lanes needed)
    (void)NB;
    const int n3m = K * L, n3d = K - 1, ntm3 = (n3m + 63) >> 6, ntd3 = (n3d + 63) >> 6, ntf3 = FT ? ((K - 1) * L + 63) >> 6 : 0;
    wtiles(ntm3 + 2 * ntd3 + ntf3, [&](int tile, int lane) {
        int k, blk;
        if (tile < ntm3) { const int e = tile * 64 + lane; if (e >= n3m) return; k = e / L; blk = e - k * L; }
        else if (tile < ntm3 + ntd3) { k = (tile - ntm3) * 64 + lane; if (k >= n3d) return; blk = L; }
        else if (tile < ntm3 + 2 * ntd3) { k = (tile - ntm3 - ntd3) * 64 + lane; if (k >= n3d) return; blk = L + 1; }
        else { const int e = (tile - ntm3 - 2 * ntd3) * 64 + lane; if (e >= (K - 1) * L) return; k = e / L; blk = L + 2 + (e - k * L); }
        cdptr pk = q.Pst + k * PS;  // [Qm 6L | Qd 21 | Qc 6 | Qf 6L]
        if (blk < L) {
            block3(pk + 6 * blk, ix.rm(k, 3 * blk), ix.v(k, 3 * blk), q.Sv + (k * L + blk) * (ST ? 1 : 6), q.Wm + (k * L + blk) * 6, stv);
        } else if (k < K - 1 && blk == L) {
            cdptr q21 = pk + 6 * L;
            double S[36], gv[9], rr[9];
#pragma unroll
            for (int a = 0; a < 9; ++a) {
                int row = ix.rd(k, a);
                rr[a] = q.rho_at(row);
                gv[a] = rr[a] * q.E[row] * q.D[ix.w(k, a)];
            }
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int d = 0; d < 6; ++d)
                    S[6 * a + d] = cc * q.D[ix.w(k, a)] * symget(q21, a, d, 6) * q.D[ix.w(k, d)] +
                                   (a == d ? sigma + gv[a] * q.E[ix.rd(k, a)] * q.D[ix.w(k, a)] : 0.0);
            inv_spd_unrolled<6>(S);
            dptr sw = q.Sw + k * (ST ? 1 : SWS);
            dptr wd = q.Wd + k * 24;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int d = a; d < 6; ++d) {
                    double si = 0.5 * (S[6 * a + d] + S[6 * d + a]);
                    sw[symidx(a, d, 6) * stw] = si;
                    wd[symidx(a, d, 6)] = (a == d ? rr[a] : 0.0) - gv[a] * si * gv[d];
                }
#pragma unroll
            for (int a = 6; a < 9; ++a) {
                double dw = q.D[ix.w(k, a)];
                double sdiag = cc * dw * c.Q_bias_dt2[a - 6] * dw + sigma + gv[a] * q.E[ix.rd(k, a)] * dw;
                sw[(21 + a - 6) * stw] = 1.0 / sdiag;
                wd[21 + a - 6] = rr[a] - gv[a] * gv[a] / sdiag;
            }
        } else if (k < K - 1 && blk == L + 1) {
            block3(pk + 6 * L + 21, ix.rv(k, 0), ix.c(k, 0), q.Sc + k * (ST ? 1 : 6), q.Wc + k * 6, stw);
        } else if (FT && k < K - 1 && blk >= L + 2) {
            const int leg = blk - L - 2;
            block3(pk + 6 * L + 27 + 6 * leg, ix.rd(k, 9 + 3 * leg), ix.w(k, 9 + 3 * leg), q.Sf + (k * L + leg) * 6, q.Wf + (k * L + leg) * 6, 1);
        }
    });
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 6);
    // 3b+3c. One lane per column j of block k: column j of PA_k = Wd_k (E A_dyn D) stays in registers and
    //     yields column j of T_kk (upper part, packed -> Sinv[k]) and column j of C_k (-> PA[k]; W_k goes
    //     to Wk[k] in 3d).  Tiles by column kind (position / velocity / bias) so that A_dyn's sparsity is
    //     compile-time structure; only the component a = j mod 3 is a run-time select.  (Was three sweeps
    //     over 1539 + 2439 + 1539 items with per-item index decoding: 57 k cycles per factorisation.)
    {
        const double dt = c.dt, hdt2 = c.hdt2;
        const int K1 = K - 1, ntx = (3 * K + 63) >> 6, ntxf = FT ? (3 * L * K + 63) >> 6 : 0;
        wtiles(3 * ntx + ntxf, [&](int tile, int lane) {
            if (FT && tile >= 3 * ntx) {
                // ---- column j = 9 + 3 leg + a of a foot-position state: it sits in its leg's Meas rows (+I, with -I on
                //      the base position), in the Dyn rows of this step (+I) and of the previous one (-I)
                const int e = (tile - 3 * ntx) * 64 + lane;
                if (e >= 3 * L * K) return;
                const int k = e / (3 * L), la = e - 3 * L * k, leg = la / 3, a = la - 3 * leg, j = 9 + la;
                const bool hn = k < K1, hp = k > 0;
                const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;
                const double dj = q.D[ix.x(k, j)];
                double em[3], dp[3], df[3], en[3], ep[3], dfn[3], wm[6], wn[6], wp[6];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    em[t] = q.E[ix.rm(k, 3 * leg + t)];
                    dp[t] = q.D[ix.x(k, t)];
                    df[t] = q.D[ix.x(k, 9 + 3 * leg + t)];
                    en[t] = q.E[ix.rd(kn, 9 + 3 * leg + t)];
                    ep[t] = q.E[ix.rd(kp, 9 + 3 * leg + t)];
                    dfn[t] = q.D[ix.x(hn ? k + 1 : k, 9 + 3 * leg + t)];
                }
#pragma unroll
                for (int t = 0; t < 6; ++t) { wm[t] = q.Wm[(k * L + leg) * 6 + t]; wn[t] = q.Wf[(kn * L + leg) * 6 + t]; wp[t] = q.Wf[(kp * L + leg) * 6 + t]; }
                for (int i = 0; i <= j; ++i) {  // upper part of column j of T_kk, mirrored
                    double v = 0.0;
                    if (i < 3) v = -em[i] * dp[i] * symget(wm, i, a, 3) * em[a] * dj;
                    else if (i >= 9 + 3 * leg) {
                        const int t = i - 9 - 3 * leg;
                        v = em[t] * df[t] * symget(wm, t, a, 3) * em[a] * dj;
                        if (hn) v += en[t] * df[t] * symget(wn, t, a, 3) * en[a] * dj;
                        if (hp) v += ep[t] * df[t] * symget(wp, t, a, 3) * ep[a] * dj;
                        if (i == j) v += sigma;
                    }
                    if (k == 0) v += cc * q.D[ix.x(0, i)] * q.Mp[NS * i + j] * dj;
                    q.Sinv[k * NS2 + NS * i + j] = v;
                    q.Sinv[k * NS2 + NS * j + i] = v;
                }
                if (hn) {  // column j of C_k: only the same foot of the next step
                    for (int i = 0; i < NS; ++i) {
                        const int t = i - 9 - 3 * leg;
                        q.PA[k * NS2 + NS * i + j] = (t >= 0 && t < 3) ? -en[t] * dfn[t] * symget(wn, t, a, 3) * en[a] * dj : 0.0;
                    }
                }
                return;
            }
            const int kind = tile < ntx ? 0 : (tile < 2 * ntx ? 1 : 2);
            const int e3 = (tile - kind * ntx) * 64 + lane;
            if (e3 >= 3 * K) return;
            const int k = e3 / 3, a = e3 - 3 * k, j = 3 * kind + a;
            const bool hn = k < K1, hp = k > 0;
            const int kn = hn ? k : 0, kp = hp ? k - 1 : 0;  // clamped: loads stay in bounds, results are selected away
            double dx[9], tc[9];  // D of x_k; column j of T_kk (rows 0..8, only i <= j is stored)
#pragma unroll
            for (int i = 0; i < 9; ++i) { dx[i] = q.D[ix.x(k, i)]; tc[i] = 0.0; }
            const double dj = q.D[ix.x(k, j)];
            // ---- this step's Dyn / VO rows
            {
                double en[9], ev[3], dn[9], Rk[9], pa[9], at[9];
                cdptr wd = q.Wd + kn * 24;
                cdptr wc = q.Wc + kn * 6;
#pragma unroll
                for (int t = 0; t < 9; ++t) { en[t] = q.E[ix.rd(kn, t)]; dn[t] = q.D[ix.x(kn + 1 < K ? kn + 1 : kn, t)]; Rk[t] = q.R[9 * kn + t]; }
#pragma unroll
                for (int t = 0; t < 3; ++t) ev[t] = q.E[ix.rv(kn, t)];
                // column j of E A_dyn D
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    double v;
                    if (kind == 0) v = (t == a) ? 1.0 : 0.0;
                    else if (kind == 1) v = (t == a) ? dt : (t == 3 + a ? 1.0 : 0.0);
                    else v = t < 3 ? -hdt2 * Rk[3 * t + a] : (t < 6 ? -dt * Rk[3 * (t - 3) + a] : (t == 6 + a ? 1.0 : 0.0));
                    at[t] = en[t] * v * dj;
                }
                // column j of PA = Wd (E A D): Wd = blkdiag(6x6 symmetric, diag 3)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int t = 0; t < 6; ++t) acc += wd[i < t ? symidx(i, t, 6) : symidx(t, i, 6)] * at[t];
                    pa[i] = acc;
                }
#pragma unroll
                for (int i = 6; i < 9; ++i) pa[i] = wd[21 + i - 6] * at[i];
                // (E A D)' PA, row i of the product: A_dyn's column i has compile-time structure
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    double s;
                    if (i < 3) s = en[i] * pa[i];
                    else if (i < 6) s = en[i - 3] * dt * pa[i - 3] + en[i] * pa[i];
                    else {
                        s = en[i] * pa[i];
#pragma unroll
                        for (int r = 0; r < 3; ++r) s -= Rk[3 * r + i - 6] * (en[r] * hdt2 * pa[r] + en[3 + r] * dt * pa[3 + r]);
                    }
                    if (kind == 0 && i < 3) s += ev[i] * symget(wc, i, a, 3) * ev[a] * dj;  // VO rows: +I on x_k[0:3]
                    tc[i] += hn ? dx[i] * s : 0.0;
                }
                // column j of C_k = -(E D_{k+1}) PA  (- the VO coupling on the position block)
                if (hn) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        double cv = -en[i] * dn[i] * pa[i];
                        if (kind == 0 && i < 3) cv -= ev[i] * dn[i] * symget(wc, i, a, 3) * ev[a] * dj;
                        q.PA[k * NS2 + NS * i + j] = cv;
                    }
                    if constexpr (FT) {
                        for (int i = 9; i < NS; ++i) q.PA[k * NS2 + NS * i + j] = 0.0;  // no coupling of a foot to the base across steps
                    }
                }
            }
            // ---- the previous step's Dyn / VO rows carry -I on x_k
            {
                cdptr wd = q.Wd + kp * 24;
                cdptr wc = q.Wc + kp * 6;
                const double ej = q.E[ix.rd(kp, j)];
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    double w;
                    if (kind == 2) w = (i == 6 + a) ? wd[21 + a] : 0.0;
                    else if (i < 6) w = symget(wd, i, j, 6);
                    else w = 0.0;
                    double s = q.E[ix.rd(kp, i)] * w * ej;
                    if (kind == 0 && i < 3) s += q.E[ix.rv(kp, i)] * symget(wc, i, a, 3) * q.E[ix.rv(kp, a)];
                    tc[i] += hp ? dx[i] * s * dj : 0.0;
                }
            }
            // ---- Meas rows (velocity block; with foot-position states the position block), arrival cost (block 0), sigma
            if constexpr (FT) {
                if (kind == 0) {
#pragma unroll
                    for (int leg = 0; leg < L; ++leg) {
                        cdptr wm = q.Wm + (k * L + leg) * 6;
                        const double ea = q.E[ix.rm(k, 3 * leg + a)];
#pragma unroll
                        for (int i = 0; i < 3; ++i) tc[i] += q.E[ix.rm(k, 3 * leg + i)] * dx[i] * symget(wm, i, a, 3) * ea * dj;
                    }
                }
            } else if (kind == 1) {
#pragma unroll
                for (int leg = 0; leg < L; ++leg) {
                    cdptr wm = q.Wm + (k * L + leg) * 6;
                    const double ea = q.E[ix.rm(k, 3 * leg + a)];
#pragma unroll
                    for (int i = 3; i < 6; ++i) tc[i] += q.E[ix.rm(k, 3 * leg + i - 3)] * dx[i] * symget(wm, i - 3, a, 3) * ea * dj;
                }
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (i > j) continue;
                double v = tc[i] + (i == j ? sigma : 0.0);
                if (k == 0) v += cc * dx[i] * q.Mp[NS * i + j] * dj;
                q.Sinv[k * NS2 + NS * i + j] = v;  // full NS x NS storage, both triangles
                q.Sinv[k * NS2 + NS * j + i] = v;
            }
        });
        DEKF_SYNC();
    }
    DEKF_PROF_MARK(q, 7);
    // 3d. TWO-SIDED block LDL' ("burn at both ends"): blocks 0..mid-1 are eliminated downwards,
    //     blocks K-1..mid+1 upwards, both fronts in the same phases (two 9x9 problems per phase),
    //     and they meet in block mid.  Same storage as a one-sided factorisation, half the depth:
    //       top     S_k = T_kk - W_{k-1} C_{k-1}',      W_k     = C_k S_k^-1        (k <  mid)
    //       bottom  S_k = T_kk - What_k C_k,            What_{k-1} = C_{k-1}' S_k^-1 (k >  mid)
    //       middle  S_m = T_mm - W_{m-1} C_{m-1}' - What_m C_m
    //     W_k lives in Wk[k] for k < mid, What_k in Wk[k] for k >= mid (it couples block k+1 to k).
    if constexpr (NS != 9) {
        // (measured for 9-state blocks with the factor in the HBM slab, PogoX: the staged generic form is 1.5 % SLOWER than the
        // DPP Gauss-Jordan below reading its operands from the slab — 84.9 k against 86.2 k steps/s)
        const bool okg = factor_blocks_generic(q);
        DEKF_PROF_MARK(q, 8);
        return okg;
    } else {
    bool ok = true;
    const int mid = mid_block(K);
    const int nph = (mid > K - 1 - mid ? mid : K - 1 - mid);
    dptr bufs[2][2] = {{q.tmp, q.tmp + 81}, {q.tmp + 176, q.tmp + 257}};
    // S of block k into dst; use_top / use_bot select the Schur terms
    auto build_s = [&](int k, bool use_top, bool use_bot, int p, dptr dst) {
        int i = p / 9, j = p - 9 * i;
        int lo_ = i < j ? i : j, hi_ = i < j ? j : i;
        double acc = q.Sinv[k * 81 + 9 * lo_ + hi_];
        if (use_top) {
            cdptr Wp = q.Wk + (k - 1) * 81;
            cdptr Cp = q.PA + (k - 1) * 81;
            double s1 = 0.0, s2 = 0.0;
            for (int t = 0; t < 9; ++t) { s1 += Wp[9 * i + t] * Cp[9 * j + t]; s2 += Wp[9 * j + t] * Cp[9 * i + t]; }
            acc -= 0.5 * (s1 + s2);
        }
        if (use_bot) {
            cdptr Wh = q.Wk + k * 81;
            cdptr Ck = q.PA + k * 81;
            double s1 = 0.0, s2 = 0.0;
            for (int t = 0; t < 9; ++t) { s1 += Wh[9 * i + t] * Ck[9 * t + j]; s2 += Wh[9 * j + t] * Ck[9 * t + i]; }
            acc -= 0.5 * (s1 + s2);
        }
        dst[p] = acc;
    };
    auto gj_step = [&](cdptr src, dptr dst, int pv, int p) {
        int i = p / 9, j = p - 9 * i;
        double d = 1.0 / src[pv * 9 + pv];
        double v;
        if (i == pv) v = (j == pv) ? d : src[p] * d;
        else if (j == pv) v = -src[p] * d;
        else v = src[p] - src[i * 9 + pv] * src[pv * 9 + j] * d;
        dst[p] = v;
    };
    auto pivot_ok = [&](cdptr src, int pv) { double piv = src[pv * 9 + pv]; return (fabs(piv) > 0.0) && (fabs(piv) < 1e300); };
    auto store_sinv = [&](int k, cdptr src, int e) {
        int p = e, i = 0;
        while (p >= 9 - i) { p -= 9 - i; ++i; }
        int j = i + p;
        const double sv = 0.5 * (src[9 * i + j] + src[9 * j + i]);
        q.Sinv[k * 81 + 9 * i + j] = sv;
        q.Sinv[k * 81 + 9 * j + i] = sv;
    };
#if DEKF_DEVICE_BUILD
    // Device: each side is ONE wavefront working through its blocks with wave-level syncs only; the
    // 9x9 inversion is a Gauss-Jordan sweep on registers (lane j < 9 holds column j, the pivot column is
    // broadcast with v_readlane), so a block costs three LDS round trips instead of eleven workgroup
    // barriers.  The host build below keeps the phase-per-pivot form (same arithmetic up to rounding).
    (void)nph; (void)bufs; (void)gj_step; (void)pivot_ok; (void)store_sinv; (void)build_s;
    // S of block k, element p (device form: three independent accumulators per sum, no symmetrisation —
    // W C' = C S^-1 C' is symmetric up to rounding and the Gauss-Jordan sweep does not need more)
    constexpr bool W_DIRECT = Q::FACTOR_LDS;
    // twp / twh: with the factor in the HBM slab, the LDS copies of W_{k-1} / W^_k that the previous block of the leg left behind
    // (a block must not wait for its predecessor's W to travel to L2 and back: 2-3 us per block of a PogoX window)
    auto build_s3 = [&](int k, bool use_top, bool use_bot, int p, dptr dst, cdptr twp, cdptr twh) -> double {
        const int i = p / 9, j = p - 9 * i;
        const int lo_ = i < j ? i : j, hi_ = i < j ? j : i;
        double acc = q.Sinv[k * 81 + 9 * lo_ + hi_];
        if (use_top) {
            cdptr Wp = (W_DIRECT ? q.Wk + (k - 1) * 81 : twp) + 9 * i;
            cdptr Cp = q.PA + (k - 1) * 81 + 9 * j;
            double s0 = Wp[0] * Cp[0] + Wp[3] * Cp[3] + Wp[6] * Cp[6];
            double s1 = Wp[1] * Cp[1] + Wp[4] * Cp[4] + Wp[7] * Cp[7];
            double s2 = Wp[2] * Cp[2] + Wp[5] * Cp[5] + Wp[8] * Cp[8];
            acc -= s0 + (s1 + s2);
        }
        if (use_bot) {
            cdptr Wh = (W_DIRECT ? q.Wk + k * 81 : twh) + 9 * i;
            cdptr Ck = q.PA + k * 81 + j;
            double s0 = Wh[0] * Ck[0] + Wh[3] * Ck[27] + Wh[6] * Ck[54];
            double s1 = Wh[1] * Ck[9] + Wh[4] * Ck[36] + Wh[7] * Ck[63];
            double s2 = Wh[2] * Ck[18] + Wh[5] * Ck[45] + Wh[8] * Ck[72];
            acc -= s0 + (s1 + s2);
        }
        dst[p] = acc;
        return acc;
    };
    // Slab-resident factor: what an entry of S_k needs from the slab — T_kk(i, j) and the nine entries of row / column j of the
    // coupling block — is requested one block AHEAD (the leg loops below), so that the loads travel while the previous block is
    // being inverted; the W rows come from the LDS copy.  Same arithmetic, same order as build_s3.
    struct SOps { double t, c[9]; };
    auto s_entry = [&](int e, int& i, int& j) {
        i = (e >= 9) + (e >= 17) + (e >= 24) + (e >= 30) + (e >= 35) + (e >= 39) + (e >= 42) + (e >= 44);
        j = e - (9 * i - ((i * (i - 1)) >> 1)) + i;
    };
    auto s_ops_load = [&](int k, bool use_top, SOps& o) {  // use_top: a block of the top leg (k > 0), else of the bottom leg (k < K - 1)
        const int e = (DEKF_LANE() & 63) < 45 ? (DEKF_LANE() & 63) : 44;
        int i, j;
        s_entry(e, i, j);
        o.t = q.Sinv[k * 81 + 9 * i + j];
        cdptr cp = use_top ? q.PA + (k - 1) * 81 + 9 * j : q.PA + k * 81 + j;
        const int st = use_top ? 1 : 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) o.c[t] = cp[st * t];
    };
    const int ldl_lane = DEKF_LANE() & 63;
    (void)ldl_lane;
    auto factor_block = [&](int k, bool use_top, bool use_bot, int wmode, dptr tb, cdptr twp, cdptr twh, dptr two, const SOps* po = nullptr) -> bool {
        // wmode 0: none (meeting block), 1: W_k = C_k S^-1 -> Wk[k], 2: What_{k-1} = C_{k-1}' S^-1 -> Wk[k-1]; two: where the LDS
        // copy of that W block goes when the factor itself lives in the HBM slab
        const int lane = ldl_lane;  // (read once in front of the leg loops: a block's lane roles and addresses are loop-invariant)
        dptr ts = tb;        // S, then the full inverse
        // S is symmetric (T_kk exactly, the Schur terms up to rounding): its 45 upper entries in ONE round of the wavefront,
        // each stored to both places (the Gauss-Jordan sweep below reads columns)
        if (lane < 45) {
            int i, j;
            s_entry(lane, i, j);
            if (po && (use_top != use_bot)) {  // operands prefetched by the leg loop
                cdptr Wr = (use_top ? twp : twh) + 9 * i;
                const double s0 = Wr[0] * po->c[0] + Wr[3] * po->c[3] + Wr[6] * po->c[6];
                const double s1 = Wr[1] * po->c[1] + Wr[4] * po->c[4] + Wr[7] * po->c[7];
                const double s2 = Wr[2] * po->c[2] + Wr[5] * po->c[5] + Wr[8] * po->c[8];
                const double acc = po->t - (s0 + (s1 + s2));
                ts[9 * i + j] = acc;
                ts[9 * j + i] = acc;
            } else if (po) {  // first block of a leg: T_kk alone
                ts[9 * i + j] = po->t;
                ts[9 * j + i] = po->t;
            } else {
                ts[9 * j + i] = build_s3(k, use_top, use_bot, 9 * i + j, ts, twp, twh);
            }
        }
        wave_sync();
        // AUGMENTED sweep: next to the nine columns of S the wavefront carries the nine columns of C' (W_k = C_k S^-1: row c of C) or
        // of C (W^_{k-1} = C_{k-1}' S^-1: column c of C); the Gauss-Jordan sweep turns a column v into S^-1 v, i.e. into row c of
        // the W block — with the SAME nine DPP instructions per pivot, which every lane executes anyway.  A DPP broadcast stays
        // inside its 16-lane row, so rows 0 and 1 both hold the columns of S (lanes 0..8) and share the eighteen others: row 0
        // lanes 9..15 take c = 0..6, row 1 lanes 9, 10 take c = 7, 8.  No copy of the inverse for a product, no product: the
        // block's critical path loses two LDS round trips and the two rounds of the 81-entry product.
        const int li = lane & 15, rw = lane >> 4;
        const int j = li < 9 ? li : 8;
        const int cidx = li - 9 + (rw == 0 ? 0 : 7);
        const bool ccol = wmode != 0 && li >= 9 && ((rw == 0) || (rw == 1 && li < 11));
        const int kw = wmode == 1 ? k : k - 1;
        // ... and the nine columns of the IDENTITY ride along too (row 1 lanes 11..15: c = 0..4, row 2 lanes 9..12: c = 5..8): they end
        // as the columns of S^-1, so the sweep needs no in-place handling of the pivot column (a multiplication by 0 / 1 and a select
        // per row and pivot): every lane runs the plain elimination a_i -= s_ip (a_p / s_pp).  Same values as the in-place form.
        const bool icol = (rw == 1 && li >= 11) || (rw == 2 && li >= 9 && li <= 12);
        const int ic = rw == 1 ? li - 11 : li - 9 + 5;
        double a[9];
        if (ccol) {  // (two loops, not one pointer select: C may sit in the HBM slab while the scratch is LDS)
            cdptr src = q.PA + kw * 81 + (wmode == 1 ? 9 * cidx : cidx);
            const int st = wmode == 1 ? 1 : 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) a[i] = src[st * i];
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) a[i] = ts[9 * i + j];
        }
        if (icol) {
#pragma unroll
            for (int i = 0; i < 9; ++i) a[i] = i == ic ? 1.0 : 0.0;
        }
        bool good = true;
        // pivot pv: every lane needs column pv (lane pv's registers).  The pivot itself goes through one
        // v_readlane pair; the other eight entries are consumed straight from lane pv by the update FMA
        // (v_fmac_f64_dpp row_newbcast:pv), with base = 0, multiplier = d on lane pv itself, whose new
        // column is -col * d.  Two wait states between the VALU write of a[i] and its DPP read: the s_nop.
#define DEKF_GJ_DPP(acc, src, mul, PV) \
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:" #PV " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul))
#define DEKF_GJ_UPD(I, PV)                                                       \
            {                                                                    \
                (void)keep;                                                      \
                DEKF_GJ_DPP(a[I], a[I], m, PV);                                  \
            }
#define DEKF_GJ_OWN(PV) false
        // The row that holds the NEXT pivot is updated first and its reciprocal started at once, so that the
        // v_rcp + Newton chain runs underneath the seven remaining (quarter-rate) DPP FMAs of this pivot.
        // Own lane: base 0 through a multiplication by `keep` (one f64 multiply instead of two 32-bit selects).
#define DEKF_GJ_PIVOT(PV, NX)                                                    \
        {                                                                        \
            const bool own = DEKF_GJ_OWN(PV);                                    \
            const double keep = own ? 0.0 : 1.0;                                 \
            const double m = own ? d : a[PV] * d;                                \
            if (NX < 9) {                                                        \
                DEKF_GJ_UPD(NX < 9 ? NX : 0, PV)                                 \
                const double pn = readlane_f64(a[NX < 9 ? NX : 0], NX < 9 ? NX : 0); \
                good = good && (fabs(pn) > 0.0) && (fabs(pn) < 1e300);           \
                d = rcp_fast(pn);                                                \
            }                                                                    \
            _Pragma("unroll") for (int i = 0; i < 9; ++i) {                      \
                if (i == PV || i == NX) continue;                                \
                DEKF_GJ_UPD(i, PV)                                               \
            }                                                                    \
            a[PV] = m;                                                           \
        }
        double d;
        {
            const double p0 = readlane_f64(a[0], 0);
            good = good && (fabs(p0) > 0.0) && (fabs(p0) < 1e300);
            d = rcp_fast(p0);
        }
        DEKF_GJ_PIVOT(0, 1) DEKF_GJ_PIVOT(1, 2) DEKF_GJ_PIVOT(2, 3) DEKF_GJ_PIVOT(3, 4) DEKF_GJ_PIVOT(4, 5)
        DEKF_GJ_PIVOT(5, 6) DEKF_GJ_PIVOT(6, 7) DEKF_GJ_PIVOT(7, 8) DEKF_GJ_PIVOT(8, 9)
#undef DEKF_GJ_UPD
#undef DEKF_GJ_OWN
#undef DEKF_GJ_PIVOT
#undef DEKF_GJ_DPP
        const bool inv_lane = icol;
        const int jc = ic;
        if (inv_lane) {
            // column jc of the inverse to S^-1[k] (full 9x9: a row is contiguous, so the solve phases address it with immediates
            // instead of packed-index arithmetic); the meeting block also leaves it in the scratch for the joint middle
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                q.Sinv[k * 81 + 9 * i + jc] = a[i];
                if (wmode == 0) ts[9 * i + jc] = a[i];
            }
        }
        if (ccol) {
#pragma unroll
            for (int t = 0; t < 9; ++t) q.Wk[kw * 81 + 9 * cidx + t] = a[t];
            if constexpr (!W_DIRECT) {  // (the factor streams from the slab: a transposed copy for the outward legs, sweeps_one_wave_rt)
#pragma unroll
                for (int t = 0; t < 9; ++t) q.WT[kw * 81 + 9 * t + cidx] = a[t];
            }
            if constexpr (!W_DIRECT) {
#pragma unroll
                for (int t = 0; t < 9; ++t) two[9 * cidx + t] = a[t];
            }
        }
        wave_sync();
        return good;
    };
    dptr fail = q.tmp + 172;  // [162, 171) is the scaled q
    if (DEKF_LANE() == 0) *fail = 0.0;
    DEKF_SYNC();
    two_waves(
        [&] {
            bool g = true;
            if constexpr (W_DIRECT) {
                for (int k = 0; k < mid; ++k) g = factor_block(k, k > 0, false, 1, q.tmp, q.tmp + 81, q.tmp + 81, q.tmp + 81) && g;
            } else {
                SOps nx;
                if (mid > 0) s_ops_load(0, false, nx);  // (block 0 has no Schur term: only its T entry is used)
                for (int k = 0; k < mid; ++k) {
                    const SOps cu = nx;
                    if (k + 1 < mid) s_ops_load(k + 1, true, nx);
                    g = factor_block(k, k > 0, false, 1, q.tmp, q.tmp + 81, q.tmp + 81, q.tmp + 81, &cu) && g;
                }
            }
            if (!g && (DEKF_LANE() & 63) == 0) *fail = 1.0;
        },
        [&] {
            bool g = true;
            if constexpr (W_DIRECT) {
                for (int k = K - 1; k > mid; --k) g = factor_block(k, false, k < K - 1, 2, q.tmp + 176, q.tmp + 257, q.tmp + 257, q.tmp + 257) && g;
            } else {
                SOps nx;
                if (K - 1 > mid) s_ops_load(K - 1, true, nx);  // (block K - 1 has no Schur term; `true` only keeps the address in range)
                for (int k = K - 1; k > mid; --k) {
                    const SOps cu = nx;
                    if (k - 1 > mid) s_ops_load(k - 1, false, nx);
                    g = factor_block(k, false, k < K - 1, 2, q.tmp + 176, q.tmp + 257, q.tmp + 257, q.tmp + 257, &cu) && g;
                }
            }
            if (!g && (DEKF_LANE() & 63) == 0) *fail = 1.0;
        });
    DEKF_SYNC();
    DEKF_PROF_MARK(q, 22);
    two_waves(
        [&] {
            bool g = factor_block(mid, mid > 0, mid < K - 1, 0, q.tmp, q.tmp + 81, q.tmp + 257, q.tmp + 81);
            if (!g && (DEKF_LANE() & 63) == 0) *fail = 1.0;
            // JOINT MIDDLE for the fixed-horizon one-wavefront solve (sweeps_one_wave): blocks m and m + 1 are solved together,
            //   [u_m; u_{m+1}] = [P11 P12; P12' P22] [f_m; f^_{m+1}],   P11 = S_m^-1 (just computed),
            //   P12 = -P11 W^_m,   P22 = S_{m+1}^-1 + W^_m' P11 W^_m = S_{m+1}^-1 - W^_m' P12,
            // which turns the meeting block's two dependent mat-vecs into one and makes both outward legs m steps long
            // (21 -> 19.3 dependent steps per iteration).  P12 replaces W^_m in W[m], P22 replaces S_{m+1}^-1: the one-wavefront
            // solve needs neither of the originals.  Only for the full window of a compile-time even horizon — every other
            // solve form (window fill, generic horizons, the lane-sequential build) keeps the plain storage.
            if constexpr (Q::NFIXED >= 4 && Q::NFIXED % 2 == 0) {
                if (K == Q::NFIXED) {
                    const int lane = DEKF_LANE() & 63;
                    cdptr P11 = q.tmp;           // factor_block left S_m^-1 there
                    dptr P12 = q.tmp + 81;
                    cdptr Wh = q.Wk + mid * 81;  // W^_m
                    for (int pp = lane; pp < 81; pp += WAVE) {
                        const int i = pp / 9, j = pp - 9 * i;
                        double s0 = P11[9 * i] * Wh[j] + P11[9 * i + 3] * Wh[27 + j] + P11[9 * i + 6] * Wh[54 + j];
                        double s1 = P11[9 * i + 1] * Wh[9 + j] + P11[9 * i + 4] * Wh[36 + j] + P11[9 * i + 7] * Wh[63 + j];
                        double s2 = P11[9 * i + 2] * Wh[18 + j] + P11[9 * i + 5] * Wh[45 + j] + P11[9 * i + 8] * Wh[72 + j];
                        P12[pp] = -(s0 + (s1 + s2));
                    }
                    wave_sync();
                    double p22[2];
                    for (int pp = lane, n = 0; pp < 81; pp += WAVE, ++n) {
                        const int i = pp / 9, j = pp - 9 * i;  // (W^_m' P12)(i, j) = sum_t W^_m(t, i) P12(t, j)
                        double s0 = Wh[i] * P12[j] + Wh[27 + i] * P12[27 + j] + Wh[54 + i] * P12[54 + j];
                        double s1 = Wh[9 + i] * P12[9 + j] + Wh[36 + i] * P12[36 + j] + Wh[63 + i] * P12[63 + j];
                        double s2 = Wh[18 + i] * P12[18 + j] + Wh[45 + i] * P12[45 + j] + Wh[72 + i] * P12[72 + j];
                        p22[n] = q.Sinv[(mid + 1) * 81 + pp] - (s0 + (s1 + s2));
                    }
                    wave_sync();  // every read of W^_m is done
                    for (int pp = lane, n = 0; pp < 81; pp += WAVE, ++n) {
                        q.Wk[mid * 81 + pp] = P12[pp];
                        if constexpr (!W_DIRECT) q.WT[mid * 81 + 9 * (pp % 9) + pp / 9] = P12[pp];
                        q.Sinv[(mid + 1) * 81 + pp] = p22[n];
                    }
                    wave_sync();
                }
            }
        },
        [&] {});
    DEKF_SYNC();
    ok = *fail == 0.0;
    DEKF_SYNC();
#else
    for (int t = 0; t <= nph; ++t) {
        const bool last = t == nph;             // the meeting block
        const int kt = last ? mid : t, kb = K - 1 - t;
        const bool vt = last || t < mid;        // side 0 active
        const bool vb = !last && kb > mid;      // side 1 active
        wfor(162, [&](int e) {
            int side = e / 81, p = e - 81 * side;
            if (side == 0 && vt) build_s(kt, kt > 0, last && mid < K - 1, p, bufs[0][0]);
            if (side == 1 && vb) build_s(kb, false, kb < K - 1, p, bufs[1][0]);
        });
        int cur = 0;
        for (int pv = 0; pv < 9; ++pv) {
            if (vt && !pivot_ok(bufs[0][cur], pv)) ok = false;
            if (vb && !pivot_ok(bufs[1][cur], pv)) ok = false;
            wfor(162, [&](int e) {
                int side = e / 81, p = e - 81 * side;
                if ((side == 0 && vt) || (side == 1 && vb)) gj_step(bufs[side][cur], bufs[side][cur ^ 1], pv, p);
            });
            cur ^= 1;
        }
        cdptr st = bufs[0][cur];
        cdptr sb = bufs[1][cur];
        wfor(252, [&](int e) {
            int side = e / 126, o = e - 126 * side;
            if (side == 0 && vt) {
                if (o < 45) store_sinv(kt, st, o);
                else if (!last) {  // W_kt = C_kt S^-1
                    int p = o - 45, i = p / 9, j = p - 9 * i;
                    cdptr Ck = q.PA + kt * 81;
                    double acc = 0.0;
                    for (int u = 0; u < 9; ++u) acc += Ck[9 * i + u] * st[9 * u + j];
                    q.Wk[kt * 81 + p] = acc;
                }
            } else if (side == 1 && vb) {
                if (o < 45) store_sinv(kb, sb, o);
                else {  // What_{kb-1} = C_{kb-1}' S^-1
                    int p = o - 45, i = p / 9, j = p - 9 * i;
                    cdptr Ck = q.PA + (kb - 1) * 81;
                    double acc = 0.0;
                    for (int u = 0; u < 9; ++u) acc += Ck[9 * u + i] * sb[9 * u + j];
                    q.Wk[(kb - 1) * 81 + p] = acc;
                }
            }
        });
    }
#endif
    DEKF_PROF_MARK(q, 8);
    return ok;
    }  // NS == 9
}

#include "mhe_admm_core.h"

// ---------------------------------------------------------------- osqp.polish: refinement against explicit residuals
// OSQP refines the polished point as polish.c does: r = rhs - K s with the UNREGULARISED matrix, K_reg ds = r, s += ds.  The
// correction is solved by the machinery that is there: one ADMM step with sigma = delta, 1 / rho = delta on the active rows and
// alpha = 1 solves  [P + delta I, A'; A, -delta I] [xt; nu] = [delta x - q; z - delta y]  — so with the state loaded as
// x := r_x / delta, q := 0, y := 0 and the rows held at z := r_y instead of at their bounds it returns xt = dx and y+ = dy.
// (Rounds 3's form iterated  K_reg s+ = rhs + delta [x; -y]  on the full point: the same recursion in exact arithmetic, but its
// error floor is relative to |s|, not to |ds| — the dual residual stopped at 1e-8 of the problem's scale where OSQP reaches 1e-13,
// and the device rejected polished points OSQP accepts.)
// Layout of q.pol: pv_xb [K NS] | pv_xs [m] | pv_y [m] | pv_lo [m] | pr_xb [K NS] | pr_xs [m] | pr_y [m]   (m = m_pad)
template <class Q>
struct PolishScratch {
    dptr pv_xb, pv_xs, pv_y, pv_lo, pr_xb, pr_xs, pr_y;
    DEKF_FN PolishScratch(const Q& q, int m_pad, int NH) {
        constexpr int NS = Q::NS;
        double* p = raw_of(q.pol);
        pv_xb = DEKF_SPAN(p, NS * NH); p += NS * NH;
        pv_xs = DEKF_SPAN(p, m_pad); p += m_pad;
        pv_y = DEKF_SPAN(p, m_pad); p += m_pad;
        pv_lo = DEKF_SPAN(p, m_pad); p += m_pad;
        pr_xb = DEKF_SPAN(p, NS * NH); p += NS * NH;
        pr_xs = DEKF_SPAN(p, m_pad); p += m_pad;
        pr_y = DEKF_SPAN(p, m_pad);
    }
};
// a VO row without bounds: not in OSQP's active set; here a free row (rho = RHO_MIN) whose multiplier stays exactly 0
// (a VO row is either an equality or the box -+1e30 E: its ORIGINAL lower bound tells which)
template <class Q>
DEKF_FN bool polish_free_row(const Q& q, int r, double lo_orig) {
    return r >= q.ix.rvb && lo_orig < -q.c.inf_thr;
}
// one step of the iteration kernels with sigma = delta, alpha = 1 from whatever the state arrays hold (a function, not a lambda: a
// lambda that is called twice is not inlined, and then the whole solve context lives in scratch memory — 1.7 KB per lane)
template <int NFIX, class Q>
DEKF_FN void polish_step(Q& q) {
#if DEKF_DEVICE_BUILD
    if constexpr (Q::R3) {
        if constexpr (Q::R4) admm_chunk_r4<NFIX>(q, 1, 1.0, q.sigma());
        else if constexpr (Q::FACTOR_LDS) admm_chunk_r3<NFIX>(q, 1, 1.0, q.sigma());
        else admm_chunk_rr(q, 1, 1.0, q.sigma());
    } else
#endif
    {
        phase_xcols(q, q.sigma());
        phase_sweeps_rows(q, 1.0, q.sigma());
    }
}
// save the polished point and the bounds; load the correction's right-hand side as the state (see above)
template <class Q>
DEKF_FN void polish_swap_in(Q& q, const PolishScratch<Q>& ps, double delta) {
    constexpr int NS = Q::NS;
    const int K = q.K, m = q.m;
    const double dinv = 1.0 / delta;
    wfor_nosync(m + K * NS, [&](int e) {
        if (e >= m) {
            const int i = e - m, k = i / NS, j = i - NS * k;
            if constexpr (Q::R3) { ps.pv_xb[i] = q.xb[i]; q.xb[i] = ps.pr_xb[i] * dinv; }
            else { const int xi = q.ix.x(k, j); ps.pv_xb[i] = q.x[xi]; q.x[xi] = ps.pr_xb[i] * dinv; }
            return;
        }
        const int r = e;
        const double lo_r = q.lo[r];
        const bool vo = r >= q.ix.rvb, fr = polish_free_row(q, r, lo_r);
        ps.pv_lo[r] = lo_r;
        if constexpr (Q::R3) {
            ps.pv_xs[r] = q.sx[r];
            ps.pv_y[r] = fr ? q.sz[r - q.ix.rvb] : q.sy[r];   // (a free row: y = 0 throughout, its z = A x is what has to be carried)
            q.sx[r] = ps.pr_xs[r] * dinv;
            q.sy[r] = 0.0;
            if (vo) q.sz[r - q.ix.rvb] = fr ? 0.0 : ps.pr_y[r];
        } else {
            int k, kind, o;
            q.dec_row(r, k, kind, o);
            const int sv = q.row_slack(k, kind, o);
            ps.pv_xs[r] = q.x[sv];
            ps.pv_y[r] = fr ? q.z[r] : q.y[r];
            q.x[sv] = ps.pr_xs[r] * dinv;
            q.y[r] = 0.0;
            q.z[r] = fr ? 0.0 : ps.pr_y[r];
        }
        if (!fr) {
            q.lo[r] = ps.pr_y[r];
            if (vo) q.hi[r - q.ix.rvb] = ps.pr_y[r];
        }
    });
    DEKF_SYNC();
}
// s := saved + correction; bounds (and z of the rows held at them) back
template <class Q>
DEKF_FN void polish_accumulate(Q& q, const PolishScratch<Q>& ps) {
    constexpr int NS = Q::NS;
    const int K = q.K, m = q.m;
    wfor_nosync(m + K * NS, [&](int e) {
        if (e >= m) {
            const int i = e - m, k = i / NS, j = i - NS * k;
            if constexpr (Q::R3) q.xb[i] = ps.pv_xb[i] + q.xb[i];
            else { const int xi = q.ix.x(k, j); q.x[xi] = ps.pv_xb[i] + q.x[xi]; }
            return;
        }
        const int r = e;
        const double lo_r = ps.pv_lo[r];
        const bool vo = r >= q.ix.rvb;
        const bool fr = polish_free_row(q, r, lo_r);  // (free rows were left alone by polish_swap_in)
        if constexpr (Q::R3) {
            q.sx[r] = ps.pv_xs[r] + q.sx[r];
            if (fr) { q.sz[r - q.ix.rvb] = ps.pv_y[r] + q.sz[r - q.ix.rvb]; q.sy[r] = 0.0; }
            else { q.sy[r] = ps.pv_y[r] + q.sy[r]; if (vo) q.sz[r - q.ix.rvb] = lo_r; }
        } else {
            int k, kind, o;
            q.dec_row(r, k, kind, o);
            const int sv = q.row_slack(k, kind, o);
            q.x[sv] = ps.pv_xs[r] + q.x[sv];
            if (fr) { q.z[r] = ps.pv_y[r] + q.z[r]; q.y[r] = 0.0; }
            else { q.y[r] = ps.pv_y[r] + q.y[r]; q.z[r] = lo_r; }
        }
        if (!fr) {
            q.lo[r] = lo_r;
            if (vo) q.hi[r - q.ix.rvb] = lo_r;
        }
    });
    DEKF_SYNC();
}

// R4: the x blocks' scaling, compact, where the solve wavefront reads it in every step (the Gauss-Jordan scratch it shares is free now)
template <class Q>
DEKF_FN void r4_fill_dxb(Q& q) {
    constexpr int NS = Q::NS, SV = 2 * NS + 3 + 3 * Q::LEGS;
    wfor(q.K * NS, [&](int e) { const int k = e / NS, j = e - NS * k; q.Dxb[e] = q.D[k * SV + j]; });
}

struct SolveInfo {
    int iters, status, rho_updates;
    double pri_res, dua_res, rho;
    int polished = 0;  // 0 polishing off or not reached, 1 the polished point was accepted, -1 rejected (the ADMM iterate stays)
    int next_fetch = 0;  // lane 0 of the workgroup: what the instance queue returned for the workgroup's NEXT instance (kernels.hip: DEKF_QUEUE_LOOP)
};

// osqp_setup + osqp_solve + extraction.  window = steps kstart .. kstart+K-1 (newest = T).
// FACTOR_LDS / PA_LDS are compile-time so that every pointer has a provable address space
// (ds_read/ds_write instead of flat_load) — see SolveLayout::factor_in_lds / pa_in_lds.
template <bool POLISH, int L, bool FACTOR_LDS, bool PA_LDS, int NFIX = 0, int FT = 0, bool R3 = false>
DEKF_FN SolveInfo solve_window_t(const DevCfg& c, const DevState& s, int b, int kstart, int K, dptr lds, dptr gws) {
    // NFIX != 0: the horizon is a compile-time constant, so every LDS array sits at a constant offset
    // (folded into the ds_read/ds_write immediates instead of living in scalar registers)
    const int NH = NFIX ? NFIX : c.N;
    constexpr int NS = 9 + 3 * L * FT, NS2 = NS * NS;
    using TM = TmpMap<NS>;
    SolveLayout lay;
    lay.init(NH, L, FT);
    Gws g;
    g.init(NH, L, FT, !FACTOR_LDS && !FT);  // (WT sits behind everything else: a kernel that does not use it sees the layout without it)
    constexpr bool R4 = R3 && FACTOR_LDS && !PA_LDS;  // four workgroups of three wavefronts per CU (SolveLayout::r4_*)
    SolveCtx<L, NFIX, FACTOR_LDS, FT, R3, POLISH, R4> q{c, s, b, K, kstart, 0, 0, IdxT<L, FT>(K)};
    q.xb = nullptr;
    q.Dxb = nullptr;
    q.WT = nullptr;
    {   // carve LDS: iterates first; xt, zt, at adjacent so PA can alias them at factor time.  Every array is handed out with its
        // extent (DEKF_SPAN: a plain pointer in the product builds, a checked one in the -DDEKF_BOUNDS build, wave.h)
        double* p = raw_of(lds);
        double* const gw = raw_of(gws);
        auto take = [&](int len) { dptr r = DEKF_SPAN(p, len); p += len; return r; };
        auto gs = [&](int off, int len) { return DEKF_SPAN(gw + off, len); };
        const int b2K = NH * NS2;
        if constexpr (R3 && !FACTOR_LDS) {
            // RR: rows in registers, run-time horizon, factor in the slab, two workgroups per CU (SolveLayout::rr_doubles)
            static_assert(!PA_LDS && NFIX == 0 && FT == 0, "RR: run-time horizon, 9 states, factor and factor-time product in the slab");
            q.R = take(9 * NH);
            q.D = take(lay.n_pad);
            q.E = take(lay.m_pad);
            q.xb = take(NS * NH);
            double* const rb = p;
            q.PA = gs(g.PA, b2K);
            q.xd = DEKF_SPAN(rb, NS * NH);
            q.at = DEKF_SPAN(rb + NS * NH, lay.m_pad);
            q.xs = DEKF_SPAN(rb + NS * NH + lay.m_pad, NS * NH);
            q.gb = DEKF_SPAN(rb + 2 * NS * NH + lay.m_pad, 3 * NH);
            q.sx = q.at;
            p += 2 * NS * NH + lay.m_pad + 3 * NH;
            q.tmp = take(TM::LEN);
            q.Sinv = gs(g.Sinv, b2K);
            q.WT = gs(g.WT, b2K);
            q.Wk = gs(g.Wk, b2K);
            q.Sf = nullptr; q.Wf = nullptr;
            q.x = gs(g.x, lay.n_pad);  // (the P column norms of the Ruiz passes)
            q.sy = gs(g.y, lay.m_pad); q.sz = gs(g.z, lay.m_pad);
            q.y = nullptr; q.z = nullptr; q.zt = nullptr; q.cf = nullptr; q.xt = nullptr;
            q.lo = gs(g.lo, lay.m_pad); q.hi = gs(g.hi, lay.m_pad);
            q.cold = true;
            q.Sv = gs(g.Sv, NH * 6 * L); q.Sw = gs(g.Sw, NH * SWS); q.Sc = gs(g.Sc, NH * 6);
            q.Wm = gs(g.Wm, NH * 6 * L); q.Wd = gs(g.Wd, NH * 24); q.Wc = gs(g.Wc, NH * 6);
        }
        if constexpr (R4) {
            static_assert(!R4 || (NFIX != 0 && FT == 0), "R4: fixed horizon, 9 states, factor in LDS, everything the rare phases read in the slab");
            q.R = take(9 * NH);
            q.xb = take(NS * NH);
            q.xd = take(NS * NH);
            q.at = take(lay.m_pad);
            q.xs = take(NS * NH);
            q.gb = take(3 * NH);
            (void)take(lay.r4_pad());
            q.Dxb = DEKF_SPAN(p + SolveLayout::R4_DXB_IN_TMP - NS * NH, NS * NH);  // ends at tmp + 96 (pad | Gauss-Jordan scratch of side 0)
            q.tmp = take(TM::LEN);
            q.Sinv = take(b2K);
            q.Wk = take(b2K);
            q.D = gs(g.D, lay.n_pad); q.E = gs(g.E, lay.m_pad);  // (in LDS during the Ruiz passes, below)
            q.PA = gs(g.PA, b2K);
            q.sx = q.at;
            q.sy = gs(g.y, lay.m_pad); q.sz = gs(g.z, lay.m_pad);
            q.Sf = nullptr; q.Wf = nullptr;
            q.x = nullptr; q.y = nullptr; q.z = nullptr; q.zt = nullptr; q.cf = nullptr; q.xt = nullptr;
            q.lo = gs(g.lo, lay.m_pad); q.hi = gs(g.hi, lay.m_pad);
            q.cold = true;
            q.Sv = gs(g.Sv, NH * 6 * L); q.Sw = gs(g.Sw, NH * SWS); q.Sc = gs(g.Sc, NH * 6);
            q.Wm = gs(g.Wm, NH * 6 * L); q.Wd = gs(g.Wd, NH * 24); q.Wc = gs(g.Wc, NH * 6);
        }
        if constexpr (R3 && FACTOR_LDS && !R4) {
            static_assert(!R3 || !FACTOR_LDS || R4 || (PA_LDS && NFIX != 0 && FT == 0), "R3: fixed horizon, 9 states, factor in LDS");
            q.R = take(9 * NH);
            q.D = take(lay.n_pad);
            q.E = take(lay.m_pad);
            q.xb = take(NS * NH);
            double* const rb = p;  // everything in this region is dead while a factorisation runs
            q.PA = DEKF_R3_PA_LDS ? DEKF_SPAN(rb, lay.r3_pa_region()) : gs(g.PA, b2K);
            q.xd = DEKF_SPAN(rb, NS * NH);
            q.at = DEKF_SPAN(rb + NS * NH, lay.m_pad);
            q.xs = DEKF_SPAN(rb + NS * NH + lay.m_pad, NS * NH);
            q.gb = DEKF_SPAN(rb + 2 * NS * NH + lay.m_pad, 3 * NH);
            q.sx = q.at;
            q.sy = DEKF_SPAN(rb + 2 * NS * NH + lay.m_pad + 3 * NH, lay.m_pad);
            q.sz = DEKF_SPAN(rb + 2 * NS * NH + 2 * lay.m_pad + 3 * NH, 3 * NH);
            p += lay.r3_pa_region();
            q.tmp = take(TM::LEN);
            q.Sinv = take(b2K);
            q.Wk = take(b2K);
            q.Sf = nullptr; q.Wf = nullptr;
            // The row state is in registers during a chunk of iterations and in LDS (sx, sy, sz) between chunks; the slab holds it
            // only across a refactorisation (whose temporaries need the LDS region): q.x <- sx, q.y <- sy, q.z <- sz.  What stays in
            // the slab for the whole solve are constants: the scaled bounds (written once, at the cold start) and the slack-block
            // inverses (once per factorisation) — 13 KB per workgroup.
            q.x = gs(g.x, lay.n_pad); q.z = gs(g.z, lay.m_pad); q.y = gs(g.y, lay.m_pad); q.zt = nullptr; q.cf = nullptr;
            q.xt = nullptr;
            q.lo = gs(g.lo, lay.m_pad); q.hi = gs(g.hi, lay.m_pad);
            q.cold = true;
            q.Sv = gs(g.Sv, NH * 6 * L); q.Sw = gs(g.Sw, NH * SWS); q.Sc = gs(g.Sc, NH * 6);
            q.Wm = gs(g.Wm, NH * 6 * L); q.Wd = gs(g.Wd, NH * 24); q.Wc = gs(g.Wc, NH * 6);
        }
        if constexpr (!R3) {
        q.x = take(lay.n_pad);
        q.z = take(lay.m_pad);
        q.y = take(lay.m_pad);
        double* const xtb = p;  // xt | zt | at: the factor-time scratch (PA, the staging panels of factor_blocks_generic) spans all three
        q.xt = DEKF_SPAN(xtb, lay.n_pad + 2 * lay.m_pad);
        q.cf = DEKF_SPAN(xtb, lay.m_pad);
        q.gb = DEKF_SPAN(xtb + lay.m_pad, lay.n_pad - lay.m_pad);  // 3 (N - 1) <= n_pad - m_pad = NS N
        p += lay.n_pad;
        q.zt = take(lay.m_pad);
        q.at = take(lay.m_pad);
        q.xs = take(NS * NH);
        q.xd = take(NS * NH);
        q.tmp = take(TM::LEN);
        q.Sf = nullptr;
        q.Wf = nullptr;
        if constexpr (FACTOR_LDS) {
            q.D = take(lay.n_pad);
            q.E = take(lay.m_pad);
            q.lo = take(lay.m_pad);
            q.hi = take(3 * NH);
            q.Sv = take(NH * 6 * L);
            q.Sw = take(NH * SWS);
            q.Sc = take(NH * 6);
            if constexpr (FT) q.Sf = take(NH * 6 * L);
            q.Sinv = take(b2K);
            q.Wk = take(b2K);
            q.R = take(NH * 9);
        } else {
            q.D = gs(g.D, lay.n_pad); q.E = gs(g.E, lay.m_pad); q.lo = gs(g.lo, lay.m_pad); q.hi = gs(g.hi, lay.m_pad);
            q.Sv = gs(g.Sv, NH * 6 * L); q.Sw = gs(g.Sw, NH * SWS); q.Sc = gs(g.Sc, NH * 6);
            if constexpr (FT) q.Sf = gs(g.Sf, NH * 6 * L);
            q.Sinv = gs(g.Sinv, b2K); q.Wk = gs(g.Wk, b2K); q.R = gs(g.rho, lay.m_pad);  // rho slot is unused: R (9K <= m_pad)
            if constexpr (!FT) q.WT = gs(g.WT, b2K);
            if (lay.gg_consts_in_lds()) {  // (run-time placement: these five become generic pointers in this instantiation)
                q.D = take(lay.n_pad);
                q.E = take(lay.m_pad);
                q.lo = take(lay.m_pad);
                q.hi = take(3 * NH);
                q.R = take(9 * NH);
            }
        }
        q.Wm = gs(g.Wm, NH * 6 * L); q.Wd = gs(g.Wd, NH * 24); q.Wc = gs(g.Wc, NH * 6);
        if constexpr (FT) q.Wf = gs(g.Wf, NH * 6 * L);
        if constexpr (PA_LDS) q.PA = q.xt;
        else q.PA = gs(g.PA, b2K);
        }
    }
    if constexpr (FACTOR_LDS) {
        // The factor-time row weights W_m, W_d, W_c [, W_f] (written by step 3a, read by 3b-3c of solve_factor) live in the W array,
        // which stays dead until the block LDL' (3d) writes it; the staged P blocks they are computed from sit in front of it, inside
        // S^-1.  (They used to make a round trip through the HBM slab per factorisation: most of the kernel's excess traffic.)
        constexpr int PSL = 6 * L + 27 + 6 * L * FT, WLEN = 6 * L + 30 + 6 * L * FT;
        if (NH * PSL + NS * (NS + 1) / 2 <= NH * NS2 && NH * WLEN <= NH * NS2) {
            double* const wb = raw_of(q.Wk);
            q.Wm = DEKF_SPAN(wb, NH * 6 * L);
            q.Wd = DEKF_SPAN(wb + NH * 6 * L, NH * 24);
            q.Wc = DEKF_SPAN(wb + NH * 6 * L + NH * 24, NH * 6);
            if constexpr (FT) q.Wf = DEKF_SPAN(wb + NH * 6 * L + NH * 30, NH * 6 * L);
        }
    }
    q.pol = DEKF_SPAN(raw_of(gws) + g.pol, 2 * NH * NS + 5 * lay.m_pad);
    q.n = (K - 1) * IdxT<L, FT>::SV + NS + IdxT<L, FT>::nm;
    q.m = (K - 1) * IdxT<L, FT>::SC + IdxT<L, FT>::nm;
    {
        const double* sn = s.snap + (size_t)c.snap_len * b;
        q.Mp = DEKF_CSPAN(sn, NS2);
        q.np = DEKF_CSPAN(sn + NS2, NS);
        q.vo = DEKF_CSPAN(sn + NS2 + NS, 4 * c.wcap);
    }
    q.cc = 1.0;
    q.lane0 = DEKF_LANE() & 63;
    q.sigma_pol = c.delta;
    q.zlo = false;
    q.Pst = DEKF_SPAN(raw_of(q.Sinv), 2 * NH * NS2);  // Sinv | Wk are adjacent in both placements and dead until a factorisation writes them
    q.staged = false;
    const int n = q.n, m = q.m;
    const auto& ix = q.ix;
    SolveInfo info{0, DEKF_SOLVE_MAX_ITER, 0, 0.0, 0.0, c.rho0};
#if defined(DEKF_BOUNDS) && DEKF_DEVICE_BUILD
    const unsigned long long bounds_hits0 = *(volatile unsigned long long*)&dekf_bounds_hits[0];
#endif

    q.prof = DEKF_SPAN(s.prof + DEKF_PROF_SLOTS * (size_t)b, DEKF_PROF_SLOTS);
#if defined(DEKF_PROFILE) && DEKF_DEVICE_BUILD
    q.prof_last = clock64();
    const long long prof_t0 = q.prof_last;
    const long long prof_w0 = wall_clock64();  // (constant 100 MHz: slot 23 / slot 13 gives the shader clock the solve really ran at)
    if (DEKF_LANE() == 0)
        for (int i = 0; i < DEKF_PROF_SLOTS; ++i) q.prof[i] = 0.0;
#endif
    wfor(K * 9, [&](int e) { q.R[e] = ld_stream(q.rec(e / 9), Rec::R + e % 9); });
    if constexpr (R3 && !FACTOR_LDS) {
        // RR: D and E in LDS for the whole solve; the passes' temporaries Dn | En overlay xb .. gb (dead until the cold start), the P
        // column norms pc live in the slab (q.x)
        double* const ob = raw_of(q.xb);
        q.xt = DEKF_SPAN(ob, lay.n_pad);                 // Dn
        q.zt = DEKF_SPAN(ob + lay.n_pad, lay.m_pad);     // En
        if (c.scaling > 0) solve_scale(q);
        else { wfor(n + m, [&](int e) { if (e < n) q.D[e] = 1.0; else q.E[e - n] = 1.0; }); }
        q.zt = nullptr; q.xt = nullptr;
        wfor(K * NS, [&](int e) { q.xb[e] = 0.0; });
    } else if constexpr (R3) {
        // D and E stay in LDS for the whole solve (R4: for the Ruiz passes, in xb .. pad, and go to the slab behind them); the Ruiz
        // passes' temporaries pc, En, Dn sit behind the staged P blocks inside S^-1 | W, which are not live before the first factorisation.
        constexpr int PSL = NFIX * (6 * L + 27) + NS * (NS + 1) / 2;
        dptr xg = q.x, ztg = nullptr;
        double* const sb = raw_of(q.Sinv);
        q.x = DEKF_SPAN(sb + PSL, lay.n_pad);                            // pc
        q.zt = DEKF_SPAN(sb + PSL + lay.n_pad, lay.m_pad);               // En
        q.xt = DEKF_SPAN(sb + PSL + lay.n_pad + lay.m_pad, lay.n_pad);   // Dn
        dptr Dg = q.D, Eg = q.E;
        if constexpr (R4) {
            q.D = DEKF_SPAN(raw_of(q.xb), lay.n_pad);
            q.E = DEKF_SPAN(raw_of(q.xb) + lay.n_pad, lay.m_pad);
        }
        if (c.scaling > 0) solve_scale(q);
        else { wfor(n + m, [&](int e) { if (e < n) q.D[e] = 1.0; else q.E[e - n] = 1.0; }); }
        if constexpr (R4) {
            cdptr Dl = q.D, El = q.E;
            wfor(n + m, [&](int e) { if (e < n) Dg[e] = Dl[e]; else Eg[e - n] = El[e - n]; });
            q.D = Dg; q.E = Eg;
        }
        q.x = xg; q.zt = ztg; q.xt = nullptr;
        wfor(K * NS, [&](int e) { q.xb[e] = 0.0; });
    } else {
    if (c.scaling > 0) solve_scale(q);
    else { wfor(n + m, [&](int e) { if (e < n) q.D[e] = 1.0; else q.E[e - n] = 1.0; }); }
    }
    DEKF_PROF_MARK(q, 0);
    q.rho = uni_pin(c.rho0c);
    // scaled bounds, cold start
    dptr x = q.x, z = q.z, y = q.y, at = q.at;
    wfor(R3 ? m : n + m, [&](int e0) {
        const int e = R3 ? e0 + n : e0;  // (R3: only the scaled bounds; the first chunk's row-block load is the cold start x = z = y = 0)
        if (e < n) { x[e] = 0.0; return; }
        int r = e - n, k, kind, o;
        q.dec_row(r, k, kind, o);
        double lb, ub;
        q.bounds(k, kind, o, lb, ub);
        q.lo[r] = lb * q.E[r];
        if (kind == 2) q.hi[r - q.ix.rvb] = ub * q.E[r];
        if constexpr (!R3) {
            z[r] = 0.0;
            y[r] = 0.0;
            at[r] = 0.0;  // u = rho z - y of the cold start
        }
    });
    DEKF_PROF_MARK(q, 19);
    bool ok = solve_factor(q);
    if constexpr (R4) r4_fill_dxb(q);
    // scaled linear cost on x_0 (LDS copy for the per-lane look-ups)
    wfor(NS, [&](int j) { q.tmp[TM::QSL + j] = q.cc * q.D[ix.x(0, j)] * q.np[j]; });
    const double sigma = c.sigma, alpha = c.alpha;
    const double cinv = uni_pin(1.0 / q.cc);
    const double eps_rel_cinv = uni_pin(c.eps_rel * cinv);  // (the compiler forms and hoists this product anyway: into a VGPR pair that lives for the whole solve)
    int iter = 0;
    bool done = false;
#if DEKF_DEVICE_BUILD && DEKF_QUEUE_MODE != 0
    // The workgroup's next instance, fetched HERE: the first wavefront runs nothing but LDS for the next 25 iterations, so the
    // atomic's round trip (microseconds when every workgroup of a lock-step round fetches at once) is waited for by nobody
    if (DEKF_LANE() == 0) info.next_fetch = atomicAdd(s.queue, 1);
#endif
    if constexpr (!R3) {  // (R3: the first chunk's load of the row blocks is the restart)
    if (ok) phase_rows<true>(q, alpha, sigma);  // cold start: cf, t = 0, w = 0 (the factorisation scratch aliased xt | zt | at)
    }
    DEKF_PROF_MARK(q, 1);
    while (ok && !done && iter < c.max_iter) {
#if DEKF_DEVICE_BUILD
        if constexpr (R3) {
            // every iteration up to the next event (termination check, rho adaptation, iteration cap) in one call: the row
            // phase's state stays in registers in between (mhe_admm_core.h: admm_chunk_r3)
            int nxt = c.max_iter;
            if (c.check_termination > 0) { const int e = (iter / c.check_termination + 1) * c.check_termination; nxt = e < nxt ? e : nxt; }
            if (c.adaptive_rho && c.adaptive_rho_interval > 0) { const int e = (iter / c.adaptive_rho_interval + 1) * c.adaptive_rho_interval; nxt = e < nxt ? e : nxt; }
            if constexpr (R4) admm_chunk_r4<NFIX>(q, nxt - iter, alpha, sigma);
            else if constexpr (FACTOR_LDS) admm_chunk_r3<NFIX>(q, nxt - iter, alpha, sigma);
            else admm_chunk_rr(q, nxt - iter, alpha, sigma);
            iter = nxt;
            DEKF_PROF_MARK(q, 9);
        } else
#endif
        {
        ++iter;
        phase_xcols(q, sigma);
        DEKF_PROF_MARK(q, 2);
        phase_sweeps_rows(q, alpha, sigma);
        DEKF_PROF_MARK(q, 9);
        }
        bool can_check = c.check_termination > 0 && (iter % c.check_termination == 0);
        bool adapt_now = c.adaptive_rho && c.adaptive_rho_interval > 0 && (iter % c.adaptive_rho_interval == 0);
        if (can_check || adapt_now || iter == c.max_iter) {
            double ra[6], va[8];
            residual_norms(q, ra, va);
#if defined(DEKF_PROFILE) && defined(DEKF_PROFILE_RESID)  // debugging aid: the 14 norms of the FIRST check into the stamp slots 16..29
            if (DEKF_LANE() == 0 && iter <= c.check_termination) {
                for (int i_ = 0; i_ < 6; ++i_) q.prof[16 + i_] = ra[i_];
                for (int i_ = 0; i_ < 8; ++i_) q.prof[22 + i_] = va[i_];
            }
#endif
            info.pri_res = uni_pin(ra[0]);
            info.dua_res = uni_pin(cinv * va[0]);
            if (can_check || iter == c.max_iter) {
                double eps_pri = c.eps_abs + c.eps_rel * dmax(ra[1], ra[2]);
                double eps_dua = c.eps_abs + eps_rel_cinv * dmax(va[1], dmax(va[2], va[3]));
                if (info.pri_res < eps_pri && info.dua_res < eps_dua) { info.status = DEKF_SOLVE_OK; done = true; }
            }
            DEKF_PROF_MARK(q, 10);
            if (!done && adapt_now) {
                double pr = ra[3] / (dmax(ra[4], ra[5]) + 1e-10);
                double du = va[4] / (dmax(va[5], dmax(va[6], va[7])) + 1e-10);
                double rho_new = dmin(dmax(q.rho * sqrt(pr / (du + 1e-10)), RHO_MIN), RHO_MAX);
                if (rho_new > q.rho * c.adaptive_rho_tolerance || rho_new < q.rho / c.adaptive_rho_tolerance) {
                    q.rho = uni_pin(rho_new);
                    info.rho_updates++;
                    DEKF_SYNC();
                    if constexpr (R3 && FACTOR_LDS && !R4) {  // the factorisation's temporaries take the LDS the row state rests in: over to the slab and back
                        // (RR, R4: the factor-time product lives in the slab, nothing of the factorisation touches the stash)
                        const int mp = lay.m_pad, nz = 3 * NH;
                        wfor(2 * mp + nz, [&](int e) {
                            if (e < mp) q.x[e] = q.sx[e];
                            else if (e < 2 * mp) q.y[e - mp] = q.sy[e - mp];
                            else q.z[e - 2 * mp] = q.sz[e - 2 * mp];
                        });
                    }
                    ok = solve_factor(q);
                    if constexpr (R4) r4_fill_dxb(q);
                    if constexpr (R3 && FACTOR_LDS && !R4) {
                        const int mp = lay.m_pad, nz = 3 * NH;
                        wfor(2 * mp + nz, [&](int e) {
                            if (e < mp) q.sx[e] = q.x[e];
                            else if (e < 2 * mp) q.sy[e - mp] = q.y[e - mp];
                            else q.sz[e - 2 * mp] = q.z[e - 2 * mp];
                        });
                    }
                    if constexpr (!R3) {
                    if (ok) phase_rows<true>(q, alpha, sigma);  // cf, t, w for the new rho (the scratch aliased xt | zt | at)
                    }
                    DEKF_PROF_MARK(q, 11);
                }
            }
        }
    }
    info.iters = iter;
    info.rho = q.rho;
    // store_solution + update() tail: x_T = D x ; v_b = R (x_T[3:6] + gyro x p_imu_2_opti)
    cdptr rT = q.rec(K - 1);
    double xT[NS];
    bool finite = ok;
    for (int j = 0; j < NS; ++j) {
        xT[j] = q.D[ix.x(K - 1, j)] * (R3 ? q.xb[NS * (K - 1) + j] : x[ix.x(K - 1, j)]);
        if (!(fabs(xT[j]) <= 1e300)) finite = false;
    }
    // ---- solution polishing (osqp.polish, OSQP paper sec. 4 / polish.c; the node's declared default, EstSub.cpp:188).
    // OSQP guesses the active set from the dual iterate, solves the equality-constrained QP on it through the regularised system
    //     [P + delta I, A_a'; A_a, -delta I] [x; y_a] = [-q; b_a]
    // and refines the result polish_refine_iter times against the unregularised matrix:  K_reg s+ = rhs + delta [x; -y].  That
    // recursion IS the ADMM step of this solver with sigma = delta, 1 / rho = delta on the active rows, alpha = 1 and z held at the
    // bound b — started from x = y = 0 (first solve: K_reg s = rhs).  Every Meas / Dyn row and every VO row with a bound is an
    // equality, hence active whatever the sign of its multiplier; a VO row without bounds stays a free row (rho = RHO_MIN; OSQP
    // leaves it out: the difference is a proximal term of 1e-6 against weights of 1e9).  So: one more factorisation, 1 +
    // polish_refine_iter iterations of the kernels above, the residuals of the result, and OSQP's acceptance test.
    if constexpr (POLISH) if (c.polish && ok && finite && info.status == DEKF_SOLVE_OK) {
        const double pri0 = info.pri_res, dua0 = info.dua_res, rho_keep = q.rho;
        DEKF_SYNC();
        q.rho = uni_pin(dmin(dmax(1.0 / (c.delta * RHO_EQ_OVER_RHO_INEQ), RHO_MIN), RHO_MAX));  // rho of an equality row: 1 / delta
        q.zlo = true;
        if constexpr (R3) {
            q.cold = true;
            wfor(K * NS, [&](int e) { q.xb[e] = 0.0; });
        } else {
            wfor(n + m, [&](int e) {
                if (e < n) { x[e] = 0.0; return; }
                const int r = e - n;
                const bool free_row = r >= q.ix.rvb && q.rho_of(q.lo[r], q.hi[r - q.ix.rvb]) == RHO_MIN;
                z[r] = free_row ? 0.0 : q.lo[r];
                y[r] = 0.0;
            });
        }
        bool okp = solve_factor(q);
        if constexpr (R4) r4_fill_dxb(q);
        if constexpr (!R3) {
            if (okp) phase_rows<true>(q, 1.0, q.sigma());
        }
        const int nref = c.polish_refine_iter > 0 ? c.polish_refine_iter : 0;
        if (okp) {
            polish_step<NFIX>(q);  // K_reg s = rhs from the cold start
            double ra[6], va[8];
            const PolishScratch<decltype(q)> ps(q, lay.m_pad, NH);
            const ResidVec rvec{ps.pr_xb, ps.pr_xs, ps.pr_y};
            for (int it = 0; it < nref; ++it) {
                residual_norms(q, ra, va, &rvec);  // r = rhs - K s (the norms are not used here)
                DEKF_SYNC();
                polish_swap_in(q, ps, c.delta);
                wfor(NS, [&](int j) { q.tmp[TM::QSL + j] = 0.0; });
                if constexpr (!R3) phase_rows<true>(q, 1.0, q.sigma());
                polish_step<NFIX>(q);  // K_reg ds = r
                polish_accumulate(q, ps);
                wfor(NS, [&](int j) { q.tmp[TM::QSL + j] = q.cc * q.D[ix.x(0, j)] * q.np[j]; });
            }
            residual_norms(q, ra, va);
            const double prp = ra[0], dup = cinv * va[0];
            // OSQP's acceptance test (polish.c) with its third clause made SYMMETRIC.  OSQP keeps a polished point on
            //     pol_dua < dua && pri < 1e-10                     (the iterate's primal residual is numerically zero: compare the dual only)
            // whatever the polished PRIMAL residual is.  This solver's iterates reach pri < 1e-10 where OSQP's do not (dense VO rows, adaptive
            // rho past 1e5: the generic sparse LDL' leaves pri_res ~ 1e-6, the structured solve 1e-11 — found by tools/fuzz_parity.py,
            // round 5), and the regularised polishing system does not converge in three refinement steps against weights of 4.4e9: the clause
            // then let a point with pri_res 8.6e-4 (x off by 0.11) — or, on a marginal dual comparison, 3e-7 (x off by 2e-5) — replace an
            // iterate at 1e-9 of the optimum, where OSQP, its own iterate less accurate, rejects the same point.  Here the clause also asks
            // the polished primal residual to be numerically zero: a polished point never trades a primal residual of 1e-11 for one of 1e-7.
            // (dekf_params.polish_accept_osqp = 1 restores polish.c's clause verbatim, for strict OSQP parity)
            const bool good = (prp < pri0 && dup < dua0) || (prp < pri0 && dua0 < 1e-10) || (dup < dua0 && pri0 < 1e-10 && (c.polish_accept_osqp || prp < 1e-10));
            double xP[NS];
            bool finp = true;
            for (int j = 0; j < NS; ++j) {
                xP[j] = q.D[ix.x(K - 1, j)] * (R3 ? q.xb[NS * (K - 1) + j] : x[ix.x(K - 1, j)]);
                if (!(fabs(xP[j]) <= 1e300)) finp = false;
            }
            if (good && finp) {  // status_polish = 1: the polished point replaces the iterate
                for (int j = 0; j < NS; ++j) xT[j] = xP[j];
                info.pri_res = prp;
                info.dua_res = dup;
                info.polished = 1;
            } else {
                info.polished = -1;
            }
        } else {
            info.polished = -1;
        }
        q.rho = rho_keep;
        q.zlo = false;
    }
    if (!finite) info.status = DEKF_SOLVE_NUMERIC;
#if defined(DEKF_BOUNDS) && DEKF_DEVICE_BUILD
    // an out-of-range dereference while this instance was being solved (by this or a concurrent workgroup: the counter is global)
    if (*(volatile unsigned long long*)&dekf_bounds_hits[0] != bounds_hits0) info.status = DEKF_SOLVE_NUMERIC;
#endif
    if (DEKF_LANE() == 0) {
        const double p_opti[3] = {0.016041, 0.089061, 0.0579875};
        double wxp[3], t[3], vb[3];
        double gy[3], Rl[9];
        for (int a = 0; a < 3; ++a) gy[a] = rT[Rec::GY + a];
        for (int a = 0; a < 9; ++a) Rl[a] = rT[Rec::R + a];
        cross3(gy, p_opti, wxp);
        for (int a = 0; a < 3; ++a) t[a] = xT[3 + a] + wxp[a];
        mv3(Rl, t, vb);
        for (int j = 0; j < NS; ++j) s.x_mhe[NS * (size_t)b + j] = xT[j];
        for (int a = 0; a < 3; ++a) s.v_b[3 * (size_t)b + a] = vb[a];
        s.status[b] = info.status;
        s.iters[b] = info.iters;
        s.rho_updates[b] = info.rho_updates;
        s.polish_status[b] = info.polished;
        s.pri_res[b] = info.pri_res;
        s.dua_res[b] = info.dua_res;
    }
    DEKF_SYNC();
#if defined(DEKF_PROFILE) && DEKF_DEVICE_BUILD
    DEKF_PROF_MARK(q, 12);
    if (DEKF_LANE() == 0) { q.prof[13] = (double)(clock64() - prof_t0); q.prof[23] = (double)(wall_clock64() - prof_w0); }
#endif
    return info;
}
// the kernels without the polishing step (osqp.polish false: parameters_go1.yaml:44 — the benchmark configuration)
template <int L, bool FACTOR_LDS, bool PA_LDS, int NFIX = 0, int FT = 0, bool R3 = false>
DEKF_FN SolveInfo solve_window(const DevCfg& c, const DevState& s, int b, int kstart, int K, dptr lds, dptr gws) {
    return solve_window_t<false, L, FACTOR_LDS, PA_LDS, NFIX, FT, R3>(c, s, b, kstart, K, lds, gws);
}

}  // namespace dekf
