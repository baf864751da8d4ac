// kernels.hip — gfx950 kernel entry points of the estimator hot path.
//   k_ekf_tick        one lane per instance              (ekf_core.h)
//   k_mhe_initialize  one wavefront per instance         (mhe_assemble_core.h)
//   k_mhe_assemble    one wavefront per instance         (mhe_assemble_core.h)
//   k_mhe_solve_*     persistent four-wavefront workgroups, grid-stride over instances; each
//                     workgroup owns one scratch slab in HBM, so a slab is only ever touched
//                     from one XCD (its L2 is the only one that caches it)
//                     (mhe_solve_core.h, mhe_admm_core.h)
//   k_kf_*            KF alternative                      (kf_core.h)
//   k_latch_vo        masked VO latch (robotSub::vo_callback for a batch)
//   k_latch4          device-to-device sensor latch of one push (IMU or leg arrays) in one launch
#include <hip/hip_runtime.h>

#include "cfg.h"
#include "ekf_core.h"
#include "go1_kin.h"
#include "kf_core.h"
#include "mhe_assemble_core.h"
#include "mhe_solve_core.h"

using namespace dekf;

#ifdef DEKF_BOUNDS
namespace dekf {
__device__ unsigned long long dekf_bounds_hits[4] = {0ull, 0ull, 0ull, 0ull};  // wave.h: BPtr
}
// proves that the checker fires: one read and one write just past a 16-element array (both are counted and redirected)
extern "C" __global__ void k_bounds_selftest(double* a) {
    dekf::dptr p = DEKF_SPAN(a, 16);
    const double v = p[16];
    (p + 3)[13] = v + 1.0;
}
#endif

// The solve kernels' instance queue (round 6).  A persistent workgroup takes its next instance from a device-side counter instead of
// striding over the batch by the grid size: instances whose solves take different numbers of ADMM iterations (a fleet that is not in
// lock-step) balance themselves, and a workgroup that the hardware could not place next to the others (four per CU fit on 96-99 % of
// the CUs, tools/probes/r4_residency_probe.hip) finds the queue empty when it finally starts instead of holding the launch for a
// whole round of its own.  queue[0]: next instance; queue[1]: workgroups that have left — the last one resets both, so the counter
// needs nothing from the host and nothing between launches (every workgroup's final fetch has returned before it counts itself out).
__device__ __forceinline__ void solve_queue_leave(const dekf::DevState& s) {
#if DEKF_QUEUE_MODE == 0
    return;
#endif
    if (threadIdx.x == 0 && atomicAdd(s.queue + 1, 1) == (int)gridDim.x - 1) {
        s.queue[0] = 0;
        s.queue[1] = 0;
    }
}

// Two forms of the loop.  STATIC FIRST (every kernel whose grid is placed at once — all of them but the four-per-CU kernels): the first
// instance of a workgroup is its own index (no stampede of gridDim.x atomics on one address when a launch starts); the counter counts
// instances beyond the first gridDim.x.  DYNAMIC (the four-per-CU kernels, whose last workgroups the hardware places late on 2-9 % of
// the CUs, profiles/r06_go1_four_per_cu_3waves.txt): every instance comes from the counter, so a workgroup that starts late holds no
// instance hostage.  In both the fetch of the NEXT instance is issued from INSIDE the current solve (solve_window_t, in front of the
// first ADMM iterations: SolveInfo::next_fetch) — in lock-step every workgroup of a round fetches at the same moment, the atomics on
// one address serialise (about 4 us for 768 of them), and a wavefront's memory operations return in order: issued where the
// wavefront runs LDS-only iterations for the next 60 us, nobody waits for it.
// Measured (Go1, 4096): a lock-step fleet pays 0.2 % for the queue (1.8558 -> 1.860 ms per launch), a mixed fleet (cameras at 5-50 Hz,
// every tenth robot blind: 21 % of the solves stop at 50 iterations, 78 % at 75) gains 4.2 % (1.92 -> 1.84 ms)
// (profiles/r06_fleet_not_in_lock_step.txt).  -DDEKF_QUEUE_MODE=0 (cfg.h): the static grid-stride of rounds 1-5, for A/B builds.
#if DEKF_QUEUE_MODE == 0
#define DEKF_QUEUE_LOOP(...) \
        for (int b = blockIdx.x; b < c.B; b += gridDim.x) { (void)(__VA_ARGS__); DEKF_WG_TRACE_COUNT; }
#define DEKF_QUEUE_LOOP_DYNAMIC(...) DEKF_QUEUE_LOOP(__VA_ARGS__)
#else
#define DEKF_QUEUE_LOOP(...)                                                                         \
        for (int b = blockIdx.x; b < c.B;) {                                                         \
            const dekf::SolveInfo si_ = (__VA_ARGS__);                                               \
            DEKF_WG_TRACE_COUNT;                                                                     \
            if (threadIdx.x == 0) next_instance = si_.next_fetch + (int)gridDim.x;                   \
            __syncthreads();                                                                         \
            b = next_instance;                                                                       \
            __syncthreads();                                                                         \
        }
#define DEKF_QUEUE_LOOP_DYNAMIC(...)                                                                 \
        int nb_ = 0;                                                                                 \
        if (threadIdx.x == 0) nb_ = atomicAdd(s.queue, 1);                                           \
        for (;;) {                                                                                   \
            if (threadIdx.x == 0) next_instance = nb_;                                               \
            __syncthreads();                                                                         \
            const int b = next_instance;                                                             \
            if (b >= c.B) break;                                                                     \
            const dekf::SolveInfo si_ = (__VA_ARGS__);                                               \
            DEKF_WG_TRACE_COUNT;                                                                     \
            nb_ = si_.next_fetch;                                                                    \
        }
#endif

// A/B builds only (-DDEKF_AB_KNOBS, tools/probes/r4_wg_trace.py): when each persistent workgroup started and left (100 MHz wall clock),
// how many instances it solved and where it ran, into the section-stamp buffer (slots 0..3 of row blockIdx.x)
#ifdef DEKF_AB_KNOBS
#define DEKF_WG_TRACE_BEGIN const long long wg_t0_ = wall_clock64(); int wg_n_ = 0;
#define DEKF_WG_TRACE_COUNT ++wg_n_
#define DEKF_WG_TRACE_END                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < c.B) {                                                                     \
        double* o_ = s.prof + (size_t)blockIdx.x * dekf::DEKF_PROF_SLOTS;                                           \
        o_[0] = (double)wg_t0_; o_[1] = (double)wall_clock64(); o_[2] = (double)wg_n_;                              \
        o_[3] = (double)((__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf) * 65536u + (__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) & 0xffffu)); \
    }
#else
#define DEKF_WG_TRACE_BEGIN
#define DEKF_WG_TRACE_COUNT (void)0
#define DEKF_WG_TRACE_END
#endif

extern "C" {

// -DDEKF_KSET_ONLY (the parallel product build, build.sh): this translation unit carries ONLY the kernels of its mask — no stubs
// for the others (another unit defines them) — and the kernels that are not solves only with bit 1024.
#if !defined(DEKF_KSET_ONLY) || (defined(DEKF_KSET) && (DEKF_KSET & 1024))
#define DEKF_MISC_KERNELS 1
#else
#define DEKF_MISC_KERNELS 0
#endif
#if DEKF_MISC_KERNELS
__global__ void __launch_bounds__(64) k_ekf_tick(DevCfg c, DevState s, int count) {
    int b = blockIdx.x * 64 + threadIdx.x;
    if (b < c.B) ekf_tick(c, s, b, count);
}

__global__ void __launch_bounds__(64) k_mhe_initialize(DevCfg c, DevState s) {
    extern __shared__ double lds[];
    assemble_initialize(c, s, blockIdx.x, lds);
    if (threadIdx.x == 0) { s.status[blockIdx.x] = DEKF_SOLVE_NONE; s.iters[blockIdx.x] = 0; }
}

// three wavefronts per SIMD (168 VGPRs, 10 spilled): 0.071 ms per step of 4096 instances against 0.076 ms with the 182 registers and two
// wavefronts the compiler takes unasked; four (128 VGPRs, 290 spilled): 0.074 ms
#ifndef DEKF_ASM_WAVES
#define DEKF_ASM_WAVES 3
#endif
__global__ void __launch_bounds__(64, DEKF_ASM_WAVES) k_mhe_assemble(DevCfg c, DevState s, int T, int pushes) {
    extern __shared__ double lds[];
    assemble_update(c, s, blockIdx.x, T, pushes, lds);
}
// the arrival cost of step T ahead of time (mhe_assemble_core.h: marginalize_early; launched on a second stream behind the assemble
// of step T - 1, so that it runs in the slots the solve of step T - 1 frees in its last round)
__global__ void __launch_bounds__(64, DEKF_ASM_WAVES) k_mhe_marginalize_early(DevCfg c, DevState s, int T) {
    extern __shared__ double lds[];
    marginalize_early(c, s, blockIdx.x, T, lds);
}
#endif  // DEKF_MISC_KERNELS

// three placements of the factor (mhe_solve_core.h: SolveLayout): _ll all in LDS (Go1, N = 20),
// _lg LDS factor with the factor-time temporary in HBM (fewer legs), _gg factor streamed from HBM
// One workgroup of DEKF_SOLVE_THREADS lanes (4 wavefronts, one per SIMD of the CU) per instance;
// the leg count is a compile-time constant of each instantiation.  The second launch bound (2 waves
// per SIMD) caps VGPR+AGPR at 256 so that TWO workgroups stay resident per CU — LDS allows exactly
// two, and at 260 registers the kernel silently dropped to one (2x slower).
// DEKF_SOLVE_MIN_WAVES: wavefronts per SIMD the generic instantiations are compiled for (2 -> 256 VGPRs; the
// residency experiment in EXPERIMENTS.md II §4.4 builds them with 3 -> 168 VGPRs)
#ifndef DEKF_SOLVE_MIN_WAVES
#define DEKF_SOLVE_MIN_WAVES 2
#endif
// Which solve kernels a build carries: -DDEKF_KSET=<mask> (default: all).  Kernels outside the mask are empty stubs, so a library
// built with a mask solves only the shapes of its kernels.  1: Go1 N = 20 (k_mhe_solve_ll_4_n20, k_mhe_solve_r3_4_n20) — what
// -DDEKF_GO1_ONLY selects, the A/B builds of the benchmark shape (tools/ab_variants.sh; builds in seconds instead of minutes);
// 2: Cassie N = 20 (k_mhe_solve_lg_2_n20, k_mhe_solve_r3_2_n20); 4 / 8 / 16 / 32: the generic kernels for 1 / 2 / 3 / 4 legs;
// 64 / 128 / 256 / 512: the foot-position kernels (leg_odom_type 1) for 1 / 2 / 3 / 4 legs.  The bounds-checked diagnostic build
// is compiled one mask at a time (tools/build_bounds.sh): with checked pointers one translation unit of everything takes hours.
#ifndef DEKF_KSET
#ifdef DEKF_GO1_ONLY
#define DEKF_KSET 1
#else
#define DEKF_KSET 0x3FF
#endif
#endif
// Every solve kernel exists twice: NAME, and NAME_pol with OSQP's polishing step behind the iterations (osqp.polish true; chosen
// by dekf_create).  Two instantiations rather than a run-time branch: compiled into k_mhe_solve_r3_4_n20 the polishing code (a
// second call site of the factorisation and of the iteration chunk) took its spilled VGPRs from 60 to 147.
#ifdef DEKF_KSET_ONLY
#define DEKF_STUB_KERNEL(NAME)
#else
#define DEKF_STUB_KERNEL(NAME)                                  \
    __global__ void NAME(DevCfg, DevState, int, int, int) {}     \
    __global__ void NAME##_pol(DevCfg, DevState, int, int, int) {}
#endif
#define DEKF_SOLVE_KERNEL_BODY_(NAME, WAVES, POLISH, ...) DEKF_SOLVE_KERNEL_BODY_T_(NAME, DEKF_SOLVE_THREADS, DEKF_QUEUE_LOOP, WAVES, POLISH, __VA_ARGS__)
#define DEKF_SOLVE_KERNEL_BODY_T_(NAME, THREADS, LOOP, WAVES, POLISH, ...)                                                   \
    __global__ void __launch_bounds__(THREADS, WAVES) NAME(DevCfg c, DevState s, int kstart, int K, int gws_len) {            \
        extern __shared__ double lds[];                                                                                      \
        __shared__ int next_instance;                                                                                        \
        double* gws = s.gws + (size_t)blockIdx.x * gws_len;                                                                  \
        DEKF_WG_TRACE_BEGIN                                                                                                  \
        LOOP(solve_window_t<POLISH, __VA_ARGS__>(c, s, b, kstart, K, lds, gws))                                              \
        solve_queue_leave(s);                                                                                                \
        DEKF_WG_TRACE_END                                                                                                    \
    }
#ifdef DEKF_NO_POLISH_KERNELS  // (A/B and diagnostic builds: half the compile time; osqp.polish true is then refused by the stubs' owner, dekf_create)
#define DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, ...)            \
    DEKF_SOLVE_KERNEL_BODY_(NAME, WAVES, false, __VA_ARGS__) \
    __global__ void NAME##_pol(DevCfg, DevState, int, int, int) {}
#else
#define DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, ...)            \
    DEKF_SOLVE_KERNEL_BODY_(NAME, WAVES, false, __VA_ARGS__) \
    DEKF_SOLVE_KERNEL_BODY_(NAME##_pol, WAVES, true, __VA_ARGS__)
#endif
#define DEKF_SOLVE_KERNEL_IF(BIT, NAME, WAVES, ...) DEKF_SOLVE_KERNEL_IF_(BIT, NAME, WAVES, __VA_ARGS__)
#define DEKF_SOLVE_KERNEL_IF_(BIT, NAME, WAVES, ...) DEKF_SOLVE_KERNEL_SEL_##BIT(NAME, WAVES, __VA_ARGS__)
// one selector per bit (the preprocessor cannot branch on an expression inside a macro body)
#if DEKF_KSET & 1
#define DEKF_SOLVE_KERNEL_SEL_0(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_0(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
#if DEKF_KSET & 2
#define DEKF_SOLVE_KERNEL_SEL_1(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_1(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
#if DEKF_KSET & 4
#define DEKF_SOLVE_KERNEL_SEL_2(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_2(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
#if DEKF_KSET & 8
#define DEKF_SOLVE_KERNEL_SEL_3(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_3(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
#if DEKF_KSET & 16
#define DEKF_SOLVE_KERNEL_SEL_4(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_4(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
#if DEKF_KSET & 32
#define DEKF_SOLVE_KERNEL_SEL_5(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_5(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
#if DEKF_KSET & 64
#define DEKF_SOLVE_KERNEL_SEL_6(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_6(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
#if DEKF_KSET & 128
#define DEKF_SOLVE_KERNEL_SEL_7(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_7(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
#if DEKF_KSET & 256
#define DEKF_SOLVE_KERNEL_SEL_8(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_8(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
#if DEKF_KSET & 512
#define DEKF_SOLVE_KERNEL_SEL_9(NAME, WAVES, ...) DEKF_SOLVE_KERNEL_BODY(NAME, WAVES, __VA_ARGS__)
#else
#define DEKF_SOLVE_KERNEL_SEL_9(NAME, WAVES, ...) DEKF_STUB_KERNEL(NAME)
#endif
// three placements of the factor per leg count: _ll all in LDS, _lg factor-time temporary in HBM, _gg factor streamed from HBM
#define DEKF_SOLVE_KERNELS(BIT, LEGS)                                                                       \
    DEKF_SOLVE_KERNEL_IF(BIT, k_mhe_solve_ll_##LEGS, DEKF_SOLVE_MIN_WAVES, LEGS, true, true)                \
    DEKF_SOLVE_KERNEL_IF(BIT, k_mhe_solve_lg_##LEGS, DEKF_SOLVE_MIN_WAVES, LEGS, true, false)               \
    DEKF_SOLVE_KERNEL_IF(BIT, k_mhe_solve_gg_##LEGS, DEKF_SOLVE_MIN_WAVES, LEGS, false, false)
// the benchmark shape (Go1, N = 20) additionally with the horizon as a compile-time constant; its full windows run THREE
// workgroups per CU (mhe_admm_core.h, admm_chunk_r3: row state in registers, 168 VGPRs); the window-fill ticks (K < N) keep the
// two-workgroup kernel
DEKF_SOLVE_KERNEL_IF(0, k_mhe_solve_ll_4_n20, 2, 4, true, true, 20)
DEKF_SOLVE_KERNEL_IF(0, k_mhe_solve_r3_4_n20, DEKF_R3_WAVES, 4, true, true, 20, 0, true)
// the same full windows on workgroups of THREE wavefronts at FOUR per CU (round 6; mhe_admm_core.h: admm_chunk_r4, SolveLayout::r4_*)
#ifdef DEKF_NO_POLISH_KERNELS
#define DEKF_SOLVE_KERNEL_R4(NAME, ...)                                              \
    DEKF_SOLVE_KERNEL_BODY_T_(NAME, DEKF_R4_THREADS, DEKF_QUEUE_LOOP_DYNAMIC, DEKF_R3_WAVES, false, __VA_ARGS__) \
    __global__ void NAME##_pol(DevCfg, DevState, int, int, int) {}
#else
#define DEKF_SOLVE_KERNEL_R4(NAME, ...)                                              \
    DEKF_SOLVE_KERNEL_BODY_T_(NAME, DEKF_R4_THREADS, DEKF_QUEUE_LOOP_DYNAMIC, DEKF_R3_WAVES, false, __VA_ARGS__) \
    DEKF_SOLVE_KERNEL_BODY_T_(NAME##_pol, DEKF_R4_THREADS, DEKF_QUEUE_LOOP_DYNAMIC, DEKF_R3_WAVES, true, __VA_ARGS__)
#endif
#if DEKF_KSET & 1
DEKF_SOLVE_KERNEL_R4(k_mhe_solve_r4_4_n20, 4, true, false, 20, 0, true)
#else
DEKF_STUB_KERNEL(k_mhe_solve_r4_4_n20)
#endif
#if DEKF_KSET & 2
DEKF_SOLVE_KERNEL_R4(k_mhe_solve_r4_2_n20, 2, true, false, 20, 0, true)
#else
DEKF_STUB_KERNEL(k_mhe_solve_r4_2_n20)
#endif
// Cassie (2 legs, N = 20; its factor-time temporary does not fit next to the vectors: _lg placement)
DEKF_SOLVE_KERNEL_IF(1, k_mhe_solve_lg_2_n20, 2, 2, true, false, 20)
DEKF_SOLVE_KERNEL_IF(1, k_mhe_solve_r3_2_n20, DEKF_R3_WAVES, 2, true, true, 20, 0, true)
DEKF_SOLVE_KERNELS(2, 1)
// one leg, long windows (PogoX, N = 100): full windows with the row state in registers at a run-time horizon, factor in the slab,
// TWO workgroups per CU instead of the one that the generic placement's 103 KB of iterates allow (mhe_admm_core.h: admm_chunk_rr)
DEKF_SOLVE_KERNEL_IF(2, k_mhe_solve_rr_1, 2, 1, false, false, 0, 0, true)
DEKF_SOLVE_KERNELS(3, 2)
DEKF_SOLVE_KERNELS(4, 3)
DEKF_SOLVE_KERNELS(5, 4)
// leg_odom_type 1: the foot positions are states (9 + 3 LEGS per window step, 21 for Go1).  Two placements: factor in
// LDS with the factor-time temporary in HBM (short windows), factor streamed from the workgroup's HBM slab (Go1, N = 20:
// S^-1 and W alone are 141 KB).
#define DEKF_SOLVE_KERNELS_FOOT(BIT, LEGS)                                                                         \
    DEKF_SOLVE_KERNEL_IF(BIT, k_mhe_solve_foot_lg_##LEGS, DEKF_SOLVE_MIN_WAVES, LEGS, true, false, 0, 1)           \
    DEKF_SOLVE_KERNEL_IF(BIT, k_mhe_solve_foot_gg_##LEGS, DEKF_SOLVE_MIN_WAVES, LEGS, false, false, 0, 1)
DEKF_SOLVE_KERNELS_FOOT(6, 1)
DEKF_SOLVE_KERNELS_FOOT(7, 2)
DEKF_SOLVE_KERNELS_FOOT(8, 3)
DEKF_SOLVE_KERNELS_FOOT(9, 4)

#if DEKF_MISC_KERNELS
__global__ void k_gap() {}
__global__ void __launch_bounds__(64) k_kf_initialize(DevCfg c, DevState s) {
    extern __shared__ double lds[];
    kf_initialize(c, s, blockIdx.x, lds);
}

__global__ void __launch_bounds__(64) k_kf_update(DevCfg c, DevState s, int pushes) {
    extern __shared__ double lds[];
    kf_update(c, s, blockIdx.x, pushes, lds);
}

__global__ void k_latch_vo(DevCfg c, DevState s, const int* mask, const double* t_pre, const double* t_now,
                           const double* dp, const double* t_pose, const double* q_vo) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B || !mask[b]) return;
    s.vo_flag[b] = 1;
    s.vo_tpre[b] = t_pre[b];
    s.vo_tnow[b] = t_now[b];
    for (int i = 0; i < 3; ++i) s.vo_dp[3 * (size_t)b + i] = dp[3 * (size_t)b + i];
    if (q_vo) {
        s.ekf_vo_flag[b] = 1;
        s.ekf_vo_t[b] = t_pose[b];
        for (int i = 0; i < 4; ++i) s.ekf_vo_q[4 * (size_t)b + i] = q_vo[4 * (size_t)b + i];
    }
}

// the device-pointer form of dekf_push_imu / dekf_push_leg: all arrays of one push in one grid-stride copy
__global__ void __launch_bounds__(256) k_latch4(LatchCopy4 a) {
    const size_t total = a.end[3], stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int seg = e < a.end[0] ? 0 : (e < a.end[1] ? 1 : (e < a.end[2] ? 2 : 3));
        const size_t off = e - (seg == 0 ? 0 : a.end[seg - 1]);
        a.dst[seg][off] = a.src[seg][off];
    }
}

__global__ void __launch_bounds__(64) k_go1_leg_odometry(DevCfg c, DevState s, const double* jp, const double* jv,
                                                         const double* force, double thr, double pibx, double piby,
                                                         double pibz) {
    int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= c.B) return;
    const double p_ib[3] = {pibx, piby, pibz};
    go1_leg_odometry(s, b, jp, jv, force, thr, p_ib);
}

// Cov_q_ of every instance as [B][4][4] from the field-major EKF state [16][B]: consecutive lanes write consecutive
// doubles, the strided side is the read (16 streams of B doubles: each is contiguous across the lanes that share i)
__global__ void __launch_bounds__(256) k_ekf_cov_out(DevCfg c, DevState s, double* out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x, B = (size_t)c.B;
    if (e >= 16 * B) return;
    const size_t b = e / 16, i = e - 16 * b;
    out[e] = s.ekf_P[i * B + b];
}

__global__ void k_reset_state(DevCfg c, DevState s) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B) return;
    size_t B = (size_t)c.B;
    for (int i = 0; i < 4; ++i) { s.ekf_q[i * B + b] = c.ekf_q0[i]; s.quat[4 * (size_t)b + i] = c.ekf_q0[i]; }
    for (int i = 0; i < 16; ++i) s.ekf_P[i * B + b] = (i % 5 == 0) ? c.ekf_P0[i / 5] : 0.0;
    s.vo_flag[b] = 0;
    s.ekf_vo_flag[b] = 0;
    s.wp_count[b] = 0;
    for (int i = 0; i < 3; ++i) s.p_vo[3 * (size_t)b + i] = 0.0;
    for (int i = 0; i < c.ns; ++i) s.x_mhe[(size_t)c.ns * b + i] = 0.0;
    for (int i = 0; i < 3; ++i) s.v_b[3 * (size_t)b + i] = 0.0;
    s.status[b] = DEKF_SOLVE_NONE;
    s.iters[b] = 0;
    s.rho_updates[b] = 0;
    s.polish_status[b] = 0;
    s.pri_res[b] = 0.0;
    s.dua_res[b] = 0.0;
    s.vo_ins_idx[b] = 0;
    s.vo_ins_dtime[b] = 0;
    s.marg_tag[b] = -1;
}

#endif  // DEKF_MISC_KERNELS

}  // extern "C"
