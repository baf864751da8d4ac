// dekf_capi.hip — implementation of the C ABI in include/dekf.h on top of the gfx950 kernels.
// No CPU fallback: without a usable HIP device dekf_create fails with DEKF_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dekf.h"
#include "cfg.h"
#include "host_common.h"
#include "kf_core.h"
#include "mhe_assemble_core.h"
#include "mhe_solve_core.h"

using namespace dekf;

extern "C" {
__global__ void k_ekf_tick(DevCfg c, DevState s, int count);
__global__ void k_mhe_initialize(DevCfg c, DevState s);
__global__ void k_mhe_assemble(DevCfg c, DevState s, int T, int pushes);
__global__ void k_mhe_marginalize_early(DevCfg c, DevState s, int T);
// every solve kernel and its twin with OSQP's polishing step (kernels.hip: DEKF_SOLVE_KERNEL_BODY)
#define DEKF_DECL_K(NAME)                                                              \
    __global__ void NAME(DevCfg c, DevState s, int kstart, int K, int gws_len);        \
    __global__ void NAME##_pol(DevCfg c, DevState s, int kstart, int K, int gws_len);
#define DEKF_DECL_SOLVE(LEGS) DEKF_DECL_K(k_mhe_solve_ll_##LEGS) DEKF_DECL_K(k_mhe_solve_lg_##LEGS) DEKF_DECL_K(k_mhe_solve_gg_##LEGS)
DEKF_DECL_K(k_mhe_solve_ll_4_n20)
DEKF_DECL_K(k_mhe_solve_lg_2_n20)
#ifndef DEKF_NO_R3
DEKF_DECL_K(k_mhe_solve_r3_4_n20)
DEKF_DECL_K(k_mhe_solve_r3_2_n20)
DEKF_DECL_K(k_mhe_solve_r4_4_n20)
DEKF_DECL_K(k_mhe_solve_r4_2_n20)
#endif
DEKF_DECL_SOLVE(1)
#ifndef DEKF_NO_RR
DEKF_DECL_K(k_mhe_solve_rr_1)
#endif
DEKF_DECL_SOLVE(2)
DEKF_DECL_SOLVE(3)
DEKF_DECL_SOLVE(4)
#define DEKF_DECL_SOLVE_FOOT(LEGS) DEKF_DECL_K(k_mhe_solve_foot_lg_##LEGS) DEKF_DECL_K(k_mhe_solve_foot_gg_##LEGS)
DEKF_DECL_SOLVE_FOOT(1)
DEKF_DECL_SOLVE_FOOT(2)
DEKF_DECL_SOLVE_FOOT(3)
DEKF_DECL_SOLVE_FOOT(4)
__global__ void k_gap();
__global__ void k_kf_initialize(DevCfg c, DevState s);
__global__ void k_kf_update(DevCfg c, DevState s, int pushes);
__global__ void k_latch_vo(DevCfg c, DevState s, const int* mask, const double* t_pre, const double* t_now,
                           const double* dp, const double* t_pose, const double* q_vo);
__global__ void k_reset_state(DevCfg c, DevState s);
__global__ void k_ekf_cov_out(DevCfg c, DevState s, double* out);
__global__ void k_latch4(LatchCopy4 a);
__global__ void k_go1_leg_odometry(DevCfg c, DevState s, const double* jp, const double* jv, const double* force,
                                   double thr, double pibx, double piby, double pibz);
}

namespace {
thread_local std::string g_err;

struct RcclApi;  // rccl_dyn.h
}  // namespace
#include "rccl_dyn.h"

struct dekf_handle_s {
    dekf_params prm;
    DevCfg c;
    DevState s;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<void*> blocks;
    // staging for host-side pushes / gets
    void* stage = nullptr;
    size_t stage_bytes = 0;
    int solve_grid = 0, gws_len = 0;
    void (*solve_kernel)(DevCfg, DevState, int, int, int) = nullptr;
    // full windows (K == N) of the fixed-horizon shapes: the three-workgroups-per-CU kernel (kernels.hip), its own grid and LDS size
    void (*solve_kernel_full)(DevCfg, DevState, int, int, int) = nullptr;
    int solve_grid_full = 0;
    int solve_threads_full = DEKF_SOLVE_THREADS;  // (the four-per-CU kernels: DEKF_R4_THREADS)
    size_t lds_solve_full = 0;
    const char *solve_name = nullptr, *solve_name_full = nullptr;  // kernel symbols, for dekf_solve_kernel_name
    size_t lds_solve = 0, lds_asm = 0, lds_kf = 0;
    int ekf_count = 0, pushes = 0, next_T = 0;
    bool initialized = false;
    // timing: 0 off, 1 every kernel class, 2 only the MHE solve (class 2)
    int timing = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[DEKF_TIMING_CLASSES];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    // Step pipelining (dekf_params.solve_pipeline): the solve of step T runs on solve_stream[T & 1] out of set T & 1 of the
    // per-solve data (outputs, scratch slabs) and copy T mod 3 of its input snapshot, so the pushes, the EKF tick and the
    // assemble of step T + 1 — and then its solve — start while the last round of step T's persistent workgroups is still
    // running (a launch of B instances on S slots runs ceil(B / S) rounds, the last one partly empty: 4096 on 768 is 5.33).
    bool pipelined = false;
    DevState sp[2];               // sp[0] == s; sp[1]: the second set
    hipStream_t solve_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_asm[2] = {nullptr, nullptr}, ev_solve[2] = {nullptr, nullptr}, ev_mark[2] = {nullptr, nullptr};
    // input snapshots by T mod 3: ev_snap_free[i] is recorded behind the solve that read copy i, the assemble of three steps on waits for it
    hipEvent_t ev_snap_free[3] = {nullptr, nullptr, nullptr};
    bool snap_busy[3] = {false, false, false};
    bool solve_pending[2] = {false, false};
    int last_par = 0;             // set the newest results are in
    // In-order mode: the arrival cost of step T + 1 is computed ahead of time on early_stream (k_mhe_marginalize_early), behind the
    // assemble of step T and beside its solve — i.e. in the slots that solve frees in its partly empty last round — and the assemble
    // of step T + 1 takes it unless a vision interval arrived (cfg.h: DevState::Mp_next).  44 of the 67 us of term construction in
    // front of every solve leave the critical path; results are the same bits.
    hipStream_t early_stream = nullptr;
    hipEvent_t ev_asm_done = nullptr, ev_early_done = nullptr, ev_early_mark = nullptr;
    bool early_pending = false;
    // RCCL: the all-gather runs on its own stream out of a snapshot of v_b, so that it overlaps the next step
    void* comm = nullptr;
    int world = 1, rank = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_vb_ready = nullptr, ev_ag_done = nullptr;
    // pipelined mode: recorded when the all-gather's snapshot copy has READ set i's v_b — what the next solve into set i waits for
    // (one event per set: the solve of step T + 1 must not wait for the copy that follows the solve of step T, or nothing overlaps)
    hipEvent_t ev_vb_read[2] = {nullptr, nullptr};
    bool vb_read_pending[2] = {false, false};
    double* vb_snapshot = nullptr;
    bool ag_pending = false;
};

namespace {

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                                     \
            return DEKF_ERR_HIP;                                                                           \
        }                                                                                                  \
    } while (0)

dekf_status fail(dekf_status st, const char* msg) {
    g_err = msg;
    return st;
}

dekf_status ensure_stage(dekf_handle h, size_t bytes) {
    if (bytes <= h->stage_bytes) return DEKF_OK;
    if (h->stage) HIPCHK(hipFree(h->stage));
    h->stage = nullptr;
    h->stage_bytes = 0;
    HIPCHK(hipMalloc(&h->stage, bytes));
    h->stage_bytes = bytes;
    return DEKF_OK;
}

// dst (device) <- src (host or device), n bytes, on the handle's stream
dekf_status put(dekf_handle h, void* dst, const void* src, size_t n, dekf_mem where) {
    if (!src) return fail(DEKF_ERR_INVALID, "null input pointer");
    HIPCHK(hipMemcpyAsync(dst, src, n, where == DEKF_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream));
    return DEKF_OK;
}
dekf_status fetch(dekf_handle h, void* dst, const void* src, size_t n, dekf_mem where) {
    if (!dst) return DEKF_OK;
    HIPCHK(hipMemcpyAsync(dst, src, n, where == DEKF_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->stream));
    return DEKF_OK;
}

// up to four device-to-device latches of one push in ONE launch (separate hipMemcpyAsync calls each cost a
// ~5 us blit kernel: seven per step were 0.9 % of the step); host sources still go through hipMemcpyAsync
dekf_status put_many(dekf_handle h, int n, double* const* dst, const double* const* src, const size_t* count, dekf_mem where) {
    for (int i = 0; i < n; ++i) if (!src[i]) return fail(DEKF_ERR_INVALID, "null input pointer");
    if (where == DEKF_HOST) {
        for (int i = 0; i < n; ++i) HIPCHK(hipMemcpyAsync(dst[i], src[i], count[i] * sizeof(double), hipMemcpyHostToDevice, h->stream));
        return DEKF_OK;
    }
    LatchCopy4 a;
    size_t total = 0;
    for (int i = 0; i < 4; ++i) {
        a.dst[i] = i < n ? dst[i] : nullptr;
        a.src[i] = i < n ? src[i] : nullptr;
        total += i < n ? count[i] : 0;
        a.end[i] = total;
    }
    if (total == 0) return DEKF_OK;
    const int blocks = (int)((total + 255) / 256);
    k_latch4<<<blocks < 2048 ? blocks : 2048, 256, 0, h->stream>>>(a);
    HIPCHK(hipGetLastError());
    return DEKF_OK;
}

// the handle's stream waits (no host block) for the solve that produces the newest results; getters then read in stream order
dekf_status await_results(dekf_handle h) {
    if (h->pipelined && h->solve_pending[h->last_par]) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_solve[h->last_par], 0));
    return DEKF_OK;
}

struct Timed {  // brackets one launch with events when timing is on
    dekf_handle h;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    Timed(dekf_handle h_, int cls_, hipStream_t st_ = nullptr) : h(h_), cls(cls_), st(st_ ? st_ : h_->stream) {
        if (!h->timing || (h->timing == 2 && cls != 2 && cls != 3)) return;  // (3: the all-gather, on the communication stream — off the step's critical path)
        if (!h->ev_pool.empty()) { a = h->ev_pool.back().first; b = h->ev_pool.back().second; h->ev_pool.pop_back(); }
        else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            if (a) (void)hipEventDestroy(a);
            a = b = nullptr;
            return;
        }
        (void)hipEventRecord(a, st);
    }
    ~Timed() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        h->ev[cls].push_back({a, b});
    }
};

}  // namespace

extern "C" {

void dekf_default_params(dekf_params* p) { default_params(p); }
int dekf_abi_version(void) { return DEKF_ABI_VERSION; }
int dekf_hip_runtime_version(void) {
    int v = 0;
    return hipRuntimeGetVersion(&v) == hipSuccess ? v : 0;
}
const char* dekf_last_error(void) { return g_err.c_str(); }

dekf_status dekf_create(const dekf_params* p, int batch, int device, void* stream, dekf_handle* out) {
    if (!p || !out) return fail(DEKF_ERR_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(DEKF_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(DEKF_ERR_INVALID, "device ordinal out of range");
    DevCfg c;
    if (const char* msg = fill_cfg(*p, batch, c)) return fail(DEKF_ERR_INVALID, msg);
#ifdef DEKF_NO_POLISH_KERNELS
    if (c.polish && c.est_type == 0) return fail(DEKF_ERR_INVALID, "this diagnostic build (-DDEKF_NO_POLISH_KERNELS) carries no polishing kernels");
#endif
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_err = std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only";
        return DEKF_ERR_NO_DEVICE;
    }
    dekf_handle h = new dekf_handle_s();
    h->prm = *p;
    h->c = c;
    h->device = device;
    if (stream) h->stream = (hipStream_t)stream;
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            h->stream = nullptr;
            dekf_destroy(h);
            return fail(DEKF_ERR_HIP, "hipStreamCreate failed");
        }
        h->own_stream = true;
    }
    SolveLayout lay;
    lay.init(c.N, c.L, c.ft);
    h->lds_solve = lay.lds_bytes();
    h->c.gws_wt = !c.ft && !lay.factor_in_lds();  // the factor streams from the slab: the outward legs read a transposed copy of W (Gws::WT)
#ifdef DEKF_PROFILE
    // diagnostic build only: DEKF_DEBUG_LDS_PAD=<bytes> inflates the request (e.g. to force one workgroup per CU)
    if (const char* pad = getenv("DEKF_DEBUG_LDS_PAD")) h->lds_solve += (size_t)atol(pad);
#endif
    typedef void (*SolveFn)(DevCfg, DevState, int, int, int);
#ifndef DEKF_R4_SELECT
#define DEKF_R4_SELECT(cap) ((cap) == 4)
#endif
    {
        struct Named { SolveFn fn; const char* name; SolveFn fn_pol; const char* name_pol; };
#define DEKF_K(sym) {sym, #sym, sym##_pol, #sym "_pol"}
        static const Named table[4][3] = {
            {DEKF_K(k_mhe_solve_ll_1), DEKF_K(k_mhe_solve_lg_1), DEKF_K(k_mhe_solve_gg_1)}, {DEKF_K(k_mhe_solve_ll_2), DEKF_K(k_mhe_solve_lg_2), DEKF_K(k_mhe_solve_gg_2)},
            {DEKF_K(k_mhe_solve_ll_3), DEKF_K(k_mhe_solve_lg_3), DEKF_K(k_mhe_solve_gg_3)}, {DEKF_K(k_mhe_solve_ll_4), DEKF_K(k_mhe_solve_lg_4), DEKF_K(k_mhe_solve_gg_4)}};
        Named pick = table[c.L - 1][lay.pa_in_lds() ? 0 : (lay.factor_in_lds() ? 1 : 2)];
        if (c.L == 4 && c.N == 20 && lay.pa_in_lds()) pick = DEKF_K(k_mhe_solve_ll_4_n20);
        if (c.L == 2 && c.N == 20 && lay.factor_in_lds() && !lay.pa_in_lds()) pick = DEKF_K(k_mhe_solve_lg_2_n20);
        if (c.ft) {  // foot-position states: their own kernel family (two rows per block in the solve)
            static const Named foot[4][2] = {{DEKF_K(k_mhe_solve_foot_lg_1), DEKF_K(k_mhe_solve_foot_gg_1)}, {DEKF_K(k_mhe_solve_foot_lg_2), DEKF_K(k_mhe_solve_foot_gg_2)},
                                             {DEKF_K(k_mhe_solve_foot_lg_3), DEKF_K(k_mhe_solve_foot_gg_3)}, {DEKF_K(k_mhe_solve_foot_lg_4), DEKF_K(k_mhe_solve_foot_gg_4)}};
            pick = foot[c.L - 1][lay.factor_in_lds() ? 0 : 1];
        }
        // osqp.polish: the twin that carries the polishing step
        h->solve_kernel = c.polish ? pick.fn_pol : pick.fn;
        h->solve_name = c.polish ? pick.name_pol : pick.name;
#ifdef DEKF_PROFILE
        // diagnostic build only: DEKF_DEBUG_PLACEMENT=1|2 forces the _lg / _gg placement (2 also shrinks the LDS request)
        if (const char* pl = getenv("DEKF_DEBUG_PLACEMENT")) {
            int p = atoi(pl);
            if ((p == 1 || p == 2) && !c.ft) {
                const Named& t = table[c.L - 1][p];
                h->solve_kernel = c.polish ? t.fn_pol : t.fn;
                h->solve_name = c.polish ? t.name_pol : t.name;
            }
            // the _gg kernels carve D, E, bounds and R behind the iterates when SolveLayout::gg_consts_in_lds() says so
            if (p == 2) h->lds_solve = (size_t)(lay.vec + (lay.gg_consts_in_lds() ? lay.gg_consts() : 0)) * sizeof(double);
            if (p == 2) h->c.gws_wt = !c.ft;
        }
#endif
    }
    h->lds_asm = (size_t)AsmScratch::len(c.L, c.ft) * sizeof(double);
    h->lds_kf = (size_t)KfScratch::len(c.L, c.ft) * sizeof(double);
    if (h->lds_solve > 160 * 1024) {
        dekf_destroy(h);
        return fail(DEKF_ERR_INVALID, "window too large: ADMM iterates exceed the 160 KiB LDS of one CU");
    }
    if (h->lds_solve > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)h->solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_solve);
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)h->solve_kernel, DEKF_SOLVE_THREADS, h->lds_solve) != hipSuccess || per_cu < 1)
        per_cu = 1;
    const int cap = p->solve_workgroups_per_cu;  // 0: no cap
    if (cap > 0 && per_cu > cap) per_cu = cap;
    long slots = (long)per_cu * prop.multiProcessorCount;
    h->solve_grid = (int)(slots < batch ? slots : batch);
#ifndef DEKF_NO_R3
    // Four workgroups of three wavefronts per CU (round 6, admm_chunk_r4): solve_workgroups_per_cu = DEKF_R4_SELECT selects them
    if (!c.ft && c.N == 20 && (c.L == 4 || c.L == 2) && DEKF_R4_SELECT(cap)) {
        const SolveFn full = c.polish ? (c.L == 4 ? k_mhe_solve_r4_4_n20_pol : k_mhe_solve_r4_2_n20_pol) : (c.L == 4 ? k_mhe_solve_r4_4_n20 : k_mhe_solve_r4_2_n20);
        const char* const full_name = c.polish ? (c.L == 4 ? "k_mhe_solve_r4_4_n20_pol" : "k_mhe_solve_r4_2_n20_pol") : (c.L == 4 ? "k_mhe_solve_r4_4_n20" : "k_mhe_solve_r4_2_n20");
        hipFuncAttributes fa;
        size_t static_lds = 512;
        if (hipFuncGetAttributes(&fa, (const void*)full) == hipSuccess) static_lds = fa.sharedSizeBytes;
        if (lay.r4_fits(c.L, static_lds)) {
            const long sf = 4L * prop.multiProcessorCount;
            int gridf = (int)(sf < batch ? sf : batch);
#ifdef DEKF_AB_KNOBS  // A/B builds only (tools/r4_check.py): another persistent grid
            if (const char* ge = getenv("DEKF_DEBUG_R4_GRID")) { const int gv = atoi(ge); if (gv > 0 && gv <= batch) gridf = gv; }
#endif
            if (gridf > h->solve_grid) {
                h->lds_solve_full = lay.r4_lds_bytes();
                h->solve_grid_full = gridf;
                h->solve_threads_full = DEKF_R4_THREADS;
                h->solve_kernel_full = full;
                h->solve_name_full = full_name;
            }
        }
    }
    if (!h->solve_kernel_full && !c.ft && c.N == 20 && (c.L == 4 || c.L == 2) && (cap == 0 || cap > per_cu)) {
        const SolveFn full = c.polish ? (c.L == 4 ? k_mhe_solve_r3_4_n20_pol : k_mhe_solve_r3_2_n20_pol) : (c.L == 4 ? k_mhe_solve_r3_4_n20 : k_mhe_solve_r3_2_n20);
        const char* const full_name = c.polish ? (c.L == 4 ? "k_mhe_solve_r3_4_n20_pol" : "k_mhe_solve_r3_2_n20_pol") : (c.L == 4 ? "k_mhe_solve_r3_4_n20" : "k_mhe_solve_r3_2_n20");
        // the kernel's static LDS (reduction scratch of wave.h) counts against the same allocation as the dynamic part
        hipFuncAttributes fa;
        size_t static_lds = 512;
        if (hipFuncGetAttributes(&fa, (const void*)full) == hipSuccess) static_lds = fa.sharedSizeBytes;
        if (lay.r3_fits(c.L, static_lds)) {
            h->lds_solve_full = lay.r3_lds_bytes();
            int pcf = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pcf, (const void*)full, DEKF_SOLVE_THREADS, h->lds_solve_full) != hipSuccess || pcf < 1)
                pcf = 1;
            if (pcf > DEKF_R3_WAVES) pcf = DEKF_R3_WAVES;  // the query over-reports: r3_fits has decided that DEKF_R3_WAVES fit
            if (cap > 0 && pcf > cap) pcf = cap;
            const long sf = (long)pcf * prop.multiProcessorCount;
            h->solve_grid_full = (int)(sf < batch ? sf : batch);
            // For EVERY batch since round 6 (it used to be only where the batch fills more slots than the two-workgroup kernel offers,
            // "below that the row state would travel through the slab for no residency gained"): measured, the rows-in-registers
            // kernel solves a full window in 0.278 ms against 0.350 ms even ALONE on its CU — 21-22 % less solve time at 64 ... 512
            // instances (profiles/r06_r3_at_small_batches.txt).  The two-workgroup kernels keep the window-fill ticks, and full
            // windows when the caller caps the residency at two.
            bool take = pcf > per_cu;
#ifdef DEKF_AB_KNOBS  // A/B builds only: the round-5 rule
            if (getenv("DEKF_DEBUG_R3_ONLY_ABOVE_512")) take = pcf > per_cu && h->solve_grid_full > h->solve_grid;
#endif
            if (take) {
                h->solve_kernel_full = full;
                h->solve_name_full = full_name;
            }
        }
    }
#endif
#ifndef DEKF_NO_RR
    // One-legged robots with long windows (PogoX: N = 100): the generic placement keeps 103 KB of iterates in LDS, one workgroup per
    // CU.  Full windows run the kernel that keeps the row state in registers (mhe_admm_core.h: admm_chunk_rr) — 77 KB, two per CU —
    // when the batch fills more slots than the generic kernel offers.
    if (!h->solve_kernel_full && !c.ft && c.L == 1 && (cap == 0 || cap > per_cu)) {
        const SolveFn full = c.polish ? k_mhe_solve_rr_1_pol : k_mhe_solve_rr_1;
        const char* const full_name = c.polish ? "k_mhe_solve_rr_1_pol" : "k_mhe_solve_rr_1";
        hipFuncAttributes fa;
        size_t static_lds = 512;
        if (hipFuncGetAttributes(&fa, (const void*)full) == hipSuccess) static_lds = fa.sharedSizeBytes;
        if (lay.rr_fits(c.L, static_lds)) {
            h->lds_solve_full = lay.rr_lds_bytes();
            if (h->lds_solve_full > 64 * 1024)
                (void)hipFuncSetAttribute((const void*)full, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_solve_full);
            int pcf = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pcf, (const void*)full, DEKF_SOLVE_THREADS, h->lds_solve_full) != hipSuccess || pcf < 1)
                pcf = 1;
            if (pcf > 2) pcf = 2;  // (rr_fits has decided that two fit; the kernel is compiled for two wavefronts per SIMD)
            if (cap > 0 && pcf > cap) pcf = cap;
            const long sf = (long)pcf * prop.multiProcessorCount;
            h->solve_grid_full = (int)(sf < batch ? sf : batch);
            // (for every batch since round 6: 10-13 % less solve time than the generic kernel at 32 ... 256 instances too,
            // profiles/r06_r3_at_small_batches.txt)
            bool take = pcf > per_cu;
#ifdef DEKF_AB_KNOBS  // A/B builds only: the round-5 rule
            if (getenv("DEKF_DEBUG_RR_ONLY_ABOVE_256")) take = pcf > per_cu && h->solve_grid_full > h->solve_grid;
#endif
            if (take) {
                h->solve_kernel_full = full;
                h->solve_name_full = full_name;
            }
        }
    }
#endif
    const int solve_slots = h->solve_kernel_full && h->solve_grid_full > h->solve_grid ? h->solve_grid_full : h->solve_grid;
    Gws g;
    g.init(c.N, c.L, c.ft, h->c.gws_wt);
    h->gws_len = g.total;
    h->pipelined = c.est_type == 0 && p->solve_pipeline == 1;
    bool ok = true;
    alloc_state(h->c, h->s, solve_slots, [&](size_t bytes) -> void* {
        void* q = nullptr;
        if (!ok) return nullptr;
        if (hipMalloc(&q, bytes ? bytes : 8) != hipSuccess) { ok = false; return nullptr; }
        if (hipMemsetAsync(q, 0, bytes ? bytes : 8, h->stream) != hipSuccess) ok = false;
        h->blocks.push_back(q);
        return q;
    }, h->pipelined ? 2 : 1);
    if (!ok) {
        dekf_destroy(h);
        return fail(DEKF_ERR_HIP, "hipMalloc failed while allocating the estimator state");
    }
    h->sp[0] = h->s;
    h->sp[1] = h->pipelined ? second_set(h->c, h->s, solve_slots) : h->s;
    // (N = 1: the record folded at step T gets its gains AT step T.  DEKF_DEBUG_NO_EARLY_MARGINALIZE in the environment: diagnostic switch,
    // everything on the handle's stream as before round 5 — same bits either way, tests/test_gpu_configs.py)
    if (!h->pipelined && c.est_type == 0 && c.N >= 2 && !getenv("DEKF_DEBUG_NO_EARLY_MARGINALIZE")) {
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        // least priority: background work that is to take the slots the solve leaves, not to compete for them when both are ready
        // (and a priority class of its own: never on a hardware queue with the caller's streams — see the solve streams below)
        if (hipStreamCreateWithPriority(&h->early_stream, hipStreamNonBlocking, prio_least) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_asm_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_early_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreate(&h->ev_early_mark) != hipSuccess) {
            dekf_destroy(h);
            return fail(DEKF_ERR_HIP, "could not create the stream of the early marginalisation");
        }
    }
    if (h->pipelined) {
        // The two solve streams are created at the greatest stream priority: HIP maps streams round-robin onto a few hardware queues
        // PER PRIORITY CLASS (GPU_MAX_HW_QUEUES, default 4), and a stream-wait is a barrier packet that holds its whole queue — with
        // one more normal-priority stream alive in the process (another handle, idle) a solve stream shared a queue with the handle's
        // own stream and the mode ran at 1.34 M instead of 2.1 M steps/s (profiles/r05_step_pipelining_queues.txt).  In their own
        // class they share queues with nothing the caller creates by default.
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        for (int i = 0; i < 2 && ok; ++i) {
            ok = hipStreamCreateWithPriority(&h->solve_stream[i], hipStreamNonBlocking, prio_greatest) == hipSuccess &&
                 hipEventCreateWithFlags(&h->ev_asm[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&h->ev_solve[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreate(&h->ev_mark[i]) == hipSuccess;
        }
        for (int i = 0; i < DEKF_SNAP_SETS && ok; ++i) ok = hipEventCreateWithFlags(&h->ev_snap_free[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            dekf_destroy(h);
            return fail(DEKF_ERR_HIP, "could not create the solve streams");
        }
    }
    // the handle is published only once it is usable: a caller that checks the status alone leaks nothing
    const dekf_status st = dekf_reset(h);
    if (st != DEKF_OK) {
        const std::string keep = g_err;
        dekf_destroy(h);
        g_err = keep;
        return st;
    }
    *out = h;
    return DEKF_OK;
}

dekf_status dekf_destroy(dekf_handle h) {
    if (!h) return DEKF_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 2; ++i) {
        if (h->solve_stream[i]) { (void)hipStreamSynchronize(h->solve_stream[i]); (void)hipStreamDestroy(h->solve_stream[i]); }
        if (h->ev_asm[i]) (void)hipEventDestroy(h->ev_asm[i]);
        if (h->ev_solve[i]) (void)hipEventDestroy(h->ev_solve[i]);
        if (h->ev_mark[i]) (void)hipEventDestroy(h->ev_mark[i]);
    }
    for (int i = 0; i < DEKF_SNAP_SETS; ++i) if (h->ev_snap_free[i]) (void)hipEventDestroy(h->ev_snap_free[i]);
    if (h->early_stream) { (void)hipStreamSynchronize(h->early_stream); (void)hipStreamDestroy(h->early_stream); }
    if (h->ev_asm_done) (void)hipEventDestroy(h->ev_asm_done);
    if (h->ev_early_done) (void)hipEventDestroy(h->ev_early_done);
    if (h->ev_early_mark) (void)hipEventDestroy(h->ev_early_mark);
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    if (h->comm) rccl_destroy(h->comm);
    if (h->ev_vb_ready) (void)hipEventDestroy(h->ev_vb_ready);
    if (h->ev_ag_done) (void)hipEventDestroy(h->ev_ag_done);
    for (int i = 0; i < 2; ++i) if (h->ev_vb_read[i]) (void)hipEventDestroy(h->ev_vb_read[i]);
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    if (h->vb_snapshot) (void)hipFree(h->vb_snapshot);
    for (auto& v : h->ev) for (auto& pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : h->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (void* q : h->blocks) (void)hipFree(q);
    if (h->stage) (void)hipFree(h->stage);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return DEKF_OK;
}

dekf_status dekf_reset(dekf_handle h) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    for (int i = 0; i < 2; ++i)   // a solve still in flight writes the outputs this clears
        if (h->solve_pending[i]) { HIPCHK(hipStreamWaitEvent(h->stream, h->ev_solve[i], 0)); h->solve_pending[i] = false; }
    for (int i = 0; i < DEKF_SNAP_SETS; ++i) h->snap_busy[i] = false;  // (every solve is behind the two events just waited for)
    if (h->early_pending) { HIPCHK(hipStreamWaitEvent(h->stream, h->ev_early_done, 0)); h->early_pending = false; }  // (k_reset_state clears its tags)
    for (int i = 0; i < 2; ++i)   // ... and an all-gather's snapshot copy on the communication stream may still be reading v_b
        if (h->vb_read_pending[i]) { HIPCHK(hipStreamWaitEvent(h->stream, h->ev_vb_read[i], 0)); h->vb_read_pending[i] = false; }
    for (int i = 0; i < (h->pipelined ? 2 : 1); ++i) {
        k_reset_state<<<(h->c.B + 255) / 256, 256, 0, h->stream>>>(h->c, h->sp[i]);
        HIPCHK(hipGetLastError());
    }
    h->last_par = 0;
    h->ekf_count = 0;
    h->pushes = 0;
    h->next_T = 0;
    h->initialized = false;
    return DEKF_OK;
}

dekf_status dekf_sync(dekf_handle h) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < 2; ++i)
        if (h->solve_stream[i]) HIPCHK(hipStreamSynchronize(h->solve_stream[i]));
    if (h->comm_stream) HIPCHK(hipStreamSynchronize(h->comm_stream));
    // (the arrival cost of the next step may still be running beside the last solve: dekf.h promises that ALL work is complete here,
    // and an asynchronous failure of that kernel must surface here too)
    if (h->early_stream) HIPCHK(hipStreamSynchronize(h->early_stream));
    return DEKF_OK;
}
int dekf_batch(dekf_handle h) { return h ? h->c.B : 0; }
void* dekf_stream(dekf_handle h) { return h ? (void*)h->stream : nullptr; }

dekf_status dekf_push_imu(dekf_handle h, const double* imu_time, const double* accel_b, const double* gyro_b, dekf_mem where) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    size_t B = h->c.B;
    double* const dst[3] = {h->s.imu_t, h->s.accel, h->s.gyro};
    const double* const src[3] = {imu_time, accel_b, gyro_b};
    const size_t cnt[3] = {B, 3 * B, 3 * B};
    return put_many(h, 3, dst, src, cnt, where);
}

dekf_status dekf_push_leg(dekf_handle h, const double* p_imu_2_foot, const double* J_imu_2_foot, const double* joint_velocity,
                          const double* contact, dekf_mem where) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    size_t B = h->c.B, L = h->c.L, nj = h->c.nj;
    double* const dst[4] = {h->s.p_foot, h->s.J, h->s.qdot, h->s.contact};
    const double* const src[4] = {p_imu_2_foot, J_imu_2_foot, joint_velocity, contact};
    const size_t cnt[4] = {3 * L * B, 3 * L * nj * B, L * nj * B, L * B};
    return put_many(h, 4, dst, src, cnt, where);
}

dekf_status dekf_push_go1_joints(dekf_handle h, const double* joint_position, const double* joint_velocity,
                                 const double* foot_force, dekf_mem where) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    if (h->c.L != 4 || h->c.nj != 3) return fail(DEKF_ERR_INVALID, "dekf_push_go1_joints needs num_legs 4, joints_per_leg 3");
    size_t B = h->c.B;
    const double *jp = joint_position, *jv = joint_velocity, *ff = foot_force;
    if (!jp || !jv || !ff) return fail(DEKF_ERR_INVALID, "null input pointer");
    if (where == DEKF_HOST) {
        dekf_status st = ensure_stage(h, 28 * B * 8);
        if (st) return st;
        double* d = (double*)h->stage;
        HIPCHK(hipMemcpyAsync(d, jp, 12 * B * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d + 12 * B, jv, 12 * B * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d + 24 * B, ff, 4 * B * 8, hipMemcpyHostToDevice, h->stream));
        jp = d; jv = d + 12 * B; ff = d + 24 * B;
    }
    k_go1_leg_odometry<<<(h->c.B + 63) / 64, 64, 0, h->stream>>>(h->c, h->s, jp, jv, ff, h->prm.contact_effort_threshold,
                                                               h->prm.p_ib[0], h->prm.p_ib[1], h->prm.p_ib[2]);
    HIPCHK(hipGetLastError());
    return DEKF_OK;
}

dekf_status dekf_push_vo(dekf_handle h, const int* mask, const double* t_pre, const double* t_now, const double* dp_body,
                         const double* t_pose, const double* q_vo, dekf_mem where) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    if (!mask || !t_pre || !t_now || !dp_body) return fail(DEKF_ERR_INVALID, "null input pointer");
    if ((q_vo == nullptr) != (t_pose == nullptr)) return fail(DEKF_ERR_INVALID, "t_pose and q_vo must be given together");
    size_t B = h->c.B;
    if (where == DEKF_HOST) {
        // mask(int) t_pre t_now dp(3) t_pose q(4): 10 doubles + 1 int per instance
        dekf_status st = ensure_stage(h, (10 * 8 + 8) * B);
        if (st) return st;
        double* d = (double*)h->stage;
        int* dm = (int*)(d + 10 * B);
        HIPCHK(hipMemcpyAsync(dm, mask, B * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d, t_pre, B * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d + B, t_now, B * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d + 2 * B, dp_body, 3 * B * 8, hipMemcpyHostToDevice, h->stream));
        if (q_vo) {
            HIPCHK(hipMemcpyAsync(d + 5 * B, t_pose, B * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(d + 6 * B, q_vo, 4 * B * 8, hipMemcpyHostToDevice, h->stream));
        }
        mask = dm; t_pre = d; t_now = d + B; dp_body = d + 2 * B;
        if (q_vo) { t_pose = d + 5 * B; q_vo = d + 6 * B; }
    }
    k_latch_vo<<<(h->c.B + 255) / 256, 256, 0, h->stream>>>(h->c, h->s, mask, t_pre, t_now, dp_body, t_pose, q_vo);
    HIPCHK(hipGetLastError());
    return DEKF_OK;
}

dekf_status dekf_push_quaternion(dekf_handle h, const double* quat, dekf_mem where) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    return put(h, h->s.quat, quat, 4 * (size_t)h->c.B * 8, where);
}

dekf_status dekf_ekf_step(dekf_handle h) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    {
        Timed t(h, 0);
        k_ekf_tick<<<(h->c.B + 63) / 64, 64, 0, h->stream>>>(h->c, h->s, h->ekf_count);
    }
    HIPCHK(hipGetLastError());
    h->ekf_count++;
    // the kernel uses the count only modulo the ring depth and to know whether the ring is full: folded long before
    // the int overflows (a 500 Hz node reaches 2^31 ticks after 49 days)
    if (h->ekf_count >= (1 << 30)) h->ekf_count = h->c.ekf_hist + h->ekf_count % h->c.ekf_hist;
    return DEKF_OK;
}

dekf_status dekf_initialize(dekf_handle h) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    if (h->initialized) return fail(DEKF_ERR_ORDER, "dekf_initialize called twice without dekf_reset");
    if (h->c.est_type == 0) {
        k_mhe_initialize<<<h->c.B, 64, h->lds_asm, h->stream>>>(h->c, h->s);
        h->pushes = 1;
    } else {
        k_kf_initialize<<<h->c.B, 64, h->lds_kf, h->stream>>>(h->c, h->s);
        h->pushes = 2;
    }
    HIPCHK(hipGetLastError());
    h->last_par = 0;
    h->initialized = true;
    h->next_T = 1;
    return DEKF_OK;
}

dekf_status dekf_update(dekf_handle h, int T) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    if (!h->initialized) return fail(DEKF_ERR_ORDER, "dekf_update before dekf_initialize");
    if (T != h->next_T) return fail(DEKF_ERR_ORDER, "update(T) must be called with T = 1, 2, 3, ... (EstSub.cpp:58-75)");
    if (h->c.est_type == 0) {
        const int par = h->pipelined ? (T & 1) : 0;
        DevState sp = h->sp[par];
        hipStream_t ss = h->pipelined ? h->solve_stream[par] : h->stream;
        const int snap_set = h->pipelined ? T % DEKF_SNAP_SETS : 0;
        if (h->pipelined) {
            // The assemble writes the solve's input snapshot.  With two copies it had to wait for the solve of step T - 2, and by
            // then step T - 1's 768 persistent workgroups hold every slot of the machine: the assemble crawled in (0.41 instead of
            // 0.07 ms) as they left, and the solve of step T started that much late — the pipeline ran on its dependencies, not on
            // the machine (profiles/r05_step_pipelining_timeline.txt).  With three copies it waits for step T - 3, which is long
            // over: the small kernels of a step run while the step before the previous one drains, and the solve of step T is
            // ready the moment a slot frees.  (Outputs and scratch slabs stay at two sets: solves on one stream are in order.)
            sp.snap = h->s.snap + (size_t)snap_set * h->c.snap_len * h->c.B;
            if (h->snap_busy[snap_set]) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_snap_free[snap_set], 0));
        }
        if (h->early_pending) { HIPCHK(hipStreamWaitEvent(h->stream, h->ev_early_done, 0)); h->early_pending = false; }
        {
            Timed t(h, 1);
            k_mhe_assemble<<<h->c.B, 64, h->lds_asm, h->stream>>>(h->c, sp, T, h->pushes);
        }
        HIPCHK(hipGetLastError());
        if (h->early_stream) HIPCHK(hipEventRecord(h->ev_asm_done, h->stream));
        if (h->pipelined) {
            HIPCHK(hipEventRecord(h->ev_asm[par], h->stream));
            HIPCHK(hipStreamWaitEvent(ss, h->ev_asm[par], 0));
            // the all-gather of step T - 2 read THIS set's v_b until its snapshot copy was through (dekf_allgather_vb)
            if (h->vb_read_pending[par]) { HIPCHK(hipStreamWaitEvent(ss, h->ev_vb_read[par], 0)); h->vb_read_pending[par] = false; }
            // A recorded event between the stream-waits and the launch: measured, not understood.  Without it the solve of step
            // T + 1 does not start under the last round of step T's solve and the mode gains nothing (2.09 M steps/s, Go1 at 4096);
            // with it the two overlap (2.16 M).  Found because bench.py's timing events had the same effect; an event recorded only
            // AFTER the launch does not (profiles/r05_step_pipelining_queues.txt).
            HIPCHK(hipEventRecord(h->ev_mark[par], ss));
        }
        int kstart = T - h->c.N + 1 > 0 ? T - h->c.N + 1 : 0;
        {
            Timed t(h, 2, ss);
            const int K = T - kstart + 1;
#ifdef DEKF_AB_KNOBS  // A/B builds only: a quiet gap (an empty kernel) between the term construction and the solve launch
            { static const int gap = getenv("DEKF_DEBUG_GAP_KERNEL") ? atoi(getenv("DEKF_DEBUG_GAP_KERNEL")) : 0;
              for (int i = 0; i < gap; ++i) k_gap<<<1, 64, 0, ss>>>(); }
#endif
            if (h->solve_kernel_full && K == h->c.N)
                h->solve_kernel_full<<<h->solve_grid_full, h->solve_threads_full, h->lds_solve_full, ss>>>(h->c, sp, kstart, K, h->gws_len);
            else
                h->solve_kernel<<<h->solve_grid, DEKF_SOLVE_THREADS, h->lds_solve, ss>>>(h->c, sp, kstart, K, h->gws_len);
        }
        if (h->pipelined) {
            HIPCHK(hipEventRecord(h->ev_solve[par], ss));
            h->solve_pending[par] = true;
            HIPCHK(hipEventRecord(h->ev_snap_free[snap_set], ss));
            h->snap_busy[snap_set] = true;
        }
        h->last_par = par;
        if (h->early_stream) {
            // behind the assemble of this step (the arrival cost and the records as it left them), beside this step's solve
            HIPCHK(hipStreamWaitEvent(h->early_stream, h->ev_asm_done, 0));
            HIPCHK(hipEventRecord(h->ev_early_mark, h->early_stream));  // (as before a pipelined solve launch: see there)
            k_mhe_marginalize_early<<<h->c.B, 64, h->lds_asm, h->early_stream>>>(h->c, h->s, T + 1);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(h->ev_early_done, h->early_stream));
            h->early_pending = true;
        }
    } else {
        Timed t(h, 1);
        k_kf_update<<<h->c.B, 64, h->lds_kf, h->stream>>>(h->c, h->s, h->pushes);
    }
    HIPCHK(hipGetLastError());
    h->pushes++;
    h->next_T++;
    return DEKF_OK;
}

dekf_status dekf_step(dekf_handle h, int T) {
    dekf_status st = dekf_ekf_step(h);
    if (st) return st;
    return T == 0 ? dekf_initialize(h) : dekf_update(h, T);
}

dekf_status dekf_get(dekf_handle h, double* x_mhe, double* v_b, double* quat, double* p_vo, int* status, dekf_mem where) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    size_t B = h->c.B;
    dekf_status st;
    if ((st = await_results(h))) return st;
    const DevState& s = h->sp[h->last_par];
    if ((st = fetch(h, x_mhe, s.x_mhe, (size_t)h->c.ns * B * 8, where))) return st;
    if ((st = fetch(h, v_b, s.v_b, 3 * B * 8, where))) return st;
    if ((st = fetch(h, quat, s.quat, 4 * B * 8, where))) return st;
    if ((st = fetch(h, p_vo, s.p_vo, 3 * B * 8, where))) return st;
    if ((st = fetch(h, status, s.status, B * 4, where))) return st;
    if (where == DEKF_HOST) HIPCHK(hipStreamSynchronize(h->stream));
    return DEKF_OK;
}

dekf_status dekf_get_ekf_cov(dekf_handle h, double* cov, dekf_mem where) {
    if (!h || !cov) return fail(DEKF_ERR_INVALID, "null argument");
    // device layout is [16][B] (field-major, one lane per instance in k_ekf_tick); hand out [B][4][4].  The transpose
    // is a kernel in stream order: with DEKF_DEVICE this call is as asynchronous as every other getter.
    const size_t B = h->c.B;
    double* dst = cov;
    if (where == DEKF_HOST) {
        dekf_status st = ensure_stage(h, 16 * B * sizeof(double));
        if (st) return st;
        dst = (double*)h->stage;
    }
    k_ekf_cov_out<<<(int)((16 * B + 255) / 256), 256, 0, h->stream>>>(h->c, h->s, dst);
    HIPCHK(hipGetLastError());
    if (where == DEKF_HOST) {
        HIPCHK(hipMemcpyAsync(cov, dst, 16 * B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return DEKF_OK;
}

dekf_status dekf_get_solver_info(dekf_handle h, int* iters, int* rho_updates, double* pri_res, double* dua_res, dekf_mem where) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    size_t B = h->c.B;
    dekf_status st;
    if ((st = await_results(h))) return st;
    const DevState& s = h->sp[h->last_par];
    if ((st = fetch(h, iters, s.iters, B * 4, where))) return st;
    if ((st = fetch(h, rho_updates, s.rho_updates, B * 4, where))) return st;
    if ((st = fetch(h, pri_res, s.pri_res, B * 8, where))) return st;
    if ((st = fetch(h, dua_res, s.dua_res, B * 8, where))) return st;
    if (where == DEKF_HOST) HIPCHK(hipStreamSynchronize(h->stream));
    return DEKF_OK;
}

dekf_status dekf_get_polish_status(dekf_handle h, int* polish_status, dekf_mem where) {
    if (!h || !polish_status) return fail(DEKF_ERR_INVALID, "null argument");
    dekf_status st;
    if ((st = await_results(h))) return st;
    if ((st = fetch(h, polish_status, h->sp[h->last_par].polish_status, (size_t)h->c.B * 4, where))) return st;
    if (where == DEKF_HOST) HIPCHK(hipStreamSynchronize(h->stream));
    return DEKF_OK;
}

dekf_status dekf_get_kf_cov(dekf_handle h, double* cov, dekf_mem where) {
    if (!h || !cov) return fail(DEKF_ERR_INVALID, "null argument");
    dekf_status st = fetch(h, cov, h->s.kf_C, (size_t)h->c.ns * h->c.ns * h->c.B * 8, where);
    if (st) return st;
    if (where == DEKF_HOST) HIPCHK(hipStreamSynchronize(h->stream));
    return DEKF_OK;
}

dekf_status dekf_timing_enable(dekf_handle h, int on) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    h->timing = on == 2 ? 2 : (on != 0 ? 1 : 0);
    // events are created here, not inside the timed region: enough pairs for a few hundred steps of three launches
    if (h->timing) {
        size_t have = h->ev_pool.size();
        for (int c = 0; c < DEKF_TIMING_CLASSES; ++c) have += h->ev[c].size();
        for (size_t i = have; i < 3 * 1024; ++i) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
                if (a) (void)hipEventDestroy(a);
                break;
            }
            h->ev_pool.push_back({a, b});
        }
    }
    return DEKF_OK;
}
dekf_status dekf_timing_read(dekf_handle h, double* ms_sum, int* launches) {
    if (!h || !ms_sum || !launches) return fail(DEKF_ERR_INVALID, "null argument");
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < 2; ++i)
        if (h->solve_stream[i]) HIPCHK(hipStreamSynchronize(h->solve_stream[i]));
    if (h->comm_stream && !h->ev[3].empty()) HIPCHK(hipStreamSynchronize(h->comm_stream));
    for (int c = 0; c < DEKF_TIMING_CLASSES; ++c) {
        double sum = 0.0;
        for (auto& pr : h->ev[c]) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
            sum += ms;
            h->ev_pool.push_back(pr);
        }
        ms_sum[c] = sum;
        launches[c] = (int)h->ev[c].size();
        h->ev[c].clear();
    }
    return DEKF_OK;
}

dekf_status dekf_launch_info(dekf_handle h, int* solve_workgroups, int* compute_units, double* clock_hz) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->device));
    if (solve_workgroups) *solve_workgroups = h->solve_kernel_full ? h->solve_grid_full : h->solve_grid;
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_hz) *clock_hz = (double)prop.clockRate * 1e3;
    return DEKF_OK;
}

const char* dekf_solve_kernel_name(dekf_handle h, int full_window) {
    if (!h || h->c.est_type != 0) return nullptr;
    return (full_window && h->solve_kernel_full) ? h->solve_name_full : h->solve_name;
}

// not part of include/dekf.h: section cycles [B][DEKF_PROF_SLOTS] of the last solve; all zero unless this
// library was built with -DDEKF_PROFILE (libdekf_prof.so, tools/profile_sections.py)
dekf_status dekf_debug_sections(dekf_handle h, double* out_host) {
    if (!h || !out_host) return fail(DEKF_ERR_INVALID, "null argument");
    { dekf_status st = await_results(h); if (st) return st; }
    HIPCHK(hipMemcpyAsync(out_host, h->sp[h->last_par].prof, DEKF_PROF_SLOTS * (size_t)h->c.B * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DEKF_OK;
}

#ifdef DEKF_BOUNDS
extern "C" __global__ void k_bounds_selftest(double* a);
dekf_status dekf_debug_bounds_selftest(void) {
    double* a = nullptr;
    HIPCHK(hipMalloc(&a, 32 * sizeof(double)));
    k_bounds_selftest<<<1, 1>>>(a);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipFree(a));
    return DEKF_OK;
}
// not part of include/dekf.h: the bounds-checked diagnostic build (libdekf_bounds.so, wave.h: BPtr).  out[0] = out-of-range
// dereferences since the library was loaded, out[1] = offset of the first one in elements from the start of its array,
// out[2] = that array's extent, out[3] = unused.  Synchronises the device.
dekf_status dekf_debug_bounds_reset(void) {
    const unsigned long long z[4] = {0ull, 0ull, 0ull, 0ull};
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(dekf::dekf_bounds_hits), z, sizeof(z)));
    return DEKF_OK;
}
dekf_status dekf_debug_bounds(unsigned long long* out4) {
    if (!out4) return fail(DEKF_ERR_INVALID, "null argument");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out4, HIP_SYMBOL(dekf::dekf_bounds_hits), 4 * sizeof(unsigned long long)));
    return DEKF_OK;
}
#endif

// ---------------------------------------------------------------- RCCL all-gather
dekf_status dekf_comm_unique_id(void* id_out) {
    if (!id_out) return fail(DEKF_ERR_INVALID, "null argument");
    const char* e = rccl_unique_id(id_out);
    if (e) return fail(DEKF_ERR_COMM, e);
    return DEKF_OK;
}
dekf_status dekf_comm_init(dekf_handle h, int world, int rank, const void* id) {
    if (!h || !id || world < 1 || rank < 0 || rank >= world) return fail(DEKF_ERR_INVALID, "bad communicator arguments");
    if (h->comm) return fail(DEKF_ERR_ORDER, "dekf_comm_init was already called on this handle");
    HIPCHK(hipSetDevice(h->device));
    // Everything the all-gather needs besides the communicator comes first and the communicator is assigned LAST, so
    // a failure on the way leaves the handle exactly as it was (h->comm == nullptr: dekf_allgather_vb refuses, a retry
    // is accepted) instead of half-initialised.
    hipStream_t cs = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    double* snap = nullptr;
    auto rollback = [&] {
        if (snap) (void)hipFree(snap);
        if (e3) (void)hipEventDestroy(e3);
        if (e2) (void)hipEventDestroy(e2);
        if (e1) (void)hipEventDestroy(e1);
        if (e0) (void)hipEventDestroy(e0);
        if (cs) (void)hipStreamDestroy(cs);
    };
    hipError_t he = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e0, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e3, hipEventDisableTiming);
    if (he == hipSuccess) he = hipMalloc((void**)&snap, 3 * (size_t)h->c.B * sizeof(double));
    if (he != hipSuccess) {
        rollback();
        g_err = std::string("dekf_comm_init: ") + hipGetErrorString(he);
        return DEKF_ERR_HIP;
    }
    void* comm = nullptr;
    const char* e = rccl_init_rank(&comm, world, rank, id);  // collective: every rank of `world` must be here
    if (e) {
        rollback();
        return fail(DEKF_ERR_COMM, e);
    }
    h->comm_stream = cs;
    h->ev_vb_ready = e0;
    h->ev_ag_done = e1;
    h->ev_vb_read[0] = e2;
    h->ev_vb_read[1] = e3;
    h->vb_read_pending[0] = h->vb_read_pending[1] = false;
    h->vb_snapshot = snap;
    h->world = world;
    h->rank = rank;
    h->ag_pending = false;
    h->comm = comm;
    return DEKF_OK;
}
dekf_status dekf_allgather_vb(dekf_handle h, double* v_b_all_dev) {
    if (!h || !v_b_all_dev) return fail(DEKF_ERR_INVALID, "null argument");
    if (!h->comm) return fail(DEKF_ERR_ORDER, "dekf_comm_init has not been called");
    HIPCHK(hipSetDevice(h->device));
    const size_t n = 3 * (size_t)h->c.B;
    if (h->pipelined) {
        // Everything on the communication stream, which is in order (the previous all-gather has read the snapshot): wait for the
        // solve that produced v_b, copy it, exchange.  The handle's stream is not involved, so the next step's pushes, EKF tick and
        // assemble are not held back; the solve that will overwrite this set's v_b (two steps on) waits for ev_vb_read (dekf_update).
        if (h->solve_pending[h->last_par]) HIPCHK(hipStreamWaitEvent(h->comm_stream, h->ev_solve[h->last_par], 0));
        else {  // (T = 0: no solve yet; v_b is what the reset / initialise kernels on the handle's stream left)
            HIPCHK(hipEventRecord(h->ev_vb_ready, h->stream));
            HIPCHK(hipStreamWaitEvent(h->comm_stream, h->ev_vb_ready, 0));
        }
        HIPCHK(hipMemcpyAsync(h->vb_snapshot, h->sp[h->last_par].v_b, n * sizeof(double), hipMemcpyDeviceToDevice, h->comm_stream));
        HIPCHK(hipEventRecord(h->ev_vb_read[h->last_par], h->comm_stream));
        h->vb_read_pending[h->last_par] = true;
    } else {
    // the previous all-gather must have read the snapshot before it is overwritten (it finished a step ago)
    if (h->ag_pending) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ag_done, 0));
    HIPCHK(hipMemcpyAsync(h->vb_snapshot, h->s.v_b, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipEventRecord(h->ev_vb_ready, h->stream));
    HIPCHK(hipStreamWaitEvent(h->comm_stream, h->ev_vb_ready, 0));
    }
    {
        Timed t(h, 3, h->comm_stream);
        const char* e = rccl_allgather_f64(h->comm, h->vb_snapshot, v_b_all_dev, n, h->comm_stream);
        if (e) return fail(DEKF_ERR_COMM, e);
    }
    HIPCHK(hipEventRecord(h->ev_ag_done, h->comm_stream));
    h->ag_pending = true;
    return DEKF_OK;
}
dekf_status dekf_comm_info(dekf_handle h, int* comm_world, int* comm_rank) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    if (!h->comm) return fail(DEKF_ERR_ORDER, "dekf_comm_init has not been called");
    int cw = 0, cr = -1;
    const char* e = rccl_comm_info(h->comm, &cw, &cr);
    if (e) return fail(DEKF_ERR_COMM, e);
    if (comm_world) *comm_world = cw;
    if (comm_rank) *comm_rank = cr;
    return DEKF_OK;
}
dekf_status dekf_comm_ranks_seen(dekf_handle h, int* ranks_seen) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    if (!h->comm) return fail(DEKF_ERR_ORDER, "dekf_comm_init has not been called");
    HIPCHK(hipSetDevice(h->device));
    // one double per rank through the very communicator and stream the v_b exchange uses
    double* buf = nullptr;
    HIPCHK(hipMalloc((void**)&buf, (size_t)(h->world + 1) * sizeof(double)));
    const double mine = (double)h->rank;
    hipError_t he = hipMemcpyAsync(buf + h->world, &mine, sizeof(double), hipMemcpyHostToDevice, h->comm_stream);
    if (he == hipSuccess) he = hipStreamSynchronize(h->comm_stream);  // (`mine` is a stack variable)
    const char* e = he == hipSuccess ? rccl_allgather_f64(h->comm, buf + h->world, buf, 1, h->comm_stream) : nullptr;
    std::vector<double> got((size_t)h->world, -1.0);
    if (he == hipSuccess && !e) he = hipMemcpyAsync(got.data(), buf, (size_t)h->world * sizeof(double), hipMemcpyDeviceToHost, h->comm_stream);
    if (he == hipSuccess && !e) he = hipStreamSynchronize(h->comm_stream);
    (void)hipFree(buf);
    if (e) return fail(DEKF_ERR_COMM, e);
    if (he != hipSuccess) { g_err = std::string("dekf_comm_ranks_seen: ") + hipGetErrorString(he); return DEKF_ERR_HIP; }
    std::vector<char> seen((size_t)h->world, 0);
    int n = 0;
    for (int i = 0; i < h->world; ++i) {
        const int r = (int)got[(size_t)i];
        if (got[(size_t)i] == (double)r && r >= 0 && r < h->world && got[(size_t)i] == (double)i && !seen[(size_t)r]) { seen[(size_t)r] = 1; ++n; }  // slot i must hold rank i
    }
    if (ranks_seen) *ranks_seen = n;
    return DEKF_OK;
}
dekf_status dekf_allgather_wait(dekf_handle h) {
    if (!h) return fail(DEKF_ERR_INVALID, "null handle");
    if (h->ag_pending) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ag_done, 0));
    return DEKF_OK;
}

}  // extern "C"
