// wave.h — the execution model the estimator cores are written against.
//
// One workgroup (1..4 wavefronts of 64 lanes) owns one robot instance.  A core is a sequence of
//   wfor(n, f)        the group's lanes split i = 0..n-1 between them, then a group-level sync
//   wred_*(n, f)      group-wide reduction of f(i); every lane receives the result
//   group-uniform scalar code in between (every lane computes the same value).
// All state a phase hands to the next one lives in memory (LDS or HBM), never in a lane's
// registers, so a phase boundary is exactly one sync.  With a single wavefront per group the
// sync is free (hipcc drops s_barrier for a 64-thread workgroup); with several it is s_barrier.
//
// The same source builds in two ways:
//   hipcc (gfx950)          lanes are real: strided loops + DPP/shuffle reductions + s_barrier
//   g++ -DDEKF_HOSTSIM      lanes are run one after the other.  This build exists ONLY for
//                           tests/hostsim (CPU container has no GPU; sanitizers run here); it is
//                           not reachable from the C ABI and is not a fallback.
#pragma once
#include <cmath>
#include <type_traits>

#if defined(__HIPCC__) && !defined(DEKF_HOSTSIM)
#include <hip/hip_runtime.h>
#define DEKF_DEVICE_BUILD 1
#define DEKF_FN __device__ __forceinline__
#define DEKF_HD __host__ __device__ __forceinline__
// The lane id goes through an empty volatile asm at every use: the compiler then cannot hoist the
// per-lane address arithmetic of a phase out of the ADMM iteration loop.  Hoisted, those values
// stayed live across every phase, pushed the kernel over its 256-register budget and came back as
// scratch (HBM-backed) reloads at the head of the sweeps; recomputing them costs a few VALU ops.
static __device__ __forceinline__ int dekf_lane_id() {
    int v = (int)threadIdx.x;
    asm volatile("" : "+v"(v));
    return v;
}
#define DEKF_LANE() dekf_lane_id()
#define DEKF_NLANES() ((int)blockDim.x)
#define DEKF_SYNC() __syncthreads()
#else
#define DEKF_DEVICE_BUILD 0
#define DEKF_FN inline
#define DEKF_HD inline
#define DEKF_LANE() 0
#define DEKF_NLANES() 1
#define DEKF_SYNC() ((void)0)
#endif

namespace dekf {

// ---------------------------------------------------------------- pointers of the solve cores
// The solve cores (mhe_solve_core.h, mhe_admm_core.h) address LDS, the workgroup's slab and the window records through `dptr` /
// `cdptr`.  In every product build these ARE double* / const double*.  The -DDEKF_BOUNDS build (csrc/libdekf_bounds.so, a
// diagnostic variant like libdekf_prof.so) makes them fat pointers that carry the extent of the array they were carved from:
// every dereference outside [lo, hi) is counted in a device-side counter, redirected to the array's first element (so the kernel
// neither faults nor corrupts a neighbour) and the instance's status becomes DEKF_SOLVE_NUMERIC.  GPU AddressSanitizer is not
// available on this pool; this is the substitute for the device-only code the lane-sequential host build cannot execute.
#if defined(DEKF_BOUNDS) && DEKF_DEVICE_BUILD
extern __device__ unsigned long long dekf_bounds_hits[4];  // [0] count, [1] first offending offset (elements from lo), [2] extent, [3] line
static __device__ __forceinline__ void dekf_bounds_report(long off, long n, int line) {
    if (atomicAdd(&dekf_bounds_hits[0], 1ull) == 0ull) {
        dekf_bounds_hits[1] = (unsigned long long)off;
        dekf_bounds_hits[2] = (unsigned long long)n;
        dekf_bounds_hits[3] = (unsigned long long)line;
    }
}
template <class T>
struct BPtr {
    T* p;
    T* lo;
    T* hi;  // lo == nullptr: unchecked (a pointer to a lane's own local array, or one that came in as a raw pointer)
    DEKF_FN BPtr() : p(nullptr), lo(nullptr), hi(nullptr) {}
    DEKF_FN BPtr(T* raw) : p(raw), lo(nullptr), hi(nullptr) {}
    DEKF_FN BPtr(decltype(nullptr)) : p(nullptr), lo(nullptr), hi(nullptr) {}
    DEKF_FN BPtr(T* p_, T* lo_, T* hi_) : p(p_), lo(lo_), hi(hi_) {}
    template <class U, class = decltype(static_cast<T*>((U*)nullptr))>
    DEKF_FN BPtr(const BPtr<U>& o) : p(o.p), lo(o.lo), hi(o.hi) {}
    DEKF_FN T& at(long i, int line = 0) const {
        T* a = p + i;
        if (lo && (a < lo || a >= hi)) { dekf_bounds_report((long)(a - lo), (long)(hi - lo), line); a = lo; }
        return *a;
    }
    template <class I> DEKF_FN T& operator[](I i) const { return at((long)i); }
    DEKF_FN T& operator*() const { return at(0); }
    template <class I> DEKF_FN BPtr operator+(I i) const { return BPtr(p + i, lo, hi); }
    template <class I> DEKF_FN BPtr operator-(I i) const { return BPtr(p - i, lo, hi); }
    template <class I> DEKF_FN BPtr& operator+=(I i) { p += i; return *this; }
    template <class U> DEKF_FN long operator-(const BPtr<U>& o) const { return (long)(p - o.p); }
    DEKF_FN explicit operator bool() const { return p != nullptr; }
    template <class U> DEKF_FN bool operator==(const BPtr<U>& o) const { return p == o.p; }
    template <class U> DEKF_FN bool operator!=(const BPtr<U>& o) const { return p != o.p; }
    DEKF_FN T* raw() const { return p; }
};
using dptr = BPtr<double>;
using cdptr = BPtr<const double>;
// the array of n doubles at raw pointer q
#define DEKF_SPAN(q_, n_) dekf::dptr((q_), (q_), (q_) + (n_))
#define DEKF_CSPAN(q_, n_) dekf::cdptr((q_), (q_), (q_) + (n_))
template <class T> DEKF_FN T* raw_of(const BPtr<T>& b) { return b.p; }
#else
using dptr = double*;
using cdptr = const double*;
#define DEKF_SPAN(q_, n_) (q_)
#define DEKF_CSPAN(q_, n_) (q_)
#endif
template <class T> DEKF_FN T* raw_of(T* p) { return p; }

constexpr int WAVE = 64;
constexpr int MAX_WAVES = 4;  // per workgroup / instance

#if DEKF_DEVICE_BUILD
template <class F>
DEKF_FN void wfor_nosync(int n, F f) {
    const int st = DEKF_NLANES();
    for (int i = DEKF_LANE(); i < n; i += st) f(i);
}
template <class F>
DEKF_FN void wfor(int n, F f) {
    const int st = DEKF_NLANES();
    for (int i = DEKF_LANE(); i < n; i += st) f(i);
    DEKF_SYNC();
}
// broadcast lane `lane`'s value of a double to the whole wavefront (two v_readlane_b32 into an SGPR pair)
DEKF_FN double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
// Wave-wide maximum, every lane gets it.  DPP all the way (row_shr 1, 2, 4, 8 inside the 16-lane rows, row_bcast 15 / 31 across
// them, then lane 63 read back): 20 instructions per value.  The __shfl_xor butterfly costs ~20 instructions PER STEP and value
// (lane-index arithmetic around two ds_bpermute): 14 maxima took 13.7 k cycles per residual check, 4.5 % of a solve (measured by
// repeating the reduction).  A maximum does not depend on the order, so the result is bit-identical.
template <int CTRL, int ROW_MASK>
DEKF_FN double dpp_max_step(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);  // lanes without a source keep their own value
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return fmax(v, __hiloint2double(hi, lo));
}
DEKF_FN double wave_max(double v) {
    v = dpp_max_step<0x111, 0xF>(v);  // row_shr:1
    v = dpp_max_step<0x112, 0xF>(v);  // row_shr:2
    v = dpp_max_step<0x114, 0xF>(v);  // row_shr:4
    v = dpp_max_step<0x118, 0xF>(v);  // row_shr:8   -> lane 15 of every row holds the row's maximum
    v = dpp_max_step<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
    return readlane_f64(v, 63);
}
// Wave-wide sum by DPP (row_shr 1, 2, 4, 8 inside the rows, row_bcast 15 / 31 across them; a lane without a source adds 0), every
// lane gets lane 63's total.  Deterministic, but not the association of the butterfly in wave_sum below.
template <int CTRL, int ROW_MASK>
DEKF_FN double dpp_add_step(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return v + __hiloint2double(hi, lo);
}
DEKF_FN double wave_sum_dpp(double v) {
    v = dpp_add_step<0x111, 0xF>(v);
    v = dpp_add_step<0x112, 0xF>(v);
    v = dpp_add_step<0x114, 0xF>(v);
    v = dpp_add_step<0x118, 0xF>(v);
    v = dpp_add_step<0x142, 0xA>(v);
    v = dpp_add_step<0x143, 0xC>(v);
    return readlane_f64(v, 63);
}
DEKF_FN double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
// NR maxima at once, all lanes get the results
template <int NR>
DEKF_FN void wave_max_n(double* v) {
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = wave_max(v[r]);  // (independent DPP chains: the scheduler interleaves them)
}
// combine NR per-wave partials across the group's wavefronts; every lane gets the result
template <int NR, bool SUM>
DEKF_FN void group_combine(double* v) {
    if (DEKF_NLANES() <= WAVE) return;  // group-uniform
    __shared__ double red[NR * MAX_WAVES];
    const int w = DEKF_LANE() >> 6, nw = DEKF_NLANES() >> 6;
    DEKF_SYNC();  // a previous use of `red` has been read by everyone
    if ((DEKF_LANE() & 63) == 0)
        for (int r = 0; r < NR; ++r) red[r * MAX_WAVES + w] = v[r];
    DEKF_SYNC();
    for (int r = 0; r < NR; ++r) {
        double a = red[r * MAX_WAVES];
        for (int i = 1; i < nw; ++i) a = SUM ? a + red[r * MAX_WAVES + i] : fmax(a, red[r * MAX_WAVES + i]);
        v[r] = a;
    }
}
template <class F>
DEKF_FN double wred_max(int n, F f) {
    double v = 0.0;
    const int st = DEKF_NLANES();
    for (int i = DEKF_LANE(); i < n; i += st) v = fmax(v, f(i));
    v = wave_max(v);
    group_combine<1, false>(&v);
    return v;
}
template <class F>
DEKF_FN double wred_sum(int n, F f) {
    double v = 0.0;
    const int st = DEKF_NLANES();
    for (int i = DEKF_LANE(); i < n; i += st) v += f(i);
    v = wave_sum(v);
    group_combine<1, true>(&v);
    return v;
}
// several maxima at once: f(i, acc) updates acc[0..NR); all lanes get the reduced values
template <int NR, class F>
DEKF_FN void wred_maxn(int n, double* out, F f) {
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
    const int st = DEKF_NLANES();
    for (int i = DEKF_LANE(); i < n; i += st) f(i, acc);
    wave_max_n<NR>(acc);
    group_combine<NR, false>(acc);
#pragma unroll
    for (int r = 0; r < NR; ++r) out[r] = acc[r];
}
// index of the largest f(i) (ties: lowest index), f(i) >= 0
template <class F>
DEKF_FN int wred_argmax(int n, F f, double* best_out) {
    double best = -1.0;
    int bi = 0x7fffffff;
    const int st = DEKF_NLANES();
    for (int i = DEKF_LANE(); i < n; i += st) {
        double v = f(i);
        if (v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(best, o, WAVE);
        int oi = __shfl_xor(bi, o, WAVE);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (DEKF_NLANES() > WAVE) {
        __shared__ double rb[MAX_WAVES];
        __shared__ int ri[MAX_WAVES];
        const int w = DEKF_LANE() >> 6, nw = DEKF_NLANES() >> 6;
        DEKF_SYNC();
        if ((DEKF_LANE() & 63) == 0) { rb[w] = best; ri[w] = bi; }
        DEKF_SYNC();
        best = rb[0]; bi = ri[0];
        for (int i = 1; i < nw; ++i)
            if (rb[i] > best || (rb[i] == best && ri[i] < bi)) { best = rb[i]; bi = ri[i]; }
    }
    if (best_out) *best_out = best;
    return bi;
}
#else
template <class F>
inline void wfor_nosync(int n, F f) {
    for (int i = 0; i < n; ++i) f(i);
}
template <class F>
inline void wfor(int n, F f) {
    for (int i = 0; i < n; ++i) f(i);
}
template <class F>
inline double wred_max(int n, F f) {
    double v = 0.0;
    for (int i = 0; i < n; ++i) v = std::fmax(v, f(i));
    return v;
}
template <class F>
inline double wred_sum(int n, F f) {
    double v = 0.0;
    for (int i = 0; i < n; ++i) v += f(i);
    return v;
}
template <int NR, class F>
inline void wred_maxn(int n, double* out, F f) {
    double acc[NR];
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
    for (int i = 0; i < n; ++i) f(i, acc);
    for (int r = 0; r < NR; ++r) out[r] = acc[r];
}
inline double wave_max(double v) { return v; }   // one sequential "lane" has seen every item already
template <int NR>
inline void wave_max_n(double*) {}
inline double wave_sum(double v) { return v; }
template <int NR, bool SUM>
inline void group_combine(double*) {}
template <class F>
inline int wred_argmax(int n, F f, double* best_out) {
    double best = -1.0;
    int bi = 0x7fffffff;
    for (int i = 0; i < n; ++i) {
        double v = f(i);
        if (v > best) { best = v; bi = i; }
    }
    if (best_out) *best_out = best;
    return bi;
}
#endif

// Serial sections run by the first wavefront alone (the other wavefronts skip to the next group
// sync): inside `if (DEKF_IN_WAVE0()) { ... }`, w0for(n, f) runs f(0..n-1) on n <= 64 lanes and
// orders its LDS traffic against the next w0for with a wave-level fence — no s_barrier.
#if DEKF_DEVICE_BUILD
#define DEKF_IN_WAVE0() (threadIdx.x < 64)
DEKF_FN void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <class F>
DEKF_FN void w0for(int n, F f) {
    for (int i = DEKF_LANE(); i < n; i += WAVE) f(i);
    wave_sync();
}
#else
#define DEKF_IN_WAVE0() true
inline void wave_sync() {}
template <class F>
inline void w0for(int n, F f) {
    for (int i = 0; i < n; ++i) f(i);
}
#endif

// Two independent serial sections, one on wavefront 0 and one on wavefront 1 of the group
// (sequentially when the group has a single wavefront, and in the lane-sequential host build).
#if DEKF_DEVICE_BUILD
template <class F0, class F1>
DEKF_FN void two_waves(F0 f0, F1 f1) {
    const int w = __builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6);  // wave-uniform: scalar branches
    if (DEKF_NLANES() <= WAVE) { f0(); f1(); }
    else if (w == 0) f0();
    else if (w == 1) f1();
}
#else
template <class F0, class F1>
inline void two_waves(F0 f0, F1 f1) {
    f0();
    f1();
}
#endif

// Wave-granular work split: `ntiles` tiles of up to 64 lanes; a tile is executed by ONE wavefront
// (tile t by wavefront t mod #wavefronts), so a branch on the tile index is wave-uniform (scalar) and
// every wavefront runs ONE straight-line body per tile.  f(tile, lane) with lane = 0..63 is called
// for all 64 lanes of the wavefront (cross-lane reads inside f are legal).  No trailing sync.
DEKF_FN int wave_count() { return DEKF_NLANES() > WAVE ? DEKF_NLANES() >> 6 : 1; }
#if DEKF_DEVICE_BUILD
// exchange a double with the partner lane of an adjacent lane pair (DPP quad_perm [1, 0, 3, 2]): no LDS, no barrier
DEKF_FN double pair_swap(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
#endif
#if DEKF_DEVICE_BUILD
template <class F>
DEKF_FN void wtiles(int ntiles, F f) {
    const int nw = DEKF_NLANES() > WAVE ? DEKF_NLANES() >> 6 : 1;
    const int lane = DEKF_LANE() & 63;
    for (int t = __builtin_amdgcn_readfirstlane(DEKF_LANE() >> 6); t < ntiles; t += nw) f(t, lane);  // t is scalar
}
#else
template <class F>
inline void wtiles(int ntiles, F f) {
    for (int t = 0; t < ntiles; ++t)
        for (int lane = 0; lane < WAVE; ++lane) f(t, lane);
}
#endif

// Loads of data that is read once per solve (window records, the solve's input snapshot).  Only the RESIDUAL CHECKS' reads carry the
// non-temporal hint (ld_stream_resid): the line is then marked for early eviction in L2 instead of pushing out the workgroup's slab
// lines, which are written once per factorisation and read back a few times per solve — measured -6 % HBM-side traffic at the same
// solve time.  The same hint on EVERY record read (staging included) was measured too: -3 % more traffic, +1.3 % solve time, not kept
// (EXPERIMENTS.md II §6) — so ld_stream is a plain load; the name marks the once-per-solve reads.
#if DEKF_DEVICE_BUILD
template <class P>
DEKF_FN double ld_stream(P p, int i) { return p[i]; }
template <class P>
DEKF_FN double ld_stream_resid(P p, int i) { return __builtin_nontemporal_load(&p[i]); }
#else
template <class P>
inline double ld_stream(P p, int i) { return p[i]; }
template <class P>
inline double ld_stream_resid(P p, int i) { return p[i]; }
#endif

// A wave-uniform double pinned into an SGPR pair (two v_readfirstlane).  The f64 pipeline is vector-only, so a uniform value that
// is COMPUTED (dt^2 / 2, 1000 rho, 1 - alpha, the cost scaling c, ...) lives in a VGPR pair, and the compiler hoists it to the
// top of the kernel and keeps it there: at 168 VGPRs (three workgroups per CU) a dozen of them were most of what the register
// allocator spilled — each spill slot a 512-byte wave access to scratch that no longer fits L2 at that residency.  From an SGPR
// pair a VALU instruction takes the value as a scalar operand.  The value is unchanged; the host build is the identity.
#if DEKF_DEVICE_BUILD
DEKF_FN double uni(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}
// (Pinning the read-back to the definition with inline assembly — so that the compiler cannot sink it to the uses and keep the
// VGPR pair alive — was tried both ways and does not work: hand-written v_readfirstlane inside the statement gave wrong results and
// erratic timing (inline assembly gets no hazard handling), an empty statement with "+s" constraints behind the intrinsic ends in
// "illegal VGPR to SGPR copy" wherever the compiler has already decided to keep that value's chain on the vector side.)
DEKF_FN double uni_pin(double v) { return uni(v); }
#else
inline double uni(double v) { return v; }
inline double uni_pin(double v) { return v; }
#endif

// 1/x and 1/sqrt(x) to double precision without the IEEE division / square-root sequences (about 25 and 65
// instructions on gfx950): hardware estimate (v_rcp_f64 / v_rsq_f64) plus two Newton steps.  Last-bit
// differences against 1.0 / x and 1.0 / sqrt(x) are possible; the host build uses those.
#if DEKF_DEVICE_BUILD
DEKF_FN double rcp_fast(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}
DEKF_FN double rsqrt_fast(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
    y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
    return y;
}
#else
inline double rcp_fast(double x) { return 1.0 / x; }
inline double rsqrt_fast(double x) { return 1.0 / std::sqrt(x); }
#endif

// a x + b y and the ADMM relaxation alpha v + (1 - alpha) w with the roundings pinned: ONE product rounded, then one fused
// multiply-add.  Left to -ffp-contract=fast the compiler picks which of the two products it fuses, and it picks differently in
// different instantiations of the same expression (measured: the three-workgroup and the two-workgroup solve kernels drifted
// apart by an ulp per iteration when their row loops were restructured).
DEKF_FN double lin2(double a, double x, double b, double y) { return fma(a, x, b * y); }
DEKF_FN double relax(double alpha, double v, double w) { return fma(alpha, v, uni(1.0 - alpha) * w); }
// a0 b0 + a1 b1 + a2 b2 as ONE rounded product and two fused multiply-adds, in this order: the form every shape of the row phase uses
// (mhe_admm_core.h is compiled with contraction off, so what is not written as fma() is not fused)
DEKF_FN double dot3(double a0, double b0, double a1, double b1, double a2, double b2) { return fma(a2, b2, fma(a1, b1, a0 * b0)); }
DEKF_FN double dmax(double a, double b) { return a > b ? a : b; }
DEKF_FN double dmin(double a, double b) { return a < b ? a : b; }

}  // namespace dekf
