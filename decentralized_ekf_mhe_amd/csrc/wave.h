// wave.h — the execution model the estimator cores are written against.
//
// One 64-lane wavefront owns one robot instance.  A core is a sequence of
//   wfor(n, f)        lanes split i = 0..n-1 between them, then a wave-level sync
//   wred_*(n, f)      wave-wide reduction of f(i)
//   wave-uniform scalar code in between (every lane computes the same value).
// All state a phase hands to the next one lives in memory (LDS or HBM), never in a lane's
// registers, so a phase boundary is exactly one sync.
//
// The same source builds in two ways:
//   hipcc (gfx950)          lanes are real: strided loops + DPP/shuffle reductions + s_barrier
//   g++ -DDEKF_HOSTSIM      lanes are run one after the other.  This build exists ONLY for
//                           tests/hostsim (CPU container has no GPU; sanitizers run here); it is
//                           not reachable from the C ABI and is not a fallback.
#pragma once
#include <cmath>

#if defined(__HIPCC__) && !defined(DEKF_HOSTSIM)
#include <hip/hip_runtime.h>
#define DEKF_DEVICE_BUILD 1
#define DEKF_FN __device__ __forceinline__
#define DEKF_HD __host__ __device__ __forceinline__
#define DEKF_LANE() ((int)(threadIdx.x & 63))
// block == one wavefront, so the workgroup barrier is a wave barrier + LDS/VMEM drain
#define DEKF_SYNC() __syncthreads()
#else
#define DEKF_DEVICE_BUILD 0
#define DEKF_FN inline
#define DEKF_HD inline
#define DEKF_LANE() 0
#define DEKF_SYNC() ((void)0)
#endif

namespace dekf {

constexpr int WAVE = 64;

#if DEKF_DEVICE_BUILD
template <class F>
DEKF_FN void wfor_nosync(int n, F f) {
    for (int i = DEKF_LANE(); i < n; i += WAVE) f(i);
}
template <class F>
DEKF_FN void wfor(int n, F f) {
    for (int i = DEKF_LANE(); i < n; i += WAVE) f(i);
    DEKF_SYNC();
}
DEKF_FN double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
    return v;
}
DEKF_FN double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
template <class F>
DEKF_FN double wred_max(int n, F f) {
    double v = 0.0;
    for (int i = DEKF_LANE(); i < n; i += WAVE) v = fmax(v, f(i));
    return wave_max(v);
}
template <class F>
DEKF_FN double wred_sum(int n, F f) {
    double v = 0.0;
    for (int i = DEKF_LANE(); i < n; i += WAVE) v += f(i);
    return wave_sum(v);
}
// several maxima at once: f(i, acc) updates acc[0..NR); all lanes get the reduced values
template <int NR, class F>
DEKF_FN void wred_maxn(int n, double* out, F f) {
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
    for (int i = DEKF_LANE(); i < n; i += WAVE) f(i, acc);
#pragma unroll
    for (int r = 0; r < NR; ++r) out[r] = wave_max(acc[r]);
}
// index of the largest f(i) (ties: lowest index), f(i) >= 0
template <class F>
DEKF_FN int wred_argmax(int n, F f, double* best_out) {
    double best = -1.0;
    int bi = 0x7fffffff;
    for (int i = DEKF_LANE(); i < n; i += WAVE) {
        double v = f(i);
        if (v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(best, o, WAVE);
        int oi = __shfl_xor(bi, o, WAVE);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
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

DEKF_FN double dmax(double a, double b) { return a > b ? a : b; }
DEKF_FN double dmin(double a, double b) { return a < b ? a : b; }

}  // namespace dekf
