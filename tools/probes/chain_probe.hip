// What does one step of the block-tridiagonal chain (9x9 mat-vec through v_readlane) cost on gfx950,
// and which part of it?  One workgroup, NW wavefronts running the same chain; variants drop one
// ingredient at a time.  hipcc --offload-arch=gfx950 -O3 -o chain_probe chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// MODE 0 full step | 1 no LDS store | 2 W kept in registers (no LDS loads) | 3 no readlane (own value)
// 4 prefetch next W before the arithmetic | 5 LDS broadcast instead of readlane (store v, wave fence, 9 loads)
template <int MODE>
__global__ void __launch_bounds__(256) chain(double* out, long long* cyc, int reps, int steps) {
    extern __shared__ double lds[];
    double* Wk = lds;               // 20 x 81
    double* xs = lds + 20 * 81;     // 4 waves x 20 x 9
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < 20 * 81; e += blockDim.x) Wk[e] = 1e-3 * ((e * 7) % 13 - 6);
    for (int e = tid; e < 4 * 180; e += blockDim.x) xs[e] = 1.0 + 1e-3 * e;
    __syncthreads();
    double* x = xs + wv * 180;
    const int i = lane < 9 ? lane : 8;
    const bool act = lane < 9;
    double v = x[i];
    double wreg[9];
    for (int t = 0; t < 9; ++t) wreg[t] = Wk[9 * i + t];
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        double wn[9];
        if (MODE == 4)
            for (int t = 0; t < 9; ++t) wn[t] = Wk[9 * i + t];
        for (int s = 1; s <= steps; ++s) {
            const double* W = Wk + s * 81 + 9 * i;
            double w[9], vt[9];
            if (MODE == 2) { for (int t = 0; t < 9; ++t) w[t] = wreg[t]; }
            else if (MODE == 4) {
                for (int t = 0; t < 9; ++t) w[t] = wn[t];
                const double* W2 = Wk + ((s + 1) % 20) * 81 + 9 * i;
                for (int t = 0; t < 9; ++t) wn[t] = W2[t];
            } else { for (int t = 0; t < 9; ++t) w[t] = W[t]; }
            const double rhs = x[9 * s + i];
            if (MODE == 3) { for (int t = 0; t < 9; ++t) vt[t] = v; }
            else if (MODE == 5) {
                if (act) x[i] = v;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                for (int t = 0; t < 9; ++t) vt[t] = x[t];
            } else { for (int t = 0; t < 9; ++t) vt[t] = readlane_f64(v, t); }
            double a0 = rhs - w[0] * vt[0];
            double a1 = w[1] * vt[1], a2 = w[2] * vt[2];
            a0 -= w[3] * vt[3]; a1 += w[4] * vt[4]; a2 += w[5] * vt[5];
            a0 -= w[6] * vt[6]; a1 += w[7] * vt[7]; a2 += w[8] * vt[8];
            v = a0 - (a1 + a2);
            if (MODE != 1 && act) x[9 * s + i] = v;
        }
    }
    long long t1 = clock64();
    out[tid] = v;
    if (lane == 0) cyc[wv] = t1 - t0;
}

// Two-set software pipeline (W and rhs of step s+1 requested before the arithmetic of step s).
// ORDER 0: compiler's own order | 1: readlanes and FMAs interleaved level by level (sched_barrier)
// MASK: operand loads only on the 9 active lanes
template <int ORDER, bool MASK>
__global__ void __launch_bounds__(256) chain_pipe(double* out, long long* cyc, int reps, int steps) {
    extern __shared__ double lds[];
    double* Wk = lds;
    double* xs = lds + 20 * 81;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < 20 * 81; e += blockDim.x) Wk[e] = 1e-3 * ((e * 7) % 13 - 6);
    for (int e = tid; e < 4 * 180; e += blockDim.x) xs[e] = 1.0 + 1e-3 * e;
    __syncthreads();
    double* x = xs + wv * 180;
    const int i = lane < 9 ? lane : 8;
    const bool act = lane < 9;
    double v = x[i];
    struct Ops { double w[9], rhs; };
    auto load = [&](int s, Ops& o) {
        const double* W = Wk + s * 81 + 9 * i;
        if (!MASK || act) {
#pragma unroll
            for (int t = 0; t < 9; ++t) o.w[t] = W[t];
            o.rhs = x[9 * s + i];
        }
    };
    auto step = [&](const Ops& c, Ops& n, int s) {
        if (s < steps) load(s + 1, n);
        double a0, a1, a2;
        if (ORDER == 0) {
            double vt[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) vt[t] = readlane_f64(v, t);
            a0 = c.rhs - c.w[0] * vt[0]; a1 = c.w[1] * vt[1]; a2 = c.w[2] * vt[2];
            a0 -= c.w[3] * vt[3]; a1 += c.w[4] * vt[4]; a2 += c.w[5] * vt[5];
            a0 -= c.w[6] * vt[6]; a1 += c.w[7] * vt[7]; a2 += c.w[8] * vt[8];
        } else {
            double v0 = readlane_f64(v, 0), v1 = readlane_f64(v, 1), v2 = readlane_f64(v, 2);
            __builtin_amdgcn_sched_barrier(0);
            a0 = c.rhs - c.w[0] * v0; a1 = c.w[1] * v1; a2 = c.w[2] * v2;
            __builtin_amdgcn_sched_barrier(0);
            double v3 = readlane_f64(v, 3), v4 = readlane_f64(v, 4), v5 = readlane_f64(v, 5);
            __builtin_amdgcn_sched_barrier(0);
            a0 -= c.w[3] * v3; a1 += c.w[4] * v4; a2 += c.w[5] * v5;
            __builtin_amdgcn_sched_barrier(0);
            double v6 = readlane_f64(v, 6), v7 = readlane_f64(v, 7), v8 = readlane_f64(v, 8);
            __builtin_amdgcn_sched_barrier(0);
            a0 -= c.w[6] * v6; a1 += c.w[7] * v7; a2 += c.w[8] * v8;
        }
        v = a0 - (a1 + a2);
        if (act) x[9 * s + i] = v;
    };
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        Ops A, B;
        load(1, A);
        int s = 1;
        for (; s + 1 <= steps; s += 2) { step(A, B, s); step(B, A, s + 1); }
        if (s <= steps) step(A, B, s);
    }
    long long t1 = clock64();
    out[tid] = v;
    if (lane == 0) cyc[wv] = t1 - t0;
}

// The same pipeline fully unrolled (compile-time step count): s_waitcnt counts stay exact, while a loop
// back-edge makes the compiler wait for lgkmcnt(0) — i.e. for the prefetch it has just issued.
template <int STEPS>
__global__ void __launch_bounds__(256) chain_unrolled(double* out, long long* cyc, int reps, int) {
    extern __shared__ double lds[];
    double* Wk = lds;
    double* xs = lds + 20 * 81;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < 20 * 81; e += blockDim.x) Wk[e] = 1e-3 * ((e * 7) % 13 - 6);
    for (int e = tid; e < 4 * 180; e += blockDim.x) xs[e] = 1.0 + 1e-3 * e;
    __syncthreads();
    double* x = xs + wv * 180;
    const int i = lane < 9 ? lane : 8;
    const bool act = lane < 9;
    double v = x[i];
    struct Ops { double w[9], rhs; };
    auto load = [&](int s, Ops& o) {
        const double* W = Wk + s * 81 + 9 * i;
#pragma unroll
        for (int t = 0; t < 9; ++t) o.w[t] = W[t];
        o.rhs = x[9 * s + i];
    };
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        Ops o[2];
        load(1, o[0]);
#pragma unroll
        for (int s = 1; s <= STEPS; ++s) {
            const Ops& c = o[(s - 1) & 1];
            if (s < STEPS) load(s + 1, o[s & 1]);
            double vt[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) vt[t] = readlane_f64(v, t);
            double a0 = c.rhs - c.w[0] * vt[0], a1 = c.w[1] * vt[1], a2 = c.w[2] * vt[2];
            a0 -= c.w[3] * vt[3]; a1 += c.w[4] * vt[4]; a2 += c.w[5] * vt[5];
            a0 -= c.w[6] * vt[6]; a1 += c.w[7] * vt[7]; a2 += c.w[8] * vt[8];
            v = a0 - (a1 + a2);
            if (act) x[9 * s + i] = v;
        }
    }
    long long t1 = clock64();
    out[tid] = v;
    if (lane == 0) cyc[wv] = t1 - t0;
}

int main() {
    double* d; long long* c;
    hipMalloc(&d, 256 * 8); hipMalloc(&c, 4 * 8);
    const int reps = 200, steps = 10;
    const size_t lds = (20 * 81 + 4 * 180) * 8;
    const char* names[] = {"full step", "no LDS store", "W in registers", "no readlane", "prefetch next W", "LDS broadcast"};
    for (int nw = 1; nw <= 4; nw *= 2)
        for (int mode = 0; mode < 6; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (mode) {
                    case 0: chain<0><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                    case 1: chain<1><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                    case 2: chain<2><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                    case 3: chain<3><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                    case 4: chain<4><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                    case 5: chain<5><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                }
                hipDeviceSynchronize();
            }
            long long h[4]; hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
            printf("waves %d  %-18s %.1f cycles per step\n", nw, names[mode], (double)h[0] / (reps * steps));
        }
    for (int nw = 1; nw <= 4; nw *= 4) {
        for (int rep = 0; rep < 2; ++rep) { chain_unrolled<10><<<1, 64 * nw, lds>>>(d, c, reps, steps); hipDeviceSynchronize(); }
        long long h[4]; hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
        printf("waves %d  %-22s %.1f cycles per step\n", nw, "pipe, fully unrolled", (double)h[0] / (reps * steps));
    }
    const char* pn[] = {"pipe", "pipe+interleave", "pipe+mask", "pipe+interleave+mask"};
    for (int nw = 1; nw <= 4; nw *= 4)
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (mode) {
                    case 0: chain_pipe<0, false><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                    case 1: chain_pipe<1, false><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                    case 2: chain_pipe<0, true><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                    case 3: chain_pipe<1, true><<<1, 64 * nw, lds>>>(d, c, reps, steps); break;
                }
                hipDeviceSynchronize();
            }
            long long h[4]; hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
            printf("waves %d  %-22s %.1f cycles per step\n", nw, pn[mode], (double)h[0] / (reps * steps));
        }
    return 0;
}
