// v_fmac_f64_dpp ... row_newbcast:t on gfx950: "acc += (v of lane t of my 16-lane row) * w" in ONE
// instruction.  Checks the semantics against the v_readlane form of the chain step and times a
// 10-step chain (fully unrolled, operands prefetched) both ways.
//   hipcc --offload-arch=gfx950 -O3 -o dpp_chain_probe dpp_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

#define DPP_FMAC(acc, v, w, T) \
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w))

// v_new = rhs - W v with v broadcast lane by lane out of row 0 (lanes 0..8 hold v)
__device__ __forceinline__ double step_dpp(double v, const double* w, double rhs) {
    double a0 = rhs, a1 = 0.0, a2 = 0.0;
    asm volatile("s_nop 1");  // VALU write of v -> DPP read needs 2 wait states; inline asm gets no hazard handling
    DPP_FMAC(a0, v, w[0], 0); DPP_FMAC(a1, v, w[1], 1); DPP_FMAC(a2, v, w[2], 2);
    DPP_FMAC(a0, v, w[3], 3); DPP_FMAC(a1, v, w[4], 4); DPP_FMAC(a2, v, w[5], 5);
    DPP_FMAC(a0, v, w[6], 6); DPP_FMAC(a1, v, w[7], 7); DPP_FMAC(a2, v, w[8], 8);
    return a0 + (a1 + a2);
}
// broadcast with v_mov_b64_dpp row_newbcast, multiply with a plain (full-rate) v_fma_f64
#define DPP_MOV(dst, v, T) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(v))
__device__ __forceinline__ double step_movdpp(double v, const double* w, double rhs) {
    double b0, b1, b2, b3, b4, b5, b6, b7, b8;
    asm volatile("s_nop 1");
    DPP_MOV(b0, v, 0); DPP_MOV(b1, v, 1); DPP_MOV(b2, v, 2); DPP_MOV(b3, v, 3); DPP_MOV(b4, v, 4);
    DPP_MOV(b5, v, 5); DPP_MOV(b6, v, 6); DPP_MOV(b7, v, 7); DPP_MOV(b8, v, 8);
    double a0 = rhs - w[0] * b0, a1 = -(w[1] * b1), a2 = -(w[2] * b2);
    a0 -= w[3] * b3; a1 -= w[4] * b4; a2 -= w[5] * b5;
    a0 -= w[6] * b6; a1 -= w[7] * b7; a2 -= w[8] * b8;
    return a0 + (a1 + a2);
}
// broadcast the two 32-bit halves with full-rate v_mov_b32_dpp, multiply with a plain (full-rate) v_fma_f64
#define DPP_MOV32(dst, v, T) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(v))
__device__ __forceinline__ double step_mov32dpp(double v, const double* w, double rhs) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int l0, l1, l2, l3, l4, l5, l6, l7, l8, h0, h1, h2, h3, h4, h5, h6, h7, h8;
    asm volatile("s_nop 1");
    DPP_MOV32(l0, lo, 0); DPP_MOV32(h0, hi, 0); DPP_MOV32(l1, lo, 1); DPP_MOV32(h1, hi, 1); DPP_MOV32(l2, lo, 2); DPP_MOV32(h2, hi, 2);
    DPP_MOV32(l3, lo, 3); DPP_MOV32(h3, hi, 3); DPP_MOV32(l4, lo, 4); DPP_MOV32(h4, hi, 4); DPP_MOV32(l5, lo, 5); DPP_MOV32(h5, hi, 5);
    DPP_MOV32(l6, lo, 6); DPP_MOV32(h6, hi, 6); DPP_MOV32(l7, lo, 7); DPP_MOV32(h7, hi, 7); DPP_MOV32(l8, lo, 8); DPP_MOV32(h8, hi, 8);
    double a0 = rhs - w[0] * __hiloint2double(h0, l0), a1 = -(w[1] * __hiloint2double(h1, l1)), a2 = -(w[2] * __hiloint2double(h2, l2));
    a0 -= w[3] * __hiloint2double(h3, l3); a1 -= w[4] * __hiloint2double(h4, l4); a2 -= w[5] * __hiloint2double(h5, l5);
    a0 -= w[6] * __hiloint2double(h6, l6); a1 -= w[7] * __hiloint2double(h7, l7); a2 -= w[8] * __hiloint2double(h8, l8);
    return a0 + (a1 + a2);
}
// nine independent DPP products, then an addition tree (depth 1 + 4 instead of 3 + 2)
__device__ __forceinline__ double step_dpp9(double v, const double* w, double rhs) {
    double a0 = rhs, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0, a8 = 0.0;
    asm volatile("s_nop 1");
    DPP_FMAC(a0, v, w[0], 0); DPP_FMAC(a1, v, w[1], 1); DPP_FMAC(a2, v, w[2], 2);
    DPP_FMAC(a3, v, w[3], 3); DPP_FMAC(a4, v, w[4], 4); DPP_FMAC(a5, v, w[5], 5);
    DPP_FMAC(a6, v, w[6], 6); DPP_FMAC(a7, v, w[7], 7); DPP_FMAC(a8, v, w[8], 8);
    return ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) + a8;
}
__device__ __forceinline__ double step_readlane(double v, const double* w, double rhs) {
    double vt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) vt[t] = readlane_f64(v, t);
    double a0 = rhs - w[0] * vt[0], a1 = -(w[1] * vt[1]), a2 = -(w[2] * vt[2]);
    a0 -= w[3] * vt[3]; a1 -= w[4] * vt[4]; a2 -= w[5] * vt[5];
    a0 -= w[6] * vt[6]; a1 -= w[7] * vt[7]; a2 -= w[8] * vt[8];
    return a0 + (a1 + a2);
}

template <int DPP, int STEPS, bool ROW0 = false>
__global__ void __launch_bounds__(256) chain(double* out, long long* cyc, int reps) {
    extern __shared__ double lds[];
    double* Wk = lds;
    double* xs = lds + 20 * 81;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < 20 * 81; e += blockDim.x) Wk[e] = 1e-2 * ((e * 7) % 13 - 6);
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
    if (!ROW0 || lane < 16)  // ROW0: only lanes 0..15 are in EXEC during the chain
    for (int r = 0; r < reps; ++r) {
        Ops o[2];
        load(1, o[0]);
#pragma unroll
        for (int s = 1; s <= STEPS; ++s) {
            const Ops& c = o[(s - 1) & 1];
            if (s < STEPS) load(s + 1, o[s & 1]);
            v = DPP == 1 ? step_dpp(v, c.w, c.rhs) : (DPP == 2 ? step_dpp9(v, c.w, c.rhs) : (DPP == 3 ? step_movdpp(v, c.w, c.rhs) : (DPP == 4 ? step_mov32dpp(v, c.w, c.rhs) : step_readlane(v, c.w, c.rhs))));
            if (act) x[9 * s + i] = v;
        }
    }
    long long t1 = clock64();
    out[tid] = v;
    if (lane == 0) cyc[wv] = t1 - t0;
}

int main() {
    double *d, h[5][256]; long long* c;
    (void)hipMalloc(&d, 256 * 8); (void)hipMalloc(&c, 4 * 8);
    const size_t lds = (20 * 81 + 4 * 180) * 8;
    for (int nw = 1; nw <= 4; nw *= 4)
        for (int dpp = 0; dpp < 5; ++dpp) {
            for (int rep = 0; rep < 2; ++rep) {
                if (dpp == 1) chain<1, 10><<<1, 64 * nw, lds>>>(d, c, 200);
                else if (dpp == 2) chain<2, 10><<<1, 64 * nw, lds>>>(d, c, 200);
                else if (dpp == 3) chain<3, 10><<<1, 64 * nw, lds>>>(d, c, 200);
                else if (dpp == 4) chain<4, 10><<<1, 64 * nw, lds>>>(d, c, 200);
                else chain<0, 10><<<1, 64 * nw, lds>>>(d, c, 200);
                (void)hipDeviceSynchronize();
            }
            long long hc[4];
            (void)hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost);
            (void)hipMemcpy(h[dpp], d, 256 * 8, hipMemcpyDeviceToHost);
            printf("waves %d  %-10s %.1f cycles per step   v[0..2] = %.15g %.15g %.15g\n", nw, dpp == 1 ? "dpp fmac" : (dpp == 2 ? "dpp x9" : (dpp == 3 ? "mov_dpp+fma" : (dpp == 4 ? "mov32_dpp+fma" : "readlane"))), (double)hc[0] / 2000.0,
                   h[dpp][0], h[dpp][1], h[dpp][2]);
        }
    for (int dpp = 0; dpp < 2; ++dpp) {
        for (int rep = 0; rep < 2; ++rep) {
            if (dpp) chain<1, 10, true><<<1, 64, lds>>>(d, c, 200);
            else chain<0, 10, true><<<1, 64, lds>>>(d, c, 200);
            (void)hipDeviceSynchronize();
        }
        long long hc[4];
        (void)hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost);
        printf("EXEC = lanes 0..15 only   %-10s %.1f cycles per step\n", dpp ? "dpp fmac" : "readlane", (double)hc[0] / 2000.0);
    }
    double md = 0.0;
    for (int l = 0; l < 9; ++l) md = fmax(md, fabs(h[0][l] - h[1][l]) / (fabs(h[0][l]) + 1e-300));
    printf("max relative difference dpp vs readlane over lanes 0..8: %.3e\n", md);
    return md < 1e-12 ? 0 : 1;
}
