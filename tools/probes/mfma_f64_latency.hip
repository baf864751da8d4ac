// Dependent-issue latency of the f64 MFMAs on gfx950 (one wavefront, s_memtime around N instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void lat(double* out, long long* cyc, int n) {
    int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = (l & 15) == 0 ? 1e-3 : 0.0;
    v4d acc = {1.0, 2.0, 3.0, 0.0};
    double s = 0.5;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);            // D -> C chain
        if (MODE == 1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[0], acc, 0, 0, 0);        // D -> B and C
        if (MODE == 2) s = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s, 0, 0, 0);                   // 4x4x4 D -> C
        if (MODE == 3) s = __builtin_amdgcn_mfma_f64_4x4x4f64(s, b, s, 0, 0, 0);                   // 4x4x4 D -> A and C
        if (MODE == 4) { acc[0] = acc[0] * a + b; acc[1] = acc[1] * a + acc[0]; }                 // 2 dependent v_fma_f64
    }
    long long t1 = clock64();
    out[l] = acc[0] + acc[1] + acc[2] + s;
    if (l == 0) *cyc = t1 - t0;
}
int main() {
    double* d; long long* c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
    const int n = 4096;
    const char* names[] = {"mfma_f64_16x16x4 D->C", "mfma_f64_16x16x4 D->B,C", "mfma_f64_4x4x4 D->C", "mfma_f64_4x4x4 D->A,C", "2x v_fma_f64 dependent"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) lat<0><<<1, 64>>>(d, c, n);
            if (mode == 1) lat<1><<<1, 64>>>(d, c, n);
            if (mode == 2) lat<2><<<1, 64>>>(d, c, n);
            if (mode == 3) lat<3><<<1, 64>>>(d, c, n);
            if (mode == 4) lat<4><<<1, 64>>>(d, c, n);
            hipDeviceSynchronize();
        }
        long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.1f cycles per iteration\n", names[mode], (double)h / n);
    }
    return 0;
}
