// Probe of v_mfma_f64_16x16x4_f64 operand / result lane maps on gfx950 (one wavefront).
// Assumed (cdna_hip_programming.md §3): A: lane l holds A[i = l&15][k = l>>4]; B: lane l holds
// B[k = l>>4][j = l&15]; C/D: lane l, reg r holds D[row = (l>>4) + 4r][col = l&15].
// Prints the max abs error of D = A*B + C under that assumption for asymmetric integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, const double* C, double* D) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];      // A is 16x4 row-major
    double b = B[(l >> 4) * 16 + (l & 15)];     // B is 4x16 row-major
    v4d c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}
int main() {
    double hA[64], hB[64], hC[256], hD[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1 + i * 3 + k * 7;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 2 + k * 5 - j * 11;
    for (int i = 0; i < 256; ++i) hC[i] = 1000 + i;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = hC[i * 16 + j]; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; ++i) { double e = hD[i] - ref[i]; if (e < 0) e = -e; if (e > err) err = e; }
    printf("mfma_f64_16x16x4 assumed-layout max abs err = %g (%s)\n", err, err == 0 ? "layout confirmed" : "LAYOUT WRONG");
    return err == 0 ? 0 : 1;
}
