// cov_mfma_probe.hip — the MFMA clause of the north star, measured: does an f64 MFMA beat the vector ALU on the two dense
// covariance contractions of the path?
//   (1) EKF predict   P <- F P F' + W C_g W'   4x4, orien_ekf.cpp:120        v_mfma_f64_4x4x4_4b_f64 (4 instances / instruction)
//   (2) KF predict    C <- A C A'              9x9, DecentralEst.cpp:783-785  v_mfma_f64_16x16x4_f64 (one instance per tile)
// Both against the form the product uses (one lane per instance for the EKF; a wavefront per instance for the KF).
// A priori: MI355X's f64 matrix peak equals its f64 vector peak (78.6 TFLOP/s each, MI355X_MICROARCH.md), so an MFMA can
// only win through fewer issue slots, never through more FLOP/s, and pays for operand layout changes.
//
//   hipcc --offload-arch=gfx950 -O3 -o cov_mfma_probe cov_mfma_probe.hip && ./cov_mfma_probe [B]
// Prints the discovered operand maps of the 4x4x4 instruction, then one JSON line per experiment.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ operand maps of v_mfma_f64_4x4x4_4b_f64
// one-hot A (lane la) x one-hot B (lane lb): which D lane lights up?
__global__ void discover(int* out) {
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            unsigned long long m = __ballot(d != 0.0);
            if (l == 0) out[la * 64 + lb] = m ? __ffsll((long long)m) - 1 + (__popcll(m) > 1 ? 1000 : 0) : -1;
        }
}

struct Map4 { int ablk[64], ai[64], ak[64], bblk[64], bk[64], bj[64], dblk[64], di[64], dj[64]; bool ok; };

static Map4 infer_map(const std::vector<int>& t) {
    // la and lb interact iff same block and same k.  Group lanes of A by the set of B lanes they interact with.
    Map4 m{};
    m.ok = true;
    // hypothesis H (the 16x16x4 maps restricted to the four diagonal 4x4 blocks): A lane l -> (blk = (l / 4) % 4, k = l / 16, i = l % 4);
    // B lane l -> (blk = (l / 4) % 4, k = l / 16, j = l % 4); D's (blk, i, j) is read off the table and checked for consistency.
    for (int l = 0; l < 64; ++l) {
        m.ablk[l] = m.bblk[l] = (l / 4) % 4;
        m.ak[l] = m.bk[l] = l / 16;
        m.ai[l] = m.bj[l] = l % 4;
        m.dblk[l] = m.di[l] = m.dj[l] = -1;
    }
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const bool expect = m.ablk[la] == m.bblk[lb] && m.ak[la] == m.bk[lb];
            const int d = t[la * 64 + lb];
            if (expect != (d >= 0) || d >= 1000) { m.ok = false; continue; }
            if (d >= 0) {
                if (m.dblk[d] == -1) { m.dblk[d] = m.ablk[la]; m.di[d] = m.ai[la]; m.dj[d] = m.bj[lb]; }
                else if (m.dblk[d] != m.ablk[la] || m.di[d] != m.ai[la] || m.dj[d] != m.bj[lb]) m.ok = false;
            }
        }
    return m;
}

// ------------------------------------------------------------------ (1) EKF covariance predict, field-major state [f][B]
__device__ __forceinline__ void build_F(const double* w, double h, double* F) {
    const double f[16] = {1, -h * w[0], -h * w[1], -h * w[2], h * w[0], 1, h * w[2], -h * w[1],
                          h * w[1], -h * w[2], 1, h * w[0], h * w[2], h * w[1], -h * w[0], 1};
    for (int i = 0; i < 16; ++i) F[i] = f[i];
}
__device__ __forceinline__ void build_W(const double* q, double h, double* W) {
    const double w[12] = {-q[1], -q[2], -q[3], q[0], -q[3], q[2], q[3], q[1], q[0], -q[2], 0.0, 0.0};
    for (int i = 0; i < 12; ++i) W[i] = h * w[i];
}

// the product's form (ekf_core.h ekf_predict): one lane per instance, everything in registers
__global__ void __launch_bounds__(64) ekf_valu(const double* gy, const double* q4, const double* Pin, double* Pout, int B, int reps,
                                               double h, double c0, double c1, double c2) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double w[3], q[4], P[16], F[16], W[12];
    for (int i = 0; i < 3; ++i) w[i] = gy[(size_t)i * B + b];
    for (int i = 0; i < 4; ++i) q[i] = q4[(size_t)i * B + b];
    for (int i = 0; i < 16; ++i) P[i] = Pin[(size_t)i * B + b];
    build_F(w, h, F);
    build_W(q, h, W);
    const double cg[3] = {c0, c1, c2};
    for (int r = 0; r < reps; ++r) {
        double FP[16], Pn[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double s = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) s += F[4 * i + t] * P[4 * t + j];
                FP[4 * i + j] = s;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double s = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) s += FP[4 * i + t] * F[4 * j + t];
#pragma unroll
                for (int t = 0; t < 3; ++t) s += W[3 * i + t] * cg[t] * W[3 * j + t];
                Pn[4 * i + j] = s;
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) P[i] = Pn[i];
    }
    for (int i = 0; i < 16; ++i) Pout[(size_t)i * B + b] = P[i];
}

// MFMA form: a wavefront takes 64 instances as 16 groups of 4; lane l of a group holds element (i = l % 4, k = l / 16) of
// instance (l / 4) % 4.  P' = F (P F') + (W C) W' as three v_mfma_f64_4x4x4: the D map (blk, i, j) is also the B map with k = i (see
// the discovery print), so  T = P F'  is formed as  T' = F P' = F P (P symmetric)  ... kept simple here: two products through the
// same register, the middle result re-enters as the B operand without any move.
//   D1 = F * P          (A = F, B = P)            lane (blk, i, j) holds (F P)[i][j]
//   D2 = D1 * F' + WCW' needs D1 as the A operand: A's lane (blk, i, k) = (l % 4, (l / 4) % 4) against D's (di, dj) — equal maps
//   iff di = l % 4 and dj = (l / 4) % 4; the probe prints which case holds and uses a transpose-free formulation for either:
//   if D is (i = (l/4)%4, j = l%4) [row-major like B], compute instead  D1 = P * F'  (A = P, B = F': lane holds F[j][k]) whose
//   result lane (blk, i, j) re-enters as B of  D2 = F * D1.
__global__ void __launch_bounds__(64) ekf_mfma(const double* gy, const double* q4, const double* Pin, double* Pout, int B, int reps,
                                               double h, double c0, double c1, double c2, int d_is_rowmajor_like_b) {
    const int l = threadIdx.x, blk = (l >> 2) & 3, r4 = l >> 4, c4 = l & 3;
    const double cg[3] = {c0, c1, c2};
    for (int g = 0; g < 16; ++g) {
        const int b = blockIdx.x * 64 + g * 4 + blk;
        const bool live = b < B;
        const int bb = live ? b : 0;
        double w[3], q[4], F[16], W[12];
        for (int i = 0; i < 3; ++i) w[i] = gy[(size_t)i * B + bb];
        for (int i = 0; i < 4; ++i) q[i] = q4[(size_t)i * B + bb];
        build_F(w, h, F);
        build_W(q, h, W);
        // operand element selectors (static indexing through selects would spill; the 16-entry tables are tiny)
        auto pick16 = [&](const double* M, int idx) { double v = M[0]; for (int t = 1; t < 16; ++t) v = idx == t ? M[t] : v; return v; };
        // A operand lane = (i = c4, k = r4); B operand lane = (k = r4, j = c4)
        const double Fa = pick16(F, 4 * c4 + r4);        // F[i][k]
        const double Ftb = pick16(F, 4 * c4 + r4);       // F'[k][j] = F[j][k]: same element index
        double WCa = 0.0, Wtb = 0.0;                      // (W C)[i][k], W'[k][j] = W[j][k], k < 3
        for (int t = 0; t < 12; ++t) {
            WCa = (t == 3 * c4 + r4 && r4 < 3) ? W[t] * cg[r4] : WCa;
            Wtb = (t == 3 * c4 + r4 && r4 < 3) ? W[t] : Wtb;
        }
        // P in the B layout: P[k][j]
        double p = Pin[(size_t)(4 * r4 + c4) * B + bb];
        for (int r = 0; r < reps; ++r) {
            const double wcw = __builtin_amdgcn_mfma_f64_4x4x4f64(WCa, Wtb, 0.0, 0, 0, 0);
            if (d_is_rowmajor_like_b) {
                // D lane (blk, i = r4, j = c4): B-shaped.  D1 = P F' needs P as A: A lane wants P[i = c4][k = r4] = P[k][i] (symmetric): p itself
                const double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(p, Ftb, 0.0, 0, 0, 0);   // (P F')[i][j] on the B-shaped lane
                p = __builtin_amdgcn_mfma_f64_4x4x4f64(Fa, d1, wcw, 0, 0, 0);                  // F (P F') + W C W'
            } else {
                // D lane (blk, i = c4, j = r4): A-shaped.  D1 = F P (A = F, B = P), D2 = D1 F' with D1 as A
                const double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(Fa, p, 0.0, 0, 0, 0);
                const double d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(d1, Ftb, wcw, 0, 0, 0);   // lane (i = c4, j = r4) = P'[i][j] = P'[j][i]
                p = d2;                                                                         // symmetric: also P'[k = r4][j = c4]
            }
        }
        if (live) Pout[(size_t)(4 * r4 + c4) * B + b] = p;  // (r4, c4) indexes either P'[r4][c4] or its mirror image: symmetric
    }
}

// ------------------------------------------------------------------ (2) KF covariance predict C <- A C A', 9x9, [B][81] row-major
// wavefront per instance, operands staged in LDS, element per lane (what kf_core.h's wmatmul does)
__global__ void __launch_bounds__(64) kf_valu(const double* A9, const double* C9, double* out, int B, int reps) {
    __shared__ double sA[81], sC[81], sT[81];
    const int l = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        for (int e = l; e < 81; e += 64) { sA[e] = A9[(size_t)b * 81 + e]; sC[e] = C9[(size_t)b * 81 + e]; }
        __syncthreads();
        for (int r = 0; r < reps; ++r) {
            for (int e = l; e < 81; e += 64) {
                const int i = e / 9, j = e - 9 * i;
                double s = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) s += sA[9 * i + t] * sC[9 * t + j];
                sT[e] = s;
            }
            __syncthreads();
            double o0 = 0, o1 = 0;
            {
                const int e = l, i = e / 9, j = e - 9 * i;
#pragma unroll
                for (int t = 0; t < 9; ++t) o0 += sT[9 * i + t] * sA[9 * j + t];
            }
            if (l + 64 < 81) {
                const int e = l + 64, i = e / 9, j = e - 9 * i;
#pragma unroll
                for (int t = 0; t < 9; ++t) o1 += sT[9 * i + t] * sA[9 * j + t];
            }
            __syncthreads();
            sC[l] = o0;
            if (l + 64 < 81) sC[l + 64] = o1;
            __syncthreads();
        }
        for (int e = l; e < 81; e += 64) out[(size_t)b * 81 + e] = sC[e];
        __syncthreads();
    }
}

// v_mfma_f64_16x16x4_f64: A operand lane l = A[i = l & 15][k = l >> 4], B operand lane l = B[k = l >> 4][j = l & 15],
// D lane l reg r = D[row = (l >> 4) + 4 r][col = l & 15]  (tools/probes/mfma_f64_probe.hip).  K = 9 -> three K chunks.
//   D' = C A'   (A operand = C[i][k], B operand = A'[k][j] = A[j][k]);  chunk c of D' as a B operand is D' register c
//   out = A D'  (A operand = A[i][k], B operand = D' regs) — the accumulator re-enters as B with no data movement
__global__ void __launch_bounds__(64) kf_mfma(const double* A9, const double* C9, double* out, int B, int reps) {
    const int l = threadIdx.x, lo = l & 15, hi = l >> 4;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const double* Ab = A9 + (size_t)b * 81;
        const double* Cb = C9 + (size_t)b * 81;
        double a_op[3], at_op[3], c_op[3];  // per K chunk c: k = 4 c + hi
        for (int c = 0; c < 3; ++c) {
            const int k = 4 * c + hi;
            const bool in = lo < 9 && k < 9;
            a_op[c] = in ? Ab[9 * lo + k] : 0.0;    // A[i = lo][k]        (A operand of the second product)
            at_op[c] = in ? Ab[9 * lo + k] : 0.0;   // A'[k][j = lo] = A[lo][k]   (B operand of the first product)
            c_op[c] = in ? Cb[9 * lo + k] : 0.0;    // C[i = lo][k]
        }
        for (int r = 0; r < reps; ++r) {
            v4d d = {0, 0, 0, 0};
            for (int c = 0; c < 3; ++c) d = __builtin_amdgcn_mfma_f64_16x16x4f64(c_op[c], at_op[c], d, 0, 0, 0);   // C A'
            v4d o = {0, 0, 0, 0};
            for (int c = 0; c < 3; ++c) o = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[c], d[c], o, 0, 0, 0);       // A (C A')
            // the new C as the A operand of the next repetition: C[i = lo][k = 4 c + hi] = C[k][i] (symmetric) = o row (hi + 4 c), col lo
            for (int c = 0; c < 3; ++c) c_op[c] = o[c];
        }
        for (int c = 0; c < 3; ++c) {
            const int k = 4 * c + hi;
            if (lo < 9 && k < 9) out[(size_t)b * 81 + 9 * k + lo] = c_op[c];
        }
    }
}

template <class Launch>
static double time_us(Launch f, int rounds = 20) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    f(); f();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < rounds; ++i) f();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / rounds;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 65536;
    // ---- layout discovery
    int* dt;
    CHK(hipMalloc(&dt, 4096 * sizeof(int)));
    discover<<<1, 64>>>(dt);
    std::vector<int> t(4096);
    CHK(hipMemcpy(t.data(), dt, 4096 * sizeof(int), hipMemcpyDeviceToHost));
    Map4 m = infer_map(t);
    printf("v_mfma_f64_4x4x4_4b_f64: A lane l = (blk (l/4)%%4, i l%%4, k l/16), B lane l = (blk (l/4)%%4, k l/16, j l%%4): %s\n",
           m.ok ? "confirmed" : "NOT confirmed");
    bool d_like_b = true, d_like_a = true;
    for (int l = 0; l < 64; ++l) {
        if (!(m.dblk[l] == (l / 4) % 4 && m.di[l] == l / 16 && m.dj[l] == l % 4)) d_like_b = false;
        if (!(m.dblk[l] == (l / 4) % 4 && m.di[l] == l % 4 && m.dj[l] == l / 16)) d_like_a = false;
    }
    printf("D lane l = (blk (l/4)%%4, i, j): %s\n", d_like_b ? "i = l/16, j = l%4 (B-shaped: a result re-enters as a B operand without a move)" : (d_like_a ? "i = l%4, j = l/16 (A-shaped)" : "other"));
    if (!m.ok || (!d_like_a && !d_like_b)) {
        for (int l = 0; l < 64; ++l) printf("D lane %d -> blk %d i %d j %d\n", l, m.dblk[l], m.di[l], m.dj[l]);
        return 1;
    }
    // ---- (1) EKF
    std::vector<double> gy(3 * (size_t)B), q4(4 * (size_t)B), P(16 * (size_t)B);
    srand(1);
    auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
    for (auto& v : gy) v = 2.0 * rnd();
    for (size_t b = 0; b < (size_t)B; ++b) {
        double q[4], n = 0;
        for (int i = 0; i < 4; ++i) { q[i] = rnd(); n += q[i] * q[i]; }
        for (int i = 0; i < 4; ++i) q4[(size_t)i * B + b] = q[i] / sqrt(n);
        double L[16];
        for (int i = 0; i < 16; ++i) L[i] = rnd();
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = i == j ? 1e-3 : 0.0;
                for (int k = 0; k < 4; ++k) s += 1e-3 * L[4 * i + k] * L[4 * j + k];
                P[(size_t)(4 * i + j) * B + b] = s;
            }
    }
    double *dgy, *dq, *dP, *dO1, *dO2;
    CHK(hipMalloc(&dgy, gy.size() * 8)); CHK(hipMalloc(&dq, q4.size() * 8)); CHK(hipMalloc(&dP, P.size() * 8));
    CHK(hipMalloc(&dO1, P.size() * 8)); CHK(hipMalloc(&dO2, P.size() * 8));
    CHK(hipMemcpy(dgy, gy.data(), gy.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dq, q4.data(), q4.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice));
    const double h = 0.5 * 0.002, c0 = 1e-4, c1 = 2e-4, c2 = 3e-4;
    const int grid = (B + 63) / 64;
    std::vector<double> o1(P.size()), o2(P.size());
    for (int reps : {1, 64}) {
        const double tv = time_us([&] { ekf_valu<<<grid, 64>>>(dgy, dq, dP, dO1, B, reps, h, c0, c1, c2); });
        const double tm = time_us([&] { ekf_mfma<<<grid, 64>>>(dgy, dq, dP, dO2, B, reps, h, c0, c1, c2, d_like_b ? 1 : 0); });
        CHK(hipMemcpy(o1.data(), dO1, o1.size() * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(o2.data(), dO2, o2.size() * 8, hipMemcpyDeviceToHost));
        double err = 0, mag = 0;
        for (size_t i = 0; i < o1.size(); ++i) { err = fmax(err, fabs(o1[i] - o2[i])); mag = fmax(mag, fabs(o1[i])); }
        printf("{\"experiment\": \"EKF P <- F P F' + W C W' (4x4)\", \"batch\": %d, \"repetitions_per_load\": %d, \"valu_lane_per_instance_us\": %.2f, "
               "\"mfma_4x4x4_4b_us\": %.2f, \"mfma_over_valu\": %.2f, \"max_abs_diff\": %.3e, \"max_abs_value\": %.3e}\n",
               B, reps, tv, tm, tm / tv, err, mag);
    }
    // ---- (2) KF
    std::vector<double> A9(81 * (size_t)B), C9(81 * (size_t)B);
    for (size_t b = 0; b < (size_t)B; ++b) {
        double L[81];
        for (int i = 0; i < 81; ++i) { L[i] = rnd(); A9[b * 81 + i] = (i % 10 == 0 ? 1.0 : 0.0) + 0.01 * rnd(); }
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) {
                double s = i == j ? 1e-2 : 0.0;
                for (int k = 0; k < 9; ++k) s += 1e-2 * L[9 * i + k] * L[9 * j + k];
                C9[b * 81 + 9 * i + j] = s;
            }
    }
    double *dA, *dC, *dK1, *dK2;
    CHK(hipMalloc(&dA, A9.size() * 8)); CHK(hipMalloc(&dC, C9.size() * 8)); CHK(hipMalloc(&dK1, C9.size() * 8)); CHK(hipMalloc(&dK2, C9.size() * 8));
    CHK(hipMemcpy(dA, A9.data(), A9.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dC, C9.data(), C9.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> k1(C9.size()), k2(C9.size());
    const int kgrid = 256 * 32;  // 8 resident wavefronts per SIMD
    for (int reps : {1, 64}) {
        const double tv = time_us([&] { kf_valu<<<kgrid, 64>>>(dA, dC, dK1, B, reps); });
        const double tm = time_us([&] { kf_mfma<<<kgrid, 64>>>(dA, dC, dK2, B, reps); });
        CHK(hipMemcpy(k1.data(), dK1, k1.size() * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(k2.data(), dK2, k2.size() * 8, hipMemcpyDeviceToHost));
        double err = 0, mag = 0;
        for (size_t i = 0; i < k1.size(); ++i) { err = fmax(err, fabs(k1[i] - k2[i])); mag = fmax(mag, fabs(k1[i])); }
        printf("{\"experiment\": \"KF C <- A C A' (9x9)\", \"batch\": %d, \"repetitions_per_load\": %d, \"valu_wave_per_instance_lds_us\": %.2f, "
               "\"mfma_16x16x4_us\": %.2f, \"mfma_over_valu\": %.2f, \"max_abs_diff\": %.3e, \"max_abs_value\": %.3e}\n",
               B, reps, tv, tm, tm / tv, err, mag);
    }
    return 0;
}
