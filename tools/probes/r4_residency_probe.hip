// Round 6: do FOUR workgroups of three wavefronts (192 threads, 168 VGPRs, 39 424 B of dynamic + 448 B of static LDS — the resource
// footprint of k_mhe_solve_r4_4_n20) really share a CU, and where do their wavefronts land?  1024 workgroups spin until all that can
// be resident have started; each records HW_ID per wavefront and its start time.
//   hipcc --offload-arch=gfx950 -O3 -o r4_residency_probe r4_residency_probe.hip && ./r4_residency_probe [lds_bytes [threads]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <string>
#include <algorithm>

template <int THREADS>
__global__ void __launch_bounds__(THREADS, 3) probe(unsigned long long* out, long long spin) {
    extern __shared__ double lds[];
    __shared__ double red[56];
    const int w = threadIdx.x >> 6;
    asm volatile("v_mov_b32 v167, 0" ::: "v167");  // the allocation of a 168-VGPR kernel
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
    lds[threadIdx.x] = hw; red[threadIdx.x & 31] = 0.0;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = out + ((size_t)blockIdx.x * 4 + w) * 2;
        o[0] = ((unsigned long long)xcc << 32) | hw;
        o[1] = (unsigned long long)t0;
    }
}

int run(int lds, int threads, int G, unsigned long long* d, bool verbose) {
    const int NW = threads / 64;
    (void)hipMemset(d, 0, (size_t)G * 4 * 2 * sizeof(unsigned long long));
    const long long spin = 100000000 / 50;  // 20 ms at the 100 MHz wall clock
    if (threads == 192) probe<192><<<G, 192, lds>>>(d, spin); else probe<256><<<G, 256, lds>>>(d, spin);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return -1; }
    std::vector<unsigned long long> h((size_t)G * 4 * 2);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int g = 0; g < G; ++g) tmin = std::min(tmin, h[(size_t)g * 8 + 1]);
    int first_round = 0;
    std::map<unsigned, std::vector<int>> cu, late;
    for (int g = 0; g < G; ++g) {
        const bool early = h[(size_t)g * 8 + 1] - tmin < (unsigned long long)spin / 2;
        first_round += early;
        unsigned hw = (unsigned)h[(size_t)g * 8], xcc = (unsigned)(h[(size_t)g * 8] >> 32) & 0xf;
        unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
        (early ? cu : late)[key].push_back(g);
    }
    if (!verbose) return first_round;
    printf("lds %d B, %d threads, grid %d: %d workgroups started in the first round (resident together) on %zu CUs\n", lds, threads, G, first_round, cu.size());
    std::map<int, int> hist;
    std::map<std::string, int> simd_load;
    int late_on_short = 0, late_total = 0;
    for (auto& kv : late) { late_total += (int)kv.second.size(); if (cu[kv.first].size() < 4) late_on_short += (int)kv.second.size(); }
    for (auto& kv : cu) {
        hist[(int)kv.second.size()]++;
        int per_simd[4] = {0, 0, 0, 0}, chain[4] = {0, 0, 0, 0};
        for (int g : kv.second)
            for (int w = 0; w < NW; ++w) { unsigned hw = (unsigned)h[((size_t)g * 4 + w) * 2]; per_simd[(hw >> 4) & 3]++; if (w == 0) chain[(hw >> 4) & 3]++; }
        char buf[64];
        snprintf(buf, sizeof buf, "waves/SIMD %d %d %d %d | wave-0s/SIMD %d %d %d %d", per_simd[0], per_simd[1], per_simd[2], per_simd[3], chain[0], chain[1], chain[2], chain[3]);
        simd_load[buf]++;
    }
    for (auto& kv : hist) printf("  CUs with %d resident workgroups: %d\n", kv.first, kv.second);
    printf("  late workgroups: %d, of which on a CU that held fewer than four: %d\n", late_total, late_on_short);
    int shown = 0;
    std::vector<std::pair<int, std::string>> sl;
    for (auto& kv : simd_load) sl.push_back({kv.second, kv.first});
    std::sort(sl.rbegin(), sl.rend());
    for (auto& e : sl) { if (shown++ < 8) printf("  %4d CUs: %s\n", e.first, e.second.c_str()); }
    return first_round;
}

int main(int argc, char** argv) {
    unsigned long long* d;
    (void)hipMalloc(&d, (size_t)4096 * 4 * 2 * sizeof(unsigned long long));
    if (argc > 1 && std::string(argv[1]) == "sweep") {  // resident workgroups of 1024 by LDS size, five launches each
        for (int lds = 32768; lds <= 40960; lds += 512) {
            printf("lds %5d B x 192 threads:", lds);
            for (int t = 0; t < 5; ++t) printf(" %4d", run(lds, 192, 1024, d, false));
            printf("\n");
        }
        return 0;
    }
    const int lds = argc > 1 ? atoi(argv[1]) : 39424, threads = argc > 2 ? atoi(argv[2]) : 192, G = argc > 3 ? atoi(argv[3]) : 1024;
    return run(lds, threads, G, d, true) < 0;
}
