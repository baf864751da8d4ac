// Where do the wavefronts of co-resident workgroups land?  512 workgroups x 256 threads x 80 KiB LDS (two per
// CU, like k_mhe_solve): every wavefront records HW_REG_HW_ID (wave slot, SIMD, CU, SE, thread-group id) and
// XCC_ID.  Answers: is wave w of a workgroup always on SIMD w, and do the two workgroups of a CU differ in TG_ID?
//   hipcc --offload-arch=gfx950 -O3 -o hwid_probe hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256, 2) probe(unsigned* out, int spin) {
    extern __shared__ double lds[];
    const int w = threadIdx.x >> 6;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
    lds[threadIdx.x] = hw;
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}  // keep every workgroup resident until all have started
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + w) * 2] = hw; out[(blockIdx.x * 4 + w) * 2 + 1] = xcc; }
}

int main() {
    const int G = 512;
    unsigned* d;
    (void)hipMalloc(&d, G * 4 * 2 * sizeof(unsigned));
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    probe<<<G, 256, 80 * 1024>>>(d, 2000000);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(G * 4 * 2);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    int simd_is_w = 0, total = 0;
    std::map<unsigned, std::vector<int>> cu_groups;  // (xcc, se, sh, cu) -> workgroups
    for (int g = 0; g < G; ++g) {
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[(g * 4 + w) * 2];
            simd_is_w += ((hw >> 4) & 3) == (unsigned)w;
            ++total;
        }
        unsigned hw = h[g * 8], xcc = h[g * 8 + 1] & 0xf;
        unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
        cu_groups[key].push_back(g);
    }
    printf("wave w on SIMD w: %d of %d wavefronts\n", simd_is_w, total);
    printf("distinct (xcc, se, sh, cu): %zu\n", cu_groups.size());
    int shown = 0, tg_differs = 0, pairs = 0, slot_differs = 0;
    for (auto& kv : cu_groups) {
        if (kv.second.size() == 2) {
            ++pairs;
            unsigned a = h[kv.second[0] * 8], b = h[kv.second[1] * 8];
            tg_differs += ((a >> 16) & 0xf) != ((b >> 16) & 0xf);
            slot_differs += (a & 0xf) != (b & 0xf);
        }
        if (shown++ < 6) {
            printf("cu key %06x:", kv.first);
            for (int g : kv.second) {
                printf("  wg %3d [", g);
                for (int w = 0; w < 4; ++w) { unsigned hw = h[(g * 4 + w) * 2]; printf(" simd%u/slot%u/tg%u", (hw >> 4) & 3, hw & 0xf, (hw >> 16) & 0xf); }
                printf(" ]");
            }
            printf("\n");
        }
    }
    printf("CUs with exactly two workgroups: %d; TG_ID differs in %d, wave slot of wave 0 differs in %d\n", pairs, tg_differs, slot_differs);
    return 0;
}
