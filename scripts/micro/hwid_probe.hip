// Development probe: which hardware registers tell the two co-resident workgroups of a CU apart?
// Launches 512 workgroups x 256 threads with 79 KB of LDS (two per CU, like the chain kernels) and dumps HW_ID,
// LDS_ALLOC and XCC_ID of wave 0 of every workgroup.   hipcc --offload-arch=gfx950 -O2 -o hwid_probe_micro hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    __shared__ float big[79 * 256];
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_ID, all 32 bits
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);    // LDS_ALLOC
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // XCC_ID
        out[blockIdx.x * 4 + 3] = (unsigned)big[17];
    }
    // stay resident so that all 512 workgroups are placed before any leaves
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 16);
    probe<<<512, 256>>>(d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2048);
    hipMemcpy(h.data(), d, 512 * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < 512; ++b) {
        const unsigned hw = h[b * 4], xcc = h[b * 4 + 2] & 0xf;
        const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xf);
        cu[key].push_back(b);
    }
    printf("distinct (xcc,se,sh,cu) keys: %zu\n", cu.size());
    int shown = 0, pairs = 0, tgdiff = 0, ldsdiff = 0;
    for (auto& kv : cu) {
        if (kv.second.size() == 2) {
            ++pairs;
            const int a = kv.second[0], b = kv.second[1];
            const unsigned tga = (h[a * 4] >> 16) & 0xf, tgb = (h[b * 4] >> 16) & 0xf;
            if ((tga & 1) != (tgb & 1)) ++tgdiff;
            if ((h[a * 4 + 1] & 0xff) != (h[b * 4 + 1] & 0xff)) ++ldsdiff;
        }
        if (shown < 12) {
            printf("key %06x:", kv.first);
            for (int b : kv.second)
                printf("  [wg %3d hw %08x tg %u wave %u simd %u lds_alloc %08x]", b, h[b * 4], (h[b * 4] >> 16) & 0xf, h[b * 4] & 0xf,
                       (h[b * 4] >> 4) & 3, h[b * 4 + 1]);
            printf("\n");
            ++shown;
        }
    }
    printf("CUs with exactly two workgroups: %d; TG_ID parity differs in %d; LDS_BASE differs in %d\n", pairs, tgdiff, ldsdiff);
    return 0;
}
