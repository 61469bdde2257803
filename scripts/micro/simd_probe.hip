// Development probe (round 5): on which SIMDs do the waves of a workgroup land?  512 workgroups x 256 threads with 80 KB of LDS (two per
// CU, the tile-resident chains' launch shape) and 256 workgroups x 512 threads with 160 KB; HW_ID of EVERY wave.
//   hipcc --offload-arch=gfx950 -O2 -o simd_probe_micro simd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <set>
template <int THREADS, int LDSKB>
__global__ __launch_bounds__(THREADS) void probe(unsigned* out, int spin) {
    __shared__ float big[LDSKB * 256];
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        out[(blockIdx.x * (THREADS / 64) + w) * 2 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_ID
        out[(blockIdx.x * (THREADS / 64) + w) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) + (unsigned)big[3] * 0u;   // XCC_ID
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
}
template <int THREADS>
static void report(const std::vector<unsigned>& h, int nwg, const char* what) {
    constexpr int NW = THREADS / 64;
    std::map<std::string, int> pattern;
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < nwg; ++b) {
        std::string p;
        for (int w = 0; w < NW; ++w) { p += char('0' + ((h[(b * NW + w) * 2] >> 4) & 3)); }
        ++pattern[p];
        const unsigned hw = h[b * NW * 2], xcc = h[b * NW * 2 + 1] & 0xf;
        cu[(xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xf)].push_back(b);
    }
    printf("%s: SIMD id of waves 0..%d of a workgroup -> number of workgroups\n", what, NW - 1);
    for (auto& kv : pattern) printf("   %s : %d\n", kv.first.c_str(), kv.second);
    // per CU: how many waves on each SIMD
    std::map<std::string, int> percu;
    for (auto& kv : cu) {
        int cnt[4] = {0, 0, 0, 0};
        for (int b : kv.second) for (int w = 0; w < NW; ++w) ++cnt[(h[(b * NW + w) * 2] >> 4) & 3];
        char buf[64]; snprintf(buf, sizeof buf, "%zu wgs: %d %d %d %d", kv.second.size(), cnt[0], cnt[1], cnt[2], cnt[3]);
        ++percu[buf];
    }
    printf("   per CU (workgroups: waves on SIMD 0 1 2 3) -> number of CUs\n");
    for (auto& kv : percu) printf("   %s : %d\n", kv.first.c_str(), kv.second);
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 8 * 8);
    std::vector<unsigned> h(512 * 8 * 2);
    probe<256, 80><<<512, 256>>>(d, 2000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 512 * 4 * 8, hipMemcpyDeviceToHost);
    report<256>(h, 512, "512 workgroups x 256 threads, 80 KB LDS");
    probe<512, 159><<<256, 512>>>(d, 2000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 256 * 8 * 8, hipMemcpyDeviceToHost);
    report<512>(h, 256, "256 workgroups x 512 threads, 159 KB LDS");
    probe<256, 159><<<256, 256>>>(d, 2000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 256 * 4 * 8, hipMemcpyDeviceToHost);
    report<256>(h, 256, "256 workgroups x 256 threads, 159 KB LDS (one per CU)");
    return 0;
}
