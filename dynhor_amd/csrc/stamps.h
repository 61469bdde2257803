// Diagnostic build only (-DDH_STAMPS, scripts/stamps.sh): per-phase s_memtime stamps of the chain kernels, written to a
// buffer nothing else reads (MI355X_MICROARCH.md "DVFS give-back" item 6 / cdna_hip_programming.md section 7).  In the
// product build DH_STAMP expands to nothing and no stamp executes.
#pragma once
#ifdef DH_STAMPS
namespace dh {
constexpr int STAMP_BLOCKS = 512, STAMP_ITERS = 2, STAMP_LAYERS = 10, STAMP_SLOTS = 8;
static __device__ unsigned long long dh_stamps[STAMP_BLOCKS * 4 * STAMP_ITERS * STAMP_LAYERS * STAMP_SLOTS];
}
// it = tile-loop iteration of this workgroup: iterations 2 and 5 are recorded (steady state)
#define DH_STAMP(it, layer, slot)                                                                                      \
    do {                                                                                                               \
        if (((it) == 2 || (it) == 5) && lane == 0 && blockIdx.x < dh::STAMP_BLOCKS)                                     \
            dh::dh_stamps[((((size_t)blockIdx.x * 4 + wave) * dh::STAMP_ITERS + ((it) == 5)) * dh::STAMP_LAYERS + (layer)) * dh::STAMP_SLOTS + (slot)] = \
                __builtin_readcyclecounter();                                                                         \
    } while (0)
#define DH_STAMP_READER(name)                                                                                          \
    extern "C" int name(unsigned long long* host, long long n) {                                                       \
        const long long total = (long long)(sizeof(dh::dh_stamps) / sizeof(unsigned long long));                        \
        if (n > total) n = total;                                                                                      \
        return hipMemcpyFromSymbol(host, HIP_SYMBOL(dh::dh_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -3; \
    }
#else
#define DH_STAMP(it, layer, slot) do { } while (0)
#define DH_STAMP_READER(name)
#endif
