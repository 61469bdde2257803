// Layouts of the hash-grid model family (BASELINE.json configs[3]; oracle/hashgrid_oracle.py): flat parameter vector
// and the small packed-weights buffer.  Host + device, constexpr only.
#pragma once
#include <stdint.h>

namespace dh {

constexpr int HM_HID = 64;        // hidden width of both small MLPs
constexpr int HM_GIN = 35;        // geometry input: 2x-1 (3) + 16 levels x 2 features
constexpr int HM_GOUT = 13;       // geometry output: sdf + 12 (all 13 are the "feature")
constexpr int HM_CIN = 32;        // colour input: feature(13) + SH4(16) + normal(3)
constexpr int HM_FEAT = 13;

// ---- flat parameter vector (floats); table first
struct HashParamOff {
    int64_t table;                         // [entries][2]
    int64_t g0_b, g0_g, g0_v;              // geometry lin0: 64 x 35
    int64_t g1_b, g1_g, g1_v;              // geometry lin1: 13 x 64
    int64_t variance;
    int64_t c0_b, c0_g, c0_v;              // colour lin0: 64 x 32
    int64_t c1_b, c1_g, c1_v;              // colour lin1: 64 x 64
    int64_t c2_b, c2_g, c2_v;              // colour lin2: 3 x 64
    int64_t total;
};
inline HashParamOff make_hash_param_off(int64_t entries) {
    HashParamOff p{};
    int64_t o = 0;
    p.table = o; o += entries * 2;
    p.g0_b = o; o += 64; p.g0_g = o; o += 64; p.g0_v = o; o += 64 * HM_GIN;
    p.g1_b = o; o += HM_GOUT; p.g1_g = o; o += HM_GOUT; p.g1_v = o; o += HM_GOUT * 64;
    p.variance = o; o += 1;
    p.c0_b = o; o += 64; p.c0_g = o; o += 64; p.c0_v = o; o += 64 * HM_CIN;
    p.c1_b = o; o += 64; p.c1_g = o; o += 64; p.c1_v = o; o += 64 * 64;
    p.c2_b = o; o += 3; p.c2_g = o; o += 3; p.c2_v = o; o += 3 * 64;
    p.total = o;
    return p;
}

// ---- packed small weights (effective W = g v/||v||, row-major, padded), loaded whole into LDS by every kernel
constexpr int HP_G0 = 0;                       // [64][36]
constexpr int HP_G0B = HP_G0 + 64 * 36;        // [64]
constexpr int HP_G1 = HP_G0B + 64;             // [16][64] (13 valid)
constexpr int HP_G1B = HP_G1 + 16 * 64;        // [16]
constexpr int HP_C0 = HP_G1B + 16;             // [64][32]
constexpr int HP_C0B = HP_C0 + 64 * 32;        // [64]
constexpr int HP_C1 = HP_C0B + 64;             // [64][64]
constexpr int HP_C1B = HP_C1 + 64 * 64;        // [64]
constexpr int HP_C2 = HP_C1B + 64;             // [4][64] (3 valid)
constexpr int HP_C2B = HP_C2 + 4 * 64;         // [4]
constexpr int HP_WEIGHTS = HP_C2B + 4;         // floats every kernel stages into LDS
constexpr int HP_RS = (HP_WEIGHTS + 3) / 4 * 4;    // g/||v|| per row: g0 64, g1 16, c0 64, c1 64, c2 4
constexpr int HP_INV = HP_RS + 212;                // 1/||v|| per row, same order
constexpr int HP_TOTAL = (HP_INV + 212 + 3) / 4 * 4;
constexpr int HP_ROW_G0 = 0, HP_ROW_G1 = 64, HP_ROW_C0 = 80, HP_ROW_C1 = 144, HP_ROW_C2 = 208;

}  // namespace dh
