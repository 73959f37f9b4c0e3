// m2v_tables.hpp — constant tables of the MI355X MPEG-2 encoder (product copy, GPU layout).
//
// Values are ISO/IEC 13818-2 tables B-9/B-10/B-12/B-13/B-14, the default intra matrix, the
// zig-zag scan and the integer DCT basis the reference uses (RTL/mpeg2encoder.v:105-112,
// 131-138, 156-163, 185-245, 258-739).  tests/test_abi.py::test_product_tables_match_oracle_tables checks every entry
// against the oracle's independent copy (through m2v_debug_table), tests/test_tables_vs_rtl.py the oracle's against the RTL's `assign` lines.
//
// VLC entries are packed (len << 8) | code; `len` bits, MSB first.
#pragma once
#include <stdint.h>

namespace m2v {

#define M2V_VLC(code, len) (uint16_t)(((len) << 8) | (code))

// forward DCT basis [frequency][sample]
static const int8_t kDctBasis[64] = {
    64,  64,  64,  64,  64,  64,  64,  64,
    89,  75,  50,  18, -18, -50, -75, -89,
    84,  35, -35, -84, -84, -35,  35,  84,
    75, -18, -89, -50,  50,  89,  18, -75,
    64, -64, -64,  64,  64, -64, -64,  64,
    50, -89,  18,  75, -75, -18,  89, -50,
    35, -84,  84, -35, -35,  84, -84,  35,
    18, -50,  75, -89,  89, -75,  50, -18,
};

// default intra quantiser matrix [v][u]
static const uint8_t kIntraW[64] = {
     8, 16, 19, 22, 26, 27, 29, 34,
    16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38,
    22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48,
    26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69,
    27, 29, 35, 38, 46, 56, 69, 83,
};

// zig-zag scan position of raster coefficient [v][u]
static const uint8_t kZigzagPos[64] = {
     0,  1,  5,  6, 14, 15, 27, 28,
     2,  4,  7, 13, 16, 26, 29, 42,
     3,  8, 12, 17, 25, 30, 41, 43,
     9, 11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54,
    20, 22, 33, 38, 46, 51, 55, 60,
    21, 34, 37, 47, 50, 56, 59, 61,
    35, 36, 48, 49, 57, 58, 62, 63,
};

// B-10 motion_code by |delta|
static const uint16_t kMotionCode[17] = {
    M2V_VLC(0x01, 1), M2V_VLC(0x01, 2), M2V_VLC(0x01, 3), M2V_VLC(0x01, 4), M2V_VLC(0x03, 6),
    M2V_VLC(0x05, 7), M2V_VLC(0x04, 7), M2V_VLC(0x03, 7), M2V_VLC(0x0b, 9), M2V_VLC(0x0a, 9),
    M2V_VLC(0x09, 9), M2V_VLC(0x11,10), M2V_VLC(0x10,10), M2V_VLC(0x0f,10), M2V_VLC(0x0e,10),
    M2V_VLC(0x0d,10), M2V_VLC(0x0c,10),
};

// B-9 coded_block_pattern by cbp (bit5 = Y00 .. bit0 = V)
static const uint16_t kCbpCode[64] = {
    M2V_VLC(0x00,0), M2V_VLC(0x0b,5), M2V_VLC(0x09,5), M2V_VLC(0x0d,6), M2V_VLC(0x0d,4), M2V_VLC(0x17,7), M2V_VLC(0x13,7), M2V_VLC(0x1f,8),
    M2V_VLC(0x0c,4), M2V_VLC(0x16,7), M2V_VLC(0x12,7), M2V_VLC(0x1e,8), M2V_VLC(0x13,5), M2V_VLC(0x1b,8), M2V_VLC(0x17,8), M2V_VLC(0x13,8),
    M2V_VLC(0x0b,4), M2V_VLC(0x15,7), M2V_VLC(0x11,7), M2V_VLC(0x1d,8), M2V_VLC(0x11,5), M2V_VLC(0x19,8), M2V_VLC(0x15,8), M2V_VLC(0x11,8),
    M2V_VLC(0x0f,6), M2V_VLC(0x0f,8), M2V_VLC(0x0d,8), M2V_VLC(0x03,9), M2V_VLC(0x0f,5), M2V_VLC(0x0b,8), M2V_VLC(0x07,8), M2V_VLC(0x07,9),
    M2V_VLC(0x0a,4), M2V_VLC(0x14,7), M2V_VLC(0x10,7), M2V_VLC(0x1c,8), M2V_VLC(0x0e,6), M2V_VLC(0x0e,8), M2V_VLC(0x0c,8), M2V_VLC(0x02,9),
    M2V_VLC(0x10,5), M2V_VLC(0x18,8), M2V_VLC(0x14,8), M2V_VLC(0x10,8), M2V_VLC(0x0e,5), M2V_VLC(0x0a,8), M2V_VLC(0x06,8), M2V_VLC(0x06,9),
    M2V_VLC(0x12,5), M2V_VLC(0x1a,8), M2V_VLC(0x16,8), M2V_VLC(0x12,8), M2V_VLC(0x0d,5), M2V_VLC(0x09,8), M2V_VLC(0x05,8), M2V_VLC(0x05,9),
    M2V_VLC(0x0c,5), M2V_VLC(0x08,8), M2V_VLC(0x04,8), M2V_VLC(0x04,9), M2V_VLC(0x07,3), M2V_VLC(0x0a,5), M2V_VLC(0x08,5), M2V_VLC(0x0c,6),
};

// B-12 / B-13 dct_dc_size (code needs up to 10 bits: kept as separate arrays)
static const uint16_t kDcSizeCode[2][12] = {
    { 0x004, 0x000, 0x001, 0x005, 0x006, 0x00e, 0x01e, 0x03e, 0x07e, 0x0fe, 0x1fe, 0x1ff },
    { 0x000, 0x001, 0x002, 0x006, 0x00e, 0x01e, 0x03e, 0x07e, 0x0fe, 0x1fe, 0x3fe, 0x3ff },
};
static const uint8_t kDcSizeLen[2][12] = {
    { 3, 2, 2, 3, 3, 4, 5, 6, 7, 8, 9, 9 },
    { 2, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10 },
};

// B-14 table zero without the sign bit: kAcCode[run * 40 + |level| - 1]; 0 = escape
#define V M2V_VLC
static const uint16_t kAcCode[32 * 40] = {
    /* run 0 */ V(0x03,2),V(0x04,4),V(0x05,5),V(0x06,7),V(0x26,8),V(0x21,8),V(0x0a,10),V(0x1d,12),V(0x18,12),V(0x13,12),
                V(0x10,12),V(0x1a,13),V(0x19,13),V(0x18,13),V(0x17,13),V(0x1f,14),V(0x1e,14),V(0x1d,14),V(0x1c,14),V(0x1b,14),
                V(0x1a,14),V(0x19,14),V(0x18,14),V(0x17,14),V(0x16,14),V(0x15,14),V(0x14,14),V(0x13,14),V(0x12,14),V(0x11,14),
                V(0x10,14),V(0x18,15),V(0x17,15),V(0x16,15),V(0x15,15),V(0x14,15),V(0x13,15),V(0x12,15),V(0x11,15),V(0x10,15),
    /* run 1 */ V(0x03,3),V(0x06,6),V(0x25,8),V(0x0c,10),V(0x1b,12),V(0x16,13),V(0x15,13),V(0x1f,15),V(0x1e,15),V(0x1d,15),
                V(0x1c,15),V(0x1b,15),V(0x1a,15),V(0x19,15),V(0x13,16),V(0x12,16),V(0x11,16),V(0x10,16),0,0,
                0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run 2 */ V(0x05,4),V(0x04,7),V(0x0b,10),V(0x14,12),V(0x14,13),0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run 3 */ V(0x07,5),V(0x24,8),V(0x1c,12),V(0x13,13),0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run 4 */ V(0x06,5),V(0x0f,10),V(0x12,12),0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run 5 */ V(0x07,6),V(0x09,10),V(0x12,13),0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run 6 */ V(0x05,6),V(0x1e,12),V(0x14,16),0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run 7 */ V(0x04,6),V(0x15,12),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run 8 */ V(0x07,7),V(0x11,12),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run 9 */ V(0x05,7),V(0x11,13),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run10 */ V(0x27,8),V(0x10,13),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run11 */ V(0x23,8),V(0x1a,16),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run12 */ V(0x22,8),V(0x19,16),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run13 */ V(0x20,8),V(0x18,16),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run14 */ V(0x0e,10),V(0x17,16),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run15 */ V(0x0d,10),V(0x16,16),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run16 */ V(0x08,10),V(0x15,16),0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
#define R1(c,l) V(c,l),0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,
    /* run17-21 */ R1(0x1f,12) R1(0x1a,12) R1(0x19,12) R1(0x17,12) R1(0x16,12)
    /* run22-26 */ R1(0x1f,13) R1(0x1e,13) R1(0x1d,13) R1(0x1c,13) R1(0x1b,13)
    /* run27-31 */ R1(0x1f,16) R1(0x1e,16) R1(0x1d,16) R1(0x1c,16) R1(0x1b,16)
#undef R1
};
#undef V
#undef M2V_VLC

// Chen-Wang IDCT constants (RTL/mpeg2encoder.v:169-174)
constexpr int kW1 = 2841, kW2 = 2676, kW3 = 2408, kW5 = 1609, kW6 = 1108, kW7 = 565;

}  // namespace m2v
