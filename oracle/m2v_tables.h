/*
 * m2v_tables.h — constant tables of the MPEG-2 I/P encoder, ORACLE copy.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  These are the ISO/IEC 13818-2
 * tables B-9, B-10, B-12, B-13, B-14, the default intra quantiser matrix, the
 * zig-zag scan and the HEVC-style integer DCT basis, written out in this repo's
 * own layout.  tests/test_tables_vs_rtl.py checks every entry against the live
 * `assign` lines of the reference RTL when /root/reference is mounted:
 *
 *   DCT basis      RTL/mpeg2encoder.v:105-112
 *   intra matrix   RTL/mpeg2encoder.v:131-138
 *   zig-zag        RTL/mpeg2encoder.v:156-163
 *   motion code    RTL/mpeg2encoder.v:185-194   (B-10)
 *   cbp            RTL/mpeg2encoder.v:202-219   (B-9)
 *   dc size Y/UV   RTL/mpeg2encoder.v:230-245   (B-12, B-13)
 *   AC run/level   RTL/mpeg2encoder.v:258-739   (B-14, table zero)
 *
 * A VLC entry is {code, len}: `len` bits, MSB first, value `code`
 * (leading zeros come from the length).
 */
#ifndef M2V_ORACLE_TABLES_H
#define M2V_ORACLE_TABLES_H

#include <stdint.h>

typedef struct { uint16_t code; uint8_t len; } m2v_vlc;

/* forward DCT basis, row = frequency, col = sample */
static const int8_t M2V_DCT_BASIS[8][8] = {
    { 64,  64,  64,  64,  64,  64,  64,  64 },
    { 89,  75,  50,  18, -18, -50, -75, -89 },
    { 84,  35, -35, -84, -84, -35,  35,  84 },
    { 75, -18, -89, -50,  50,  89,  18, -75 },
    { 64, -64, -64,  64,  64, -64, -64,  64 },
    { 50, -89,  18,  75, -75, -18,  89, -50 },
    { 35, -84,  84, -35, -35,  84, -84,  35 },
    { 18, -50,  75, -89,  89, -75,  50, -18 },
};

/* MPEG-2 default intra quantiser matrix, [vertical freq][horizontal freq] */
static const uint8_t M2V_INTRA_W[8][8] = {
    {  8, 16, 19, 22, 26, 27, 29, 34 },
    { 16, 16, 22, 24, 27, 29, 34, 37 },
    { 19, 22, 26, 27, 29, 34, 34, 38 },
    { 22, 22, 26, 27, 29, 34, 37, 40 },
    { 22, 26, 27, 29, 32, 35, 40, 48 },
    { 26, 27, 29, 32, 35, 40, 48, 58 },
    { 26, 27, 29, 34, 38, 46, 56, 69 },
    { 27, 29, 35, 38, 46, 56, 69, 83 },
};

/* scan position of coefficient [v][u] in the (non-alternate) zig-zag order */
static const uint8_t M2V_ZIGZAG_POS[8][8] = {
    {  0,  1,  5,  6, 14, 15, 27, 28 },
    {  2,  4,  7, 13, 16, 26, 29, 42 },
    {  3,  8, 12, 17, 25, 30, 41, 43 },
    {  9, 11, 18, 24, 31, 40, 44, 53 },
    { 10, 19, 23, 32, 39, 45, 52, 54 },
    { 20, 22, 33, 38, 46, 51, 55, 60 },
    { 21, 34, 37, 47, 50, 56, 59, 61 },
    { 35, 36, 48, 49, 57, 58, 62, 63 },
};

/* B-10 motion_code, index = |delta| (sign bit follows when delta != 0) */
static const m2v_vlc M2V_MOTION_CODE[17] = {
    {0x01, 1}, {0x01, 2}, {0x01, 3}, {0x01, 4}, {0x03, 6}, {0x05, 7}, {0x04, 7}, {0x03, 7},
    {0x0b, 9}, {0x0a, 9}, {0x09, 9}, {0x11,10}, {0x10,10}, {0x0f,10}, {0x0e,10}, {0x0d,10},
    {0x0c,10},
};

/* B-9 coded_block_pattern, index = cbp (bit5 = Y00 ... bit0 = V); cbp 0 has no code */
static const m2v_vlc M2V_CBP_CODE[64] = {
    {0x00,0}, {0x0b,5}, {0x09,5}, {0x0d,6}, {0x0d,4}, {0x17,7}, {0x13,7}, {0x1f,8},
    {0x0c,4}, {0x16,7}, {0x12,7}, {0x1e,8}, {0x13,5}, {0x1b,8}, {0x17,8}, {0x13,8},
    {0x0b,4}, {0x15,7}, {0x11,7}, {0x1d,8}, {0x11,5}, {0x19,8}, {0x15,8}, {0x11,8},
    {0x0f,6}, {0x0f,8}, {0x0d,8}, {0x03,9}, {0x0f,5}, {0x0b,8}, {0x07,8}, {0x07,9},
    {0x0a,4}, {0x14,7}, {0x10,7}, {0x1c,8}, {0x0e,6}, {0x0e,8}, {0x0c,8}, {0x02,9},
    {0x10,5}, {0x18,8}, {0x14,8}, {0x10,8}, {0x0e,5}, {0x0a,8}, {0x06,8}, {0x06,9},
    {0x12,5}, {0x1a,8}, {0x16,8}, {0x12,8}, {0x0d,5}, {0x09,8}, {0x05,8}, {0x05,9},
    {0x0c,5}, {0x08,8}, {0x04,8}, {0x04,9}, {0x07,3}, {0x0a,5}, {0x08,5}, {0x0c,6},
};

/* B-12 / B-13 dct_dc_size, index = number of bits of |dc difference| */
static const m2v_vlc M2V_DC_SIZE_LUMA[12] = {
    {0x004,3}, {0x000,2}, {0x001,2}, {0x005,3}, {0x006,3}, {0x00e,4},
    {0x01e,5}, {0x03e,6}, {0x07e,7}, {0x0fe,8}, {0x1fe,9}, {0x1ff,9},
};
static const m2v_vlc M2V_DC_SIZE_CHROMA[12] = {
    {0x000,2}, {0x001,2}, {0x002,2}, {0x006,3}, {0x00e,4}, {0x01e,5},
    {0x03e,6}, {0x07e,7}, {0x0fe,8}, {0x1fe,9}, {0x3fe,10}, {0x3ff,10},
};

/*
 * B-14 (table zero) run/level codes WITHOUT the trailing sign bit.
 * M2V_AC_CODE[run][|level|-1]; len == 0 means "no VLC: use the 24-bit escape".
 * Run 0 / level 1 is the "11" form (the "1" first-coefficient form is handled
 * by the caller).  Runs >= 32 always escape.
 */
#define M2V_AC_MAX_LEVEL 40
static const m2v_vlc M2V_AC_CODE[32][M2V_AC_MAX_LEVEL] = {
    /* run 0 */ { {0x03,2},{0x04,4},{0x05,5},{0x06,7},{0x26,8},{0x21,8},{0x0a,10},{0x1d,12},{0x18,12},{0x13,12},
                  {0x10,12},{0x1a,13},{0x19,13},{0x18,13},{0x17,13},{0x1f,14},{0x1e,14},{0x1d,14},{0x1c,14},{0x1b,14},
                  {0x1a,14},{0x19,14},{0x18,14},{0x17,14},{0x16,14},{0x15,14},{0x14,14},{0x13,14},{0x12,14},{0x11,14},
                  {0x10,14},{0x18,15},{0x17,15},{0x16,15},{0x15,15},{0x14,15},{0x13,15},{0x12,15},{0x11,15},{0x10,15} },
    /* run 1 */ { {0x03,3},{0x06,6},{0x25,8},{0x0c,10},{0x1b,12},{0x16,13},{0x15,13},{0x1f,15},{0x1e,15},{0x1d,15},
                  {0x1c,15},{0x1b,15},{0x1a,15},{0x19,15},{0x13,16},{0x12,16},{0x11,16},{0x10,16} },
    /* run 2 */ { {0x05,4},{0x04,7},{0x0b,10},{0x14,12},{0x14,13} },
    /* run 3 */ { {0x07,5},{0x24,8},{0x1c,12},{0x13,13} },
    /* run 4 */ { {0x06,5},{0x0f,10},{0x12,12} },
    /* run 5 */ { {0x07,6},{0x09,10},{0x12,13} },
    /* run 6 */ { {0x05,6},{0x1e,12},{0x14,16} },
    /* run 7 */ { {0x04,6},{0x15,12} },
    /* run 8 */ { {0x07,7},{0x11,12} },
    /* run 9 */ { {0x05,7},{0x11,13} },
    /* run10 */ { {0x27,8},{0x10,13} },
    /* run11 */ { {0x23,8},{0x1a,16} },
    /* run12 */ { {0x22,8},{0x19,16} },
    /* run13 */ { {0x20,8},{0x18,16} },
    /* run14 */ { {0x0e,10},{0x17,16} },
    /* run15 */ { {0x0d,10},{0x16,16} },
    /* run16 */ { {0x08,10},{0x15,16} },
    /* run17 */ { {0x1f,12} }, /* run18 */ { {0x1a,12} }, /* run19 */ { {0x19,12} },
    /* run20 */ { {0x17,12} }, /* run21 */ { {0x16,12} },
    /* run22 */ { {0x1f,13} }, /* run23 */ { {0x1e,13} }, /* run24 */ { {0x1d,13} },
    /* run25 */ { {0x1c,13} }, /* run26 */ { {0x1b,13} },
    /* run27 */ { {0x1f,16} }, /* run28 */ { {0x1e,16} }, /* run29 */ { {0x1d,16} },
    /* run30 */ { {0x1c,16} }, /* run31 */ { {0x1b,16} },
};

/* Chen-Wang IDCT constants 2048*sqrt(2)*cos(k*pi/16)  (RTL/mpeg2encoder.v:169-174) */
#define M2V_W1 2841
#define M2V_W2 2676
#define M2V_W3 2408
#define M2V_W5 1609
#define M2V_W6 1108
#define M2V_W7  565

#endif
