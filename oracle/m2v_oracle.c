/*
 * m2v_oracle.c — CPU restatement of RTL/mpeg2encoder.v, plain C99, single thread.
 *
 * TEST INFRASTRUCTURE ONLY (see m2v_oracle.h: "PARITY UNPINNED").
 *
 * Every function cites the RTL lines it restates as `RTL:n` =
 * /root/reference/RTL/mpeg2encoder.v:n.  The RTL is a 64-clock-per-macroblock
 * shift-register pipeline; this file restates WHAT each stage computes, with the
 * register widths, wrap-arounds, tie-breaks and rounding quirks kept.
 */
#include "m2v_oracle.h"
#include "m2v_tables.h"

#include <stdlib.h>
#include <string.h>

/* Mutation switch (tests/test_oracle_mutants.py; oracle/Makefile target `mutants`).  The oracle is a hand restatement that
 * nothing in this image can check against an executed RTL, so its guard is a SECOND, structurally different restatement
 * (tests/rtl_stage_*.py, tests/rtl_module.py).  -DM2V_ORACLE_MUTANT=k builds this file with ONE deliberate mis-reading of
 * an RTL quirk; the test asserts every such library is caught by the second restatement.  0 (the
 * default, the only value the checker is ever built with) = the reading the parity tests rely on. */
#ifndef M2V_ORACLE_MUTANT
#define M2V_ORACLE_MUTANT 0
#endif
#define MUT(k) (M2V_ORACLE_MUTANT == (k))
int m2v_oracle_mutant(void) { return M2V_ORACLE_MUTANT; }

/* ------------------------------------------------------------------------------------------
 * geometry (RTL:55-72, 985-1006)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int W, H;             /* clamped frame size in pixels     */
    int mbw, mbh;         /* macroblocks per row / column     */
    int max_x16, max_y16; /* index of the last MB column/row  */
    int UR, YR;           /* chroma / luma search range       */
    int Q;                /* Q_LEVEL                          */
} geom_t;

static int params_ok(const m2v_oracle_params *p)
{
    return p && p->XL >= 4 && p->XL <= 7 && p->YL >= 4 && p->YL <= 7 &&
           p->VECTOR_LEVEL >= 1 && p->VECTOR_LEVEL <= 3 && p->Q_LEVEL >= 1 && p->Q_LEVEL <= 4;
}

/* RTL:985-991: sizes above 2^XL clamp to 2^XL blocks, below 4 clamp to 4 blocks */
static int clamp_size16(unsigned size16, int L)
{
    unsigned lim = 1u << L;
    if (size16 > lim) return (int)lim - 1;
    if (size16 < 4)   return 3;
    return (int)size16 - 1;
}

static int make_geom(const m2v_oracle_params *p, unsigned xsize16, unsigned ysize16, geom_t *g)
{
    if (!params_ok(p)) return -1;
    /* the ports are XL+1 / YL+1 bits wide (RTL:20-21): wider values cannot be driven */
    xsize16 &= (2u << p->XL) - 1;
    ysize16 &= (2u << p->YL) - 1;
    g->max_x16 = clamp_size16(xsize16, p->XL);
    g->max_y16 = clamp_size16(ysize16, p->YL);
    g->mbw = g->max_x16 + 1;
    g->mbh = g->max_y16 + 1;
    g->W = 16 * g->mbw;
    g->H = 16 * g->mbh;
    g->UR = p->VECTOR_LEVEL;
    g->YR = 2 * p->VECTOR_LEVEL;
    g->Q = p->Q_LEVEL;
    return 0;
}

int m2v_oracle_geometry(const m2v_oracle_params *p, unsigned xsize16, unsigned ysize16,
                        int *width, int *height)
{
    geom_t g;
    if (make_geom(p, xsize16, ysize16, &g)) return -1;
    if (width) *width = g.W;
    if (height) *height = g.H;
    return 0;
}

size_t m2v_oracle_frame_count(const m2v_oracle_params *p, unsigned xsize16, unsigned ysize16,
                              size_t nbeats)
{
    geom_t g;
    if (make_geom(p, xsize16, ysize16, &g)) return 0;
    size_t bpf = (size_t)g.W * g.H / 4;
    return (nbeats + bpf - 1) / bpf;
}

/* ------------------------------------------------------------------------------------------
 * small helpers (RTL:750-795)
 * ---------------------------------------------------------------------------------------- */
static inline int mean2(int a, int b)                 { return (a + b + 1) >> 1; }      /* RTL:750-757 */
/* Optional standard-conformant reconstruction loop (NOT a mode of the reference, it pins the GPU
 * path's option "conformant"): ISO/IEC 13818-2 where the RTL deviates from it - mean4 rounds with +2 (7.6.4), the
 * 4:2:0 chroma vector is mv / 2 truncated toward zero (7.6.3.7), inverse quantisation truncates toward zero,
 * saturates to [-2048, 2047] and applies mismatch control (7.4.2.3, 7.4.3, 7.4.4), the IDCT keeps its row pass in
 * full width and saturates to [-256, 255] (Annex A).  Process-global, tests switch it serially. */
static int g_conformant = 0;
void m2v_oracle_set_conformant(int on) { g_conformant = on != 0; }
int  m2v_oracle_get_conformant(void)   { return g_conformant; }

static inline int mean4(int a, int b, int c, int d, int r4) { return (a + b + c + d + r4) >> 2; } /* RTL:760-767: r4 = 1, not 2 (2 = ISO, conformant mode only) */
static inline int absdiff(int a, int b)               { return a > b ? a - b : b - a; } /* RTL:770-775 */
static inline int32_t sext(int32_t v, int bits)
{
    uint32_t m = 1u << (bits - 1);
    uint32_t x = (uint32_t)v & ((m << 1) - 1);
    return (int32_t)((x ^ m) - m);
}

/* ------------------------------------------------------------------------------------------
 * stages U/V: bit packer (RTL:2888-2956) and output word order (RTL:2963-2994)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *buf;
    size_t   cap;
    uint64_t nbits;
} bitw_t;

/* append `len` bits of `v`, MSB first (fields of stage T, RTL:2897-2904) */
static void bw_put(bitw_t *w, uint32_t v, int len)
{
    while (len > 0) {
        int bitpos = (int)(w->nbits & 7);
        int room = 8 - bitpos;
        int take = len < room ? len : room;
        uint32_t chunk = (v >> (len - take)) & ((1u << take) - 1u);
        size_t byte = (size_t)(w->nbits >> 3);
        if (byte < w->cap) {
            if (bitpos == 0) w->buf[byte] = 0;
            w->buf[byte] |= (uint8_t)(chunk << (room - take));
        }
        w->nbits += (unsigned)take;
        len -= take;
    }
}

/* t_align: zero-pad to a byte boundary before a header (RTL:2940-2943) */
static void bw_align(bitw_t *w)
{
    if (w->nbits & 7) bw_put(w, 0, 8 - (int)(w->nbits & 7));
}

/* end of sequence: the residual (<256 bits, possibly none) always leaves as one more
 * zero-padded 32-byte word with o_last (RTL:2932-2937) */
static size_t bw_finish(bitw_t *w)
{
    size_t words = MUT(13) ? (size_t)((w->nbits + 255) / 256)      /* mutant 13: no extra word when the stream ends on a word boundary */
                           : (size_t)(w->nbits / 256) + 1;
    size_t total = words * 32;
    size_t used = (size_t)((w->nbits + 7) >> 3);
    for (size_t i = used; i < total && i < w->cap; ++i) w->buf[i] = 0;
    return total;
}

/* ------------------------------------------------------------------------------------------
 * stages A-C: 4:4:4 -> 4:2:0 (RTL:1086-1089 horizontal mean2, RTL:1116-1125 + 1167-1170
 * vertical mean2 of the two horizontally-averaged rows, stored on odd rows RTL:1211)
 * ---------------------------------------------------------------------------------------- */
void m2v_oracle_subsample(const uint8_t *p, int W, int H, uint8_t *o)
{
    for (int j = 0; j < H / 2; ++j)
        for (int k = 0; k < W / 2; ++k) {
            int h0 = mean2(p[(2 * j) * W + 2 * k], p[(2 * j) * W + 2 * k + 1]);
            int h1 = mean2(p[(2 * j + 1) * W + 2 * k], p[(2 * j + 1) * W + 2 * k + 1]);
            o[j * (W / 2) + k] = (uint8_t)mean2(h1, h0);
            if (MUT(16))    /* mutant: one rounding over the four samples instead of the RTL's two stages (RTL:1086-1089, 1167-1170) */
                o[j * (W / 2) + k] = (uint8_t)((p[(2 * j) * W + 2 * k] + p[(2 * j) * W + 2 * k + 1] + p[(2 * j + 1) * W + 2 * k] +
                                                p[(2 * j + 1) * W + 2 * k + 1] + 2) >> 2);
        }
}

/* ------------------------------------------------------------------------------------------
 * stage G: forward DCT (RTL:2029-2062) and quantiser (RTL:2065-2077)
 * ---------------------------------------------------------------------------------------- */
void m2v_oracle_fdct(const int16_t x[64], int32_t c[64])
{
    int32_t r1[64];
    /* phase 1: R1 = X * DCTM^T, 19-bit signed (RTL:2029-2036) */
    for (int r = 0; r < 8; ++r)
        for (int j = 0; j < 8; ++j) {
            int32_t t = 0;
            for (int k = 0; k < 8; ++k) t += x[r * 8 + k] * M2V_DCT_BASIS[j][k];
            r1[r * 8 + j] = sext(t, 19);
        }
    /* phase 2: C = (DCTM * R1 + 2048) >> 12, kept in 17 bits (RTL:2054-2062) */
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            int32_t t = 0;
            for (int k = 0; k < 8; ++k) t += M2V_DCT_BASIS[i][k] * r1[k * 8 + j];
            t = sext(t, 29);
            t = (t >> 12) + ((t >> 11) & 1);
            c[i * 8 + j] = sext(t, 17);
        }
}

void m2v_oracle_quant(const int32_t c[64], int inter, int Q, int16_t q[64])
{
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            int32_t v = c[i * 8 + j];
            uint32_t a = (uint32_t)(v < 0 ? -v : v) & 0xFFFFu;          /* g_t3 is 16 bits, RTL:2068 */
            if (inter) {
                a = ((a + (MUT(19) ? 0u : 2u)) & 0xFFFFu) >> (4 + Q);  /* RTL:2070 (mutant 19: no "+ 2") */
            } else if (i != 0 || j != 0) {
                uint32_t w = M2V_INTRA_W[i][j];
                a = ((a + ((w * ((3u << Q) + 2u)) >> 3)) >> Q) / w;    /* RTL:2072 */
                a &= 0xFFFFu;
            } else {
                a = (a >> 4) + (MUT(9) ? 0u : (a >> 3) & 1u);         /* RTL:2074 (mutant 9: truncation, no a[3] rounding) */
            }
            if (a > 2047u) a = 2047u;                                  /* RTL:2075 */
            q[i * 8 + j] = (int16_t)(v < 0 ? -(int32_t)a : (int32_t)a); /* RTL:2076 */
        }
}

/* ------------------------------------------------------------------------------------------
 * stage H: inverse quantiser (RTL:2129-2150), 17-bit signed temporary, no mismatch control
 * ---------------------------------------------------------------------------------------- */
static void dequant_conformant(const int16_t q[64], int inter, int Q, int16_t d[64])
{
    const int qs = 2 << Q;                                  /* quantiser_scale: the slice header carries code 1 << Q, q_scale_type 0 */
    int sum = 0;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            const int32_t qf = q[i * 8 + j];
            int32_t t;
            if (inter)                 t = ((2 * qf + (qf > 0 ? 1 : qf < 0 ? -1 : 0)) * 16 * qs) / 32;   /* 7.4.2.3, k = sign */
            else if (i != 0 || j != 0) t = (2 * qf * M2V_INTRA_W[i][j] * qs) / 32;                      /* k = 0; "/" truncates */
            else                       t = 2 * qf;                                                      /* intra_dc_mult, 10 bit */
            t = t < -2048 ? -2048 : t > 2047 ? 2047 : t;                                                /* 7.4.3 */
            d[i * 8 + j] = (int16_t)t;
            sum += t;
        }
    if ((sum & 1) == 0) d[63] ^= 1;                         /* 7.4.4: odd -> minus 1, even -> plus 1 = toggle the LSB */
}

void m2v_oracle_dequant(const int16_t q[64], int inter, int Q, int16_t d[64])
{
    if (g_conformant) { dequant_conformant(q, inter, Q, d); return; }
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            int32_t t = q[i * 8 + j];
            if (inter) {
                t = sext(t * 2, 17);                                    /* RTL:2134 (x << 1) */
                t = sext(t + (t < 0 ? -1 : t > 0 ? 1 : 0), 17);         /* RTL:2135 */
                t = sext((int32_t)((uint32_t)t << Q), 17);              /* RTL:2136 */
                t = t < -2047 ? -2047 : t > 2047 ? 2047 : t;            /* RTL:2137 */
            } else if (i != 0 || j != 0) {
                t = sext((int32_t)((uint32_t)t * M2V_INTRA_W[i][j]), MUT(8) ? 32 : 17); /* RTL:2139, modulo 2^17 (mutant 8: no wrap) */
                if (Q >= 3) t = sext((int32_t)((uint32_t)t << (Q - 3)), MUT(8) ? 32 : 17); /* RTL:2141 */
                else if (MUT(17)) t = t / (1 << (3 - Q));               /* mutant 17: toward zero, as ISO 7.4.2.3 would */
                else        t = t >> (3 - Q);                           /* RTL:2143, arithmetic */
                t = t < -2047 ? -2047 : t > 2047 ? 2047 : t;            /* RTL:2144 */
            } else {
                t = sext(t * 2, 17);                                    /* RTL:2146 (x << 1) */
            }
            d[i * 8 + j] = (int16_t)sext(t, 13);                        /* h_iquant is 13 bits */
        }
}

/* ------------------------------------------------------------------------------------------
 * stages J/K/M: Chen-Wang inverse DCT (RTL:844-972).  32-bit wrapping arithmetic; the row
 * pass result is stored in 18 bits (RTL:886, 2170); the column pass clips to +-255 (RTL:781).
 * ---------------------------------------------------------------------------------------- */
#define WRAP(x) ((int32_t)(uint32_t)(x))
static inline int32_t mul32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

static inline int32_t row_store(int32_t v, int conf) { return conf || MUT(7) ? v : sext(v, 18); }   /* RTL:886, 2170: 18-bit register (mutant 7: full width) */

static void idct_row(const int16_t a[8], int32_t r[8], int conf)
{
    int32_t x0 = a[0], x1 = a[4], x2 = a[6], x3 = a[2], x4 = a[1], x5 = a[7], x6 = a[5], x7 = a[3], x8;
    x0 = (int32_t)((uint32_t)x0 << 11);
    x1 = (int32_t)((uint32_t)x1 << 11);
    x0 |= 128;                                            /* RTL:859  x0[7] = 1 */
    /* step 1, RTL:861-866 */
    x8 = mul32(M2V_W7, add32(x4, x5));
    x4 = add32(x8, mul32(M2V_W1 - M2V_W7, x4));
    x5 = sub32(x8, mul32(M2V_W1 + M2V_W7, x5));
    x8 = mul32(M2V_W3, add32(x6, x7));
    x6 = sub32(x8, mul32(M2V_W3 - M2V_W5, x6));
    x7 = sub32(x8, mul32(M2V_W3 + M2V_W5, x7));
    /* step 2, RTL:868-876 */
    x8 = add32(x0, x1);
    x0 = sub32(x0, x1);
    x1 = mul32(M2V_W6, add32(x3, x2));
    x2 = sub32(x1, mul32(M2V_W2 + M2V_W6, x2));
    x3 = add32(x1, mul32(M2V_W2 - M2V_W6, x3));
    x1 = add32(x4, x6);
    x4 = sub32(x4, x6);
    x6 = add32(x5, x7);
    x5 = sub32(x5, x7);
    /* step 3, RTL:890-895 */
    x7 = add32(x8, x3);
    x8 = sub32(x8, x3);
    x3 = add32(x0, x2);
    x0 = sub32(x0, x2);
    x2 = add32(mul32(181, add32(x4, x5)), 128) >> 8;
    x4 = add32(mul32(181, sub32(x4, x5)), 128) >> 8;
    /* step 4, RTL:897-904: >>>8 then stored in 18 bits */
    r[0] = row_store(add32(x7, x1) >> 8, conf);
    r[1] = row_store(add32(x3, x2) >> 8, conf);
    r[2] = row_store(add32(x0, x4) >> 8, conf);
    r[3] = row_store(add32(x8, x6) >> 8, conf);
    r[4] = row_store(sub32(x8, x6) >> 8, conf);
    r[5] = row_store(sub32(x0, x4) >> 8, conf);
    r[6] = row_store(sub32(x3, x2) >> 8, conf);
    r[7] = row_store(sub32(x7, x1) >> 8, conf);
}

static inline int clip255(int32_t v, int conf)             /* RTL:778-783; the argument port is 28 bits */
{
    if (conf) return v < -256 ? -256 : v > 255 ? 255 : v;      /* Annex A saturation */
    v = sext(v, 28);
    return v < -255 ? -255 : v > 255 ? 255 : v;
}

static void idct_col(const int32_t a[8], int16_t r[8], int conf)
{
    int32_t x0 = a[0], x1 = a[4], x2 = a[6], x3 = a[2], x4 = a[1], x5 = a[7], x6 = a[5], x7 = a[3], x8;
    x0 = (int32_t)((uint32_t)x0 << 8);
    x1 = (int32_t)((uint32_t)x1 << 8);
    x0 = add32(x0, 8192);
    /* step 1, RTL:928-933 */
    x8 = add32(mul32(M2V_W7, add32(x4, x5)), 4);
    x4 = add32(x8, mul32(M2V_W1 - M2V_W7, x4)) >> 3;
    x5 = sub32(x8, mul32(M2V_W1 + M2V_W7, x5)) >> 3;
    x8 = add32(mul32(M2V_W3, add32(x6, x7)), 4);
    x6 = sub32(x8, mul32(M2V_W3 - M2V_W5, x6)) >> 3;
    x7 = sub32(x8, mul32(M2V_W3 + M2V_W5, x7)) >> 3;
    /* step 2, RTL:935-943 */
    x8 = add32(x0, x1);
    x0 = sub32(x0, x1);
    x1 = add32(mul32(M2V_W6, add32(x3, x2)), 4);
    x2 = sub32(x1, mul32(M2V_W2 + M2V_W6, x2)) >> 3;
    x3 = add32(x1, mul32(M2V_W2 - M2V_W6, x3)) >> 3;
    x1 = add32(x4, x6);
    x4 = sub32(x4, x6);
    x6 = add32(x5, x7);
    x5 = sub32(x5, x7);
    /* step 3, RTL:956-961 */
    x7 = add32(x8, x3);
    x8 = sub32(x8, x3);
    x3 = add32(x0, x2);
    x0 = sub32(x0, x2);
    x2 = add32(mul32(181, add32(x4, x5)), 128) >> 8;
    x4 = add32(mul32(181, sub32(x4, x5)), 128) >> 8;
    /* step 4, RTL:963-970 */
    r[0] = (int16_t)clip255(add32(x7, x1) >> 14, conf);
    r[1] = (int16_t)clip255(add32(x3, x2) >> 14, conf);
    r[2] = (int16_t)clip255(add32(x0, x4) >> 14, conf);
    r[3] = (int16_t)clip255(add32(x8, x6) >> 14, conf);
    r[4] = (int16_t)clip255(sub32(x8, x6) >> 14, conf);
    r[5] = (int16_t)clip255(sub32(x0, x4) >> 14, conf);
    r[6] = (int16_t)clip255(sub32(x3, x2) >> 14, conf);
    r[7] = (int16_t)clip255(sub32(x7, x1) >> 14, conf);
}

void m2v_oracle_idct(const int16_t d[64], int16_t r[64])
{
    int32_t rows[64];
    const int conf = g_conformant;
    for (int i = 0; i < 8; ++i) idct_row(d + 8 * i, rows + 8 * i, conf);   /* stage J, RTL:2159-2189 */
    for (int j = 0; j < 8; ++j) {                                    /* stages K/M, RTL:2238-2279 */
        int32_t col[8];
        int16_t out[8];
        for (int i = 0; i < 8; ++i) col[i] = rows[i * 8 + j];
        idct_col(col, out, conf);
        for (int i = 0; i < 8; ++i) r[i * 8 + j] = out[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * stage F: motion estimation, half-pel refinement, intra/inter decision, prediction
 * (RTL:1589-1918).  `ref*` = reconstruction of the previous frame.
 * ---------------------------------------------------------------------------------------- */
typedef struct { const uint8_t *Y, *U, *V; } planes_t;

static inline int pix(const uint8_t *p, int stride, int w, int h, int y, int x)
{
    /* samples outside the frame can never be selected (masks below); the RTL holds stale
       register contents there, any value is equivalent */
    if (y < 0 || x < 0 || y >= h || x >= w) return 0;
    return p[y * stride + x];
}

/* 10-way argmin with the RTL's tree tie-breaks (RTL:804-840) */
static int find_min_in_10_values(const int v[10])
{
    int wi1 = v[1] < v[0], w01 = wi1 ? v[1] : v[0];
    int wi3 = v[3] < v[2], w23 = wi3 ? v[3] : v[2];
    int wi5 = v[5] < v[4], w45 = wi5 ? v[5] : v[4];
    int wi7 = v[7] < v[6], w67 = wi7 ? v[7] : v[6];
    int wi9 = v[9] < v[8], w89 = wi9 ? v[9] : v[8];
    int xi23 = w23 < w01, x0123 = xi23 ? w23 : w01;
    int xi67 = w67 < w45, x4567 = xi67 ? w67 : w45;
    if (w89 <= x0123 && w89 <= x4567) return 8 + wi9;
    if (x0123 < x4567) return xi23 ? 2 + wi3 : 0 + wi1;
    return xi67 ? 6 + wi7 : 4 + wi5;
}

/* column-serial 12-bit SAD accumulator with sticky 13th bit (RTL:1664-1671, 1779-1786).
 * `col[c]` = sum of the 16 absolute differences of pixel column c (<= 4080).
 * Returns {over, diff[11:0]} as a 13-bit number. */
static int accumulate_sad13(const int col[16], int masked)
{
    int over = masked, diff = 0;
    for (int c = 0; c < 16; ++c)
        if (!over) {
            int s = diff + col[c];
            over = MUT(4) ? s >= 4095 : s >> 12;          /* mutant 4: the kill threshold one too low */
            diff = s & 0xFFF;
        }
    return (over << 12) | diff;
}

static void motion_stage(const geom_t *g, const uint8_t cy[256], const planes_t *ref, int bx, int by,
                         int *o_inter, int *o_mvx, int *o_mvy,
                         uint8_t py[256], uint8_t pu[64], uint8_t pv[64])
{
    const int YR = g->YR, W = g->W, H = g->H;
    const int x0 = 16 * bx, y0 = 16 * by;
    const int r4 = g_conformant || MUT(1) ? 2 : 1;        /* rounding of the four-sample mean: 1 = RTL:764 (mutant 1: the ISO + 2) */

    /* ---- full-pel search (RTL:1634-1715) ---- */
    int have = 0, best = 0, fy = 0, fx = 0;
    for (int dy = -YR; dy <= YR; ++dy)
        for (int dx = -YR; dx <= YR; ++dx) {
            int masked = (bx == 0 && dx < 0) || (bx == g->max_x16 && dx > 0) ||   /* RTL:1642-1645 */
                         (by == 0 && dy < 0) || (by == g->max_y16 && dy > 0);
            if (masked) continue;
            int col[16];
            for (int c = 0; c < 16; ++c) {
                int s = 0;
                for (int r = 0; r < 16; ++r)
                    s += absdiff(cy[r * 16 + c], pix(ref->Y, W, W, H, y0 + r + dy, x0 + c + dx));
                col[c] = s;
            }
            int v = accumulate_sad13(col, 0);
            if (MUT(3)) { v = 0; for (int c = 0; c < 16; ++c) v += col[c]; }     /* mutant 3: a large SAD never kills a candidate */
            else
            if (v >> 12) continue;                               /* SAD >= 4096: candidate dead (RTL:1669-1670) */
            /* minimum (RTL:1675-1691); among equal minima the largest dy, then the largest dx
               survive (last-assignment-wins loops, RTL:1694-1710) */
            if (!have || (MUT(5) ? v < best : v <= best)) { have = 1; best = v; fy = dy; fx = dx; }   /* mutant 5: the FIRST minimum wins */
        }
    /* no live candidate: f_mvy = f_mvx = 0 (RTL:1695, 1707) */

    /* ---- T = matched block with a 1-px border (RTL:1712-1740) ---- */
    int T[18][18];
    for (int y = -1; y <= 16; ++y)
        for (int x = -1; x <= 16; ++x)
            T[y + 1][x + 1] = pix(ref->Y, W, W, H, y0 + y + fy, x0 + x + fx);

    /* ---- 33x33 half-pel grid (RTL:1746-1752), index -1..31 stored at +1 ---- */
    static const int dummy = 0; (void)dummy;
    uint8_t hg[33][33];
    for (int i = -1; i <= 31; ++i)
        for (int j = -1; j <= 31; ++j) {
            int y = (i + 2) / 2 - 1, yo = i - 2 * y;
            int x = (j + 2) / 2 - 1, xo = j - 2 * x;
            int a = T[y + 1][x + 1];
            int v;
            if (!yo && !xo)      v = a;
            else if (!yo)        v = mean2(a, T[y + 1][x + 2]);
            else if (!xo)        v = mean2(a, T[y + 2][x + 1]);
            else                 v = mean4(a, T[y + 1][x + 2], T[y + 2][x + 1], T[y + 2][x + 2], r4);
            hg[i + 1][j + 1] = (uint8_t)v;
        }

    /* ---- 9 half-pel SADs (RTL:1754-1787) ---- */
    int v10[10];
    for (int hy = -1; hy <= 1; ++hy)
        for (int hx = -1; hx <= 1; ++hx) {
            int masked = ((bx == 0          || fx == -YR) && hx < 0) ||          /* RTL:1757-1760 */
                         ((bx == g->max_x16 || fx ==  YR) && hx > 0) ||
                         ((by == 0          || fy == -YR) && hy < 0) ||
                         ((by == g->max_y16 || fy ==  YR) && hy > 0);
            if (MUT(15))    /* mutant 15: half-pel candidates masked by the block position only, not by the search range */
                masked = (bx == 0 && hx < 0) || (bx == g->max_x16 && hx > 0) || (by == 0 && hy < 0) || (by == g->max_y16 && hy > 0);
            int col[16];
            for (int c = 0; c < 16; ++c) {
                int s = 0;
                for (int r = 0; r < 16; ++r)
                    s += absdiff(cy[r * 16 + c], hg[2 * r + hy + 1][2 * c + hx + 1]);
                col[c] = s;
            }
            v10[(hy + 1) * 3 + (hx + 1)] = accumulate_sad13(col, masked);
        }

    /* ---- "intra cost": accumulates on top of the pixel sum in 16 bits
            (RTL:1600, 1662, 1744, 1774-1777, 1791) ---- */
    uint32_t S = 0;
    for (int k = 0; k < 256; ++k) S += cy[k];
    int m = (int)((S >> 8) & 0xFF);
    if (MUT(6)) S = 0;                                    /* mutant 6: the deviation sum starts from zero, not on top of the pixel sum */
    for (int c = 0; c < 16; ++c) {
        int s = 0;
        for (int r = 0; r < 16; ++r) s += absdiff(cy[r * 16 + c], m);
        S = (S + (uint32_t)s) & 0xFFFFu;
    }
    v10[9] = (S >> 12) == 0 ? (int)(S & 0xFFF) : 0xFFF;

    /* ---- decision (RTL:1794-1815) and final vector (RTL:1826-1829) ---- */
    int idx = find_min_in_10_values(v10);
    int inter = idx != 9;
    int hy = inter ? idx / 3 - 1 : 0;
    int hx = inter ? idx % 3 - 1 : 0;
    int mvy = sext(2 * fy + hy, 5);
    int mvx = sext(2 * fx + hx, 5);
    *o_inter = inter;
    *o_mvx = mvx;
    *o_mvy = mvy;
    if (!inter) return;                     /* prediction = 0x80 (RTL:1894-1903), filled by the caller */

    /* ---- luma prediction: half-pel grid sample (RTL:1847-1897) ---- */
    for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x)
            py[y * 16 + x] = hg[2 * y + hy + 1][2 * x + hx + 1];

    /* ---- chroma prediction (RTL:1854-1888 integer part mv>>>2, RTL:1904-1916 half flag = bit 1) ---- */
    const int cw = W / 2, ch = H / 2;
    /* chroma vector in chroma half samples: RTL floor (mv >>> 1), ISO 7.6.3.7 truncation toward zero (mv / 2) */
    const int cmvy = g_conformant || MUT(2) ? mvy / 2 : mvy >> 1, cmvx = g_conformant || MUT(2) ? mvx / 2 : mvx >> 1;   /* mutant 2: the ISO division */
    int cy_i = cmvy >> 1, cx_i = cmvx >> 1;             /* floor */
    int fyh = cmvy & 1, fxh = cmvx & 1;
    for (int pl = 0; pl < 2; ++pl) {
        const uint8_t *rp = pl ? ref->V : ref->U;
        uint8_t *out = pl ? pv : pu;
        for (int y = 0; y < 8; ++y)
            for (int x = 0; x < 8; ++x) {
                int yy = 8 * by + y + cy_i, xx = 8 * bx + x + cx_i;
                int a = pix(rp, cw, cw, ch, yy, xx);
                int v;
                if (fyh && fxh) v = mean4(a, pix(rp, cw, cw, ch, yy, xx + 1), pix(rp, cw, cw, ch, yy + 1, xx),
                                          pix(rp, cw, cw, ch, yy + 1, xx + 1), r4);
                else if (fxh)   v = mean2(a, pix(rp, cw, cw, ch, yy, xx + 1));
                else if (fyh)   v = mean2(a, pix(rp, cw, cw, ch, yy + 1, xx));
                else            v = a;
                out[y * 8 + x] = (uint8_t)v;
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * stage T: headers (RTL:2590-2716).  Written field by field in ISO/IEC 13818-2 terms; the
 * constants are the RTL's.
 * ---------------------------------------------------------------------------------------- */
static void put_sequence_headers(bitw_t *w, const geom_t *g)
{
    /* sequence_header (RTL:2598-2602) */
    bw_align(w);
    bw_put(w, 0x000001B3, 32);
    bw_put(w, (uint32_t)g->W, 12);       /* horizontal_size_value   */
    bw_put(w, (uint32_t)g->H, 12);       /* vertical_size_value     */
    bw_put(w, 1, 4);                     /* aspect_ratio_information: square samples */
    bw_put(w, 2, 4);                     /* frame_rate_code: 24 fps */
    bw_put(w, 10000, 18);                /* bit_rate_value (x400 bit/s) */
    bw_put(w, 1, 1);                     /* marker */
    bw_put(w, 0, 10);                    /* vbv_buffer_size_value */
    bw_put(w, 0, 3);                     /* constrained_parameters, load_intra, load_non_intra */
    /* sequence_extension (RTL:2602-2604, 2611) */
    bw_put(w, 0x000001B5, 32);
    bw_put(w, 1, 4);                     /* extension id: sequence extension */
    bw_put(w, 0x44, 8);                  /* profile_and_level_indication */
    bw_put(w, 0, 1);                     /* progressive_sequence */
    bw_put(w, 1, 2);                     /* chroma_format 4:2:0 */
    bw_put(w, 0, 4);                     /* size extensions */
    bw_put(w, 0, 12);                    /* bit_rate_extension */
    bw_put(w, 1, 1);                     /* marker */
    bw_put(w, 0, 8);                     /* vbv_buffer_size_extension */
    bw_put(w, 0, 8);                     /* low_delay, frame_rate_extension_n/d */
    /* sequence_display_extension (RTL:2612-2617) */
    bw_put(w, 0x000001B5, 32);
    bw_put(w, 2, 4);                     /* extension id: sequence display extension */
    bw_put(w, 1, 3);                     /* video_format */
    bw_put(w, 1, 1);                     /* colour_description */
    bw_put(w, 5, 8);                     /* colour_primaries */
    bw_put(w, 5, 8);                     /* transfer_characteristics */
    bw_put(w, 5, 8);                     /* matrix_coefficients */
    bw_put(w, (uint32_t)g->W, 14);       /* display_horizontal_size */
    bw_put(w, 1, 1);                     /* marker */
    bw_put(w, (uint32_t)g->H, 14);       /* display_vertical_size */
}

/* GOP header, closed_gop = 1; time code of frame number n at 24 frames/s (RTL:2645-2656, 2685-2698) */
static void put_gop_header(bitw_t *w, size_t n)
{
    uint32_t pic = (uint32_t)(n % 24), sec = (uint32_t)((n / 24) % 60), min = (uint32_t)((n / 1440) % 60);
    size_t hh = n / 86400;
    uint32_t hour = hh > 63 ? 63u : (uint32_t)hh;      /* saturates at 63 (RTL:2694) */
    bw_align(w);
    bw_put(w, 0x000001B8, 32);
    bw_put(w, hour, 6);                  /* drop_frame_flag + hours */
    bw_put(w, min, 6);
    bw_put(w, 1, 1);                     /* marker */
    bw_put(w, sec, 6);
    bw_put(w, pic, 6);
    bw_put(w, 2, 2);                     /* closed_gop = 1, broken_link = 0 */
}

/* picture header + picture coding extension (RTL:2670-2682) */
static void put_picture_header(bitw_t *w, int i_frame)
{
    bw_align(w);
    bw_put(w, 0x00000100, 32);
    bw_put(w, (uint32_t)i_frame, 10);    /* temporal_reference */
    if (i_frame == 0) {
        bw_put(w, 1, 3);                 /* picture_coding_type I */
        bw_put(w, 0, 16);                /* vbv_delay */
        bw_put(w, 0, 3);                 /* extra_bit_picture + stuffing */
    } else {
        bw_put(w, 2, 3);                 /* picture_coding_type P */
        bw_put(w, 0, 16);                /* vbv_delay */
        bw_put(w, 0, 1);                 /* full_pel_forward_vector */
        bw_put(w, 7, 3);                 /* forward_f_code */
        bw_put(w, 0, 7);                 /* extra_bit_picture + stuffing */
    }
    bw_put(w, 0x000001B5, 32);
    bw_put(w, 8, 4);                     /* extension id: picture coding extension */
    bw_put(w, 0x1111, 16);               /* f_code[0][0], [0][1], [1][0], [1][1] = 1 */
    bw_put(w, 2, 2);                     /* intra_dc_precision: 10 bit */
    bw_put(w, 3, 2);                     /* picture_structure: frame */
    bw_put(w, 1, 1);                     /* top_field_first */
    bw_put(w, 1, 1);                     /* frame_pred_frame_dct */
    bw_put(w, 0, 8);                     /* concealment_mv, q_scale_type, intra_vlc_format, alternate_scan,
                                            repeat_first_field, chroma_420_type, progressive_frame,
                                            composite_display_flag */
    bw_put(w, 0, 6);                     /* stuffing to the byte boundary */
}

static void put_slice_header(bitw_t *w, const geom_t *g, int by)   /* RTL:2708-2710 */
{
    bw_align(w);
    bw_put(w, 0x000001, 24);
    bw_put(w, (uint32_t)(by + 1), 8);    /* slice_vertical_position */
    bw_put(w, 1u << g->Q, 5);            /* quantiser_scale_code */
    bw_put(w, 0, 1);                     /* extra_bit_slice */
}

/* ------------------------------------------------------------------------------------------
 * stage T: macroblock layer (RTL:2718-2847)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int prev_mvx, prev_mvy;              /* t_prev_mvx/y  */
    int prev_dc[3];                      /* t_prev_Y_dc, t_prev_U_dc, t_prev_V_dc */
} slice_pred_t;

/* put_AC (RTL:2525-2547): run/level VLC + sign, or the 24-bit escape */
static void put_ac(bitw_t *w, int v, int run)
{
    int absv = v < 0 ? -v : v;
    if (run < (MUT(11) ? 31 : 32) && absv <= M2V_AC_MAX_LEVEL && M2V_AC_CODE[run][absv - 1].len) {   /* mutant 11: run 31 escapes */
        const m2v_vlc *c = &M2V_AC_CODE[run][absv - 1];
        bw_put(w, c->code, c->len);
        bw_put(w, v < 0, 1);
    } else {
        bw_put(w, 1, 6);                                 /* escape */
        bw_put(w, (uint32_t)run & 63u, 6);
        bw_put(w, (uint32_t)v & 0xFFFu, 12);
    }
}

static void put_motion_delta(bitw_t *w, int mv, int prev)        /* RTL:2736-2748 */
{
    int d = mv - prev;
    if (MUT(21)) { if (d > 16) d -= 32; else if (d < -15) d += 32; }      /* mutant 21: the wrap window shifted by one */
    else
    if (d > 15) d -= 32;
    else if (d < -16) d += 32;
    int a = d < 0 ? -d : d;
    bw_put(w, M2V_MOTION_CODE[a].code, M2V_MOTION_CODE[a].len);
    if (d != 0) bw_put(w, d < 0, 1);
}

static void put_macroblock(bitw_t *w, slice_pred_t *sp, int i_frame, int inter, int mvx, int mvy,
                           int cbp, const int16_t zig[6][64])
{
    /* macroblock_address_increment '1' + macroblock_type (RTL:2722-2731) */
    if (!inter && i_frame != 0) bw_put(w, 0x23, 6);      /* intra in a P picture     */
    else if (inter && cbp == 0) bw_put(w, 0x09, 4);      /* MC, not coded            */
    else                        bw_put(w, 0x03, 2);      /* I: intra; P: MC + coded  */

    if (inter) {                                         /* RTL:2734-2770 */
        put_motion_delta(w, mvx, sp->prev_mvx);
        put_motion_delta(w, mvy, sp->prev_mvy);
        bw_put(w, M2V_CBP_CODE[cbp].code, M2V_CBP_CODE[cbp].len);
        sp->prev_mvx = mvx;
        sp->prev_mvy = mvy;
    } else if (!MUT(20)) {                               /* RTL:2771-2774 (mutant 20: an intra macroblock leaves the predictors alone) */
        sp->prev_mvx = 0;
        sp->prev_mvy = 0;
    }

    for (int t = 0; t < 6; ++t) {                        /* PUT_TILE, RTL:2777-2847 */
        int coded = (cbp >> (5 - t)) & 1;
        int comp = t < 4 ? 0 : t - 3;
        int val = zig[t][0];
        int diff_dc = val - sp->prev_dc[comp];
        if (!(MUT(14) && inter))                         /* mutant 14: an inter macroblock leaves the DC predictors alone */
        sp->prev_dc[comp] = inter ? 0 : val;             /* updated for every tile (RTL:2786-2792) */
        int run = 0;
        if (inter) {                                     /* RTL:2795-2806 */
            if (val == 0) run = MUT(12) ? 0 : 1;         /* mutant 12: the first position does not count as a zero */
            else if (coded) {
                if ((val == 1 || val == -1) && !MUT(10)) { bw_put(w, 1, 1); bw_put(w, val < 0, 1); }   /* mutant 10: no '1s' rule */
                else put_ac(w, val, 0);
            }
        } else if (coded) {                              /* RTL:2807-2822 */
            int a = diff_dc < 0 ? -diff_dc : diff_dc;
            int size = 0;
            for (int b = 0; b < 12; ++b) if ((a >> b) & 1) size = b + 1;
            uint32_t bits = (uint32_t)diff_dc & 0xFFFu;
            if (diff_dc < 0) bits = (bits + ((1u << size) - 1u)) & 0xFFFu;
            const m2v_vlc *c = t < 4 ? &M2V_DC_SIZE_LUMA[size] : &M2V_DC_SIZE_CHROMA[size];
            bw_put(w, c->code, c->len);
            bw_put(w, bits, size);
        }
        for (int k = 1; k < 64; ++k) {                   /* RTL:2824-2834 */
            int v = zig[t][k];
            if (v != 0) {
                if (coded) put_ac(w, v, run);
                run = 0;
            } else {
                run = (run + 1) & 63;
            }
        }
        if (coded) bw_put(w, 2, 2);                      /* end_of_block '10' (RTL:2835) */
    }
}

/* ------------------------------------------------------------------------------------------
 * one frame: stages D..T in macroblock raster order
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int8_t *inter, *mvx, *mvy;
    uint8_t *cbp;
    int16_t *coef;
    uint32_t *mb_bits;
} frame_dump_t;

static void encode_frame(const geom_t *g, size_t n, int i_frame, const planes_t *cur, const planes_t *ref,
                         uint8_t *recY, uint8_t *recU, uint8_t *recV, bitw_t *w, const frame_dump_t *fd)
{
    const int W = g->W, cw = W / 2;
    if (i_frame == 0) put_gop_header(w, n);              /* RTL:2645-2657 */
    put_picture_header(w, i_frame);                      /* RTL:2663-2682 */

    for (int by = 0; by < g->mbh; ++by) {
        slice_pred_t sp;
        memset(&sp, 0, sizeof sp);                       /* RTL:2713-2715 */
        put_slice_header(w, g, by);
        for (int bx = 0; bx < g->mbw; ++bx) {
            uint8_t cy[256], cu[64], cv[64], py[256], pu[64], pv[64];
            for (int y = 0; y < 16; ++y) memcpy(cy + 16 * y, cur->Y + (16 * by + y) * W + 16 * bx, 16);
            for (int y = 0; y < 8; ++y) {
                memcpy(cu + 8 * y, cur->U + (8 * by + y) * cw + 8 * bx, 8);
                memcpy(cv + 8 * y, cur->V + (8 * by + y) * cw + 8 * bx, 8);
            }
            int inter = 0, mvx = 0, mvy = 0;
            if (i_frame != 0)                            /* I frame: f_inter = 0, mv = 0 (RTL:1820-1825) */
                motion_stage(g, cy, ref, bx, by, &inter, &mvx, &mvy, py, pu, pv);
            if (!inter) {
                memset(py, 0x80, sizeof py);
                memset(pu, 0x80, sizeof pu);
                memset(pv, 0x80, sizeof pv);
                mvx = mvy = 0;                           /* never transmitted (RTL:2734, 2771-2774) */
            }

            /* six tiles: Y00, Y01, Y10, Y11, U, V (RTL:1980-2014) */
            int16_t zig[6][64];
            int cbp = 0;
            for (int t = 0; t < 6; ++t) {
                int16_t x[64], q[64], d[64], r[64];
                int32_t c[64];
                const uint8_t *cp, *pp;
                int cs, ox = 0, oy = 0;
                if (t < 4) { cp = cy; pp = py; cs = 16; ox = (t & 1) * 8; oy = (t >> 1) * 8; }
                else if (t == 4) { cp = cu; pp = pu; cs = 8; }
                else { cp = cv; pp = pv; cs = 8; }
                for (int y = 0; y < 8; ++y)
                    for (int xx = 0; xx < 8; ++xx)
                        x[y * 8 + xx] = (int16_t)(cp[(oy + y) * cs + ox + xx] - pp[(oy + y) * cs + ox + xx]);
                m2v_oracle_fdct(x, c);
                m2v_oracle_quant(c, inter, g->Q, q);
                int nz = !inter;                         /* stage S (RTL:2461-2467) */
                for (int i = 0; i < 8; ++i)
                    for (int j = 0; j < 8; ++j) {
                        zig[t][M2V_ZIGZAG_POS[i][j]] = q[i * 8 + j];
                        nz |= q[i * 8 + j] != 0;
                    }
                cbp = (cbp << 1) | nz;
                /* reconstruction loop (stages H..P) */
                if (g_conformant && inter && !nz) {
                    memset(r, 0, sizeof r);                 /* a block that is not coded is not reconstructed (7.6.8): no mismatch toggle */
                } else {
                    m2v_oracle_dequant(q, inter, g->Q, d);
                    m2v_oracle_idct(d, r);
                }
                for (int y = 0; y < 8; ++y)
                    for (int xx = 0; xx < 8; ++xx) {
                        int v = pp[(oy + y) * cs + ox + xx] + r[y * 8 + xx];   /* add_clip_0_255, RTL:786-795 */
                        uint8_t rv = (uint8_t)(v > 255 ? 255 : v < 0 ? 0 : v);
                        if (t < 4)       recY[(16 * by + oy + y) * W + 16 * bx + ox + xx] = rv;
                        else if (t == 4) recU[(8 * by + y) * cw + 8 * bx + xx] = rv;
                        else             recV[(8 * by + y) * cw + 8 * bx + xx] = rv;
                    }
            }

            uint64_t b0 = w->nbits;
            put_macroblock(w, &sp, i_frame, inter, mvx, mvy, cbp, (const int16_t (*)[64])zig);

            if (fd) {
                size_t mb = (size_t)by * g->mbw + bx;
                if (fd->inter) fd->inter[mb] = (int8_t)inter;
                if (fd->mvx) fd->mvx[mb] = (int8_t)mvx;
                if (fd->mvy) fd->mvy[mb] = (int8_t)mvy;
                if (fd->cbp) fd->cbp[mb] = (uint8_t)cbp;
                if (fd->coef) memcpy(fd->coef + mb * 384, zig, sizeof zig);
                if (fd->mb_bits) fd->mb_bits[mb] = (uint32_t)(w->nbits - b0);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * whole sequence: stage A sequence control (RTL:1027-1095) + everything above
 * ---------------------------------------------------------------------------------------- */
size_t m2v_oracle_encode(const m2v_oracle_params *p,
                         unsigned xsize16, unsigned ysize16, unsigned pframes_count,
                         const uint8_t *frames444, size_t nbeats,
                         uint8_t *out, size_t cap, const m2v_oracle_dump *dump)
{
    geom_t g;
    if (make_geom(p, xsize16, ysize16, &g)) return (size_t)-1;
    if (nbeats == 0) return 0;                            /* stop while idle does nothing (RTL:1090) */
    pframes_count &= 0xFF;                                /* 8-bit port */

    const size_t npix = (size_t)g.W * g.H, cpix = npix / 4, bpf = npix / 4;
    const size_t nframes = (nbeats + bpf - 1) / bpf;
    const size_t mbs = (size_t)g.mbw * g.mbh;

    uint8_t *tmp = (uint8_t *)malloc(npix * 3);           /* one 4:4:4 frame with black fill */
    uint8_t *c420 = (uint8_t *)malloc(npix + 2 * cpix);
    uint8_t *ra = (uint8_t *)malloc(npix + 2 * cpix);
    uint8_t *rb = (uint8_t *)malloc(npix + 2 * cpix);
    if (!tmp || !c420 || !ra || !rb) { free(tmp); free(c420); free(ra); free(rb); return (size_t)-1; }
    memset(ra, 0, npix + 2 * cpix);
    memset(rb, 0, npix + 2 * cpix);

    bitw_t w = { out, out ? cap : 0, 0 };
    put_sequence_headers(&w, &g);                         /* on sequence_start (RTL:2591-2618) */

    uint8_t *refbuf = ra, *recbuf = rb;
    for (size_t n = 0; n < nframes; ++n) {
        int i_frame = (int)(n % (pframes_count + 1u));    /* a_i_frame (RTL:1078) */
        const uint8_t *src = frames444 + n * npix * 3;
        size_t first_beat = n * bpf;
        const uint8_t *sy, *su, *sv;
        if (first_beat + bpf <= nbeats) {
            sy = src; su = src + npix; sv = src + 2 * npix;
        } else {
            /* i_sequence_stop inside the frame: the rest is Y=0, U=V=0x80 (RTL:1036-1056) */
            size_t valid = (nbeats - first_beat) * 4;     /* pixels in raster order */
            memset(tmp, 0x00, npix);
            memset(tmp + npix, 0x80, 2 * npix);
            memcpy(tmp, src, valid);
            memcpy(tmp + npix, src + npix, valid);
            memcpy(tmp + 2 * npix, src + 2 * npix, valid);
            sy = tmp; su = tmp + npix; sv = tmp + 2 * npix;
        }
        memcpy(c420, sy, npix);
        m2v_oracle_subsample(su, g.W, g.H, c420 + npix);
        m2v_oracle_subsample(sv, g.W, g.H, c420 + npix + cpix);

        planes_t cur = { c420, c420 + npix, c420 + npix + cpix };
        planes_t ref = { refbuf, refbuf + npix, refbuf + npix + cpix };
        frame_dump_t fd, *pfd = NULL;
        if (dump) {
            fd.inter = dump->mb_inter ? dump->mb_inter + n * mbs : NULL;
            fd.mvx = dump->mb_mvx ? dump->mb_mvx + n * mbs : NULL;
            fd.mvy = dump->mb_mvy ? dump->mb_mvy + n * mbs : NULL;
            fd.cbp = dump->mb_cbp ? dump->mb_cbp + n * mbs : NULL;
            fd.coef = dump->coef ? dump->coef + n * mbs * 384 : NULL;
            fd.mb_bits = dump->mb_bits ? dump->mb_bits + n * mbs : NULL;
            pfd = &fd;
            if (dump->yuv420) memcpy(dump->yuv420 + n * (npix + 2 * cpix), c420, npix + 2 * cpix);
        }
        encode_frame(&g, n, i_frame, &cur, &ref, recbuf, recbuf + npix, recbuf + npix + cpix, &w, pfd);
        if (dump && dump->recon) memcpy(dump->recon + n * (npix + 2 * cpix), recbuf, npix + 2 * cpix);
        uint8_t *t = refbuf; refbuf = recbuf; recbuf = t;  /* ref(f+1) = recon(f) */
    }

    bw_align(&w);                                          /* sequence_end_code (RTL:2621-2628) */
    bw_put(&w, 0x000001B7, 32);
    size_t total = bw_finish(&w);

    free(tmp); free(c420); free(ra); free(rb);
    return total;
}

/* ------------------------------------------------------------------------------------------
 * table accessors
 * ---------------------------------------------------------------------------------------- */
int m2v_oracle_tab_dct(int i, int k)     { return M2V_DCT_BASIS[i][k]; }
int m2v_oracle_tab_intra_w(int i, int j) { return M2V_INTRA_W[i][j]; }
int m2v_oracle_tab_zigzag(int i, int j)  { return M2V_ZIGZAG_POS[i][j]; }
void m2v_oracle_tab_motion(int idx, int *code, int *len) { *code = M2V_MOTION_CODE[idx].code; *len = M2V_MOTION_CODE[idx].len; }
void m2v_oracle_tab_cbp(int idx, int *code, int *len)    { *code = M2V_CBP_CODE[idx].code; *len = M2V_CBP_CODE[idx].len; }
void m2v_oracle_tab_dc(int chroma, int idx, int *code, int *len)
{
    const m2v_vlc *c = chroma ? &M2V_DC_SIZE_CHROMA[idx] : &M2V_DC_SIZE_LUMA[idx];
    *code = c->code; *len = c->len;
}
void m2v_oracle_tab_ac(int run, int abslevel, int *code, int *len)
{
    if (run < 0 || run > 31 || abslevel < 1 || abslevel > M2V_AC_MAX_LEVEL) { *code = 0; *len = 0; return; }
    *code = M2V_AC_CODE[run][abslevel - 1].code;
    *len = M2V_AC_CODE[run][abslevel - 1].len;
}
