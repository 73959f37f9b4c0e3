/*
 * m2v_oracle.h — CPU restatement of RTL/mpeg2encoder.v (the parity oracle).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (fpga-mpeg2-encoder_amd/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker.
 *
 * PARITY UNPINNED: the reference ships no golden bitstreams (its only test is
 * "open the .m2v in VLC", README.md:352), its input clips (SIM/data.zip) are not
 * in the checkout, and no Verilog simulator exists in the build image, so this
 * restatement cannot be checked against an execution of the RTL here.  It is
 * pinned only by (i) a line-by-line reading of RTL/mpeg2encoder.v, cited at each
 * function, (ii) the RTL's own constant tables, compared entry by entry when
 * /root/reference is mounted (tests/test_tables_vs_rtl.py), (iii) hand-derived
 * known-answer bitstreams (tests/golden/), and (iv) an independent MPEG-2
 * decoder written from ISO/IEC 13818-2 that must reproduce the encoder's own
 * reconstruction (tests/m2v_decode.py).  tools/run_rtl_oracle.sh runs the real
 * RTL under iverilog and compares, wherever iverilog exists.
 */
#ifndef M2V_ORACLE_H
#define M2V_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* module parameters, RTL/mpeg2encoder.v:11-14 */
typedef struct m2v_oracle_params {
    int XL;            /* 4..7 : max width  = 16 << XL */
    int YL;            /* 4..7 : max height = 16 << YL */
    int VECTOR_LEVEL;  /* 1..3 : luma search range +-2*VECTOR_LEVEL px */
    int Q_LEVEL;       /* 1..4 */
} m2v_oracle_params;

/*
 * Optional per-stage dumps (any pointer may be NULL).  `frames` below is the
 * number of frames actually encoded (see m2v_oracle_frame_count).
 */
typedef struct m2v_oracle_dump {
    int8_t   *mb_inter;  /* [frames][mbs]          1 = inter (stage F f_inter)            */
    int8_t   *mb_mvx;    /* [frames][mbs]          half-pel vector as transmitted, 0 if intra */
    int8_t   *mb_mvy;
    uint8_t  *mb_cbp;    /* [frames][mbs]          coded flags, bit5 = Y00 ... bit0 = V    */
    int16_t  *coef;      /* [frames][mbs][6][64]   quantised levels, zig-zag order         */
    uint8_t  *recon;     /* [frames][W*H*3/2]      reconstructed Y, U, V planes            */
    uint8_t  *yuv420;    /* [frames][W*H*3/2]      subsampled input Y, U, V planes         */
    uint32_t *mb_bits;   /* [frames][mbs]          bits of the macroblock layer of each MB */
} m2v_oracle_dump;

/* NOT a reference mode: switches the reconstruction loop to ISO/IEC 13818-2 where the RTL deviates from it (+2
 * rounding of the four-sample average, chroma vector = mv / 2 toward zero, truncating inverse quantiser with
 * [-2048, 2047] saturation and mismatch control, full-width IDCT row pass, [-256, 255] saturation).  It exists
 * to check the GPU path's option "conformant" (SURVEY.md 8(f4)).  Process-global; default off. */
void m2v_oracle_set_conformant(int on);
int  m2v_oracle_get_conformant(void);

/* Size clamp of RTL/mpeg2encoder.v:985-991.  Returns 0, or -1 for bad params. */
int m2v_oracle_geometry(const m2v_oracle_params *p, unsigned xsize16, unsigned ysize16,
                        int *width, int *height);

/* frames the module encodes when `nbeats` 4-pixel beats arrive before the stop pulse */
size_t m2v_oracle_frame_count(const m2v_oracle_params *p, unsigned xsize16, unsigned ysize16,
                              size_t nbeats);

/*
 * Encode one video sequence exactly as the RTL would.
 *   frames444 : planar Y plane, U plane, V plane per frame, each W*H bytes, W/H being
 *               the CLAMPED geometry (the layout of SIM/tb_mpeg2encoder.v:210-218).
 *   nbeats    : number of 4-pixel beats pushed before i_sequence_stop.  A partial last
 *               frame is completed with Y=0,U=V=0x80 (RTL:1036-1056); pixels of that
 *               frame beyond `nbeats` are never read.
 *   out/cap   : output buffer.  The return value is the stream length in bytes (always
 *               a multiple of 32); if it exceeds `cap` the buffer holds the first `cap`
 *               bytes.  0 is returned when nbeats == 0 (the sequence never starts).
 *               (size_t)-1 on invalid parameters.
 */
size_t m2v_oracle_encode(const m2v_oracle_params *p,
                         unsigned xsize16, unsigned ysize16, unsigned pframes_count,
                         const uint8_t *frames444, size_t nbeats,
                         uint8_t *out, size_t cap,
                         const m2v_oracle_dump *dump);

/* ---- single-stage entry points, used by the unit tests of the HIP kernels ---- */

/* 4:4:4 -> 4:2:0, RTL:1086-1089 + 1167-1170 */
void m2v_oracle_subsample(const uint8_t *plane444, int W, int H, uint8_t *plane420);

/* forward DCT of an 8x8 residual tile (s9) -> s17, RTL:2029-2062 */
void m2v_oracle_fdct(const int16_t x[64], int32_t c[64]);

/* quantise / dequantise one tile (raster order), RTL:2065-2077 / 2129-2150 */
void m2v_oracle_quant(const int32_t c[64], int inter, int q_level, int16_t q[64]);
void m2v_oracle_dequant(const int16_t q[64], int inter, int q_level, int16_t d[64]);

/* Chen-Wang inverse DCT with the RTL's 18-bit row store and +-255 clip, RTL:844-972 */
void m2v_oracle_idct(const int16_t d[64], int16_t r[64]);

/* table accessors for tests/test_tables_vs_rtl.py */
int  m2v_oracle_tab_dct(int i, int k);
int  m2v_oracle_tab_intra_w(int i, int j);
int  m2v_oracle_tab_zigzag(int i, int j);
void m2v_oracle_tab_motion(int idx, int *code, int *len);
void m2v_oracle_tab_cbp(int idx, int *code, int *len);
void m2v_oracle_tab_dc(int chroma, int idx, int *code, int *len);
/* run/level code without the sign bit; *len == 0 -> escape */
void m2v_oracle_tab_ac(int run, int abslevel, int *code, int *len);

#ifdef __cplusplus
}
#endif
#endif
