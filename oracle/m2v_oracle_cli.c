/*
 * m2v_oracle_cli — file-to-file driver of the CPU oracle; the counterpart of
 * SIM/tb_mpeg2encoder.v (planar yuv444p in, .m2v out).  TEST INFRASTRUCTURE ONLY.
 *
 *   m2v_oracle_cli in.yuv WIDTH HEIGHT out.m2v [pframes=23] [XL=7] [YL=7] [VL=3] [Q=2] [stop_beats]
 *
 * Like the testbench (TB:220, the !$feof loop) only complete frames of the file are pushed,
 * then i_sequence_stop is pulsed; `stop_beats` (optional) stops after that many beats instead.
 */
#define _POSIX_C_SOURCE 200809L
#include "m2v_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s in.yuv W H out.m2v [pframes] [XL] [YL] [VL] [Q] [stop_beats]\n", argv[0]);
        return 2;
    }
    int W = atoi(argv[2]), H = atoi(argv[3]);
    unsigned pframes = argc > 5 ? (unsigned)atoi(argv[5]) : 23;
    m2v_oracle_params p = { argc > 6 ? atoi(argv[6]) : 7, argc > 7 ? atoi(argv[7]) : 7,
                            argc > 8 ? atoi(argv[8]) : 3, argc > 9 ? atoi(argv[9]) : 2 };
    if (W < 64 || H < 64 || W % 16 || H % 16 || W > (16 << p.XL) || H > (16 << p.YL)) {   /* TB:189-201 */
        fprintf(stderr, "*** size %dx%d invalid for XL=%d YL=%d\n", W, H, p.XL, p.YL);
        return 2;
    }
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { fprintf(stderr, "*** couldn't open input file\n"); return 1; }
    fseek(fi, 0, SEEK_END);
    long fsz = ftell(fi);
    fseek(fi, 0, SEEK_SET);
    size_t fbytes = (size_t)W * H * 3;
    size_t nframes = (size_t)fsz / fbytes;
    size_t nbeats = nframes * ((size_t)W * H / 4);
    if (argc > 10) {
        size_t sb = (size_t)atoll(argv[10]);
        if (sb < nbeats) nbeats = sb;
    }
    if (nbeats == 0) { fprintf(stderr, "*** no complete frame in input\n"); return 1; }
    uint8_t *in = (uint8_t *)malloc(nframes * fbytes);
    if (!in || fread(in, 1, nframes * fbytes, fi) != nframes * fbytes) { fprintf(stderr, "*** read error\n"); return 1; }
    fclose(fi);
    size_t cap = nframes * fbytes + 4096;
    uint8_t *out = (uint8_t *)malloc(cap);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    size_t n = m2v_oracle_encode(&p, (unsigned)W / 16, (unsigned)H / 16, pframes, in, nbeats, out, cap, NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (n == (size_t)-1 || n > cap) { fprintf(stderr, "*** encode failed\n"); return 1; }
    FILE *fo = fopen(argv[4], "wb");
    if (!fo) { fprintf(stderr, "*** couldn't open output file\n"); return 1; }
    fwrite(out, 1, n, fo);
    fclose(fo);
    double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    fprintf(stderr, "%zu beats (%zu frames) -> %zu bytes in %.3f s (%.3f MPixels/s)\n", nbeats,
            m2v_oracle_frame_count(&p, (unsigned)W / 16, (unsigned)H / 16, nbeats), n, s,
            (double)nbeats * 4 / s * 1e-6);
    free(in); free(out);
    return 0;
}
