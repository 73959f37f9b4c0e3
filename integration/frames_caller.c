/*
 * frames_caller.c - the testbench's file loop (SIM/tb_mpeg2encoder.v:206-266: read a frame, drive it in, write what comes out) from plain
 * C, one thread, with both port groups in ONE call per GOP:
 *
 *     per GOP:   fread into a page-locked buffer -> m2v_push_frames_pull(e, xsize16, ysize16, pframes, frames, n, out, cap, &last) -> fwrite
 *     at EOF:    m2v_sequence_stop(e), then m2v_pull until the o_last word
 *
 * The stream bytes of the chunks that are complete are copied into `out` while the call's frames cross PCIe; frames in page-locked memory
 * (hipHostMalloc here; a capture card's DMA buffer would do) are uploaded from where they lie.  Two frame buffers alternate so that the next
 * GOP can be read from the file while nothing else is going on (the call returns when its frames have been read).
 *
 *     cc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include integration/frames_caller.c -Lfpga-mpeg2-encoder_amd -lm2v_mi355x \
 *        -L/opt/rocm/lib -lamdhip64 -o frames_caller
 *     frames_caller in.yuv WIDTH HEIGHT out.m2v [pframes [frames_per_call [pageable]]]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "m2v_mi355x.h"

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s in.yuv W H out.m2v [pframes [frames_per_call [pageable]]]\n", argv[0]);
        return 2;
    }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    const unsigned pframes = argc > 5 ? (unsigned)atoi(argv[5]) : 8u;
    const size_t per_call = argc > 6 && atoi(argv[6]) > 0 ? (size_t)atoi(argv[6]) : (size_t)pframes + 1;      /* one GOP */
    const int pageable = argc > 7 && atoi(argv[7]);
    if (W % 16 || H % 16 || W < 64 || H < 64) { fprintf(stderr, "frames_caller: sizes must be multiples of 16, >= 64\n"); return 2; }
    FILE *fin = fopen(argv[1], "rb"), *fout = fopen(argv[4], "wb");
    if (!fin || !fout) { perror("frames_caller"); return 2; }
    const size_t frame_bytes = (size_t)3 * W * H, cap = per_call * frame_bytes / 2 + 4096;
    unsigned char *buf[2] = {NULL, NULL}, *out = (unsigned char *)malloc(cap);
    for (int i = 0; i < 2; ++i) {
        if (pageable) buf[i] = (unsigned char *)malloc(per_call * frame_bytes);
        else if (hipHostMalloc((void **)&buf[i], per_call * frame_bytes, hipHostMallocDefault) != hipSuccess) buf[i] = NULL;
    }
    if (!buf[0] || !buf[1] || !out) { fprintf(stderr, "frames_caller: no memory\n"); return 2; }

    int err = 0, last = 0;
    m2v_enc *e = m2v_create(7, 7, 3, 2, 0, &err);
    if (!e) { fprintf(stderr, "frames_caller: m2v_create failed (%d): %s\n", err, m2v_last_error(NULL)); return 1; }
    m2v_set_option(e, "batch_frames", (long long)pframes + 1);          /* a chunk per GOP: the shortest tail behind the last upload */

    long calls = 0, frames = 0;
    long long bytes = 0;
    for (int k = 0;; k ^= 1) {
        const size_t got = fread(buf[k], frame_bytes, per_call, fin);  /* complete frames only (TB:220) */
        if (!got) break;
        const long long n = m2v_push_frames_pull(e, (unsigned)(W / 16), (unsigned)(H / 16), pframes, buf[k], got, out, cap, &last);
        if (n < 0) { fprintf(stderr, "frames_caller: m2v_push_frames_pull: %s\n", m2v_last_error(e)); return 1; }
        fwrite(out, 1, (size_t)n, fout);
        bytes += n; frames += (long)got; ++calls;
    }
    if (m2v_sequence_stop(e) < 0) { fprintf(stderr, "frames_caller: m2v_sequence_stop: %s\n", m2v_last_error(e)); return 1; }
    while (m2v_busy(e)) {
        const long long n = m2v_pull(e, out, cap, &last);
        if (n < 0) { fprintf(stderr, "frames_caller: m2v_pull: %s\n", m2v_last_error(e)); return 1; }
        fwrite(out, 1, (size_t)n, fout);
        bytes += n;
    }
    printf("frames_caller: %ld frames in %ld calls, %lld bytes, last=%d\n", frames, calls, bytes, last);
    m2v_destroy(e);
    for (int i = 0; i < 2; ++i) { if (pageable) free(buf[i]); else (void)hipHostFree(buf[i]); }
    free(out);
    fclose(fin);
    fclose(fout);
    return last ? 0 : 1;
}
