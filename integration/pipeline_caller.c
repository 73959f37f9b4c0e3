/* integration/pipeline_caller.c - several sequences through TWO encoder handles taking turns, from plain C.
 *
 * The submission form bench.py's timed loop uses ("two sequences in flight"): m2v_encode_resident_begin queues a
 * whole sequence - every macroblock launch, the scans, the stream assembly - on the handle's own streams and returns;
 * m2v_encode_resident_end waits for it and hands back the byte count.  With two handles the stream assembly of sequence k runs
 * beside the first macroblock kernels of sequence k + 1.  Here: the same clip N times (each into a buffer of its own), every stream
 * compared with the first one and written once.
 *
 *   pipeline_caller in.yuv444p W H pframes sequences out.m2v [XL YL VECTOR_LEVEL Q_LEVEL]
 *
 * gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include pipeline_caller.c -lm2v_mi355x -lamdhip64
 * (HIP only for hipMalloc / hipMemcpy of the caller's own buffers: the encoder's ABI carries plain pointers).
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "m2v_mi355x.h"

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s in.yuv444p W H pframes sequences out.m2v [XL YL VECTOR_LEVEL Q_LEVEL]\n", argv[0]); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]), pf = atoi(argv[4]), nseq = atoi(argv[5]);
    const int XL = argc > 7 ? atoi(argv[7]) : 7, YL = argc > 8 ? atoi(argv[8]) : 7, VL = argc > 9 ? atoi(argv[9]) : 3, Q = argc > 10 ? atoi(argv[10]) : 2;
    if (nseq < 1 || nseq > 64) { fprintf(stderr, "1..64 sequences\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t fb = (size_t)W * H * 3, nframes = (size_t)ftell(f) / fb;      /* complete frames only (TB:220) */
    fseek(f, 0, SEEK_SET);
    unsigned char *host = (unsigned char *)malloc(nframes * fb);
    if (!host || fread(host, fb, nframes, f) != nframes) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);

    enum { HANDLES = 2 };
    const size_t cap = nframes * fb / 2 + 65536;           /* the library's worst case: 1.5 bytes per pixel and the headers */
    void *d_frames = NULL, *d_out[HANDLES] = {NULL, NULL};
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&d_frames, nframes * fb) != hipSuccess ||
        hipMemcpy(d_frames, host, nframes * fb, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "no GPU memory\n"); return 1; }
    m2v_enc *enc[HANDLES];
    for (int h = 0; h < HANDLES; ++h) {
        int err = 0;
        enc[h] = m2v_create(XL, YL, VL, Q, 0, &err);
        if (!enc[h] || hipMalloc(&d_out[h], cap) != hipSuccess) { fprintf(stderr, "m2v_create / hipMalloc failed (%d)\n", err); return 1; }
        m2v_set_option(enc[h], "batch_frames", (long long)nframes);       /* the whole sequence as one chunk */
        m2v_set_option(enc[h], "split_streams", 1);                       /* the sequences are what overlaps: one stream each */
    }
    unsigned char *first = NULL, *cur = (unsigned char *)malloc(cap);
    size_t first_bytes = 0;
    int busy[HANDLES] = {0, 0}, rc = 0, done = 0;
    /* sequence i goes to handle i % 2 as soon as that handle's previous sequence has been collected */
    for (int i = 0; i < nseq + HANDLES && rc == 0; ++i) {
        const int h = i % HANDLES;
        if (busy[h]) {
            size_t bytes = 0;
            rc = m2v_encode_resident_end(enc[h], &bytes);
            if (rc < 0) { fprintf(stderr, "m2v_encode_resident_end: %s\n", m2v_last_error(enc[h])); break; }
            busy[h] = 0;
            if (hipMemcpy(cur, d_out[h], bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = -1; break; }
            if (!first) { first = (unsigned char *)malloc(bytes); memcpy(first, cur, bytes); first_bytes = bytes; }
            else if (bytes != first_bytes || memcmp(first, cur, bytes) != 0) { fprintf(stderr, "sequence %d differs from the first\n", done); rc = -1; break; }
            ++done;
        }
        if (i < nseq) {
            rc = m2v_encode_resident_begin(enc[h], (unsigned)(W / 16), (unsigned)(H / 16), (unsigned)pf, d_frames, nframes, d_out[h], cap, NULL);
            if (rc < 0) { fprintf(stderr, "m2v_encode_resident_begin: %s\n", m2v_last_error(enc[h])); break; }
            busy[h] = 1;
        }
    }
    if (rc == 0 && done == nseq) {
        FILE *o = fopen(argv[6], "wb");
        if (!o || fwrite(first, 1, first_bytes, o) != first_bytes) { perror(argv[6]); rc = -1; }
        if (o) fclose(o);
        printf("%d sequences of %zu frames %dx%d through %d handles: %zu bytes each, all identical\n", done, nframes, W, H, HANDLES, first_bytes);
    } else if (rc == 0) rc = -1;
    for (int h = 0; h < HANDLES; ++h) { m2v_destroy(enc[h]); (void)hipFree(d_out[h]); }
    (void)hipFree(d_frames);
    free(host); free(cur); free(first);
    return rc < 0 ? 1 : 0;
}
