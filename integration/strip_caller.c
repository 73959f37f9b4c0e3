/* integration/strip_caller.c - config c5 from plain C: ONE sequence cut into macroblock-row strips, one rank per strip.
 *
 * What a multi-GPU caller of include/m2v_mi355x.h writes, in the form that runs on a 1-GPU box: the
 * ranks are threads of this process, all on GPU 0, talking through an in-process communicator (m2v_comm_init_local); with one
 * process per GPU the only difference is m2v_comm_init_rccl(id, rank, nranks, device) and the device ordinal.  Every rank makes ONE
 * call, m2v_strip_encode: the GOP steps, the exchange of the +-2*VECTOR_LEVEL luma / +-VECTOR_LEVEL chroma boundary rows with the
 * two neighbours, the size all-gather, the strips to rank 0 and the final assembly all happen inside it.
 *
 *   strip_caller in.yuv444p W H pframes nranks out.m2v [XL YL VECTOR_LEVEL Q_LEVEL]
 *
 * gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include strip_caller.c -lm2v_mi355x -lamdhip64 -lpthread
 * (HIP only for hipMalloc / hipMemcpy of the caller's own buffers: the encoder's ABI carries plain pointers).
 */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "m2v_mi355x.h"

typedef struct {
    int rank, nranks, rc;
    m2v_enc *enc;
    m2v_comm *comm;
    unsigned xs16, ys16, pframes;
    const void *d_frames;
    size_t nframes;
    void *d_out;
    size_t cap, bytes;
} rank_t;

static void *rank_main(void *p)
{
    rank_t *r = (rank_t *)p;
    r->rc = m2v_strip_encode(r->enc, r->comm, r->rank, r->nranks, /*dst_rank=*/0, r->xs16, r->ys16, r->pframes, r->d_frames, r->nframes,
                             r->rank == 0 ? r->d_out : NULL, r->rank == 0 ? r->cap : 0, &r->bytes, NULL);
    if (r->rc < 0) fprintf(stderr, "rank %d: m2v_strip_encode failed (%d): %s\n", r->rank, r->rc, m2v_last_error(r->enc));
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s in.yuv444p W H pframes nranks out.m2v [XL YL VECTOR_LEVEL Q_LEVEL]\n", argv[0]); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]), pf = atoi(argv[4]), nranks = atoi(argv[5]);
    const int XL = argc > 7 ? atoi(argv[7]) : 7, YL = argc > 8 ? atoi(argv[8]) : 7, VL = argc > 9 ? atoi(argv[9]) : 3, Q = argc > 10 ? atoi(argv[10]) : 2;
    if (nranks < 1 || nranks > 16) { fprintf(stderr, "1..16 ranks\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t fb = (size_t)W * H * 3, nframes = (size_t)ftell(f) / fb;      /* complete frames only (TB:220) */
    fseek(f, 0, SEEK_SET);
    unsigned char *host = (unsigned char *)malloc(nframes * fb);
    if (!host || fread(host, fb, nframes, f) != nframes) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);

    void *d_frames = NULL, *d_out = NULL;
    const size_t cap = nframes * ((size_t)(W / 16) * (H / 16) * 1216 + (size_t)(H / 16) * 8 + 64) + 256;   /* worst case */
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&d_frames, nframes * fb) != hipSuccess || hipMalloc(&d_out, cap) != hipSuccess ||
        hipMemcpy(d_frames, host, nframes * fb, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "no GPU memory\n"); return 1; }

    int err = 0;
    m2v_comm *comm = nranks > 1 ? m2v_comm_init_local(nranks, &err) : NULL;
    if (nranks > 1 && !comm) { fprintf(stderr, "m2v_comm_init_local: %s\n", m2v_comm_last_error()); return 1; }
    rank_t ranks[16];
    pthread_t th[16];
    for (int r = 0; r < nranks; ++r) {
        memset(&ranks[r], 0, sizeof ranks[r]);
        ranks[r].enc = m2v_create(XL, YL, VL, Q, /*device=*/0, &err);            /* one encoder instance per rank */
        if (!ranks[r].enc) { fprintf(stderr, "m2v_create: %s\n", m2v_last_error(NULL)); return 1; }
        ranks[r].rank = r; ranks[r].nranks = nranks; ranks[r].comm = comm;
        ranks[r].xs16 = (unsigned)(W / 16); ranks[r].ys16 = (unsigned)(H / 16); ranks[r].pframes = (unsigned)pf;
        ranks[r].d_frames = d_frames; ranks[r].nframes = nframes; ranks[r].d_out = d_out; ranks[r].cap = cap;
    }
    for (int r = 0; r < nranks; ++r) pthread_create(&th[r], NULL, rank_main, &ranks[r]);
    int bad = 0;
    for (int r = 0; r < nranks; ++r) { pthread_join(th[r], NULL); bad |= ranks[r].rc < 0; }
    if (!bad) {
        unsigned char *out = (unsigned char *)malloc(ranks[0].bytes);
        if (hipMemcpy(out, d_out, ranks[0].bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        FILE *o = fopen(argv[6], "wb");
        if (!o || fwrite(out, 1, ranks[0].bytes, o) != ranks[0].bytes) { perror(argv[6]); return 1; }
        fclose(o);
        printf("%zu frames %dx%d, %d strips -> %zu bytes\n", nframes, W, H, nranks, ranks[0].bytes);
        free(out);
    }
    for (int r = 0; r < nranks; ++r) m2v_destroy(ranks[r].enc);
    if (comm) m2v_comm_destroy(comm);
    (void)hipFree(d_frames); (void)hipFree(d_out);
    free(host);
    return bad;
}
