/* integration/strip_inflight_caller.c - config c5 from plain C with SEVERAL sequences in flight: every rank keeps K strip sequences
 * under way from its ONE thread (m2v_strip_encode_begin / m2v_strip_encode_end on K handles taking turns), the halo rows travel as peer
 * stores (m2v_comm_init_peer: a landing block per handle) over ONE shared base communicator, and the output rank rotates
 * (sequence i is assembled on rank i mod nranks).  INTEGRATION.md section 5; what `bench.py --mode strips --transport peer --rotate-dst`
 * runs.  In the form that runs on a 1-GPU box: the ranks are threads of this process on GPU 0 and the base communicator is the
 * in-process one (m2v_comm_init_local); with one process per GPU the base is m2v_comm_init_rccl(id, rank, nranks, device).
 *
 *   strip_inflight_caller in.yuv444p W H pframes nranks nsequences K out_prefix     -> out_prefix.<i>.m2v for every sequence i
 *
 * Every sequence encodes the same clip, so every out file must hold the same bytes - the oracle's.
 * gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include strip_inflight_caller.c -lm2v_mi355x -lamdhip64 -lpthread
 */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "m2v_mi355x.h"

#define MAXK 4

typedef struct {
    int rank, nranks, nseq, K, rc;
    m2v_comm *base;
    unsigned xs16, ys16, pframes;
    int XL, YL, VL, Q;
    const void *d_frames;
    size_t nframes, cap;
    const char *prefix;
} rank_t;

static int write_stream(const char *prefix, int seq, const void *d_out, size_t bytes)
{
    char name[1024];
    unsigned char *out = (unsigned char *)malloc(bytes);
    snprintf(name, sizeof name, "%s.%d.m2v", prefix, seq);
    if (!out || hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    FILE *o = fopen(name, "wb");
    const int ok = o && fwrite(out, 1, bytes, o) == bytes;
    if (o) fclose(o);
    free(out);
    return ok ? 0 : -1;
}

static void *rank_main(void *p)
{
    rank_t *r = (rank_t *)p;
    m2v_enc *enc[MAXK] = {0};
    m2v_comm *comm[MAXK] = {0};
    void *d_out[MAXK] = {0};
    int busy[MAXK], seq_of[MAXK], err = 0;
    r->rc = 0;
    if (hipSetDevice(0) != hipSuccess) { r->rc = -1; return NULL; }
    for (int k = 0; k < r->K; ++k) {
        busy[k] = 0;
        enc[k] = m2v_create(r->XL, r->YL, r->VL, r->Q, /*device=*/0, &err);
        /* a landing block per handle; creating + connecting is collective over the base: every rank, the same order */
        comm[k] = enc[k] && r->nranks > 1 ? m2v_comm_init_peer(r->base, r->rank, /*device=*/0, 0, &err) : NULL;
        if (!enc[k] || (r->nranks > 1 && (!comm[k] || m2v_comm_peer_connect_all(comm[k]) < 0)) || hipMalloc(&d_out[k], r->cap) != hipSuccess) {
            fprintf(stderr, "rank %d: set-up failed: %s / %s\n", r->rank, m2v_last_error(NULL), m2v_comm_last_error());
            r->rc = -1;
            return NULL;      /* (a real caller would tell the other ranks; here they fail in their next collective) */
        }
    }
    for (int i = 0; i < r->nseq + r->K && r->rc == 0; ++i) {
        const int k = i % r->K;
        if (busy[k]) {                                  /* collect sequence i - K: sizes, strips, assembly on its output rank */
            size_t bytes = 0;
            const int rc = m2v_strip_encode_end(enc[k], &bytes);
            if (rc < 0) { fprintf(stderr, "rank %d: m2v_strip_encode_end: %s\n", r->rank, m2v_last_error(enc[k])); r->rc = rc; break; }
            if (seq_of[k] % r->nranks == r->rank && write_stream(r->prefix, seq_of[k], d_out[k], bytes) < 0) r->rc = -1;
            busy[k] = 0;
        }
        if (i < r->nseq) {                              /* start sequence i: nothing is waited for */
            const int rc = m2v_strip_encode_begin(enc[k], comm[k], r->rank, r->nranks, /*dst_rank=*/i % r->nranks, r->xs16, r->ys16, r->pframes,
                                                  r->d_frames, r->nframes, d_out[k], r->cap, NULL);
            if (rc < 0) { fprintf(stderr, "rank %d: m2v_strip_encode_begin: %s\n", r->rank, m2v_last_error(enc[k])); r->rc = rc; break; }
            busy[k] = 1;
            seq_of[k] = i;
        }
    }
    for (int k = 0; k < r->K; ++k) {
        if (enc[k]) m2v_destroy(enc[k]);               /* waits for what is left on the handle's stream */
        /* every rank has been through the last sequence's sizes all-gather, which sits behind every rank's kernels of that sequence:
         * no neighbour is storing into this landing block any more.  The peer communicators go before their base. */
        if (comm[k]) m2v_comm_destroy(comm[k]);
        if (d_out[k]) (void)hipFree(d_out[k]);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: %s in.yuv444p W H pframes nranks nsequences K out_prefix\n", argv[0]); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]), pf = atoi(argv[4]), nranks = atoi(argv[5]), nseq = atoi(argv[6]), K = atoi(argv[7]);
    if (nranks < 1 || nranks > 16 || K < 1 || K > MAXK || nseq < 1) { fprintf(stderr, "1..16 ranks, 1..%d in flight\n", MAXK); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t fb = (size_t)W * H * 3, nframes = (size_t)ftell(f) / fb;
    fseek(f, 0, SEEK_SET);
    unsigned char *host = (unsigned char *)malloc(nframes * fb);
    if (!host || fread(host, fb, nframes, f) != nframes) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    void *d_frames = NULL;
    const size_t cap = nframes * ((size_t)(W / 16) * (H / 16) * 1216 + (size_t)(H / 16) * 8 + 64) + 256;   /* worst case */
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&d_frames, nframes * fb) != hipSuccess ||
        hipMemcpy(d_frames, host, nframes * fb, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "no GPU memory\n"); return 1; }
    int err = 0;
    m2v_comm *base = nranks > 1 ? m2v_comm_init_local(nranks, &err) : NULL;
    if (nranks > 1 && !base) { fprintf(stderr, "m2v_comm_init_local: %s\n", m2v_comm_last_error()); return 1; }
    rank_t ranks[16];
    pthread_t th[16];
    for (int r = 0; r < nranks; ++r) {
        memset(&ranks[r], 0, sizeof ranks[r]);
        ranks[r].rank = r; ranks[r].nranks = nranks; ranks[r].nseq = nseq; ranks[r].K = K; ranks[r].base = base;
        ranks[r].xs16 = (unsigned)(W / 16); ranks[r].ys16 = (unsigned)(H / 16); ranks[r].pframes = (unsigned)pf;
        ranks[r].XL = 7; ranks[r].YL = 7; ranks[r].VL = 3; ranks[r].Q = 2;
        ranks[r].d_frames = d_frames; ranks[r].nframes = nframes; ranks[r].cap = cap; ranks[r].prefix = argv[8];
        pthread_create(&th[r], NULL, rank_main, &ranks[r]);
    }
    int bad = 0;
    for (int r = 0; r < nranks; ++r) { pthread_join(th[r], NULL); bad |= ranks[r].rc != 0; }
    if (!bad) printf("%d sequences of %zu frames %dx%d, %d strips, %d in flight per rank\n", nseq, nframes, W, H, nranks, K);
    if (base) m2v_comm_destroy(base);
    (void)hipFree(d_frames);
    free(host);
    return bad;
}
