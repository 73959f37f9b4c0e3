// m2v_tb — file-to-file driver over the C-ABI; the counterpart of SIM/tb_mpeg2encoder.v.
//
//   m2v_tb [-XL n] [-YL n] [-VL n] [-Q n] [-p pframes] [-d device] [-bubbles] [-conformant] [-ps] [-ts]
//          in.yuv W H out.m2v  [in2.yuv W2 H2 out2.m2v ...]
//
// -conformant switches the encoder's option "conformant" on (ISO reconstruction loop; NOT byte-identical to the RTL).
// -ps / -ts additionally write out.m2v.mpg / out.m2v.ts: the same elementary stream in an MPEG-2 program / transport
// stream (include/m2v_container.h), so the result plays in an ordinary player.
//
// Like the testbench it encodes the listed videos back to back on ONE encoder instance (TB:150:
// "verify the module can end a sequence and start the next"), pushes only the complete frames of
// each file (TB:220), 4 pixels per beat in raster order (TB:224-229), pulses stop with i_en = 0
// (TB:249-252) and writes o_data byte 0 first (TB:260-262).  Defaults are the testbench's:
// XL=7 YL=6 VECTOR_LEVEL=3 Q_LEVEL=2 i_pframes_count=23 (TB:23-24, 98-106).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/m2v_container.h"
#include "../../include/m2v_mi355x.h"

int main(int argc, char **argv)
{
    int XL = 7, YL = 6, VL = 3, Q = 2, pf = 23, dev = 0, bubbles = 0, conformant = 0, want_ps = 0, want_ts = 0;
    int i = 1;
    for (; i < argc && argv[i][0] == '-'; ++i) {
        if (!strcmp(argv[i], "-bubbles")) { bubbles = 1; continue; }
        if (!strcmp(argv[i], "-conformant")) { conformant = 1; continue; }
        if (!strcmp(argv[i], "-ps")) { want_ps = 1; continue; }
        if (!strcmp(argv[i], "-ts")) { want_ts = 1; continue; }
        if (i + 1 >= argc) break;
        int v = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-XL")) XL = v; else if (!strcmp(argv[i], "-YL")) YL = v;
        else if (!strcmp(argv[i], "-VL")) VL = v; else if (!strcmp(argv[i], "-Q")) Q = v;
        else if (!strcmp(argv[i], "-p")) pf = v; else if (!strcmp(argv[i], "-d")) dev = v;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
        ++i;
    }
    if ((argc - i) < 4 || (argc - i) % 4) {
        fprintf(stderr, "usage: %s [-XL n] [-YL n] [-VL n] [-Q n] [-p pframes] [-d dev] in.yuv W H out.m2v ...\n", argv[0]);
        return 2;
    }
    int err = 0;
    m2v_enc *e = m2v_create(XL, YL, VL, Q, dev, &err);
    if (!e) { fprintf(stderr, "*** m2v_create failed (%d): an MI355X is required, there is no CPU fallback\n", err); return 1; }
    if (conformant) m2v_set_option(e, "conformant", 1);
    int num_video = 0;
    for (; i + 3 < argc; i += 4) {
        ++num_video;
        const char *in = argv[i], *out = argv[i + 3];
        const int xsize = atoi(argv[i + 1]), ysize = atoi(argv[i + 2]);
        printf("start to encode video %d (%4dx%4d)\n", num_video, xsize, ysize);
        FILE *fi = fopen(in, "rb");
        if (!fi) { printf("*** couldn't open input file\n"); return 1; }                       // TB:175-180
        FILE *fo = fopen(out, "wb");
        if (!fo) { printf("*** couldn't open output file\n"); return 1; }                      // TB:182-187
        if (xsize < 64 || xsize > (16 << XL) || xsize % 16) {                                  // TB:189-194
            printf("*** xsize=%4d is invalid, which must in range [64,%4d], and must be a multiple of 16\n", xsize, 16 << XL);
            return 1;
        }
        if (ysize < 64 || ysize > (16 << YL) || ysize % 16) {                                  // TB:196-201
            printf("*** ysize=%4d is invalid, which must in range [64,%4d], and must be a multiple of 16\n", ysize, 16 << YL);
            return 1;
        }
        const size_t fb = (size_t)xsize * ysize * 3;
        std::vector<uint8_t> frame(fb), word(1 << 20), es;
        size_t frames = 0, bytes = 0;
        const auto t0 = std::chrono::steady_clock::now();
        auto drain = [&](bool until_last) {
            for (;;) {
                int last = 0;
                long long n = m2v_pull(e, word.data(), word.size(), &last);
                if (n < 0) { fprintf(stderr, "*** m2v_pull: %s\n", m2v_last_error(e)); exit(1); }
                if (n) {
                    fwrite(word.data(), 1, (size_t)n, fo);
                    bytes += (size_t)n;
                    if (want_ps || want_ts) es.insert(es.end(), word.begin(), word.begin() + n);
                }
                if (last || (!until_last && n == 0)) break;
                if (until_last && n == 0 && !m2v_busy(e)) break;
            }
        };
        while (fread(frame.data(), 1, fb, fi) == fb) {                                         // complete frames only (TB:220)
            printf("  start to encode video %d frame %3zu\n", num_video, frames);
            int r;
            if (!bubbles) {
                r = m2v_push_frames(e, (uint32_t)xsize / 16, (uint32_t)ysize / 16, (uint32_t)pf, frame.data(), 1);
            } else {                                                                           // beat-level, odd batch sizes
                const size_t npix = (size_t)xsize * ysize;
                size_t b = 0, nb = npix / 4;
                r = 0;
                while (b < nb && r == 0) {
                    size_t take = 1 + (b * 7919) % 61;
                    if (take > nb - b) take = nb - b;
                    r = m2v_push_beats(e, (uint32_t)xsize / 16, (uint32_t)ysize / 16, (uint32_t)pf, frame.data() + b * 4,
                                       frame.data() + npix + b * 4, frame.data() + 2 * npix + b * 4, take, 0);
                    b += take;
                }
            }
            if (r < 0) { fprintf(stderr, "*** push failed: %s\n", m2v_last_error(e)); return 1; }
            ++frames;
            drain(false);
        }
        if (m2v_sequence_stop(e) < 0) { fprintf(stderr, "*** stop failed: %s\n", m2v_last_error(e)); return 1; }
        drain(true);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fclose(fi);
        fclose(fo);
        for (int kind = 0; kind < 2; ++kind) {
            if (!(kind ? want_ts : want_ps) || es.empty()) continue;
            size_t need = 0;
            int r = kind ? m2vc_mux_ts(es.data(), es.size(), nullptr, 0, &need) : m2vc_mux_ps(es.data(), es.size(), nullptr, 0, &need);
            std::vector<uint8_t> mux(need);
            if (r == 0) r = kind ? m2vc_mux_ts(es.data(), es.size(), mux.data(), mux.size(), &need)
                                 : m2vc_mux_ps(es.data(), es.size(), mux.data(), mux.size(), &need);
            if (r < 0) { fprintf(stderr, "*** multiplexer failed (%d)\n", r); return 1; }
            const std::string name = std::string(out) + (kind ? ".ts" : ".mpg");
            FILE *fm = fopen(name.c_str(), "wb");
            if (!fm) { printf("*** couldn't open %s\n", name.c_str()); return 1; }
            fwrite(mux.data(), 1, need, fm);
            fclose(fm);
            printf("  %s: %zu bytes\n", name.c_str(), need);
        }
        printf("end of video %d: %zu frames -> %zu bytes, %.3f s, %.1f MPixels/s incl. file I/O and PCIe\n", num_video, frames,
               bytes, s, (double)frames * xsize * ysize / s * 1e-6);
    }
    m2v_destroy(e);
    return 0;
}
