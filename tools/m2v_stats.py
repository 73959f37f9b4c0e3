#!/usr/bin/env python3
"""Size / type statistics of an .m2v elementary stream in the style of the reference's README:735-768 (bytes per clip,
PSNR), optional PSNR against the source through the repo's decoder, optional PS / TS multiplexing.

    python tools/m2v_stats.py out.m2v [--yuv src.yuv] [--ps out.mpg] [--ts out.ts]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import m2v_load


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("m2v")
    ap.add_argument("--yuv", help="planar yuv444p source (SIM/tb_mpeg2encoder.v:210-218 layout) for PSNR")
    ap.add_argument("--ps", help="write an MPEG-2 program stream")
    ap.add_argument("--ts", help="write an MPEG-2 transport stream")
    ap.add_argument("--pictures", action="store_true", help="one line per picture")
    args = ap.parse_args()
    M = m2v_load.load()
    C = M.container
    es = open(args.m2v, "rb").read()
    info, pics = C.scan(es)
    num, den = C.frame_rate(info.frame_rate_code)
    secs = info.pictures * den / num
    px = info.width * info.height * info.pictures
    print("%s: %dx%d, %d pictures (%d I, %d P, %d GOPs), %.3f s at %d/%d fps" % (
        args.m2v, info.width, info.height, info.pictures, info.i_pictures, info.p_pictures, info.gops, secs, num, den))
    print("  %d bytes (+%d padding) = %.4f bit/pixel, %.1f kbit/s; I pictures %.1f %% of the bytes" % (
        info.bytes, info.padding_bytes, info.bytes * 8 / px, info.bytes * 8 / secs / 1e3,
        100.0 * sum(p.bytes for p in pics if p.coding_type == 1) / max(info.bytes, 1)))
    sizes = np.array([p.bytes for p in pics], dtype=np.float64)
    for name, t in (("I", 1), ("P", 2)):
        s = sizes[[p.coding_type == t for p in pics]]
        if s.size:
            print("  %s pictures: mean %.0f  min %.0f  max %.0f bytes" % (name, s.mean(), s.min(), s.max()))
    if args.pictures:
        for k, p in enumerate(pics):
            print("  %5d  %s  tref %3d  %8d bytes  %d slices%s" % (k, "IP"[p.coding_type - 1], p.temporal_reference, p.bytes,
                                                                p.slices, "  GOP" if p.gop_start else ""))
    if args.yuv:
        m2v_decode = M.decoder
        dec = m2v_decode.decode(es, quirks=True)           # the encoder's own reconstruction (see fpga-mpeg2-encoder_amd/decoder.py)
        W, H = info.width, info.height
        src = np.fromfile(args.yuv, np.uint8)
        n = min(src.size // (3 * W * H), len(dec.frames))
        src = src[:n * 3 * W * H].reshape(n, 3, H, W)
        ps = [m2v_decode.psnr(src[k, 0], dec.frames[k][0]) for k in range(n)]
        print("  luma PSNR over %d frames: mean %.2f dB  min %.2f dB" % (n, float(np.mean(ps)), float(np.min(ps))))
    if args.ps:
        d = C.mux_ps(es)
        open(args.ps, "wb").write(d)
        print("  program stream  : %s, %d bytes (+%.2f %%)" % (args.ps, len(d), 100.0 * (len(d) - info.bytes) / info.bytes))
    if args.ts:
        d = C.mux_ts(es)
        open(args.ts, "wb").write(d)
        print("  transport stream: %s, %d bytes (+%.2f %%)" % (args.ts, len(d), 100.0 * (len(d) - info.bytes) / info.bytes))


if __name__ == "__main__":
    main()
