"""CPU: libm2v_container.so (include/m2v_container.h) - elementary-stream scan, MPEG-2 PS and TS multiplexers.
The streams come from the oracle; the demultiplexers below are written here, independently, from ISO/IEC 13818-1,
and must give back the elementary stream byte for byte with well-formed headers, monotonic clocks and one PTS per
picture."""
import ctypes

import numpy as np
import pytest

import m2v_load
from oracle import m2v_oracle_ctypes as orc


@pytest.fixture(scope="module")
def env():
    M = m2v_load.load()
    import importlib
    C = importlib.import_module(M.__name__ + ".container")
    orc.build()
    clips = {}
    f = M.synth.clip(96, 64, 11, clip_index=40)
    clips["ip"] = (orc.encode(f, 6, 4, 3, XL=6, YL=6), 11, 96, 64, 3)          # 3 GOPs of 1 I + 3 P (last one short)
    f = M.synth.clip(64, 64, 3, clip_index=41)
    clips["i"] = (orc.encode(f, 4, 4, 0, XL=4, YL=4), 3, 64, 64, 0)            # intra only
    f = M.synth.clip(320, 240, 6, clip_index=42)
    clips["big"] = (orc.encode(f, 20, 15, 255, XL=6, YL=6, Q=1), 6, 320, 240, 255)   # pictures of several packs each
    return M, C, clips


def test_exports_and_frame_rates(env):
    M, C, _ = env
    L = C.lib()
    for name in ("m2vc_frame_rate", "m2vc_scan", "m2vc_mux_ps", "m2vc_mux_ts"):
        assert hasattr(L, name)
    assert C.frame_rate(2) == (24, 1) and C.frame_rate(4) == (30000, 1001) and C.frame_rate(8) == (60, 1)
    with pytest.raises(C.ContainerError):
        C.frame_rate(0)


@pytest.mark.parametrize("key", ["ip", "i", "big"])
def test_scan(env, key):
    M, C, clips = env
    es, n, W, H, pf = clips[key]
    info, pics = C.scan(es)
    assert (info.width, info.height) == (W, H)
    assert info.frame_rate_code == 2                                          # the RTL writes 24 fps (RTL:2598-2617)
    assert info.pictures == n == len(pics)
    gop = pf + 1
    assert info.i_pictures == (n + gop - 1) // gop and info.p_pictures == n - info.i_pictures
    assert info.gops == info.i_pictures and info.slices == n * (H // 16)
    assert info.has_sequence_end == 1 and info.bytes + info.padding_bytes == len(es) and len(es) % 32 == 0
    pos = pics[0].offset
    for k, p in enumerate(pics):
        assert p.offset == pos and p.slices == H // 16
        assert p.coding_type == (1 if k % gop == 0 else 2) and p.temporal_reference == k % gop
        assert p.gop_start == (1 if k % gop == 0 else 0)
        assert es[p.offset:p.offset + 4] == (b"\x00\x00\x01\xb8" if p.gop_start else b"\x00\x00\x01\x00")
        pos += p.bytes
    assert pos + 4 == info.bytes and es[pos:pos + 4] == b"\x00\x00\x01\xb7"


def test_scan_rejects_garbage(env):
    M, C, clips = env
    es = clips["i"][0]
    with pytest.raises(C.ContainerError):
        C.scan(b"\x00\x00\x01\xb4" + es[4:])
    with pytest.raises(C.ContainerError):
        C.scan(es + b"\x01")                                                  # something other than zero padding at the end
    with pytest.raises(C.ContainerError):
        C.mux_ps(b"")


def _ts33(b, prefix):
    """5-byte time stamp: 4-bit prefix, 3+15+15 bits, marker after each group"""
    assert b[0] >> 4 == prefix and b[0] & 1 and b[2] & 1 and b[4] & 1
    return ((b[0] >> 1) & 7) << 30 | (b[1] << 7 | b[2] >> 1) << 15 | (b[3] << 7 | b[4] >> 1)


def demux_ps(ps):
    """-> (elementary stream, [(es offset of the packet's first payload byte, PTS)], [SCR in 27 MHz], mux_rate)"""
    es, stamps, scrs, rates = bytearray(), [], [], set()
    p, first = 0, True
    while True:
        assert ps[p:p + 3] == b"\x00\x00\x01"
        code = ps[p + 3]
        if code == 0xB9:                                                      # MPEG_program_end_code
            assert p + 4 == len(ps)
            break
        assert code == 0xBA, "a pack_header starts every pack"
        h = ps[p + 4:p + 14]
        assert h[0] >> 6 == 1 and h[0] & 4 and h[2] & 4 and h[4] & 4 and h[5] & 1 and h[8] & 3 == 3
        base = ((h[0] >> 3) & 7) << 30 | (h[0] & 3) << 28 | h[1] << 20 | (h[2] >> 3) << 15 | (h[2] & 3) << 13 | h[3] << 5 | h[4] >> 3
        ext = (h[4] & 3) << 7 | h[5] >> 1
        assert ext < 300
        scrs.append(base * 300 + ext)
        rates.add(h[6] << 14 | h[7] << 6 | h[8] >> 2)
        assert h[9] >> 3 == 0x1F and h[9] & 7 == 0
        pack_start = p
        p += 14
        if ps[p:p + 4] == b"\x00\x00\x01\xbb":
            assert first, "system header in the first pack only"
            n = ps[p + 4] << 8 | ps[p + 5]
            s = ps[p + 6:p + 6 + n]
            assert n == 9 and s[0] & 0x80 and s[2] & 1 and s[4] & 0x20 and s[4] & 0x1F == 1 and s[5] == 0x7F
            assert s[6] == 0xE0 and s[7] >> 6 == 3 and s[7] & 0x20
            p += 6 + n
        first = False
        assert ps[p:p + 4] == b"\x00\x00\x01\xe0"
        n = ps[p + 4] << 8 | ps[p + 5]
        f1, f2, hl = ps[p + 6], ps[p + 7], ps[p + 8]
        assert f1 >> 6 == 2 and f2 & 0x3F == 0 and hl in (0, 5)
        if f2 >> 6 == 2:
            assert hl == 5
            stamps.append((len(es), _ts33(ps[p + 9:p + 14], 2)))
            assert f1 & 4, "data_alignment_indicator"
            assert ps[p + 9 + hl:p + 9 + hl + 3] == b"\x00\x00\x01"
        else:
            assert f2 >> 6 == 0 and hl == 0
        es += ps[p + 9 + hl:p + 6 + n]
        p += 6 + n
        assert p - pack_start <= 2048
    assert len(rates) == 1
    return bytes(es), stamps, scrs, rates.pop()


@pytest.mark.parametrize("key", ["ip", "i", "big"])
def test_program_stream(env, key):
    M, C, clips = env
    es, n, W, H, pf = clips[key]
    info, pics = C.scan(es)
    ps = C.mux_ps(es)
    got, stamps, scrs, rate = demux_ps(ps)
    assert got == es[:info.bytes]                                             # the zero padding after the end code is not muxed
    assert len(stamps) == n, "one PTS per picture"
    for k, ((off, pts), pic) in enumerate(zip(stamps, pics)):
        assert off == (0 if k == 0 else pic.offset), "every picture starts a PES packet; the headers go with the first"
        assert pts - stamps[0][1] == k * 3750                                 # 90 kHz / 24 fps
    assert all(b > a for a, b in zip(scrs, scrs[1:])) and scrs[0] == 0
    # every picture has completely arrived (SCR of the pack after its last byte) before it is presented
    assert rate * 50 * 8 >= 1_000_000
    total_time = len(ps) / (rate * 50)
    assert stamps[0][1] / 90000 > 0 and stamps[-1][1] / 90000 + 1 / 24 >= total_time * 0.5


def _crc32_mpeg(data):
    c = 0xFFFFFFFF
    for b in data:
        c ^= b << 24
        for _ in range(8):
            c = ((c << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if c & 0x80000000 else (c << 1) & 0xFFFFFFFF
    return c


def demux_ts(ts):
    """-> (elementary stream, [PTS per PES packet], [PCR 27 MHz], [byte position of each PCR packet])"""
    assert len(ts) % 188 == 0
    cc = {}
    es, pts, pcrs, pcr_pos = bytearray(), [], [], []
    pmt_pid = video_pid = None
    pes_hdr_left = 0
    for k in range(0, len(ts), 188):
        p = ts[k:k + 188]
        assert p[0] == 0x47 and not p[1] & 0x80
        pusi, pid = bool(p[1] & 0x40), (p[1] & 0x1F) << 8 | p[2]
        afc, c = (p[3] >> 4) & 3, p[3] & 15
        assert p[3] >> 6 == 0 and afc in (1, 3)
        assert c == (cc.get(pid, -1) + 1) & 15 or pid not in cc, "continuity_counter"
        cc[pid] = c
        q = 4
        if afc == 3:
            afl = p[4]
            assert afl <= 183
            if afl:
                flags = p[5]
                assert flags & 0xEF == 0
                if flags & 0x10:
                    b = p[6:12]
                    base = b[0] << 25 | b[1] << 17 | b[2] << 9 | b[3] << 1 | b[4] >> 7
                    ext = (b[4] & 1) << 8 | b[5]
                    assert ext < 300 and (b[4] >> 1) & 0x3F == 0x3F
                    pcrs.append(base * 300 + ext)
                    pcr_pos.append(k)
                    assert all(x == 0xFF for x in p[12:5 + afl])
                else:
                    assert all(x == 0xFF for x in p[6:5 + afl])
            q = 5 + afl
        pay = p[q:]
        if pid == 0 or pid == pmt_pid:
            assert pusi and pay[0] == 0
            sec = pay[1:]
            n = (sec[1] & 15) << 8 | sec[2]
            sec = sec[:3 + n]
            assert sec[1] & 0x80 and _crc32_mpeg(sec) == 0, "PSI CRC"
            assert all(x == 0xFF for x in pay[1 + 3 + n:])
            if pid == 0:
                assert sec[0] == 0 and (sec[8] << 8 | sec[9]) == 1
                pmt_pid = (sec[10] & 0x1F) << 8 | sec[11]
            else:
                assert sec[0] == 2 and sec[12] == 2, "one MPEG-2 video stream"
                video_pid = (sec[13] & 0x1F) << 8 | sec[14]
                assert (sec[8] & 0x1F) << 8 | sec[9] == video_pid, "the PCR rides on the video PID"
        else:
            assert pid == video_pid, "PAT and PMT come first"
            if pusi:
                assert afc == 3 and pcr_pos and pcr_pos[-1] == k, "a PES packet starts with a PCR"
                assert pay[:4] == b"\x00\x00\x01\xe0" and pay[4:6] == b"\x00\x00"
                assert pay[6] >> 6 == 2 and pay[7] == 0x80 and pay[8] == 5
                pts.append(_ts33(pay[9:14], 2))
                pay = pay[14:]
            es += pay
    return bytes(es), pts, pcrs, pcr_pos


@pytest.mark.parametrize("key", ["ip", "i", "big"])
def test_transport_stream(env, key):
    M, C, clips = env
    es, n, W, H, pf = clips[key]
    info, pics = C.scan(es)
    ts = C.mux_ts(es)
    got, pts, pcrs, pcr_pos = demux_ts(ts)
    assert got == es[:info.bytes]
    assert len(pts) == n and [t - pts[0] for t in pts] == [k * 3750 for k in range(n)]
    assert all(b > a for a, b in zip(pcrs, pcrs[1:]))
    # constant multiplex rate: the PCRs are proportional to their byte positions
    if len(pcrs) > 2:
        r = [(pcrs[i] - pcrs[0]) / (pcr_pos[i] - pcr_pos[0]) for i in range(1, len(pcrs))]
        assert max(r) - min(r) < 1e-3 * max(r) + 1
    # each picture is complete (PCR of the next PES packet's first byte) before its PTS
    for k in range(n - 1):
        assert pcrs[k + 1] / 300 < pts[k]


def test_size_query_and_small_buffer(env):
    M, C, clips = env
    es = clips["ip"][0]
    L = C.lib()
    n = ctypes.c_size_t()
    assert L.m2vc_mux_ps(es, len(es), None, 0, ctypes.byref(n)) == 0 and n.value > len(es) - 64
    buf = ctypes.create_string_buffer(16)
    assert L.m2vc_mux_ps(es, len(es), buf, 16, ctypes.byref(n)) == -3
    info = C.StreamInfo()
    pics = (C.Picture * 2)()
    assert L.m2vc_scan(es, len(es), ctypes.byref(info), pics, 2, ctypes.byref(n)) == -3 and n.value == 11
    assert pics[1].coding_type == 2
