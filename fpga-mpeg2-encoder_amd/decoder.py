"""A small MPEG-2 video elementary-stream decoder written from the ISO/IEC 13818-2 syntax (clause 6.2) and
decoding process (clause 7), for the subset this encoder emits: frame pictures, I and P, 4:2:0, frame prediction
with one forward vector, f_code 1, intra_vlc_format 0, zig-zag scan, q_scale_type 0, default matrices.

It is an INDEPENDENT check of the stream layer (every header field, macroblock types, vector and DC prediction, the
run/level coding, end-of-slice padding) and, with `quirks=True`, of the encoder's reconstruction loop: the RTL
deviates from ISO in a few places and a conformant decoder would drift from the encoder's own
reference frames, so the quirks can be switched on to reproduce the encoder's reconstruction exactly:
   * half-pel 4-sample average rounds with +1 instead of +2                      (RTL:764)
   * chroma vector = floor(mv / 2) instead of mv / 2 truncated toward zero       (RTL:1854-1916)
   * intra AC dequantisation floors negatives; no mismatch control               (RTL:2132-2148)
   * IDCT = Chen-Wang with an 18-bit row store and a +-255 output clip           (RTL:844-972)
Everything here, the VLC tables of Annex B included, is written from the standard: nothing is shared with the
encoder, the oracle or the RTL.  It is an analysis tool (tools/m2v_stats.py: PSNR of a stream against its source) and
the independent checker of the parity tests; it runs on the CPU and is not on the encode path.
"""
import numpy as np


class BitReader:
    def __init__(self, data):
        self.d = data
        self.pos = 0                 # bit position

    def bits(self, n):
        v = 0
        for _ in range(n):
            byte = self.d[self.pos >> 3]
            v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def peek(self, n):
        p = self.pos
        v = self.bits(n)
        self.pos = p
        return v

    def aligned(self):
        return (self.pos & 7) == 0

    def align(self):
        pad = (-self.pos) & 7
        assert self.bits(pad) == 0, "non-zero stuffing before a start code"

    def next_start_code(self):
        """next_start_code(): zero stuffing to the byte boundary, then zero bytes, then 0x000001"""
        self.align()
        while self.peek(24) != 1:
            assert self.bits(8) == 0, "garbage before a start code"
        return self.peek(32) & 0xFF


def _vlc_decoder(entries):
    """entries: list of (code, length, symbol) -> function(BitReader) -> symbol"""
    table = {}
    for code, length, sym in entries:
        assert (length, code) not in table, "duplicate code"
        table[(length, code)] = sym
    maxlen = max(l for _, l, _ in entries)
    for (l1, c1) in table:                                # a VLC table must be prefix free
        for l2 in range(1, l1):
            assert (l2, c1 >> (l1 - l2)) not in table, "code %s/%d has a prefix in the table" % (bin(c1), l1)

    def dec(br):
        v = 0
        for l in range(1, maxlen + 1):
            v = (v << 1) | br.bits(1)
            if (l, v) in table:
                return table[(l, v)]
        raise ValueError("invalid VLC at bit %d" % br.pos)
    return dec


# ---------------------------------------------------------------------------------------------------------------
# ISO/IEC 13818-2 Annex B, typed in from the standard's tables (NOT taken from the encoder, the oracle or the RTL:
# tests/test_decoder_tables.py checks that the three agree entry by entry)
# ---------------------------------------------------------------------------------------------------------------
_B10_MOTION = """0:1 1:01 2:001 3:0001 4:000011 5:0000101 6:0000100 7:0000011 8:000001011 9:000001010 10:000001001
11:0000010001 12:0000010000 13:0000001111 14:0000001110 15:0000001101 16:0000001100"""

_B12_DC_LUMA = "0:100 1:00 2:01 3:101 4:110 5:1110 6:11110 7:111110 8:1111110 9:11111110 10:111111110 11:111111111"
_B13_DC_CHROMA = "0:00 1:01 2:10 3:110 4:1110 5:11110 6:111110 7:1111110 8:11111110 9:111111110 10:1111111110 11:1111111111"

_B9_CBP = """60:111 4:1101 8:1100 16:1011 32:1010 12:10011 48:10010 20:10001 40:10000 28:01111 44:01110 52:01101 56:01100
1:01011 61:01010 2:01001 62:01000 24:001111 36:001110 3:001101 63:001100 5:0010111 9:0010110 17:0010101 33:0010100
6:0010011 10:0010010 18:0010001 34:0010000 7:00011111 11:00011110 19:00011101 35:00011100 13:00011011 49:00011010
21:00011001 41:00011000 14:00010111 50:00010110 22:00010101 42:00010100 15:00010011 51:00010010 23:00010001 43:00010000
25:00001111 37:00001110 26:00001101 38:00001100 29:00001011 45:00001010 53:00001001 57:00001000 30:00000111 46:00000110
54:00000101 58:00000100 31:000000111 47:000000110 55:000000101 59:000000100 27:000000011 39:000000010"""

# Table B-14 (DCT coefficients table zero), "run,level:code" without the sign bit; (0,1) is listed in its
# "not first coefficient" form '11' - the first-coefficient form '1s' is handled in the block loop
_B14_AC = """0,1:11 1,1:011 0,2:0100 2,1:0101 0,3:00101 3,1:00111 4,1:00110 1,2:000110 5,1:000111 6,1:000101 7,1:000100
0,4:0000110 2,2:0000100 8,1:0000111 9,1:0000101
0,5:00100110 0,6:00100001 1,3:00100101 3,2:00100100 10,1:00100111 11,1:00100011 12,1:00100010 13,1:00100000
0,7:0000001010 1,4:0000001100 2,3:0000001011 4,2:0000001111 5,2:0000001001 14,1:0000001110 15,1:0000001101 16,1:0000001000
0,8:000000011101 0,9:000000011000 0,10:000000010011 0,11:000000010000 1,5:000000011011 2,4:000000010100 3,3:000000011100
4,3:000000010010 6,2:000000011110 7,2:000000010101 8,2:000000010001 17,1:000000011111 18,1:000000011010 19,1:000000011001
20,1:000000010111 21,1:000000010110
0,12:0000000011010 0,13:0000000011001 0,14:0000000011000 0,15:0000000010111 1,6:0000000010110 1,7:0000000010101
2,5:0000000010100 3,4:0000000010011 5,3:0000000010010 9,2:0000000010001 10,2:0000000010000 22,1:0000000011111
23,1:0000000011110 24,1:0000000011101 25,1:0000000011100 26,1:0000000011011
0,16:00000000011111 0,17:00000000011110 0,18:00000000011101 0,19:00000000011100 0,20:00000000011011 0,21:00000000011010
0,22:00000000011001 0,23:00000000011000 0,24:00000000010111 0,25:00000000010110 0,26:00000000010101 0,27:00000000010100
0,28:00000000010011 0,29:00000000010010 0,30:00000000010001 0,31:00000000010000
0,32:000000000011000 0,33:000000000010111 0,34:000000000010110 0,35:000000000010101 0,36:000000000010100
0,37:000000000010011 0,38:000000000010010 0,39:000000000010001 0,40:000000000010000 1,8:000000000011111
1,9:000000000011110 1,10:000000000011101 1,11:000000000011100 1,12:000000000011011 1,13:000000000011010 1,14:000000000011001
1,15:0000000000010011 1,16:0000000000010010 1,17:0000000000010001 1,18:0000000000010000 6,3:0000000000010100
11,2:0000000000011010 12,2:0000000000011001 13,2:0000000000011000 14,2:0000000000010111 15,2:0000000000010110
16,2:0000000000010101 27,1:0000000000011111 28,1:0000000000011110 29,1:0000000000011101 30,1:0000000000011100
31,1:0000000000011011"""
_B14_EOB, _B14_ESCAPE = "10", "000001"

# 7.3 figure 7-2: zig-zag scan (alternate_scan = 0), scan position of raster [v][u]
_ZIGZAG = [0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
           10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63]
# 6.3.11: default intra quantiser matrix, raster [v][u]
_INTRA_W = [8, 16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37, 19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
            22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58, 26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83]


def _parse(text, key=int):
    """'sym:bits sym:bits ...' -> [(code, length, sym)]"""
    out = []
    for item in text.split():
        sym, bits = item.split(":")
        out.append((int(bits, 2), len(bits), key(sym)))
    return out


def iso_tables():
    """The raw Annex B tables as lists of (code, length, symbol); used by tests/test_decoder_tables.py"""
    ac = _parse(_B14_AC, key=lambda t: tuple(int(x) for x in t.split(",")))
    return dict(motion=_parse(_B10_MOTION), cbp=_parse(_B9_CBP), dcy=_parse(_B12_DC_LUMA), dcc=_parse(_B13_DC_CHROMA), ac=ac,
                zigzag=list(_ZIGZAG), intra_w=list(_INTRA_W))


def _tables():
    t = iso_tables()
    ac = [e for e in t["ac"] if e[2] != (0, 1)]          # '11' is matched before the table look-up (see the block loop)
    ac.append((int(_B14_EOB, 2), len(_B14_EOB), "EOB"))
    ac.append((int(_B14_ESCAPE, 2), len(_B14_ESCAPE), "ESC"))
    zz = np.zeros(64, np.int64)                          # scan position -> raster index
    for raster, pos in enumerate(_ZIGZAG):
        zz[pos] = raster
    return dict(motion=_vlc_decoder(t["motion"]), cbp=_vlc_decoder(t["cbp"]), dcy=_vlc_decoder(t["dcy"]),
                dcc=_vlc_decoder(t["dcc"]), ac=_vlc_decoder(ac), zz=zz, W=np.array(_INTRA_W, np.int64))


_T = None


def tables():
    global _T
    if _T is None:
        _T = _tables()
    return _T


# ---------------------------------------------------------------------------------------------------------------
# inverse DCT
# ---------------------------------------------------------------------------------------------------------------
def _s32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def _sext(x, bits):
    x &= (1 << bits) - 1
    return x - (1 << bits) if x >> (bits - 1) else x


W1, W2, W3, W5, W6, W7 = 2841, 2676, 2408, 1609, 1108, 565


def _idct_1d(a, row, quirks):
    """mpeg2decode's idctrow / idctcol (Chen-Wang), 32-bit arithmetic"""
    if row:
        x0, x1 = _s32((a[0] << 11) + 128), _s32(a[4] << 11)
    else:
        x0, x1 = _s32((a[0] << 8) + 8192), _s32(a[4] << 8)
    x2, x3, x4, x5, x6, x7 = a[6], a[2], a[1], a[7], a[5], a[3]
    r = 0 if row else 4
    sh = 0 if row else 3
    x8 = _s32(W7 * (x4 + x5) + r)
    x4 = _s32(x8 + (W1 - W7) * x4) >> sh
    x5 = _s32(x8 - (W1 + W7) * x5) >> sh
    x8 = _s32(W3 * (x6 + x7) + r)
    x6 = _s32(x8 - (W3 - W5) * x6) >> sh
    x7 = _s32(x8 - (W3 + W5) * x7) >> sh
    x8 = _s32(x0 + x1)
    x0 = _s32(x0 - x1)
    x1 = _s32(W6 * (x3 + x2) + r)
    x2 = _s32(x1 - (W2 + W6) * x2) >> sh
    x3 = _s32(x1 + (W2 - W6) * x3) >> sh
    x1 = _s32(x4 + x6)
    x4 = _s32(x4 - x6)
    x6 = _s32(x5 + x7)
    x5 = _s32(x5 - x7)
    x7 = _s32(x8 + x3)
    x8 = _s32(x8 - x3)
    x3 = _s32(x0 + x2)
    x0 = _s32(x0 - x2)
    x2 = _s32(181 * (x4 + x5) + 128) >> 8
    x4 = _s32(181 * (x4 - x5) + 128) >> 8
    out = [x7 + x1, x3 + x2, x0 + x4, x8 + x6, x8 - x6, x0 - x4, x3 - x2, x7 - x1]
    out = [_s32(v) >> (8 if row else 14) for v in out]
    if row:
        return [_sext(v, 18) for v in out] if quirks else out
    lo = -255 if quirks else -256
    return [max(lo, min(255, v)) for v in out]


def idct(F, quirks):
    rows = [_idct_1d([int(v) for v in F[i * 8:i * 8 + 8]], True, quirks) for i in range(8)]
    out = np.zeros(64, np.int64)
    for j in range(8):
        col = _idct_1d([rows[i][j] for i in range(8)], False, quirks)
        for i in range(8):
            out[i * 8 + j] = col[i]
    return out


# ---------------------------------------------------------------------------------------------------------------
# the decoder
# ---------------------------------------------------------------------------------------------------------------
class Decoded:
    def __init__(self):
        self.frames = []            # (Y, U, V) uint8 arrays
        self.pictures = []          # dicts of header fields
        self.mbs = []               # per picture: list of dict(type, mv, cbp)
        self.width = self.height = 0
        self.gops = []


def decode(data, quirks=True):
    T = tables()
    br = BitReader(data)
    out = Decoded()
    # ---- sequence_header ----
    assert br.next_start_code() == 0xB3
    br.bits(32)
    out.width, out.height = br.bits(12), br.bits(12)
    hdr = dict(aspect=br.bits(4), frame_rate_code=br.bits(4), bit_rate=br.bits(18))
    assert br.bits(1) == 1
    hdr.update(vbv=br.bits(10), constrained=br.bits(1))
    assert br.bits(1) == 0 and br.bits(1) == 0, "quantiser matrices are not loaded by this encoder"
    # ---- sequence_extension ----
    assert br.next_start_code() == 0xB5
    br.bits(32)
    assert br.bits(4) == 1
    hdr.update(profile_level=br.bits(8), progressive_sequence=br.bits(1), chroma_format=br.bits(2))
    assert br.bits(2) == 0 and br.bits(2) == 0 and br.bits(12) == 0 and br.bits(1) == 1
    br.bits(8)
    hdr.update(low_delay=br.bits(1))
    br.bits(7)
    assert hdr["chroma_format"] == 1
    # ---- sequence_display_extension ----
    assert br.next_start_code() == 0xB5
    br.bits(32)
    assert br.bits(4) == 2
    br.bits(3)
    if br.bits(1):
        br.bits(24)
    dw = br.bits(14)
    assert br.bits(1) == 1
    dh = br.bits(14)
    assert (dw, dh) == (out.width, out.height)
    out.sequence = hdr
    W, H = out.width, out.height
    mbw, mbh = W // 16, H // 16
    ref = None
    ended = False
    while True:
        code = br.next_start_code()
        if code == 0xB7:
            br.bits(32)
            ended = True
            break
        if code == 0xB8:                                        # group_of_pictures_header
            br.bits(32)
            tc = dict(drop=br.bits(1), hours=br.bits(5), minutes=br.bits(6))
            assert br.bits(1) == 1
            tc.update(seconds=br.bits(6), pictures=br.bits(6), closed_gop=br.bits(1), broken_link=br.bits(1))
            out.gops.append(tc)
            code = br.next_start_code()
        assert code == 0x00, hex(code)
        # ---- picture_header ----
        br.bits(32)
        pic = dict(temporal_reference=br.bits(10), type=br.bits(3), vbv_delay=br.bits(16))
        assert pic["type"] in (1, 2)
        if pic["type"] == 2:
            assert br.bits(1) == 0
            pic["forward_f_code"] = br.bits(3)
        assert br.bits(1) == 0                                  # extra_bit_picture
        # ---- picture_coding_extension ----
        assert br.next_start_code() == 0xB5
        br.bits(32)
        assert br.bits(4) == 8
        pic["f_code"] = [br.bits(4) for _ in range(4)]
        pic.update(intra_dc_precision=br.bits(2), picture_structure=br.bits(2), top_field_first=br.bits(1),
                   frame_pred_frame_dct=br.bits(1), concealment_motion_vectors=br.bits(1), q_scale_type=br.bits(1),
                   intra_vlc_format=br.bits(1), alternate_scan=br.bits(1), repeat_first_field=br.bits(1),
                   chroma_420_type=br.bits(1), progressive_frame=br.bits(1), composite_display_flag=br.bits(1))
        assert pic["picture_structure"] == 3 and pic["frame_pred_frame_dct"] == 1
        assert pic["q_scale_type"] == 0 and pic["intra_vlc_format"] == 0 and pic["alternate_scan"] == 0
        assert pic["concealment_motion_vectors"] == 0 and pic["intra_dc_precision"] == 2
        assert pic["f_code"][0] == 1 and pic["f_code"][1] == 1
        out.pictures.append(pic)
        if pic["type"] == 2:
            assert ref is not None, "P picture without a reference"
        Y = np.zeros((H, W), np.int64)
        U = np.zeros((H // 2, W // 2), np.int64)
        V = np.zeros((H // 2, W // 2), np.int64)
        mbs = []
        for row in range(mbh):
            assert br.next_start_code() == row + 1, "slices must come one per macroblock row, in order"
            br.bits(32)
            qscale = 2 * br.bits(5)                             # q_scale_type 0
            assert br.bits(1) == 0                              # extra_bit_slice
            dc_pred = [0, 0, 0]                                 # relative to the reset value 1 << (7 + precision)
            pmv = [0, 0]
            for col in range(mbw):
                assert br.bits(1) == 1, "macroblock_address_increment must be 1 (no skipped macroblocks)"
                # macroblock_type, tables B-2 / B-3 (without the quant variants, which this encoder never emits)
                if pic["type"] == 1:
                    assert br.bits(1) == 1
                    intra, mc, pattern = True, False, False
                else:
                    if br.bits(1):
                        intra, mc, pattern = False, True, True          # '1'   MC, coded
                    elif br.bits(1):
                        raise AssertionError("'01' No-MC coded is never emitted by this encoder")
                    elif br.bits(1):
                        intra, mc, pattern = False, True, False         # '001' MC, not coded
                    else:
                        assert br.bits(2) == 0b11                       # '00011' intra
                        intra, mc, pattern = True, False, False
                mv = [0, 0]
                if mc:
                    for t in range(2):                          # motion_code, no residual for f_code 1 (7.6.3.1)
                        m = T["motion"](br)
                        if m and br.bits(1):
                            m = -m
                        v = pmv[t] + m
                        if v < -16:
                            v += 32
                        elif v > 15:
                            v -= 32
                        pmv[t] = mv[t] = v
                else:
                    pmv = [0, 0]                                # 7.6.3.4: reset by an intra macroblock
                cbp = 63 if intra else (T["cbp"](br) if pattern else 0)
                # prediction
                if intra:
                    py = np.full((16, 16), 128, np.int64)
                    pu = np.full((8, 8), 128, np.int64)
                    pv = np.full((8, 8), 128, np.int64)
                else:
                    dc_pred = [0, 0, 0]                         # 7.2.1: reset by a non-intra macroblock
                    py = _predict(ref[0], 16 * col, 16 * row, mv[0], mv[1], 16, quirks)
                    if quirks:
                        cmv = [mv[0] >> 1, mv[1] >> 1]          # floor
                    else:
                        cmv = [int(mv[0] / 2), int(mv[1] / 2)]  # toward zero (7.6.3.7)
                    pu = _predict(ref[1], 8 * col, 8 * row, cmv[0], cmv[1], 8, quirks)
                    pv = _predict(ref[2], 8 * col, 8 * row, cmv[0], cmv[1], 8, quirks)
                for b in range(6):
                    res = np.zeros(64, np.int64)
                    if (cbp >> (5 - b)) & 1:
                        QF = np.zeros(64, np.int64)
                        k = 0
                        comp = 0 if b < 4 else b - 3
                        if intra:
                            size = (T["dcy"] if b < 4 else T["dcc"])(br)
                            diff = 0
                            if size:
                                d = br.bits(size)
                                diff = d if d >> (size - 1) else d - (1 << size) + 1
                            dc_pred[comp] += diff
                            QF[0] = dc_pred[comp]
                            k = 1
                        first = not intra
                        while True:
                            if first and br.peek(1) == 1:       # '1s': run 0 level +-1 as first coefficient
                                br.bits(1)
                                run, lvl = 0, 1
                                lvl = -lvl if br.bits(1) else lvl
                            else:
                                sym = T["ac"](br) if not (br.peek(2) == 0b11) else (br.bits(2), (0, 1))[1]
                                if sym == "EOB":
                                    assert not first, "a coded non-intra block cannot start with EOB"
                                    break
                                if sym == "ESC":
                                    run = br.bits(6)
                                    lvl = br.bits(12)
                                    lvl = lvl - 4096 if lvl & 0x800 else lvl
                                    assert lvl not in (0, -2048)
                                else:
                                    run, lvl = sym
                                    lvl = -lvl if br.bits(1) else lvl
                            first = False
                            k += run
                            assert k < 64, "coefficient index beyond 63"
                            QF[T["zz"][k]] = lvl
                            k += 1
                        F = _dequant(QF, intra, qscale, T["W"], quirks)
                        res = idct(F, quirks)
                    res = res.reshape(8, 8)
                    if b < 4:
                        oy, ox = 8 * (b >> 1), 8 * (b & 1)
                        Y[16 * row + oy:16 * row + oy + 8, 16 * col + ox:16 * col + ox + 8] = np.clip(
                            py[oy:oy + 8, ox:ox + 8] + res, 0, 255)
                    elif b == 4:
                        U[8 * row:8 * row + 8, 8 * col:8 * col + 8] = np.clip(pu + res, 0, 255)
                    else:
                        V[8 * row:8 * row + 8, 8 * col:8 * col + 8] = np.clip(pv + res, 0, 255)
                mbs.append(dict(intra=intra, mv=tuple(mv), cbp=cbp))
            # the slice must end here: only zero stuffing up to the next start code
            pad = (-br.pos) & 7
            assert br.peek(pad) == 0 if pad else True
        ref = (Y, U, V)
        out.frames.append(tuple(p.astype(np.uint8) for p in ref))
        out.mbs.append(mbs)
    assert ended
    rest = data[(br.pos + 7) >> 3:]
    assert not any(rest) and len(data) % 32 == 0, "only zero padding may follow sequence_end_code"
    return out


def _dequant(QF, intra, qscale, W, quirks):
    F = np.zeros(64, np.int64)
    if intra:
        F[0] = 2 * QF[0]                                        # intra_dc_mult for 10-bit precision, minus the 1024 offset
        for i in range(1, 64):
            v = 2 * int(QF[i]) * int(W[i]) * qscale
            F[i] = (v >> 5) if quirks else int(v / 32)          # RTL floors (>>>), ISO truncates toward zero
    else:
        for i in range(64):
            q = int(QF[i])
            v = (2 * q + (1 if q > 0 else -1 if q < 0 else 0)) * 16 * qscale
            F[i] = int(v / 32)
    if quirks:
        F = np.clip(F, -2047, 2047)
        F[0] = 2 * QF[0] if intra else F[0]
    else:
        F = np.clip(F, -2048, 2047)
        if (int(F.sum()) & 1) == 0:                             # mismatch control (7.4.4)
            F[63] += -1 if F[63] & 1 else 1
    return F


def _predict(plane, x0, y0, mvx, mvy, n, quirks):
    """half-pel prediction of an n x n block (7.6.4); mv in half-sample units of this plane"""
    ix, iy, hx, hy = mvx >> 1, mvy >> 1, mvx & 1, mvy & 1
    H, W = plane.shape
    ys = np.arange(y0 + iy, y0 + iy + n + 1)
    xs = np.arange(x0 + ix, x0 + ix + n + 1)
    assert ys[0] >= 0 and xs[0] >= 0 and ys[n - 1 + hy] < H and xs[n - 1 + hx] < W, "vector points outside the picture"
    ys, xs = np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1)
    p = plane[np.ix_(ys, xs)].astype(np.int64)
    a, b, c, d = p[:n, :n], p[:n, 1:n + 1], p[1:n + 1, :n], p[1:n + 1, 1:n + 1]
    if hx and hy:
        return (a + b + c + d + (1 if quirks else 2)) >> 2
    if hx:
        return (a + b + 1) >> 1
    if hy:
        return (a + c + 1) >> 1
    return a


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
