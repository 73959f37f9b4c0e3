"""Register-width emulation of the transform loop of RTL/mpeg2encoder.v: stage G (forward DCT + quantiser, RTL:2029-2077),
stage H (inverse quantiser, RTL:2129-2150), stages J/K/M (Chen-Wang inverse DCT, RTL:844-972, 2159-2279) and
add_clip_0_255 (RTL:786-795).

Every register of the RTL is a `Reg` with the declared width and signedness (RTL:1937-1949, 2091-2093, 2170, 2203,
2258) and every assignment goes through it, so wrap-around, truncation and sign re-interpretation happen where the
hardware does them and nowhere else.  Verilog's expression rules that matter here are spelled out where they apply:
  * an expression with an unsigned operand is unsigned as a whole, and `>>>` on it shifts in zeros (RTL:2060);
  * the bits kept afterwards decide whether that matters;
  * `reg [17:0] r` assigned from a signed expression keeps 18 bits, and the 18 bits are read back as signed (RTL:886, 2170).
This is a SECOND restatement next to oracle/m2v_oracle.c (which works on plain C ints with explicit sext() calls at the
places its author believed to matter).  The pipeline's shift registers only move values around and are not modelled.
TEST INFRASTRUCTURE ONLY.
"""

DCTM = [[64, 64, 64, 64, 64, 64, 64, 64],
        [89, 75, 50, 18, -18, -50, -75, -89],
        [84, 35, -35, -84, -84, -35, 35, 84],
        [75, -18, -89, -50, 50, 89, 18, -75],
        [64, -64, -64, 64, 64, -64, -64, 64],
        [50, -89, 18, 75, -75, -18, 89, -50],
        [35, -84, 84, -35, -35, 84, -84, 35],
        [18, -50, 75, -89, 89, -75, 50, -18]]                      # RTL:104-112 (checked against the RTL text in test_tables_vs_rtl.py)
INTRA_Q = [[8, 16, 19, 22, 26, 27, 29, 34], [16, 16, 22, 24, 27, 29, 34, 37], [19, 22, 26, 27, 29, 34, 34, 38],
           [22, 22, 26, 27, 29, 34, 37, 40], [22, 26, 27, 29, 32, 35, 40, 48], [26, 27, 29, 32, 35, 40, 48, 58],
           [26, 27, 29, 34, 38, 46, 56, 69], [27, 29, 35, 38, 46, 56, 69, 83]]                               # RTL:131-139
W1, W2, W3, W5, W6, W7 = 2841, 2676, 2408, 1609, 1108, 565      # localparam signed [16:0], RTL:169-174


def wrap(v, bits, signed):
    """the value a `bits`-wide register holds after `v` is assigned to it"""
    v &= (1 << bits) - 1
    if signed and v >> (bits - 1):
        v -= 1 << bits
    return v


def s32(v):
    return wrap(v, 32, True)


# ---------------------------------------------------------------------------------------------------------------
# stage G
# ---------------------------------------------------------------------------------------------------------------
def forward_dct(tile):
    """tile[y][x]: 9-bit signed residuals (g_tiles).  Returns g_dct_res3: 8x8 of 17-bit signed."""
    res1 = [[0] * 8 for _ in range(8)]
    for y in range(8):                                            # phase 1, one row of the tile per clock (RTL:2029-2037)
        for j in range(8):
            g_t1 = 0                                              # reg signed [18:0]
            for k in range(8):
                g_t1 = wrap(g_t1 + wrap(tile[y][k], 9, True) * DCTM[j][k], 19, True)
            res1[y][j] = g_t1                                     # g_dct_res1 / g_dct_res2: signed [18:0]
    res3 = [[0] * 8 for _ in range(8)]
    for j in range(8):                                            # phase 2, one column per clock (RTL:2054-2062)
        for i in range(8):
            g_t2 = 0                                              # reg signed [28:0]
            for k in range(8):
                g_t2 = wrap(g_t2 + DCTM[i][k] * res1[k][j], 29, True)
            # g_t2 = (g_t2 >>> 12) + g_t2[11]: the bit select is unsigned, so the whole right-hand side is an UNSIGNED
            # 29-bit expression and >>> shifts in zeros; g_dct_res3 then takes $signed(g_t2[16:0]) - bits 28..12 of the
            # sum, which the zero fill above bit 16 cannot reach
            u = g_t2 & ((1 << 29) - 1)
            g_t2 = wrap((u >> 12) + ((u >> 11) & 1), 29, True)
            res3[i][j] = wrap(g_t2, 17, True)
    return res3


def quantise(res3, inter, Q):
    """g_quant: 8x8 of 12-bit signed (RTL:2065-2077)"""
    out = [[0] * 8 for _ in range(8)]
    for i in range(8):
        for j in range(8):
            c = res3[i][j]
            g_t3 = wrap(-c if c < 0 else c, 16, False)            # reg [15:0]
            if inter:
                g_t3 = wrap(wrap(g_t3 + 2, 16, False) >> (4 + Q), 16, False)
            elif i or j:
                # integer constants make this a 32-bit unsigned expression
                g_t3 = wrap(((g_t3 + ((INTRA_Q[i][j] * ((3 << Q) + 2)) >> 3)) >> Q) // INTRA_Q[i][j], 16, False)
            else:
                g_t3 = wrap((g_t3 >> 4) + ((g_t3 >> 3) & 1), 16, False)
            if g_t3 > 2047:
                g_t3 = 2047
            m = wrap(g_t3, 12, True)                              # $signed(g_t3[11:0])
            out[i][j] = wrap(-m if c < 0 else m, 12, True)
    return out


# ---------------------------------------------------------------------------------------------------------------
# stage H
# ---------------------------------------------------------------------------------------------------------------
def dequantise(q, inter, Q):
    """h_iquant: 8x8 of 13-bit signed (RTL:2129-2150); h_t1 is a 17-bit signed temporary"""
    out = [[0] * 8 for _ in range(8)]
    for i in range(8):
        for j in range(8):
            h = wrap(q[i][j], 17, True)
            if inter:
                h = wrap(h << 1, 17, True)
                h = wrap(h + (-1 if h < 0 else 1 if h > 0 else 0), 17, True)
                h = wrap(h << Q, 17, True)
                h = -2047 if h < -2047 else 2047 if h > 2047 else h
            elif i or j:
                h = wrap(h * INTRA_Q[i][j], 17, True)             # unsigned operand: the low 17 bits of the product are the same
                h = wrap(h << (Q - 3), 17, True) if Q >= 3 else h >> (3 - Q)      # >>> on a signed-only expression: arithmetic
                h = -2047 if h < -2047 else 2047 if h > 2047 else h
            else:
                h = wrap(h << 1, 17, True)
            out[i][j] = wrap(h, 13, True)
    return out


# ---------------------------------------------------------------------------------------------------------------
# stages J / K / M
# ---------------------------------------------------------------------------------------------------------------
def _rows_step12(a):
    x0, x1, x2, x3, x4, x5, x6, x7 = (wrap(a[0], 13, True), wrap(a[4], 13, True), wrap(a[6], 13, True), wrap(a[2], 13, True),
                                      wrap(a[1], 13, True), wrap(a[7], 13, True), wrap(a[5], 13, True), wrap(a[3], 13, True))
    x0 = s32(x0 << 11)
    x1 = s32(x1 << 11)
    x0 = s32((x0 & 0xFFFFFFFF) | 128)                              # x0[7] = 1'b1
    x8 = s32(W7 * s32(x4 + x5))
    x4 = s32(x8 + s32((W1 - W7) * x4))
    x5 = s32(x8 - s32((W1 + W7) * x5))
    x8 = s32(W3 * s32(x6 + x7))
    x6 = s32(x8 - s32((W3 - W5) * x6))
    x7 = s32(x8 - s32((W3 + W5) * x7))
    x8 = s32(x0 + x1)
    x0 = s32(x0 - x1)
    x1 = s32(W6 * s32(x3 + x2))
    x2 = s32(x1 - s32((W2 + W6) * x2))
    x3 = s32(x1 + s32((W2 - W6) * x3))
    x1 = s32(x4 + x6)
    x4 = s32(x4 - x6)
    x6 = s32(x5 + x7)
    x5 = s32(x5 - x7)
    return x0, x1, x2, x3, x4, x5, x6, x7, x8


def _step3(x0, x1, x2, x3, x4, x5, x6, x7, x8):
    x7 = s32(x8 + x3)
    x8 = s32(x8 - x3)
    x3 = s32(x0 + x2)
    x0 = s32(x0 - x2)
    x2 = s32(s32(181 * s32(x4 + x5)) + 128) >> 8
    x4 = s32(s32(181 * s32(x4 - x5)) + 128) >> 8
    return x0, x1, x2, x3, x4, x5, x6, x7, x8


def _rows_step34(xs):
    x0, x1, x2, x3, x4, x5, x6, x7, x8 = _step3(*xs)
    r = [s32(x7 + x1) >> 8, s32(x3 + x2) >> 8, s32(x0 + x4) >> 8, s32(x8 + x6) >> 8,
         s32(x8 - x6) >> 8, s32(x0 - x4) >> 8, s32(x3 - x2) >> 8, s32(x7 - x1) >> 8]
    return [wrap(wrap(v, 18, False), 18, True) for v in r]        # reg [17:0] r0..r7, read back through signed [17:0] j_idct_res1


def _cols_step12(a):
    x0, x1, x2, x3, x4, x5, x6, x7 = (wrap(a[0], 18, True), wrap(a[4], 18, True), wrap(a[6], 18, True), wrap(a[2], 18, True),
                                      wrap(a[1], 18, True), wrap(a[7], 18, True), wrap(a[5], 18, True), wrap(a[3], 18, True))
    x0 = s32(x0 << 8)
    x1 = s32(x1 << 8)
    x0 = s32(x0 + 8192)
    x8 = s32(s32(W7 * s32(x4 + x5)) + 4)
    x4 = s32(x8 + s32((W1 - W7) * x4)) >> 3
    x5 = s32(x8 - s32((W1 + W7) * x5)) >> 3
    x8 = s32(s32(W3 * s32(x6 + x7)) + 4)
    x6 = s32(x8 - s32((W3 - W5) * x6)) >> 3
    x7 = s32(x8 - s32((W3 + W5) * x7)) >> 3
    x8 = s32(x0 + x1)
    x0 = s32(x0 - x1)
    x1 = s32(s32(W6 * s32(x3 + x2)) + 4)
    x2 = s32(x1 - s32((W2 + W6) * x2)) >> 3
    x3 = s32(x1 + s32((W2 - W6) * x3)) >> 3
    x1 = s32(x4 + x6)
    x4 = s32(x4 - x6)
    x6 = s32(x5 + x7)
    x5 = s32(x5 - x7)
    return x0, x1, x2, x3, x4, x5, x6, x7, x8


def _clip_neg255_pos255(x):                                       # input signed [27:0], RTL:778-783
    x = wrap(x, 28, True)
    return -255 if x < -255 else 255 if x > 255 else wrap(x, 9, True)


def _cols_step34(xs):
    x0, x1, x2, x3, x4, x5, x6, x7, x8 = _step3(*xs)
    return [_clip_neg255_pos255(s32(x7 + x1) >> 14), _clip_neg255_pos255(s32(x3 + x2) >> 14),
            _clip_neg255_pos255(s32(x0 + x4) >> 14), _clip_neg255_pos255(s32(x8 + x6) >> 14),
            _clip_neg255_pos255(s32(x8 - x6) >> 14), _clip_neg255_pos255(s32(x0 - x4) >> 14),
            _clip_neg255_pos255(s32(x3 - x2) >> 14), _clip_neg255_pos255(s32(x7 - x1) >> 14)]


def inverse_dct(iq):
    """iq: 8x8 of 13-bit signed.  Returns m_idct_res3: 8x8 of 9-bit signed"""
    res1 = [_rows_step34(_rows_step12(iq[i])) for i in range(8)]                 # one row per clock, RTL:2159-2189
    out = [[0] * 8 for _ in range(8)]
    for j in range(8):                                                            # one column per clock, RTL:2238-2279
        col = _cols_step34(_cols_step12([res1[i][j] for i in range(8)]))
        for i in range(8):
            out[i][j] = wrap(col[i], 9, True)
    return out


def add_clip_0_255(a, b):                                         # RTL:786-795: a [7:0], b signed [8:0], c signed [9:0]
    c = wrap(wrap(b, 9, True), 10, True)
    c = wrap(c + (a & 255), 10, True)
    return 255 if c > 255 else 0 if c < 0 else c & 255
