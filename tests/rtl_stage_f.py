"""Register-level emulation of stage F of RTL/mpeg2encoder.v (motion estimation FSM, RTL:1589-1918), state by state:
the window registers f_Y_ref / f_Y_tmp / f_Y_hlf / f_U_prd are numpy arrays that are shifted exactly like the RTL's
for-loops shift them, f_diff / f_over are updated per clock, CALC_MIN runs the 6-clock bit-serial elimination, the
vectors are picked by the last-assignment-wins loops, and PREDICT reads the shifted arrays.

This is a SECOND restatement next to oracle/m2v_oracle.c, which uses a DERIVED model (direct SAD sums, "samples that can
never be selected may hold anything", closed-form chroma vectors).  Here nothing is derived: positions outside the
frame are filled with random garbage by the caller, and if the derivations were wrong the two would disagree.
TEST INFRASTRUCTURE ONLY.
"""
import numpy as np


def mean2(a, b):
    return (a.astype(np.int64) + b + 1) >> 1


def mean4(a, b, c, d):
    return (a.astype(np.int64) + b + c + d + 1) >> 2          # +1, RTL:764


def find_min_in_10_values(v):                                  # RTL:804-840
    wi1 = v[1] < v[0]; w01 = v[1] if wi1 else v[0]
    wi3 = v[3] < v[2]; w23 = v[3] if wi3 else v[2]
    wi5 = v[5] < v[4]; w45 = v[5] if wi5 else v[4]
    wi7 = v[7] < v[6]; w67 = v[7] if wi7 else v[6]
    wi9 = v[9] < v[8]; w89 = v[9] if wi9 else v[8]
    xi23 = w23 < w01; x0123 = w23 if xi23 else w01
    xi67 = w67 < w45; x4567 = w67 if xi67 else w45
    if w89 <= x0123 and w89 <= x4567:
        return 8 + int(wi9)
    if x0123 < x4567:
        return (2 + int(wi3)) if xi23 else int(wi1)
    return (6 + int(wi7)) if xi67 else (4 + int(wi5))


def stage_f(Yblk, Yref, Uref, Vref, x16, y16, max_x16, max_y16, i_frame, VL):
    """Yblk: 16x16 current luma.  Yref: (16+2YR) x (16+16+YR) window = f_Y_ref[-YR:16+YR-1][-YR:31],
    Uref/Vref: (8+2UR) x (8+8+UR) = f_U_ref[-UR:8+UR-1][-UR:15].  Returns inter, mvx, mvy, Yprd, Uprd, Vprd."""
    UR, YR = VL, 2 * VL
    NY = 16 + 2 * YR
    f_Y_blk = Yblk.astype(np.int64).copy()
    # ---- PREPARE_SEARCH_FULL (RTL:1634-1647) ----
    f_Y_tmp = Yref[:, :NY].astype(np.int64).copy()             # [-YR..16+YR-1][-YR..16+YR-1]
    nc = 2 * YR + 1
    f_diff = np.zeros((nc, nc), np.int64)
    f_over = np.zeros((nc, nc), bool)
    for yi in range(nc):
        for xi in range(nc):
            y, x = yi - YR, xi - YR
            f_over[yi, xi] = (x16 == 0 and x < 0) or (x16 == max_x16 and x > 0) or (y16 == 0 and y < 0) or (y16 == max_y16 and y > 0)
    f_Y_sum = 0
    # ---- CALC_DIFF, 16 clocks (RTL:1650-1672) ----
    for _ in range(16):
        col = f_Y_blk[:, 0]
        f_Y_sum = (f_Y_sum + int(col.sum())) & 0xFFFF
        new_diff, new_over = f_diff.copy(), f_over.copy()
        for yi in range(nc):
            for xi in range(nc):
                if not f_over[yi, xi]:
                    d = int(np.abs(col - f_Y_tmp[yi:yi + 16, xi]).sum()) & 0xFFF       # rows yt+y, column x (index offset YR)
                    s = int(f_diff[yi, xi]) + d
                    new_over[yi, xi] = bool(s >> 12)
                    new_diff[yi, xi] = s & 0xFFF
        f_diff, f_over = new_diff, new_over
        f_Y_blk = np.roll(f_Y_blk, -1, axis=1)                   # cyclic left shift
        f_Y_tmp[:, :-1] = f_Y_tmp[:, 1:].copy()                  # left shift, last column keeps its value
    # ---- CALC_MIN, 6 clocks (RTL:1675-1691) ----
    for _ in range(6):
        b11, b10 = (f_diff >> 11) & 1, (f_diff >> 10) & 1
        tmpbit1 = bool(np.all(f_over | (b11 == 1)))
        tmpbit2 = bool(np.all(f_over | ((b11 == 1) & (not tmpbit1)) | (b10 == 1)))
        f_over = f_over | ((b11 == 1) & (not tmpbit1)) | ((b10 == 1) & (not tmpbit2))
        f_diff = (f_diff << 2) & 0xFFF
    # ---- CALC_MOTION_VECTOR_Y / X (RTL:1694-1715): last assignment wins ----
    f_mvy = 0
    for yi in range(nc):
        if not np.all(f_over[yi]):
            f_mvy = yi - YR
    f_mvx = 0
    for xi in range(nc):
        if not f_over[f_mvy + YR, xi]:
            f_mvx = xi - YR
    f_Y_tmp = Yref[:, :NY].astype(np.int64).copy()
    # ---- REF_SHIFT_Y, YR clocks (RTL:1719-1728); array row index r <-> y = r - YR ----
    for cnt in range(YR):
        if f_mvy > 0 and cnt < f_mvy:
            f_Y_tmp[YR - 1:NY - 1, :] = f_Y_tmp[YR:NY, :].copy()            # tmp[y-1] <= tmp[y], y = 0 .. 16+YR-1
        elif f_mvy < 0 and cnt < -f_mvy:
            f_Y_tmp[1:YR + 17, :] = f_Y_tmp[0:YR + 16, :].copy()            # tmp[y+1] <= tmp[y], y = -YR .. 15
    # ---- REF_SHIFT_X, YR clocks (RTL:1731-1740); rows y = -1 .. 16 only ----
    r0, r1 = YR - 1, YR + 17
    for cnt in range(YR):
        if f_mvx > 0 and cnt < f_mvx:
            f_Y_tmp[r0:r1, YR - 1:NY - 1] = f_Y_tmp[r0:r1, YR:NY].copy()    # tmp[y][x-1] <= tmp[y][x], x = 0 .. 16+YR-1
        elif f_mvx < 0 and cnt < -f_mvx:
            f_Y_tmp[r0:r1, 1:YR + 17] = f_Y_tmp[r0:r1, 0:YR + 16].copy()    # tmp[y][x+1] <= tmp[y][x], x = -YR .. 15

    def build_hlf():
        """f_Y_hlf[-1:31][-1:31] from f_Y_tmp (RTL:1746-1752); index i <-> array i + 1"""
        T = f_Y_tmp[YR - 1:YR + 17, YR - 1:YR + 17]                          # y, x = -1 .. 16
        h = np.zeros((33, 33), np.int64)
        a, b, c, d = T[:17, :17], T[:17, 1:18], T[1:18, :17], T[1:18, 1:18]   # (y,x), (y,x+1), (y+1,x), (y+1,x+1) for y,x = -1..15
        h[1::2, 1::2] = T[1:17, 1:17]                                        # hlf[2y][2x], y,x = 0..15
        h[1::2, 0::2] = mean2(a, b)[1:17, :]                                 # hlf[2y][2x+1], x = -1..15
        h[0::2, 1::2] = mean2(a, c)[:, 1:17]                                 # hlf[2y+1][2x], y = -1..15
        h[0::2, 0::2] = mean4(a, b, c, d)                                    # hlf[2y+1][2x+1]
        return h
    # ---- PREPARE_SEARCH_HALF (RTL:1743-1762) ----
    f_Y_mean = (f_Y_sum >> 8) & 0xFF
    f_Y_hlf = build_hlf()
    h_diff = np.zeros((3, 3), np.int64)
    h_over = np.zeros((3, 3), bool)
    for yi in range(3):
        for xi in range(3):
            y, x = yi - 1, xi - 1
            h_over[yi, xi] = ((x16 == 0 or f_mvx == -YR) and x < 0) or ((x16 == max_x16 or f_mvx == YR) and x > 0) or \
                             ((y16 == 0 or f_mvy == -YR) and y < 0) or ((y16 == max_y16 or f_mvy == YR) and y > 0)
    # ---- CALC_DIFF_HALF, 16 clocks (RTL:1765-1787) ----
    for _ in range(16):
        col = f_Y_blk[:, 0]
        f_Y_sum = (f_Y_sum + (int(np.abs(col - f_Y_mean).sum()) & 0xFFF)) & 0xFFFF
        nd, no = h_diff.copy(), h_over.copy()
        for yi in range(3):
            for xi in range(3):
                if not h_over[yi, xi]:
                    rows = (yi - 1) + 2 * np.arange(16) + 1                  # hlf[y + 2*yt][x]
                    d = int(np.abs(col - f_Y_hlf[rows, xi - 1 + 1]).sum()) & 0xFFF
                    s = int(h_diff[yi, xi]) + d
                    no[yi, xi] = bool(s >> 12)
                    nd[yi, xi] = s & 0xFFF
        h_diff, h_over = nd, no
        f_Y_blk = np.roll(f_Y_blk, -1, axis=1)
        f_Y_hlf[:, :31] = f_Y_hlf[:, 2:33].copy()                            # left shift by 2 (x = -1 .. 29)
    # ---- CALC_MIN_HALF1 (RTL:1790-1816) ----
    cost = f_Y_sum & 0xFFF if (f_Y_sum >> 12) == 0 else 0xFFF
    v = [(int(h_over[yi, xi]) << 12) | int(h_diff[yi, xi]) for yi in range(3) for xi in range(3)] + [cost]
    idx = find_min_in_10_values(v)
    if idx <= 8:
        f_mvyh, f_mvxh, f_inter = idx // 3 - 1, idx % 3 - 1, True
    else:
        f_mvyh, f_mvxh, f_inter = 0, 0, False
    # ---- CALC_MIN_HALF2 (RTL:1819-1844) ----
    if i_frame == 0:
        f_inter, f_mvyh, f_mvxh, f_mvy, f_mvx = False, 0, 0, 0, 0
    else:
        f_mvy, f_mvx = 2 * f_mvy + f_mvyh, 2 * f_mvx + f_mvxh
    f_Y_hlf = build_hlf()
    NU = 8 + 2 * UR
    f_U_prd, f_V_prd = Uref[:, :NU].astype(np.int64).copy(), Vref[:, :NU].astype(np.int64).copy()
    # ---- REF_UV_SHIFT_Y, 3 clocks (RTL:1847-1866) ----
    for cnt in range(3):
        if (cnt == 0 and f_mvyh >= 0) or (cnt == 1 and f_mvyh >= 1):
            f_Y_hlf[0:32, :] = f_Y_hlf[1:33, :].copy()                       # up shift
        if f_mvy > 0 and cnt < (f_mvy >> 2):
            for P in (f_U_prd, f_V_prd):
                P[UR:NU - 1, :] = P[UR + 1:NU, :].copy()                     # prd[y-1] <= prd[y], y = 1 .. 8+UR-1
        elif f_mvy < 0 and cnt < -(f_mvy >> 2):
            for P in (f_U_prd, f_V_prd):
                P[1:UR + 9, :] = P[0:UR + 8, :].copy()                       # prd[y+1] <= prd[y], y = -UR .. 7
    # ---- REF_UV_SHIFT_X, 3 clocks (RTL:1869-1888) ----
    for cnt in range(3):
        if (cnt == 0 and f_mvxh >= 0) or (cnt == 1 and f_mvxh >= 1):
            f_Y_hlf[0:31, 0:32] = f_Y_hlf[0:31, 1:33].copy()                 # left shift, rows -1 .. 29
        if f_mvx > 0 and cnt < (f_mvx >> 2):
            for P in (f_U_prd, f_V_prd):
                P[UR:UR + 9, UR:NU - 1] = P[UR:UR + 9, UR + 1:NU].copy()     # rows 0..8: prd[y][x-1] <= prd[y][x], x = 1 .. 8+UR-1
        elif f_mvx < 0 and cnt < -(f_mvx >> 2):
            for P in (f_U_prd, f_V_prd):
                P[UR:UR + 9, 1:UR + 9] = P[UR:UR + 9, 0:UR + 8].copy()       # prd[y][x+1] <= prd[y][x], x = -UR .. 7
    # ---- PREDICT (RTL:1891-1917) ----
    if not f_inter:
        return False, 0, 0, np.full((16, 16), 128), np.full((8, 8), 128), np.full((8, 8), 128), (f_mvx, f_mvy)
    Yprd = f_Y_hlf[0:32:2, 0:32:2].copy()                                    # hlf[2y-1][2x-1]
    fy, fx = (f_mvy >> 1) & 1, (f_mvx >> 1) & 1
    out = []
    for P in (f_U_prd, f_V_prd):
        a, b = P[UR:UR + 8, UR:UR + 8], P[UR:UR + 8, UR + 1:UR + 9]
        c, d = P[UR + 1:UR + 9, UR:UR + 8], P[UR + 1:UR + 9, UR + 1:UR + 9]
        if fy and fx:
            out.append(mean4(a, b, c, d))
        elif fx:
            out.append(mean2(a, b))
        elif fy:
            out.append(mean2(a, c))
        else:
            out.append(a.copy())
    return True, f_mvx, f_mvy, Yprd, out[0], out[1], (f_mvx, f_mvy)
