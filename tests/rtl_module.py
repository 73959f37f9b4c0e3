"""Clock-by-clock model of the WHOLE module RTL/mpeg2encoder.v, from i_en beats to o_data words: the per-stage
emulations of this directory chained through every register that decides WHEN something happens.

What is modelled register by register, with the RTL's non-blocking semantics (every right-hand side sees the values of
the previous clock; memories are read before they are written):
  A      sequence FSM, beat counters, horizontal chroma mean                         RTL:1027-1095
  B, C   line buffer (read-before-write), vertical chroma mean                       RTL:1105-1171
  D, E   slice double buffer mem_dbuf_Y/U/V, flip / flop, column-first read-out, the shifting
         16x16 / 8x8 block registers with their combinational last entries          RTL:1177-1304
  X,Y,Z  reference prefetch of the NEXT block from mem_ref_Y / mem_ref_UV (wrapping row address,
         frame wrap at the bottom-right block), the serpentine z_*_ref shift registers RTL:1314-1425
  F      the motion-estimation FSM with its exact cycle counts; f_Y_ref / f_U_ref / f_V_ref shift
         and load on e_en_blk; the arithmetic inside one block is tests/rtl_stage_f.py  RTL:1462-1918
  G      g_cnt, the block registers latched on f_en_blk, g_en_tile / g_num_tile; the arithmetic of
         one tile (DCT + quantiser) is tests/rtl_stage_gm.py                         RTL:1928-2077
  H..M   inverse quantiser, Chen-Wang row / column passes as the RTL pipelines them, including the
         enables that only update while the previous stage is enabled (j_en_tile, m_idct_en3)  RTL:2085-2279
  N, P   prediction + residual, one tile row per clock, line counter                 RTL:2286-2357
  Q, R   mem_delay (the same address is read and written in one clock), write-back into
         mem_ref_Y / mem_ref_UV one slice up                                         RTL:2364-2424
  S      zig-zag reorder, coded flags                                                RTL:2434-2468
  T,U,V  tests/rtl_stage_tuv.py (clock-level model of the entropy coder and the packer), started on
         s_en_blk; T must be back in PUT_IDLE when the next block arrives           RTL:2480-2994
Every memory and every data register starts as RANDOM GARBAGE, so anything the RTL reads before it wrote it shows up
as a difference from the oracle (oracle/m2v_oracle.c derives "ref(f+1) = recon(f)" instead of modelling these
memories: SURVEY.md 3.4).  This is a reading of the RTL, not a simulation of it: no Verilog simulator exists in this
image.  TEST INFRASTRUCTURE ONLY.
"""
import numpy as np

import rtl_stage_gm as gm
from rtl_stage_f import stage_f
from rtl_stage_tuv import StageTUV

IDLE, DURING, ENDING, ENDED = 0, 1, 2, 3
(MV_IDLE, PREPARE_SEARCH_FULL, CALC_DIFF, CALC_MIN, CALC_MOTION_VECTOR_Y, CALC_MOTION_VECTOR_X, REF_SHIFT_Y, REF_SHIFT_X,
 PREPARE_SEARCH_HALF, CALC_DIFF_HALF, CALC_MIN_HALF1, CALC_MIN_HALF2, REF_UV_SHIFT_Y, REF_UV_SHIFT_X, PREDICT) = range(15)
ZIGZAG = [[0, 1, 5, 6, 14, 15, 27, 28], [2, 4, 7, 13, 16, 26, 29, 42], [3, 8, 12, 17, 25, 30, 41, 43], [9, 11, 18, 24, 31, 40, 44, 53],
          [10, 19, 23, 32, 39, 45, 52, 54], [20, 22, 33, 38, 46, 51, 55, 60], [21, 34, 37, 47, 50, 56, 59, 61],
          [35, 36, 48, 49, 57, 58, 62, 63]]                                            # RTL:156-163


def mean2(a, b):
    return (a + b + 1) >> 1


class CountingTUV(StageTUV):
    def __init__(self, *a):
        super().__init__(*a)
        self.clocks = 0

    def clock(self, *a, **k):
        self.clocks += 1
        return super().clock(*a, **k)


class RtlModule:
    def __init__(self, XL, YL, VL, Q, seed=1, sabotage=None):
        """sabotage: None, or one deliberate deviation from the RTL that the tests use to show that the comparison with the
        oracle notices it: "same_slice" (write-back into the block's own slice instead of one slice up, RTL:2407),
        "no_delay" (the reconstruction bypasses mem_delay, RTL:2389), "no_frame_wrap" (the bottom-right block prefetches
        the row below the frame instead of the next frame's top-left block, RTL:1351-1353), "late_prefetch" (the prefetch
        of a block starts when the block itself starts, one block late, RTL:1350)."""
        self.sabotage = sabotage
        self.XL, self.YL, self.VL, self.Q = XL, YL, VL, Q
        self.UR, self.YR = VL, 2 * VL
        rng = self.rng = np.random.default_rng(seed)
        XS, YS = 16 << XL, 16 << YL
        self.XS, self.YS = XS, YS
        g8 = lambda *shape: rng.integers(0, 256, shape).astype(np.int64)       # noqa: E731  garbage bytes
        YR, UR = self.YR, self.UR
        # ---- A ----
        self.pframes_count = 0
        self.max_x16 = self.max_y16 = 0
        self.a_i_frame = self.a_x4 = self.a_y = 0
        self.a_en = 0
        self.a_Y, self.a_U, self.a_V = [0] * 4, [0x80, 0x80], [0x80, 0x80]
        self.sequence_start = 0
        self.sequence_state = IDLE
        # ---- B, C ----
        self.mem_lbuf_U, self.mem_lbuf_V = g8(XS // 4, 2), g8(XS // 4, 2)
        self.b_i_frame = self.b_x4 = self.b_y = self.b_en = 0
        self.b_Y, self.b_U, self.b_V, self.b_Uu, self.b_Vu = g8(4), g8(2), g8(2), g8(2), g8(2)
        self.c_i_frame = self.c_x4 = self.c_y = self.c_en = 0
        self.c_Y, self.c_U, self.c_V = g8(4), g8(2), g8(2)
        # ---- D, E ----
        self.mem_dbuf_Y, self.mem_dbuf_U, self.mem_dbuf_V = g8(2, 16, XS // 4, 4), g8(2, 8, XS // 4, 2), g8(2, 8, XS // 4, 2)
        self.c_flip = self.d_flop = 0
        self.d_i_frame = self.d_x4 = self.d_y16 = self.d_y_16 = 0
        self.e_i_frame = self.e_x16 = self.e_y16 = 0
        self.e_start_blk = self.e_en_blk = self.e_Y_en = self.e_UV_en = 0
        self.e_Y_rd, self.e_U_rd, self.e_V_rd = g8(4), g8(2), g8(2)
        self.e_Y_blk, self.e_U_blk, self.e_V_blk = g8(16, 16), g8(8, 8), g8(8, 8)    # [15][12:16], [7][6:8] are wires
        # ---- X, Y, Z ----
        self.mem_ref_Y, self.mem_ref_UV = g8(YS, XS // 8, 8), g8(YS // 2, XS // 8, 8)
        self.x_cnt, self.x_x16, self.x_x8_2, self.x_y = 0x1F, 0, 0, 0
        self.y_Y_en = self.y_U_en = self.y_V_en = 0
        self.z_Y_en = self.z_U_en = self.z_V_en = 0
        self.y_Y_rd, self.y_UV_rd, self.z_Y_rd, self.z_UV_rd = g8(8), g8(8), g8(8), g8(8)
        self.z_Y_ref, self.z_U_ref, self.z_V_ref = g8(16 + 2 * YR, 16), g8(8 + 2 * UR, 8), g8(8 + 2 * UR, 8)
        # ---- F ----
        self.f_stat, self.f_cnt, self.f_en_blk = MV_IDLE, 0, 0
        self.f_i_frame = self.f_x16 = self.f_y16 = 0
        self.f_Y_blk, self.f_U_blk, self.f_V_blk = g8(16, 16), g8(8, 8), g8(8, 8)
        self.f_Y_ref, self.f_U_ref, self.f_V_ref = g8(16 + 2 * YR, 32 + YR), g8(8 + 2 * UR, 16 + UR), g8(8 + 2 * UR, 16 + UR)
        self.f_inter, self.f_mvx, self.f_mvy = 0, 0, 0
        self.f_Y_prd, self.f_U_prd, self.f_V_prd = g8(16, 16), g8(8, 8), g8(8, 8)
        self.f_spacing = 1 << 30
        # ---- G ----
        self.g_cnt, self.g_en_tile, self.g_num_tile = 0, 0, 0
        self.g_i_frame = self.g_x16 = self.g_y16 = self.g_inter = self.g_mvx = self.g_mvy = 0
        self.g_tiles_prd = g8(48, 8)
        self.g_res = g8(48, 8) - 128                                   # the residual tiles as latched on f_en_blk
        self.g_quant = [[int(v) for v in row] for row in g8(8, 8) - 128]
        # ---- H .. M ----
        self.h_num_tile = self.h_en = self.h_cnt = 0
        self.h_iquant = [[int(v) for v in row] for row in g8(8, 8)]
        self.j1_en = self.j1_num_tile = self.j1_en_tile = 0
        self.j1_x = tuple(int(v) for v in g8(9))
        self.j_num_tile = self.j_en_tile = 0
        self.j_idct_res1 = [[int(v) for v in row] for row in g8(8, 8)]
        self.k_num_tile = self.k_en = self.k_cnt = 0
        self.k_idct_res2 = [[int(v) for v in row] for row in g8(8, 8)]
        self.m1_en = self.m1_idct_en3 = self.m1_num_tile = 0
        self.m1_x = tuple(int(v) for v in g8(9))
        self.m_idct_en3 = self.m_num_tile = 0
        self.m_idct_res3 = [[int(v) for v in row] for row in g8(8, 8) - 128]
        # ---- N, P ----
        self.n_x16 = self.n_y16 = self.n_num_tiles_line = 0
        self.n_tiles_prd = g8(48, 8)
        self.n_idct_res4 = [[int(v) for v in row] for row in g8(8, 8) - 128]
        self.n_en = self.n_cnt = 0
        self.p_wdata = g8(8)
        self.p_en = self.p_x16 = self.p_y16 = self.p_num_tiles_line = 0
        # ---- Q, R ----
        self.mem_delay = g8(64, XS // 16, 8)       # the RTL declares 48 lines; the line counter is 6 bits wide
        self.q_rd, self.r_rd = g8(8), g8(8)
        self.q_en = self.q_x16 = self.q_y16 = self.q_num_tiles_line = 0
        self.r_en = self.r_x16 = self.r_y16 = self.r_num_tiles_line = 0
        # ---- S ----
        self.s_nzflags = int(rng.integers(0, 64))
        self.s_zig_blk = [[int(v) for v in row] for row in g8(6, 64) - 128]
        self.s_en_blk = 0
        # ---- T, U, V ----
        self.tuv = None
        self.t_ended = True                     # t_stat == PUT_ENDED
        self.t_idle_at = 0                      # first cycle at which t_stat is PUT_IDLE again
        self.t_x16 = self.t_y16 = 0
        self.o_last_at = -1
        self.cycle = 0
        self.out = bytearray()
        self.out_done = False
        self.blocks = self.inter_blocks = 0

    # ------------------------------------------------------------------------------------------------------------
    def clock(self, i_en=0, iY=None, iU=None, iV=None, i_stop=0, xsize16=0, ysize16=0, pframes=0):
        """one rising edge.  Returns (o_en words appended this clock are in self.out), o_last of THIS clock (before the edge)"""
        s = self
        YR, UR = s.YR, s.UR
        max_x4, max_y = 4 * (s.max_x16 + 1) - 1, 16 * (s.max_y16 + 1) - 1
        o_last = 1 if s.cycle == s.o_last_at else 0
        wr = []                                                        # deferred memory writes: every read of this clock sees the old content

        # ================= T / U / V =================
        if s.t_ended:
            if s.sequence_start:
                s.tuv = CountingTUV(16 * (s.max_x16 + 1), 16 * (s.max_y16 + 1), s.Q)
                s.tuv.sequence_start()
                s.t_ended = False
                s.t_idle_at = s.cycle + 2                               # PUT_SEQ_HEADER2, then PUT_IDLE
        elif s.cycle >= s.t_idle_at:                                    # t_stat == PUT_IDLE
            if s.t_y16 == s.max_y16 and s.t_x16 == s.max_x16 and s.sequence_state == ENDED:
                s.out = bytearray(s.tuv.sequence_end())
                s.t_ended = True
                s.o_last_at = s.cycle + 4                               # t_end_seq, u_end_seq1, u_end_seq2, v_last
            elif s.s_en_blk:
                before = s.tuv.clocks
                s.tuv.macroblock(s.g_i_frame, s.g_x16, s.g_y16, bool(s.g_inter), s.g_mvx, s.g_mvy, s.s_nzflags,
                                 [list(z) for z in s.s_zig_blk])
                s.t_idle_at = s.cycle + (s.tuv.clocks - before)
                s.t_x16, s.t_y16 = s.g_x16, s.g_y16
        else:
            assert not s.s_en_blk, "cycle %d: a block reaches stage T while it is still busy with the previous one" % s.cycle

        # ================= S =================
        n_s_en_blk = 1 if (s.g_en_tile and s.g_num_tile == 5) else 0
        if s.g_en_tile:
            nz = 0 if s.g_inter else 1
            newt = [0] * 64
            for i in range(8):
                for j in range(8):
                    newt[ZIGZAG[i][j]] = s.g_quant[i][j]
                    nz |= 1 if s.g_quant[i][j] != 0 else 0
            s.s_zig_blk = s.s_zig_blk[1:] + [newt]
            s.s_nzflags = ((s.s_nzflags << 1) | nz) & 63
        s.s_en_blk = n_s_en_blk

        # ================= R: write-back, one slice up =================
        if s.r_en:
            ntl = s.r_num_tiles_line
            item = s.r_x16 * 2 + ((ntl >> 3) & 1)
            if not (ntl >> 5) & 1:
                wr.append((s.mem_ref_Y, (s.r_y16 * 16 + ((ntl >> 4) & 1) * 8 + (ntl & 7), item), s.r_rd.copy()))
            else:
                wr.append((s.mem_ref_UV, (s.r_y16 * 8 + (ntl & 7), item), s.r_rd.copy()))
        up = s.q_y16 if s.sabotage == "same_slice" else (s.max_y16 if s.q_y16 == 0 else s.q_y16 - 1)
        n_r = (s.q_en, s.q_x16, up, s.q_num_tiles_line, s.q_rd.copy())
        # ================= Q =================
        n_q = (s.p_en, s.p_x16, s.p_y16, s.p_num_tiles_line,
               (s.p_wdata if s.sabotage == "no_delay" else s.mem_delay[s.p_num_tiles_line, s.p_x16]).copy())
        if s.p_en:
            wr.append((s.mem_delay, (s.p_num_tiles_line, s.p_x16), s.p_wdata.copy()))
        s.r_en, s.r_x16, s.r_y16, s.r_num_tiles_line, s.r_rd = n_r
        s.q_en, s.q_x16, s.q_y16, s.q_num_tiles_line, s.q_rd = n_q

        # ================= P, N =================
        if s.n_en:
            s.p_wdata = np.array([gm.add_clip_0_255(int(s.n_tiles_prd[0][x]), s.n_idct_res4[0][x]) for x in range(8)], np.int64)
            s.p_x16, s.p_y16, s.p_num_tiles_line = s.n_x16, s.n_y16, s.n_num_tiles_line
        s.p_en = s.n_en
        first = s.m_idct_en3 and s.m_num_tile == 0
        if first:
            n_tiles_prd = s.g_tiles_prd.copy()
            n_x16, n_y16, n_ntl = s.g_x16, s.g_y16, 0
        elif s.n_en:
            n_tiles_prd = np.vstack([s.n_tiles_prd[1:], s.n_tiles_prd[47:48]])      # row 47 keeps its value
            n_x16, n_y16, n_ntl = s.n_x16, s.n_y16, (s.n_num_tiles_line + 1) & 63
        else:
            n_tiles_prd, n_x16, n_y16, n_ntl = s.n_tiles_prd, s.n_x16, s.n_y16, s.n_num_tiles_line
        if s.m_idct_en3:
            n_res4 = [row[:] for row in s.m_idct_res3]
            n_n_en, n_n_cnt = 1, 0
        else:
            n_res4 = s.n_idct_res4[1:] + [[0] * 8]
            n_n_cnt = (s.n_cnt + 1) & 7
            n_n_en = 0 if s.n_cnt == 7 else s.n_en
        s.n_tiles_prd, s.n_x16, s.n_y16, s.n_num_tiles_line = n_tiles_prd, n_x16, n_y16, n_ntl
        s.n_idct_res4, s.n_en, s.n_cnt = n_res4, n_n_en, n_n_cnt

        # ================= M (updates only while m1_en) =================
        if s.m1_en:
            col = gm._cols_step34(s.m1_x)
            s.m_idct_res3 = [s.m_idct_res3[i][1:] + [gm.wrap(col[i], 9, True)] for i in range(8)]
            s.m_idct_en3, s.m_num_tile = s.m1_idct_en3, s.m1_num_tile
        # ================= M1, K =================
        n_m1_idct_en3, n_m1_num_tile = 0, s.m1_num_tile
        if s.k_en and s.k_cnt == 7:
            n_m1_idct_en3, n_m1_num_tile = 1, s.k_num_tile
        n_m1_x = gm._cols_step12([s.k_idct_res2[i][0] for i in range(8)]) if s.k_en else s.m1_x
        n_m1_en = s.k_en
        if s.j_en_tile:
            n_k = (s.j_num_tile, 1, 0, [row[:] for row in s.j_idct_res1])
        else:
            n_k = (s.k_num_tile, 0 if s.k_cnt == 7 else s.k_en, (s.k_cnt + 1) & 7, [row[1:] + [0] for row in s.k_idct_res2])
        s.m1_en, s.m1_idct_en3, s.m1_num_tile, s.m1_x = n_m1_en, n_m1_idct_en3, n_m1_num_tile, n_m1_x
        s.k_num_tile, s.k_en, s.k_cnt, s.k_idct_res2 = n_k
        # ================= J (updates only while j1_en) =================
        if s.j1_en:
            s.j_idct_res1 = s.j_idct_res1[1:] + [gm._rows_step34(s.j1_x)]
            s.j_num_tile, s.j_en_tile = s.j1_num_tile, s.j1_en_tile
        # ================= J1, H =================
        n_j1_en_tile, n_j1_num_tile = 0, s.j1_num_tile
        if s.h_en and s.h_cnt == 7:
            n_j1_en_tile, n_j1_num_tile = 1, s.h_num_tile
        n_j1_x = gm._rows_step12(s.h_iquant[0]) if s.h_en else s.j1_x
        n_j1_en = s.h_en
        if s.g_en_tile:
            n_h = (1, 0, s.g_num_tile, gm.dequantise(s.g_quant, bool(s.g_inter), s.Q))
        else:
            n_h = (0 if s.h_cnt == 7 else s.h_en, (s.h_cnt + 1) & 7, s.h_num_tile, s.h_iquant[1:] + [[0] * 8])
        s.j1_en, s.j1_en_tile, s.j1_num_tile, s.j1_x = n_j1_en, n_j1_en_tile, n_j1_num_tile, n_j1_x
        s.h_en, s.h_cnt, s.h_num_tile, s.h_iquant = n_h

        # ================= G =================
        n_g_en_tile, n_g_num_tile = 0, s.g_num_tile
        if s.g_cnt in (18, 26, 34, 42, 50, 58):
            t = (s.g_cnt >> 3) - 2
            n_g_en_tile, n_g_num_tile = 1, t
            tile = [[int(v) for v in row] for row in s.g_res[8 * t:8 * t + 8]]
            s.g_quant = gm.quantise(gm.forward_dct(tile), bool(s.g_inter), s.Q)
        if s.f_en_blk:
            assert s.g_cnt == 0 or s.g_cnt > 58, "cycle %d: f_en_blk while stage G is still transforming (g_cnt %d)" % (s.cycle, s.g_cnt)
            n_g_cnt = 1
            s.blocks += 1
            s.inter_blocks += int(s.f_inter)
            s.g_i_frame, s.g_x16, s.g_y16 = s.f_i_frame, s.f_x16, s.f_y16
            s.g_inter, s.g_mvx, s.g_mvy = s.f_inter, s.f_mvx, s.f_mvy
            prd = np.zeros((48, 8), np.int64)
            blk = np.zeros((48, 8), np.int64)
            for t, (y0, x0) in enumerate(((0, 0), (0, 8), (8, 0), (8, 8))):
                prd[8 * t:8 * t + 8] = s.f_Y_prd[y0:y0 + 8, x0:x0 + 8]
                blk[8 * t:8 * t + 8] = s.f_Y_blk[y0:y0 + 8, x0:x0 + 8]
            prd[32:40], prd[40:48] = s.f_U_prd, s.f_V_prd
            blk[32:40], blk[40:48] = s.f_U_blk, s.f_V_blk
            s.g_tiles_prd, s.g_res = prd, blk - prd
        else:
            n_g_cnt = (s.g_cnt + 1) & 63 if s.g_cnt != 0 else 0
        s.g_cnt, s.g_en_tile, s.g_num_tile = n_g_cnt, n_g_en_tile, n_g_num_tile

        # ================= F =================
        st, cnt = s.f_stat, s.f_cnt
        n_stat, n_cnt, n_f_en_blk = st, 0, 0
        e_Y_blk = s.e_Y_blk.copy(); e_Y_blk[15, 12:16] = s.e_Y_rd
        e_U_blk = s.e_U_blk.copy(); e_U_blk[7, 6:8] = s.e_U_rd
        e_V_blk = s.e_V_blk.copy(); e_V_blk[7, 6:8] = s.e_V_rd
        s.f_spacing += 1
        if st == MV_IDLE:
            s.f_Y_blk, s.f_U_blk, s.f_V_blk = e_Y_blk, e_U_blk, e_V_blk
            if s.e_en_blk:
                n_stat = PREPARE_SEARCH_FULL
                s.f_i_frame, s.f_x16, s.f_y16 = s.e_i_frame, s.e_x16, s.e_y16
                s.f_Y_ref = np.hstack([s.f_Y_ref[:, 16:], s.z_Y_ref])          # left shift by 16, new reference on the right
                s.f_U_ref = np.hstack([s.f_U_ref[:, 8:], s.z_U_ref])
                s.f_V_ref = np.hstack([s.f_V_ref[:, 8:], s.z_V_ref])
        else:
            assert not s.e_en_blk, "cycle %d: a block is complete in stage E while stage F is busy (state %d)" % (s.cycle, st)
            if st == PREPARE_SEARCH_FULL:
                n_stat = CALC_DIFF
            elif st == CALC_DIFF:
                n_stat, n_cnt = (CALC_DIFF, cnt + 1) if cnt < 15 else (CALC_MIN, 0)
            elif st == CALC_MIN:
                n_stat, n_cnt = (CALC_MIN, cnt + 1) if cnt < 5 else (CALC_MOTION_VECTOR_Y, 0)
            elif st == CALC_MOTION_VECTOR_Y:
                n_stat = CALC_MOTION_VECTOR_X
            elif st == CALC_MOTION_VECTOR_X:
                n_stat = REF_SHIFT_Y
            elif st == REF_SHIFT_Y:
                n_stat, n_cnt = (REF_SHIFT_Y, cnt + 1) if cnt < YR - 1 else (REF_SHIFT_X, 0)
            elif st == REF_SHIFT_X:
                n_stat, n_cnt = (REF_SHIFT_X, cnt + 1) if cnt < YR - 1 else (PREPARE_SEARCH_HALF, 0)
            elif st == PREPARE_SEARCH_HALF:
                n_stat = CALC_DIFF_HALF
            elif st == CALC_DIFF_HALF:
                n_stat, n_cnt = (CALC_DIFF_HALF, cnt + 1) if cnt < 15 else (CALC_MIN_HALF1, 0)
            elif st == CALC_MIN_HALF1:
                n_stat = CALC_MIN_HALF2
            elif st == CALC_MIN_HALF2:
                n_stat = REF_UV_SHIFT_Y
            elif st == REF_UV_SHIFT_Y:
                n_stat, n_cnt = (REF_UV_SHIFT_Y, cnt + 1) if cnt < 2 else (REF_UV_SHIFT_X, 0)
            elif st == REF_UV_SHIFT_X:
                n_stat, n_cnt = (REF_UV_SHIFT_X, cnt + 1) if cnt < 2 else (PREDICT, 0)
            else:                                                        # PREDICT: the block's results are complete
                n_stat, n_f_en_blk = MV_IDLE, 1
                inter, mvx, mvy, Yp, Up, Vp, _ = stage_f(s.f_Y_blk, s.f_Y_ref, s.f_U_ref, s.f_V_ref, s.f_x16, s.f_y16,
                                                         s.max_x16, s.max_y16, s.f_i_frame, s.VL)
                s.f_inter, s.f_mvx, s.f_mvy = int(inter), int(mvx), int(mvy)
                s.f_Y_prd, s.f_U_prd, s.f_V_prd = np.asarray(Yp, np.int64), np.asarray(Up, np.int64), np.asarray(Vp, np.int64)
        s.f_stat, s.f_cnt, s.f_en_blk = n_stat, n_cnt, n_f_en_blk

        # ================= Z, Y, X: reference prefetch =================
        if s.z_Y_en:
            z = s.z_Y_ref
            nz = np.empty_like(z)
            nz[:, 0:8] = z[:, 8:16]
            nz[:-1, 8:16] = z[1:, 0:8]
            nz[-1, 8:16] = s.z_Y_rd
            s.z_Y_ref = nz
        if s.z_U_en:
            s.z_U_ref = np.vstack([s.z_U_ref[1:], s.z_UV_rd[None, :]])
        if s.z_V_en:
            s.z_V_ref = np.vstack([s.z_V_ref[1:], s.z_UV_rd[None, :]])
        s.z_Y_en, s.z_U_en, s.z_V_en = s.y_Y_en, s.y_U_en, s.y_V_en
        s.z_Y_rd, s.z_UV_rd = s.y_Y_rd, s.y_UV_rd
        s.y_Y_rd = s.mem_ref_Y[s.x_y % s.YS, s.x_x16 * 2 + s.x_x8_2].copy()
        s.y_UV_rd = s.mem_ref_UV[(s.x_y >> 1) % (s.YS // 2), s.x_x16 * 2 + s.x_x8_2].copy()
        n_y = (0, 0, 0)
        if s.e_start_blk:
            if s.sabotage == "late_prefetch":
                s.x_x16, y16 = s.e_x16, s.e_y16
            elif s.e_y16 == s.max_y16 and s.e_x16 == s.max_x16 and s.sabotage != "no_frame_wrap":
                s.x_x16, y16 = 0, 0                                      # the next frame's top-left block
            elif s.e_x16 == s.max_x16:
                s.x_x16, y16 = 0, s.e_y16 + 1
            else:
                s.x_x16, y16 = s.e_x16 + 1, s.e_y16
            s.x_y = ((y16 << 4) - YR) & (s.YS - 1)                       # reg [YB-1:0]: rows above the frame wrap around
            s.x_x8_2, s.x_cnt = 0, 0
        elif s.x_cnt < 16 + 2 * YR:
            n_y = (1, (1 - (s.x_y & 1)) & (1 - s.x_x8_2), (1 - (s.x_y & 1)) & s.x_x8_2)
            if s.x_x8_2:
                s.x_cnt += 1
                s.x_y = (s.x_y + 1) & (s.YS - 1)
            s.x_x8_2 ^= 1
        s.y_Y_en, s.y_U_en, s.y_V_en = n_y

        # ================= E =================
        avail = s.c_flip != s.d_flop
        if s.e_Y_en:
            b = s.e_Y_blk.copy(); b[15, 12:16] = s.e_Y_rd
            nb = np.empty_like(b)
            nb[:15] = b[1:]
            nb[15, :12] = b[0, 4:16]
            s.e_Y_blk = nb                                               # [15][12:16] of the new state is the wire again
        if s.e_UV_en:
            for name, rd in (("e_U_blk", s.e_U_rd), ("e_V_blk", s.e_V_rd)):
                b = getattr(s, name).copy(); b[7, 6:8] = rd
                nb = np.empty_like(b)
                nb[:7] = b[1:]
                nb[7, :6] = b[0, 2:8]
                setattr(s, name, nb)
        s.e_Y_rd = s.mem_dbuf_Y[s.d_flop, s.d_y_16, s.d_x4].copy()
        s.e_U_rd = s.mem_dbuf_U[s.d_flop, s.d_y_16 >> 1, s.d_x4].copy()
        s.e_V_rd = s.mem_dbuf_V[s.d_flop, s.d_y_16 >> 1, s.d_x4].copy()
        s.e_i_frame, s.e_x16, s.e_y16 = s.d_i_frame, s.d_x4 >> 2, s.d_y16
        s.e_start_blk = 1 if (avail and (s.d_x4 & 3) == 0 and s.d_y_16 == 0) else 0
        s.e_en_blk = 1 if (avail and (s.d_x4 & 3) == 3 and s.d_y_16 == 15) else 0
        s.e_Y_en = 1 if avail else 0
        s.e_UV_en = 1 if (avail and (s.d_y_16 & 1)) else 0
        # ================= D =================
        if avail:
            if s.d_y_16 == 15:
                if s.d_x4 < max_x4:
                    s.d_x4 += 1
                else:
                    s.d_x4 = 0
                    s.d_flop ^= 1
            s.d_y_16 = (s.d_y_16 + 1) & 15
        if s.c_en:
            wr.append((s.mem_dbuf_Y, (s.c_flip, s.c_y & 15, s.c_x4), s.c_Y.copy()))
            if s.c_y & 1:
                wr.append((s.mem_dbuf_U, (s.c_flip, (s.c_y >> 1) & 7, s.c_x4), s.c_U.copy()))
                wr.append((s.mem_dbuf_V, (s.c_flip, (s.c_y >> 1) & 7, s.c_x4), s.c_V.copy()))
            if s.c_x4 == max_x4 and (s.c_y & 15) == 15:
                s.d_i_frame, s.d_y16 = s.c_i_frame, s.c_y >> 4
                s.c_flip ^= 1
        # ================= C, B =================
        s.c_i_frame, s.c_x4, s.c_y, s.c_en = s.b_i_frame, s.b_x4, s.b_y, s.b_en
        s.c_Y = s.b_Y.copy()
        s.c_U = mean2(s.b_U, s.b_Uu)
        s.c_V = mean2(s.b_V, s.b_Vu)
        s.b_i_frame, s.b_x4, s.b_y, s.b_en = s.a_i_frame, s.a_x4, s.a_y, s.a_en
        s.b_Y, s.b_U, s.b_V = np.array(s.a_Y, np.int64), np.array(s.a_U, np.int64), np.array(s.a_V, np.int64)
        s.b_Uu, s.b_Vu = s.mem_lbuf_U[s.a_x4].copy(), s.mem_lbuf_V[s.a_x4].copy()
        if s.a_en:
            wr.append((s.mem_lbuf_U, (s.a_x4,), np.array(s.a_U, np.int64)))
            wr.append((s.mem_lbuf_V, (s.a_x4,), np.array(s.a_V, np.int64)))
        # ================= A =================
        n_seq_start, n_a_en = 0, 0
        n_Y, n_U, n_V = [0] * 4, [0x80, 0x80], [0x80, 0x80]
        if s.sequence_state == ENDED:
            if o_last:
                s.sequence_state = IDLE
        elif s.sequence_state == ENDING:
            if s.a_x4 < max_x4:
                s.a_x4 += 1
                n_a_en = 1
            elif s.a_y < max_y:
                s.a_x4 = 0
                s.a_y += 1
                n_a_en = 1
            else:
                s.sequence_state = ENDED
        elif i_en:
            if s.sequence_state == IDLE:
                s.sequence_state = DURING
                n_seq_start = 1
                s.pframes_count = pframes & 0xFF
                lim_x, lim_y = 1 << s.XL, 1 << s.YL
                xs, ys = xsize16 & ((2 << s.XL) - 1), ysize16 & ((2 << s.YL) - 1)
                s.max_x16 = lim_x - 1 if xs > lim_x else 3 if xs < 4 else xs - 1
                s.max_y16 = lim_y - 1 if ys > lim_y else 3 if ys < 4 else ys - 1
                s.a_x4 = s.a_y = 0
                s.a_i_frame = 0
            else:
                if s.a_x4 < max_x4:
                    s.a_x4 += 1
                else:
                    s.a_x4 = 0
                    if s.a_y < max_y:
                        s.a_y += 1
                    else:
                        s.a_y = 0
                        s.a_i_frame = s.a_i_frame + 1 if s.a_i_frame < s.pframes_count else 0
            if i_stop:
                s.sequence_state = ENDING
            n_a_en = 1
            n_Y = [int(v) for v in iY]
            n_U = [mean2(int(iU[0]), int(iU[1])), mean2(int(iU[2]), int(iU[3]))]
            n_V = [mean2(int(iV[0]), int(iV[1])), mean2(int(iV[2]), int(iV[3]))]
        elif i_stop and s.sequence_state == DURING:
            s.sequence_state = ENDING
        s.sequence_start, s.a_en, s.a_Y, s.a_U, s.a_V = n_seq_start, n_a_en, n_Y, n_U, n_V

        for mem, idx, val in wr:
            mem[idx] = val
        s.cycle += 1
        return o_last


def encode(frames444, xsize16, ysize16, pframes, XL, YL, VL, Q, nbeats=None, bubbles=0, seed=1, max_cycles=None, sabotage=None):
    """The testbench's protocol against the model (TB:206-266): beats in raster order, a stop pulse with i_en = 0 after the
    last beat, words collected until o_last.  bubbles: every bubbles-th clock carries no beat.  Returns the stream bytes."""
    m = RtlModule(XL, YL, VL, Q, seed, sabotage)
    f = np.ascontiguousarray(frames444, np.uint8)
    n, _, H, W = f.shape
    bpf = W * H // 4
    total = n * bpf if nbeats is None else nbeats
    Y, U, V = (f[:, p].reshape(n, bpf, 4) for p in range(3))
    k, clk, stop_sent = 0, 0, False
    limit = max_cycles or (total + bpf) * (2 if bubbles else 1) + 200000
    while True:
        clk += 1
        assert clk < limit, "the model did not finish"
        if k < total and not (bubbles and clk % bubbles == 0):
            fr, b = divmod(k, bpf)
            o_last = m.clock(1, Y[fr, b], U[fr, b], V[fr, b], 0, xsize16, ysize16, pframes)
            k += 1
        elif k >= total and not stop_sent:
            o_last = m.clock(0, i_stop=1)
            stop_sent = True
        else:
            o_last = m.clock(0)
        if o_last:
            return bytes(m.out), m
