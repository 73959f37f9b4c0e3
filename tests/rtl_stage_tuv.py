"""Clock-by-clock model of stages T, U and V of RTL/mpeg2encoder.v (RTL:2480-2994), written directly from the RTL's
sequential logic: the stage-T state machine emits up to seven (bits, length) fields plus an optional '10' per clock
(RTL:2589-2848), stage U concatenates them (RTL:2888-2909), stage V aligns, accumulates 255 bits and emits 256-bit
words, with the end-of-sequence flush (RTL:2924-2956), and the word is byte-swapped on the way out (RTL:2963-2994).

It is a SECOND, structurally different restatement of the entropy coder next to oracle/m2v_oracle.c (which walks
coefficient by coefficient and appends to a flat bit string): same tables, different decomposition - seven zig-zag
positions per clock with the run length carried in t_runlen, field packing without masking, explicit word framing.
Inputs are the per-macroblock decisions and levels (from the oracle's dumps).  TEST INFRASTRUCTURE ONLY.
"""
import ctypes


def _tables():
    from oracle import m2v_oracle_ctypes as orc
    L = orc.lib()

    def pair(fn, *a):
        c, n = ctypes.c_int(), ctypes.c_int()
        fn(*a, ctypes.byref(c), ctypes.byref(n))
        return c.value, n.value
    return dict(mv=[pair(L.m2v_oracle_tab_motion, k) for k in range(17)],
                cbp=[pair(L.m2v_oracle_tab_cbp, k) for k in range(64)],
                dcy=[pair(L.m2v_oracle_tab_dc, 0, k) for k in range(12)],
                dcc=[pair(L.m2v_oracle_tab_dc, 1, k) for k in range(12)],
                ac=lambda run, a: pair(L.m2v_oracle_tab_ac, run, a))


def put_AC(T, v, rl):
    """function put_AC (RTL:2525-2547) -> (bits, lens)"""
    absv = (-v if v < 0 else v) - 1
    if (rl == 0 and absv < 40) or (rl == 1 and absv < 18) or (rl == 2 and absv < 5) or (rl == 3 and absv < 4) or \
            (rl <= 6 and absv < 3) or (rl <= 16 and absv < 2) or (rl <= 31 and absv < 1):
        code, ln = T["ac"](rl, absv + 1)
        assert ln > 0
        return (code << 1) | (1 if v < 0 else 0), ln + 1
    return (1 << 18) | (rl << 12) | (v & 0xFFF), 24


class StageTUV:
    def __init__(self, W, H, Q):
        self.T = _tables()
        self.size_x, self.size_y, self.Q = W, H, Q
        # stage V registers
        self.v_bits, self.v_lens = 0, 0                     # 255-bit MSB-aligned accumulator
        self.out = bytearray()
        # stage T registers
        self.tc = [0, 0, 0, 0]                              # hour, minute, second, insec
        self.prev_mv = [0, 0]
        self.prev_dc = [0, 0, 0]

    # ---- stages U + V: one clock with the given stage-T outputs ----
    def clock(self, fields, align=False, append_b10=False):
        """fields: list of up to 7 (bits, lens)"""
        assert len(fields) <= 7
        ut_bits, ut_lens = 0, 0
        if append_b10:
            ut_bits, ut_lens = 0b10, 2
        for bits, lens in reversed(list(fields) + [(0, 0)] * (7 - len(fields))):      # i = 6 downto 0 (RTL:2901-2904)
            assert bits < (1 << 24) and lens <= 24
            ut_bits |= bits << ut_lens                      # no masking: a value wider than its length would corrupt its neighbour
            assert bits < (1 << lens) or lens == 0 and bits == 0, "field %x does not fit %d bits" % (bits, lens)
            ut_lens += lens
        assert ut_lens <= 170
        vt_lens = self.v_lens
        if align and vt_lens & 7:
            vt_lens = (vt_lens | 7) + 1                     # RTL:2940-2943
        vt_lens += ut_lens
        vt_bits = (self.v_bits << 177) | (ut_bits << (432 - vt_lens))               # RTL:2945
        assert vt_lens <= 432
        if vt_lens >> 8:                                    # RTL:2947-2949
            v_data = vt_bits >> 176                         # {v_data, v_bits} <= {vt_bits, 79'h0}
            self._emit(v_data)
            self.v_bits = (vt_bits & ((1 << 176) - 1)) << 79     # v_bits <= {vt_bits[175:0], 79'h0}
        else:
            self.v_bits = vt_bits >> 177
        self.v_lens = vt_lens & 0xFF

    def _emit(self, v_data):
        # o_data byte k (first written by the testbench) = v_data[255-8k -: 8] (RTL:2963-2994, TB:260-262)
        self.out += int(v_data).to_bytes(32, "big")

    def flush(self):
        """u_end_seq2: v_data <= {v_bits, 1'b0}, always one more word (RTL:2932-2937)"""
        self._emit(self.v_bits << 1)
        self.v_bits, self.v_lens = 0, 0

    # ---- stage T ----
    def sequence_start(self):
        sx, sy = self.size_x, self.size_y
        self.tc = [0, 0, 0, 0]
        self.clock([(0x000001, 24), (0xB3, 8), ((sx << 12) | sy, 24), (0x1209c4, 24), (0x200000, 24), (0x0001B5, 24),
                    (0x144200, 24)], align=True)                                     # RTL:2598-2604
        self.clock([(0x010000, 24), (0x000001, 24), (0xB52305, 24), (0x0505, 16), (sx, 14), (1, 1), (sy, 14)])   # RTL:2611-2617

    def macroblock(self, i_frame, x16, y16, inter, mvx, mvy, nzflags, zig):
        """zig: 6 lists of 64 levels in zig-zag order"""
        T = self.T
        # PUT_IDLE (RTL:2630-2660)
        if x16 == 0 and y16 == 0 and i_frame == 0:
            h, m, s_, p = self.tc
            self.clock([(0x000001, 24), (0xB8, 8), (h, 6), (m, 6), ((1 << 6) | s_, 7), (p, 6), (0x2, 2)], align=True)
        else:
            self.clock([])
        if x16 == 0 and y16 == 0:                           # PUT_FRAME_HEADER (RTL:2663-2699)
            f = [(0x000001, 24), (i_frame, 18), (0x10000, 19), (0x0, 3), (0x000001, 24), (0xB58111, 24), (0x1BC000, 24)]
            if i_frame != 0:
                f[2] = (0x20000, 19)
                f[3] = (0x380, 11)
            self.clock(f, align=True)
            h, m, s_, p = self.tc
            p += 1
            if p == 24:
                p = 0
                s_ += 1
                if s_ == 60:
                    s_ = 0
                    m += 1
                    if m == 60:
                        m = 0
                        if h < 63:
                            h += 1
            self.tc = [h, m, s_, p]
        if x16 == 0:                                        # PUT_SLICE_HEADER (RTL:2701-2716)
            self.clock([(0x000001, 24), (1 + y16, 8), (2 << self.Q, 6)], align=True)
            self.prev_dc = [0, 0, 0]
            self.prev_mv = [0, 0]
        # PUT_BLOCK_INFO (RTL:2718-2775)
        f = [(0, 0)] * 7
        if not inter and i_frame != 0:
            f[0] = (0x23, 6)
        elif inter and nzflags == 0:
            f[0] = (0x09, 4)
        else:
            f[0] = (0x03, 2)
        if inter:
            for k, mv in enumerate((mvx, mvy)):
                dmv = mv - self.prev_mv[k]
                if dmv > 15:
                    dmv -= 32
                elif dmv < -16:
                    dmv += 32
                c, n = T["mv"][abs(dmv)]
                f[1 + 2 * k] = (c, n)
                if dmv != 0:
                    f[2 + 2 * k] = (1 if dmv < 0 else 0, 1)
            f[5] = T["cbp"][nzflags]
            self.prev_mv = [mvx, mvy]
        else:
            self.prev_mv = [0, 0]
        self.clock(f)
        # PUT_TILE: 6 tiles x 10 clocks (RTL:2777-2847)
        t_nz = nzflags
        for t in range(6):
            z = list(zig[t])
            nzflag = (t_nz >> 5) & 1
            t_runlen = 0
            for t_cnt in range(10):
                f = [(0, 0)] * 7
                b10 = False
                if t_cnt == 0:
                    val = z[0]
                    comp = 0 if t < 4 else t - 3
                    diff_dc = val - self.prev_dc[comp]
                    self.prev_dc[comp] = 0 if inter else val
                    nxt_runlen = 0
                    if inter:
                        if val == 0:
                            nxt_runlen = 1
                        elif val in (1, -1):
                            if nzflag:
                                f[0] = (0b10 | (1 if val < 0 else 0), 2)
                        elif nzflag:
                            f[0] = put_AC(T, val, 0)
                    else:
                        a = -diff_dc if diff_dc < 0 else diff_dc
                        vallen = a.bit_length()
                        tmp = diff_dc & 0xFFF
                        if diff_dc < 0:
                            tmp = (tmp + ((1 << vallen) - 1)) & 0xFFF
                        if nzflag:
                            f[0] = (T["dcy"] if t < 4 else T["dcc"])[vallen]
                            f[1] = (tmp, vallen)
                    t_runlen = nxt_runlen
                else:
                    runlen = t_runlen
                    for i in range(7):
                        val = z[i + 1]                      # t_zig_blk[0][i+1]; the array shifts by 7 per clock (RTL:2865-2866)
                        if val != 0:
                            if nzflag:
                                f[i] = put_AC(T, val, runlen)
                            runlen = 0
                        else:
                            runlen = (runlen + 1) & 63
                    t_runlen = runlen
                    b10 = bool(nzflag) and t_cnt == 9
                    if t_cnt < 9:
                        z = [z[0]] + z[8:] + [0] * 7        # shift AC values by 7
                self.clock(f, append_b10=b10)
            t_nz = (t_nz << 1) & 63

    def sequence_end(self):
        self.clock([(0x000001, 24), (0xB7, 8)], align=True)   # RTL:2621-2628
        self.clock([])                                          # u stage
        self.flush()
        return bytes(self.out)
