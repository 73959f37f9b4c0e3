"""Helpers for the -m gpu parity tests: drive the C-ABI on cuda:0, compare with the oracle."""
import numpy as np

import m2v_load
from oracle import m2v_oracle_ctypes as orc

M = m2v_load.load()


def resident_encode(frames, xs16, ys16, pframes, XL=7, YL=7, VL=3, Q=2, batch_frames=None, debug=False, enc=None):
    """frames: uint8 [n,3,H,W] in the clamped geometry.  Returns bytes (and dumps if debug)."""
    import torch
    own = enc is None
    if own:
        # debug=True: the -DM2V_DEBUG build of the same sources (level dump, every frame's reconstruction kept)
        enc = M.Mpeg2Encoder(XL, YL, VL, Q, device=0, debug=debug)
    try:
        if batch_frames:
            enc.set_option("batch_frames", batch_frames)
        if debug:
            enc.set_option("keep_recon", 1)
        W, H = enc.geometry(xs16, ys16)
        f = np.ascontiguousarray(frames, np.uint8).reshape(-1, 3, H, W)
        n = f.shape[0]
        d_in = torch.from_numpy(f).to("cuda:0")
        cap = n * W * H * 3 + (1 << 16)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        nbytes = enc.encode_resident(d_in.data_ptr(), n, d_out.data_ptr(), cap, xs16, ys16, pframes)
        data = d_out[:nbytes].cpu().numpy().tobytes()
        if not debug:
            return data
        mbs = (W // 16) * (H // 16)
        info = enc.debug_read(0, n * mbs * 4, np.uint32).reshape(n, mbs)
        coef = enc.debug_read(1, n * mbs * 768, np.int16).reshape(n, mbs, 6, 64)
        bits = enc.debug_read(2, n * mbs * 4, np.uint32).reshape(n, mbs)
        recon = None
        if pframes > 0:
            recon = enc.debug_read(3, n * (W * H * 3 // 2), np.uint8).reshape(n, -1)
        dumps = dict(mb_inter=(info & 1).astype(np.int8), mb_cbp=((info >> 1) & 63).astype(np.uint8),
                     mb_mvx=((info >> 8) & 255).astype(np.uint8).view(np.int8),
                     mb_mvy=((info >> 16) & 255).astype(np.uint8).view(np.int8),
                     coef=coef, mb_bits=bits, recon=recon)
        return data, dumps
    finally:
        if own:
            enc.close()


def first_diff(a, b):
    a = np.asarray(a).reshape(-1)
    b = np.asarray(b).reshape(-1)
    n = min(a.size, b.size)
    d = np.nonzero(a[:n] != b[:n])[0]
    if d.size:
        return int(d[0])
    return None if a.size == b.size else n


def compare_stages(frames, xs16, ys16, pframes, XL=7, YL=7, VL=3, Q=2, batch_frames=None):
    """Stage-by-stage comparison; returns a list of human-readable mismatch strings (empty = parity)."""
    W, H = M.clamp_geometry(xs16, ys16, XL, YL)
    mbw = W // 16
    ref_bytes, ref = orc.encode(frames, xs16, ys16, pframes, XL, YL, VL, Q, dump=True)
    got_bytes, got = resident_encode(frames, xs16, ys16, pframes, XL, YL, VL, Q, batch_frames, debug=True)
    problems = []

    def where(flat_mb):
        f, mb = divmod(flat_mb, ref["mb_inter"].shape[1])
        return "frame %d mb (x=%d,y=%d)" % (f, mb % mbw, mb // mbw)

    for key in ("mb_inter", "mb_mvx", "mb_mvy", "mb_cbp"):
        d = first_diff(ref[key], got[key])
        if d is not None:
            problems.append("%s differs first at %s: oracle %d gpu %d (%d MBs differ)" % (
                key, where(d), ref[key].reshape(-1)[d], got[key].reshape(-1)[d],
                int((ref[key] != got[key]).sum())))
    d = first_diff(ref["coef"], got["coef"])
    if d is not None:
        mb, rest = divmod(d, 384)
        problems.append("levels differ first at %s tile %d zz %d: oracle %d gpu %d" % (
            where(mb), rest // 64, rest % 64, ref["coef"].reshape(-1)[d], got["coef"].reshape(-1)[d]))
    if pframes > 0 and got["recon"] is not None:
        n = ref["recon"].shape[0]
        gop = pframes + 1
        for f in range(n):
            needed = (f % gop) < pframes and f != n - 1
            if needed and not np.array_equal(ref["recon"][f], got["recon"][f]):
                d = first_diff(ref["recon"][f], got["recon"][f])
                problems.append("recon of frame %d differs first at byte %d" % (f, d))
                break
    # GPU mb_len includes the 38-bit slice header on the first MB of each row
    gb = got["mb_bits"].astype(np.int64).copy()
    gb.reshape(gb.shape[0], -1, mbw)[:, :, 0] -= 38
    d = first_diff(ref["mb_bits"], gb)
    if d is not None:
        problems.append("mb bit length differs first at %s: oracle %d gpu %d" % (
            where(d), ref["mb_bits"].reshape(-1)[d], gb.reshape(-1)[d]))
    if ref_bytes != got_bytes:
        d = first_diff(np.frombuffer(ref_bytes, np.uint8), np.frombuffer(got_bytes, np.uint8))
        problems.append("stream differs: oracle %d bytes, gpu %d bytes, first difference at byte %s" % (
            len(ref_bytes), len(got_bytes), d))
    return problems
