"""Seeded synthetic yuv444p clips (the reference's SIM/data.zip clips are not redistributable
and absent from the checkout; SURVEY.md 8(d) specifies these stand-ins).

Content: low-frequency sinusoids + 8x8-block-constant offsets + noise, translated per frame by an
integer+half-pel global vector in [-5, 5] (exercises full- and half-pel vectors, ties, edge
masking), a few independently moving 32x32 objects, a scene cut every 23 frames (intra blocks in
P frames), and one dark region (un-saturated intra cost).  numpy only.
"""
import numpy as np

SEED0 = 0x4D325631


def _texture(rng, H, W, pad):
    hh, ww = H + 2 * pad, W + 2 * pad
    y, x = np.mgrid[0:hh, 0:ww].astype(np.float32)
    t = np.zeros((hh, ww), np.float32)
    for _ in range(4):
        fx, fy = rng.uniform(0.005, 0.08, 2)
        ph = rng.uniform(0, 6.28)
        t += rng.uniform(10, 30) * np.sin(fx * x + fy * y + ph)
    blocks = rng.uniform(-24, 24, ((hh + 7) // 8, (ww + 7) // 8)).astype(np.float32)
    t += np.kron(blocks, np.ones((8, 8), np.float32))[:hh, :ww]
    return t


def clip(W, H, nframes, clip_index=0, scene_len=23):
    """-> uint8 array [nframes, 3, H, W] (Y, U, V planes, 4:4:4)"""
    rng = np.random.default_rng(SEED0 + clip_index)
    pad = 16
    out = np.empty((nframes, 3, H, W), np.uint8)
    base = None
    pos = np.zeros(2)
    objs = []
    for f in range(nframes):
        if f % scene_len == 0:
            base = [128 + _texture(rng, H, W, pad) * (1.0 if p == 0 else 0.5) for p in range(3)]
            pos = np.zeros(2)
            objs = [dict(p=rng.uniform([0, 0], [H - 32, W - 32]), v=rng.uniform(-3, 3, 2),
                         c=rng.uniform(30, 220, 3)) for _ in range(4)]
        else:
            pos = np.clip(pos + rng.integers(-10, 11, 2) / 2.0, -pad + 1, pad - 2)
        iy, ix = np.floor(pos).astype(int)
        fy, fx = pos - np.floor(pos)
        for p in range(3):
            b = base[p]
            a = b[pad + iy:pad + iy + H + 1, pad + ix:pad + ix + W + 1]
            img = a[:H, :W]
            if fx:
                img = (img + a[:H, 1:W + 1]) / 2
            if fy:
                img2 = a[1:H + 1, :W]
                if fx:
                    img2 = (img2 + a[1:H + 1, 1:W + 1]) / 2
                img = (img + img2) / 2
            img = img + rng.integers(-4, 5, (H, W))
            for o in objs:
                oy, ox = int(o["p"][0]), int(o["p"][1])
                img[oy:oy + 32, ox:ox + 32] = o["c"][p] + rng.integers(-2, 3, img[oy:oy + 32, ox:ox + 32].shape)
            if p == 0:
                img[H // 2:H // 2 + 48, W // 4:W // 4 + 64] *= 0.05   # dark region
            out[f, p] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        for o in objs:
            o["p"] = np.clip(o["p"] + o["v"], [0, 0], [H - 33, W - 33])
    return out


def degenerate(kind, W, H, nframes=2):
    """constant gray / black / full-range checkerboard clips"""
    out = np.empty((nframes, 3, H, W), np.uint8)
    if kind == "gray":
        out[:] = 128
    elif kind == "black":
        out[:, 0] = 0
        out[:, 1:] = 128
    elif kind == "checker":
        y, x = np.mgrid[0:H, 0:W]
        for f in range(nframes):
            out[f, 0] = np.where((x + y + f) & 1, 255, 0)
            out[f, 1] = np.where((x // 2 + y) & 1, 255, 0)
            out[f, 2] = np.where((x + y // 2) & 1, 0, 255)
    elif kind == "noise":
        rng = np.random.default_rng(SEED0 ^ 0x5555)
        out[:] = rng.integers(0, 256, out.shape, dtype=np.uint8)
    else:
        raise ValueError(kind)
    return out


def clip_torch(W, H, nframes, clip_index=0, device="cuda:0", scene_len=23):
    """Same recipe as clip() generated with torch on `device` (fast enough for 90 x 1920x1152).

    Not value-identical to clip(); used where the clip only has to be resident in HBM.
    -> uint8 tensor [nframes, 3, H, W]
    """
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(SEED0 + clip_index)
    rng = np.random.default_rng(SEED0 + clip_index)
    pad = 16
    hh, ww = H + 2 * pad, W + 2 * pad
    yy = torch.arange(hh, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(ww, device=device, dtype=torch.float32)[None, :]
    out = torch.empty((nframes, 3, H, W), dtype=torch.uint8, device=device)
    base, pos, objs = None, np.zeros(2), []
    for f in range(nframes):
        if f % scene_len == 0:
            base = []
            for p in range(3):
                t = torch.zeros((hh, ww), device=device)
                for _ in range(4):
                    fx, fy = rng.uniform(0.005, 0.08, 2)
                    t += float(rng.uniform(10, 30)) * torch.sin(float(fx) * xx + float(fy) * yy + float(rng.uniform(0, 6.28)))
                blocks = (torch.rand(((hh + 7) // 8, (ww + 7) // 8), device=device, generator=g) - 0.5) * 48
                t += blocks.repeat_interleave(8, 0).repeat_interleave(8, 1)[:hh, :ww]
                base.append(128 + t * (1.0 if p == 0 else 0.5))
            pos = np.zeros(2)
            objs = [dict(p=rng.uniform([0, 0], [H - 32, W - 32]), v=rng.uniform(-3, 3, 2), c=rng.uniform(30, 220, 3))
                    for _ in range(4)]
        else:
            pos = np.clip(pos + rng.integers(-10, 11, 2) / 2.0, -pad + 1, pad - 2)
        iy, ix = np.floor(pos).astype(int)
        fy, fx = pos - np.floor(pos)
        for p in range(3):
            a = base[p][pad + iy:pad + iy + H + 1, pad + ix:pad + ix + W + 1]
            img = a[:H, :W]
            if fx:
                img = (img + a[:H, 1:W + 1]) / 2
            if fy:
                img2 = a[1:H + 1, :W]
                if fx:
                    img2 = (img2 + a[1:H + 1, 1:W + 1]) / 2
                img = (img + img2) / 2
            img = img + torch.randint(-4, 5, (H, W), device=device, generator=g)
            for o in objs:
                oy, ox = int(o["p"][0]), int(o["p"][1])
                img[oy:oy + 32, ox:ox + 32] = float(o["c"][p]) + torch.randint(-2, 3, (32, 32), device=device, generator=g)
            if p == 0:
                img[H // 2:H // 2 + 48, W // 4:W // 4 + 64] *= 0.05
            out[f, p] = img.round().clamp(0, 255).to(torch.uint8)
        for o in objs:
            o["p"] = np.clip(o["p"] + o["v"], [0, 0], [H - 33, W - 33])
    return out
