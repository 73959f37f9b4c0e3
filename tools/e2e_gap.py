import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, m2v_load
M = m2v_load.load()
W, H, PF, n = 1920, 1152, 8, 90
gop = PF + 1
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0").cpu()
pinned_t = clip.pin_memory(); frames = pinned_t.numpy()
for batch, one in ((18, 0), (9, 0), (9, 1), (18, 0), (9, 0), (9, 1)):
    enc = M.Mpeg2Encoder(7, 7, 3, 2)
    enc.set_option("batch_frames", batch)
    outbuf = np.empty(n * W * H * 3 // 2, np.uint8)
    for rep in range(4):
        T = []
        pos = 0
        t0 = time.perf_counter()
        for k in range(0, n, gop):
            a = time.perf_counter()
            if one:
                pos += enc.push_frames_pull(W // 16, H // 16, PF, frames[k:k + gop], outbuf, pos)[0]
                b = time.perf_counter()
            else:
                enc.push_frames(W // 16, H // 16, PF, frames[k:k + gop])
                b = time.perf_counter()
                pos += enc.pull_into(outbuf, pos)[0]
            c = time.perf_counter()
            T.append((a, b, c))
        a = time.perf_counter(); enc.sequence_stop(); b = time.perf_counter()
        last = False
        while not last:
            m, last = enc.pull_into(outbuf, pos); pos += m
        c = time.perf_counter()
        ok = __import__("hashlib").sha1(outbuf[:pos].tobytes()).hexdigest()[:8]
        if rep >= 2:
            print(("one call " if one else "two calls") + " batch %2d total %.2f ms | push us %s | pull us %s | gap us %s | stop %.0f drain %.0f  sha %s" % (batch, (c - t0) * 1e3,
                  [int((y - x) * 1e6) for x, y, z in T], [int((z - y) * 1e6) for x, y, z in T],
                  [int((T[i + 1][0] - T[i][2]) * 1e6) for i in range(len(T) - 1)], (b - a) * 1e6, (c - b) * 1e6, ok), file=sys.stderr)
    enc.close()
