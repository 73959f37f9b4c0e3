"""-m gpu: strip mode (config c5) on ONE GPU with emulated ranks: N encoder handles, each owning a strip of
macroblock rows; the halo "exchange" is a device copy.  The assembled stream must equal the oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_emulated_strips(M, d_clip, W, H, pf, VL, world, split):
    """`world` encoder handles on ONE GPU, each owning a strip of macroblock rows; the halo "exchange" is a device copy.
    Returns the assembled stream bytes."""
    import torch
    encs = [M.Mpeg2Encoder(7, 7, VL, 2) for _ in range(world)]
    try:
        shared = torch.cuda.Stream()                       # the emulated ranks share one stream: the "exchange" is a plain copy
        engines = [M.parallel.GpuStripEngine(e, d_clip, W // 16, H // 16, pf, "cuda:0", stream=shared) for e in encs]
        rows = M.parallel.partition_rows(H // 16, world)
        info = [eng.begin(*rows[r]) for r, eng in enumerate(engines)]
        steps, hb = info[0]
        assert all(i == info[0] for i in info)
        torch.cuda.synchronize()
        up = [eng.alloc(hb) for eng in engines]
        down = [eng.alloc(hb) for eng in engines]
        for j in range(steps):
            if split:
                nb = [eng.step_edges(j, up[r], down[r]) for r, eng in enumerate(engines)]
                for eng in engines:
                    eng.step_interior(j)
            else:
                nb = [eng.step(j, up[r], down[r]) for r, eng in enumerate(engines)]
            assert len(set(nb)) == 1
            if nb[0]:
                for r, eng in enumerate(engines):       # rank r receives rank r-1's bottom rows and rank r+1's top rows
                    eng.halo_in(j, down[r - 1] if r > 0 else None, up[r + 1] if r < world - 1 else None)
        outs = [eng.finish() for eng in engines]
        stream = engines[0].assemble([o[0] for o in outs], [o[1] for o in outs])
        torch.cuda.synchronize()
        return stream.cpu().numpy().tobytes()
    finally:
        for e in encs:
            e.close()


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("world,W,H,pf,VL", [(2, 128, 96, 4, 3), (3, 96, 160, 2, 2), (4, 160, 128, 3, 1), (8, 64, 128, 1, 3)])
def test_strips_equal_single_encoder(world, W, H, pf, VL, split):
    """split: the step in two parts (edge rows + halo pack, then interior rows) as encode_strips() issues it around the
    exchange; strips of 1, 2, 3 and 4 rows"""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    n = 2 * (pf + 1) + 1
    clip = M.synth.clip(W, H, n, clip_index=50 + world)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, 2)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    assert run_emulated_strips(M, d_clip, W, H, pf, VL, world, split) == want


def test_config_c5_full_size_8_strips_of_16_rows():
    """BASELINE config c5 at its real size on one GPU: ONE 2048x2048 sequence (XL = YL = 7, 128 x 128 macroblocks), one
    whole GOP of 1 I + 8 P frames, VECTOR_LEVEL 3, cut into 8 strips of 16 macroblock rows with the +-6 luma / +-3
    chroma halo (9 rows x 2048 bytes per frame per direction), edge rows first as encode_strips() issues them.
    Byte-identical to the oracle (about 5 s of CPU) and to the single-handle encode of the same clip."""
    import torch
    import m2v_load
    import gpu_util as G
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W = H = 2048
    pf, n = 8, 9
    d_clip = M.synth.clip_torch(W, H, n, clip_index=55, device="cuda:0", scene_len=5)      # a scene cut inside the GOP
    clip = d_clip.cpu().numpy()
    want = orc.encode(clip, 128, 128, pf, 7, 7, 3, 2)
    got = run_emulated_strips(M, d_clip, W, H, pf, 3, 8, True)
    assert len(got) == len(want)
    assert got == want
    assert G.resident_encode(clip, 128, 128, pf, 7, 7, 3, 2) == want


def test_encode_strips_world1_equals_resident():
    """parallel.encode_strips with one rank (no exchange) == m2v_encode_resident == oracle."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    clip = M.synth.clip(160, 96, 7, clip_index=60)
    want = orc.encode(clip, 10, 6, 2, 7, 7, 3, 2)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    enc = M.Mpeg2Encoder(7, 7, 3, 2)
    try:
        eng = M.parallel.GpuStripEngine(enc, d_clip, 10, 6, 2, "cuda:0")
        out = M.parallel.encode_strips(eng, 0, 1)
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == want
    finally:
        enc.close()


# ---- the native loop: m2v_strip_encode, ranks = threads of this process talking through a local communicator ----
def run_native_strips(M, d_clip, W, H, pf, VL, world, profile=False, general=False, Q=2, conformant=False):
    """`world` handles, one host thread each, all on GPU 0; every thread makes ONE call (m2v_strip_encode) - the GOP steps,
    the halo exchange (mailboxes + device copies behind the same interface RCCL sits behind), the size all-gather, the
    strips to rank 0 and the final assembly all happen inside it.  Returns (stream bytes, [strip_stats of every rank])."""
    import threading
    import torch
    encs = [M.Mpeg2Encoder(7, 7, VL, Q) for _ in range(world)]
    if conformant:       # option conformant (NOT the reference's arithmetic): the general form of the step, checked against the oracle's conformant mode
        for e in encs:
            e.set_option("conformant", 1)
    if general:          # the general form of the step (pack / unpack kernels, exchange on its own stream): what option conformant
        for e in encs:   # and dct_mfma = 0 run instead of the fused edge-row kernel
            e.set_option("dct_mfma", 0)
    comm = M.StripComm.local(world) if world > 1 else None
    out = torch.empty(M.parallel.strip_output_bound(int(d_clip.shape[0]), W, H), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    res, errs = [None] * world, []

    def work(r):
        try:
            if profile:
                encs[r].set_option("profile", 1)
            res[r] = M.parallel.encode_strips_native(encs[r], comm, r, world, d_clip, W // 16, H // 16, pf, out if r == 0 else None)
        except Exception as ex:  # noqa: BLE001
            errs.append((r, ex))
    try:
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in th), "a rank is stuck in the exchange"
        assert not errs, errs
        stats = [e.strip_stats() for e in encs]
        return res[0].cpu().numpy().tobytes(), stats
    finally:
        for e in encs:
            e.close()
        if comm is not None:
            comm.close()


@pytest.mark.parametrize("world,W,H,pf,VL", [(1, 96, 64, 2, 3), (2, 128, 96, 4, 3), (3, 96, 160, 2, 2), (4, 160, 128, 3, 1), (8, 64, 128, 1, 3)])
def test_native_strip_loop_equals_oracle(world, W, H, pf, VL):
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    n = 2 * (pf + 1) + 1
    clip = M.synth.clip(W, H, n, clip_index=70 + world)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, 2)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    for _ in range(2):                                   # twice: the handles' and the communicator's state is reusable
        got, stats = run_native_strips(M, d_clip, W, H, pf, VL, world)
        assert got == want
    assert all(s["steps"] == pf + 1 for s in stats)
    got, _ = run_native_strips(M, d_clip, W, H, pf, VL, world, general=True)
    assert got == want


def test_native_strip_loop_config_c5_full_size_8_ranks():
    """config c5 at its real size through the native loop: 2048x2048, one GOP of 1 I + 8 P, 8 ranks x 16 rows (threads on one
    GPU, local communicator), byte-identical to the oracle; with option profile the per-step host time is reported."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W = H = 2048
    pf, n = 8, 9
    d_clip = M.synth.clip_torch(W, H, n, clip_index=56, device="cuda:0", scene_len=5)
    want = orc.encode(d_clip.cpu().numpy(), 128, 128, pf, 7, 7, 3, 2)
    got, stats = run_native_strips(M, d_clip, W, H, pf, 3, 8, profile=True)
    assert got == want
    print("host us per GOP step, per rank:", [round(s["host_us_per_step"], 1) for s in stats])


def test_rccl_communicator_on_one_gpu():
    """What can be run of the RCCL transport on a 1-GPU box: librccl is found and dlopen()ed, a 1-rank communicator is
    created from a fresh ncclUniqueId, and one ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd addressed to the rank itself
    moves bytes on the device; m2v_strip_encode accepts the communicator (world = 1: nothing to exchange)."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    comm = M.StripComm.rccl(0, 1, 0)
    try:
        a = torch.arange(1 << 16, dtype=torch.int32, device="cuda:0").view(torch.uint8)
        b = torch.zeros_like(a)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        comm.selftest(0, a.data_ptr(), b.data_ptr(), a.numel(), s.cuda_stream)
        s.synchronize()
        assert torch.equal(a, b)
        clip = M.synth.clip(96, 64, 5, clip_index=77)
        d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
        out = torch.empty(M.parallel.strip_output_bound(5, 96, 64), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        enc = M.Mpeg2Encoder(7, 7, 3, 2)
        try:
            got = M.parallel.encode_strips_native(enc, comm, 0, 1, d_clip, 6, 4, 2, out)
            assert got.cpu().numpy().tobytes() == orc.encode(clip, 6, 4, 2, 7, 7, 3, 2)
        finally:
            enc.close()
    finally:
        comm.close()


def test_a_failing_rank_does_not_leave_the_others_waiting():
    """three ranks (threads) on an in-process communicator; rank 1 makes a call that is refused (no output buffer on the output
    rank's position is fine for 0 and 2 - rank 1 claims to BE the output rank without one): the communicator is aborted and
    the two ranks that are already waiting for rank 1's rows come back with an error instead of hanging."""
    import ctypes
    import threading
    import torch
    import m2v_load
    M = m2v_load.load()
    W, H, pf, n = 96, 96, 2, 6
    d_clip = torch.from_numpy(np.ascontiguousarray(M.synth.clip(W, H, n, clip_index=90))).to("cuda:0")
    out = torch.empty(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    encs = [M.Mpeg2Encoder(7, 7, 3, 2) for _ in range(3)]
    comm = M.StripComm.local(3)
    rc, msg = [None] * 3, [None] * 3

    def work(r):
        nb = ctypes.c_size_t(0)
        dst = 1 if r == 1 else 0                       # rank 1: "I am the output rank", but hands in no buffer
        rc[r] = encs[r]._L.m2v_strip_encode(encs[r]._h, comm.handle, r, 3, dst, W // 16, H // 16, pf, d_clip.data_ptr(), n,
                                            out.data_ptr() if r == 0 else None, out.numel() if r == 0 else 0, ctypes.byref(nb), None)
        msg[r] = encs[r]._L.m2v_last_error(encs[r]._h)
    try:
        th = [threading.Thread(target=work, args=(r,)) for r in (0, 2)]
        for t in th:
            t.start()
        import time
        time.sleep(0.5)                                # ranks 0 and 2 are inside the exchange of the first step by now
        work(1)
        for t in th:
            t.join(timeout=60)
        assert not any(t.is_alive() for t in th), "a rank is still waiting for the one that failed"
        assert rc[1] == -1 and b"d_out" in msg[1]
        # (rank 1 keeps the call order and marks its sizes - "rank 1 of the job failed" -, then aborts the communicator on its way out:
        # a rank still inside the size exchange at that moment reads "another rank ... has failed" instead)
        assert rc[0] < 0 and rc[2] < 0 and (b"another rank" in msg[0] + msg[2] or b"rank 1 of the job failed" in msg[0] + msg[2])
        for e in encs:                                 # the handles are usable afterwards
            assert not e.busy and e.encode(d_clip[:2].cpu().numpy(), W // 16, H // 16, pf)
    finally:
        for e in encs:
            e.close()
        comm.close()


@pytest.mark.parametrize("world", [1, 2, 4])
def test_native_strip_loop_in_conformant_mode(world):
    """option conformant (ISO reconstruction loop, clearly non-parity) through strip mode: the halo rows are the conformant
    reconstruction's, the stream equals the oracle's conformant mode."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, VL, n = 160, 128, 3, 3, 9
    clip = M.synth.clip(W, H, n, clip_index=91, scene_len=4)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, 2, conformant=True)
    assert want != orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, 2)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    got, _ = run_native_strips(M, d_clip, W, H, pf, VL, world, conformant=True)
    assert got == want



def test_a_rank_whose_own_work_fails_keeps_the_call_order_and_everybody_returns_an_error():
    """The failure protocol of m2v_strip_encode (the one RCCL peers depend on: nobody may be left waiting inside an exchange): rank 1 of
    three fails locally right after its plan (injected, -DM2V_DEBUG library).  It still takes part in every halo exchange and in the
    all-gather of the sizes, where it marks its row; every rank sees the mark after the same call, skips the gather and returns an
    error - rank 1 its own, the others "rank 1 of the job failed".  The handles and a fresh communicator work afterwards."""
    import ctypes
    import threading
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, n = 96, 96, 2, 6
    clip = M.synth.clip(W, H, n, clip_index=92)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    out = torch.empty(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    encs = [M.Mpeg2Encoder(7, 7, 3, 2, debug=True) for _ in range(3)]
    L = encs[0]._L

    def run(comm, bad):
        rc, msg = [None] * 3, [None] * 3

        def work(r):
            nb = ctypes.c_size_t(0)
            rc[r] = L.m2v_strip_encode(encs[r]._h, comm.handle, r, 3, 0, W // 16, H // 16, pf, d_clip.data_ptr(), n,
                                       out.data_ptr() if r == 0 else None, out.numel() if r == 0 else 0, ctypes.byref(nb), None)
            msg[r] = L.m2v_last_error(encs[r]._h)
            if r == 0:
                rc.append(nb.value)
        if bad is not None:
            encs[bad].set_option("ablate", 1 << 21)
        th = [threading.Thread(target=work, args=(r,)) for r in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=60)
        assert not any(t.is_alive() for t in th), "a rank is still waiting"
        if bad is not None:
            encs[bad].set_option("ablate", 0)
        return rc, msg
    comm = M.StripComm.local(3, debug=True)
    try:
        rc, msg = run(comm, 1)
        assert rc[1] < 0 and b"injected failure" in msg[1]
        assert rc[0] < 0 and rc[2] < 0 and b"rank 1 of the job failed" in msg[0] and b"rank 1 of the job failed" in msg[2]
    finally:
        comm.close()
    comm = M.StripComm.local(3, debug=True)
    try:
        rc, msg = run(comm, None)
        assert rc[:3] == [0, 0, 0], msg
        assert out[:rc[3]].cpu().numpy().tobytes() == want
    finally:
        comm.close()
        for e in encs:
            e.close()
