"""-m gpu, only where the box has MORE THAN ONE GPU (skipped on the 1-GPU boxes of this pool): the first contact of the
multi-device paths with two devices - handles created from threads on two GPUs, the in-process communicator with its two ranks
on two GPUs (peer copies, events that belong to the right device), and the two multi-process bench modes with real RCCL."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        return 0


need2 = pytest.mark.skipif(_devices() < 2, reason="needs two GPUs (config c4 / c5 across devices)")


@need2
def test_two_handles_created_from_two_threads_on_two_devices():
    """config c4 in one process: a handle per GPU, created and driven from a thread each at the same time (the constant tables are
    uploaded once per DEVICE); both streams are the oracle's"""
    import threading
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    clips = [M.synth.clip(160, 96, 7, clip_index=300 + k, scene_len=4) for k in range(2)]
    wants = [orc.encode(c, 10, 6, 2, 6, 6, 3, 2) for c in clips]
    got, errs = [None, None], []

    def work(k):
        try:
            dev = "cuda:%d" % k
            enc = M.Mpeg2Encoder(6, 6, 3, 2, device=k)
            try:
                d_in = torch.from_numpy(np.ascontiguousarray(clips[k])).to(dev)
                d_out = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize(dev)
                for _ in range(3):
                    n = enc.encode_resident(d_in.data_ptr(), 7, d_out.data_ptr(), d_out.numel(), 10, 6, 2)
                got[k] = d_out[:n].cpu().numpy().tobytes()
                assert enc.encode(clips[k], 10, 6, 2) == wants[k]
            finally:
                enc.close()
        except Exception as ex:  # noqa: BLE001
            errs.append((k, ex))
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    assert got == wants


@need2
@pytest.mark.parametrize("world", [2, 4])
def test_in_process_communicator_with_ranks_on_two_devices(world):
    """m2v_strip_encode, ranks = threads, rank r on GPU r % 2: the halo rows and the strips cross from one device to the other
    (hipMemcpyPeerAsync behind events of the right device); the assembled stream is the oracle's"""
    import threading
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, n = 160, 128, 3, 9
    clip = M.synth.clip(W, H, n, clip_index=310, scene_len=4)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    devs = [r % 2 for r in range(world)]
    d_clips = [torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:%d" % d) for d in range(2)]
    out = torch.empty(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0")
    for d in range(2):
        torch.cuda.synchronize(d)
    encs = [M.Mpeg2Encoder(7, 7, 3, 2, device=devs[r]) for r in range(world)]
    comm = M.StripComm.local(world)
    res, errs = [None] * world, []

    def work(r):
        try:
            res[r] = M.parallel.encode_strips_native(encs[r], comm, r, world, d_clips[devs[r]], W // 16, H // 16, pf, out if r == 0 else None)
        except Exception as ex:  # noqa: BLE001
            errs.append((r, ex))
    try:
        for _ in range(2):
            th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
            for t in th:
                t.start()
            for t in th:
                t.join(timeout=120)
            assert not any(t.is_alive() for t in th), "a rank is stuck in the exchange"
            assert not errs, errs
            assert res[0].cpu().numpy().tobytes() == want
    finally:
        for e in encs:
            e.close()
        comm.close()


def _bench(argv, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "M2V_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@need2
def test_bench_strips_on_two_gpus_with_real_rccl():
    """config c5 over two devices, one process per GPU, RCCL send / recv of the halo rows from C++ (strip_loop native): the
    assembled stream is the oracle's"""
    r, lines = _bench(["--gpus", "2", "--mode", "strips", "--rotate-dst", "--steps", "6", "--warmup", "1", "--prewarm", "0.2", "--gops", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = lines[-1]
    assert d["n_gpus"] == 2 and d["config"]["strip_loop"] == "native", d["config"]
    assert d["parity_check"]["identical_to_oracle"] is True
    # two sequences in flight per rank from one thread on the ONE RCCL communicator, output rank fixed and rotating
    assert d["sequences_in_flight"] == 2 and d["in_flight_output_rank_0"]["identical_to_the_blocking_call"] is True
    assert d["in_flight_output_rank_rotating"]["value"] > 0


@need2
def test_bench_sequences_on_two_gpus():
    """config c4 with two ranks: independent sequences, no collective on the data path; both ranks' streams checked"""
    r, lines = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm", "0.2", "--gops", "2", "--no-cpu-baseline", "--no-e2e", "--strips-steps", "6", "--long-gops", "8"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    d = lines[-1]
    assert d["n_gpus"] == 2 and d["parity_check"]["identical_to_oracle"] is True
    # the one line also carries config c5 over the two devices: RCCL halo, then peer stores across xGMI - numbers, or a stated error that
    # did not cost the c4 line (printed either way: this is the first contact)
    print("strips legs on two GPUs:", json.dumps(d["strips"])[:2000])
    assert set(d["strips"]) == {"rccl", "peer"}
    for name in ("rccl", "peer"):
        leg = d["strips"][name]
        assert "error" in leg or (leg["n_gpus"] == 2 and leg["identical_to_oracle"] is True and leg["value"] > 0), leg
    assert "error" not in d["strips"]["rccl"], d["strips"]["rccl"]


@need2
@pytest.mark.parametrize("world", [2, 4])
def test_peer_transport_with_ranks_on_two_devices(world):
    """The peer transport's first contact with a second GPU: ranks = threads, rank r on GPU r % 2, every landing block in fine-grained
    memory of its own device, the neighbours' blocks reached through hipDeviceEnablePeerAccess; the edge blocks of one GPU store
    into the memory of the other (write-through, system scope) and count their arrival there.  Byte-identical to the oracle, three
    sequences; what the communicator says about waits that ran out of budget is printed (none are expected: every rank has a GPU
    or at least a stream of its own)."""
    import threading
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, n = 256, 256, 4, 15
    clip = M.synth.clip(W, H, n, clip_index=311, scene_len=6)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    devs = [r % 2 for r in range(world)]
    d_clips = [torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:%d" % d) for d in range(2)]
    out = torch.empty(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0")
    for d in range(2):
        torch.cuda.synchronize(d)
    encs = [M.Mpeg2Encoder(7, 7, 3, 2, device=devs[r]) for r in range(world)]
    base = M.StripComm.local(world)
    peers, got, errs = [None] * world, [], []

    def work(r):
        try:
            peers[r] = M.StripComm.peer(base, r, devs[r])
            for _ in range(3):
                o = M.parallel.encode_strips_native(encs[r], peers[r], r, world, d_clips[devs[r]], W // 16, H // 16, pf, out if r == 0 else None)
                if r == 0:
                    got.append(o.cpu().numpy().tobytes())
        except Exception as ex:  # noqa: BLE001
            errs.append((r, ex))
    try:
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in th), "a rank is stuck"
        assert not errs, errs
        assert got == [want] * 3
        stats = [p.peer_stats() for p in peers]
        print("peer transport across two devices:", stats[0])
        assert all(s == stats[0] for s in stats) and stats[0]["peer_sequences"] >= 1
    finally:
        for p in peers:
            if p is not None:
                p.close()
        for e in encs:
            e.close()
        base.close()


@need2
def test_bench_strips_on_two_gpus_with_the_peer_transport():
    """config c5 over two devices, one process per GPU: landing blocks mapped through hipIpc handles (all-gathered over RCCL), the
    halo rows stored across xGMI by the edge-row kernel itself; sizes and strips through RCCL.  The stream is the oracle's."""
    r, lines = _bench(["--gpus", "2", "--mode", "strips", "--transport", "peer", "--rotate-dst", "--steps", "6", "--warmup", "1", "--prewarm", "0.2", "--gops", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = lines[-1]
    assert d["n_gpus"] == 2 and d["config"]["strip_loop"] == "native" and d["config"]["transport"] == "peer+rccl", d["config"]
    assert d["parity_check"]["identical_to_oracle"] is True
    print("bench --transport peer on two GPUs:", d["config"]["peer"], d["config"]["gop_steps_ran_as"], d["value"])
