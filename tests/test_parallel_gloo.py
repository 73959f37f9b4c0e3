"""Multi-process (gloo, CPU) tests of the N > 1 paths: the strip orchestration of parallel.encode_strips — which
rows go to which neighbour at which GOP step, the size all-gather, the gather of the strips and the final assembly —
and the per-rank work split of the independent-sequence mode.  The GPU kernels are replaced by an engine that serves
rows / slices computed by the CPU oracle, and that ASSERTS every halo it receives equals the neighbour's true rows."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import m2v_load  # noqa: E402

M = m2v_load.load()


class OracleStripEngine:
    def __init__(self, clip, xs16, ys16, pf, VL=3, Q=2):
        import torch
        from oracle import m2v_oracle_ctypes as orc
        self.torch = torch
        self.n, self.pf, self.VL = clip.shape[0], pf, VL
        self.W, self.H = orc.geometry(xs16, ys16)
        self.mbw, self.mbh = self.W // 16, self.H // 16
        self.stream_bytes, d = orc.encode(clip, xs16, ys16, pf, 7, 7, VL, Q, dump=True)
        self.recon = d["recon"]
        bits = d["mb_bits"].reshape(self.n, self.mbh, self.mbw).astype(np.int64)
        self.slice_bytes = (bits.sum(axis=2) + 38 + 7) // 8                       # [frame][row]
        self.hdr = [(8 + 17) if f % (pf + 1) == 0 else 18 for f in range(self.n)]
        self.frame_pos, pos = [], 34
        for f in range(self.n):
            self.frame_pos.append(pos)
            pos += self.hdr[f] + int(self.slice_bytes[f].sum())
        self.checked = 0

    # step structure identical to the C++ planner: step j = j-th frame of every GOP; halo = frames referenced later
    def _frames(self, j):
        gop = self.pf + 1
        return [g * gop + j for g in range((self.n + gop - 1) // gop) if g * gop + j < self.n and j < gop]

    def _halo_frames(self, j):
        return [f for f in self._frames(j) if f % (self.pf + 1) < self.pf and f != self.n - 1]

    def _rows(self, f, y0, yc0):
        YR, UR, W, H = 2 * self.VL, self.VL, self.W, self.H
        r = self.recon[f]
        Y = r[:W * H].reshape(H, W)
        U = r[W * H:W * H + W * H // 4].reshape(H // 2, W // 2)
        V = r[W * H + W * H // 4:].reshape(H // 2, W // 2)
        return np.concatenate([Y[y0:y0 + YR].reshape(-1), U[yc0:yc0 + UR].reshape(-1), V[yc0:yc0 + UR].reshape(-1)])

    def begin(self, row0, row1):
        self.phase = None
        self.row0, self.row1 = row0, row1
        self.chunk = 3 * self.VL * self.W
        steps = min(self.n, self.pf + 1)
        return steps, max(len(self._halo_frames(j)) for j in range(steps)) * self.chunk

    def alloc(self, nbytes):
        return self.torch.zeros(max(nbytes, 1), dtype=self.torch.uint8)

    def step(self, j, send_up, send_down):
        YR, UR = 2 * self.VL, self.VL
        for k, f in enumerate(self._halo_frames(j)):
            if self.row0 > 0:
                send_up[k * self.chunk:(k + 1) * self.chunk] = self.torch.from_numpy(self._rows(f, 16 * self.row0, 8 * self.row0))
            if self.row1 < self.mbh:
                send_down[k * self.chunk:(k + 1) * self.chunk] = self.torch.from_numpy(
                    self._rows(f, 16 * self.row1 - YR, 8 * self.row1 - UR))
        return len(self._halo_frames(j)) * self.chunk

    # the two-part step encode_strips() prefers (edge rows + halo first, interior rows while the halo travels);
    # the order of the calls is part of the contract
    def step_edges(self, j, send_up, send_down):
        assert self.phase == ("interior", j - 1) or (j == 0 and self.phase is None), self.phase
        self.phase = ("edges", j)
        return self.step(j, send_up, send_down)

    def step_interior(self, j):
        assert self.phase == ("edges", j), self.phase
        self.phase = ("interior", j)

    def halo_in(self, j, from_up, from_down):
        assert self.phase == ("interior", j), "halo_in comes after the interior rows were queued"
        YR, UR = 2 * self.VL, self.VL
        for k, f in enumerate(self._halo_frames(j)):
            if from_up is not None:
                assert np.array_equal(from_up[k * self.chunk:(k + 1) * self.chunk].numpy(),
                                      self._rows(f, 16 * self.row0 - YR, 8 * self.row0 - UR)), ("up", j, f)
                self.checked += 1
            if from_down is not None:
                assert np.array_equal(from_down[k * self.chunk:(k + 1) * self.chunk].numpy(),
                                      self._rows(f, 16 * self.row1, 8 * self.row1)), ("down", j, f)
                self.checked += 1

    def finish(self):
        parts, off = [], [0]
        for f in range(self.n):
            a = self.frame_pos[f] + self.hdr[f] + int(self.slice_bytes[f][:self.row0].sum())
            b = a + int(self.slice_bytes[f][self.row0:self.row1].sum())
            parts.append(self.stream_bytes[a:b])
            off.append(off[-1] + b - a)
        return self.torch.from_numpy(np.frombuffer(b"".join(parts), np.uint8).copy()), np.array(off, np.int64)

    def assemble(self, strips, offs):
        out = bytearray(self.stream_bytes[:34])
        for f in range(self.n):
            out += self.stream_bytes[self.frame_pos[f]:self.frame_pos[f] + self.hdr[f]]
            for s, o in zip(strips, offs):
                out += s[int(o[f]):int(o[f + 1])].numpy().tobytes()
        out += b"\x00\x00\x01\xb7"
        out += b"\x00" * ((len(out) // 32 + 1) * 32 - len(out))
        return bytes(out)


def _worker(rank, world, port, W, H, n, pf, VL, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clip = M.synth.clip(W, H, n, clip_index=70)
        eng = OracleStripEngine(clip, W // 16, H // 16, pf, VL)

        def no_padded_gather(*a, **k):          # the strips travel as sized point-to-point transfers, not as a padded gather
            raise AssertionError("encode_strips must not use dist.gather")
        dist.gather = no_padded_gather
        sent = []
        real_batch = dist.batch_isend_irecv

        def counting_batch(ops):
            sent.extend((op.op.__name__, int(op.tensor.numel()), op.peer) for op in ops)
            return real_batch(ops)
        dist.batch_isend_irecv = counting_batch
        out = M.parallel.encode_strips(eng, rank, world, dist, dst=0)
        # the last batch is the gather of the strips: rank 0 receives exactly every other rank's strip size, the others send theirs
        strip_len = len(eng.finish()[0])
        if rank != 0:
            assert sent[-1] == ("isend", strip_len, 0), sent[-1]
        else:
            assert [x[0] for x in sent[-(world - 1):]] == ["irecv"] * (world - 1) and [x[2] for x in sent[-(world - 1):]] == list(range(1, world))
        ok = True if rank != 0 else (out == eng.stream_bytes)
        q.put((rank, bool(ok), eng.checked, M.parallel.partition_rows(H // 16, world)[rank]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,W,H,n,pf,VL", [(2, 96, 96, 7, 2, 3), (3, 64, 160, 5, 4, 1)])
def test_encode_strips_gloo(world, W, H, n, pf, VL):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, n, pf, VL, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    # interior ranks check two neighbours, edge ranks one; every frame that is referenced later is exchanged once
    halo_frames = sum(1 for f in range(n) if f % (pf + 1) < pf and f != n - 1)
    for rank, _, checked, _ in res:
        assert checked == halo_frames * (2 if 0 < rank < world - 1 else 1)
    assert [r[3] for r in res] == M.parallel.partition_rows(H // 16, world)


def _turns_worker(rank, world, port, q):
    """Two engines (two different sequences) taking turns on ONE process group: begin A, begin B, end A, begin A', end B, ... with the
    output rank rotating - the call order of m2v_strip_encode_begin / _end on two handles.  Every collective is logged; the log must
    be the same sequence of (kind, sequence) on every rank, and a sequence's sizes all-gather must come right before its strips."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H, pf, VL = 96, 96, 2, 2
        clips = [M.synth.clip(W, H, 7, clip_index=71), M.synth.clip(W, H, 5, clip_index=72)]
        engs = [OracleStripEngine(c, W // 16, H // 16, pf, VL) for c in clips]
        log, current = [], [None]
        real_ag, real_batch = dist.all_gather, dist.batch_isend_irecv

        def ag(out, t, *a, **k):
            log.append(("sizes", current[0]))
            return real_ag(out, t, *a, **k)

        def batch(ops):
            log.append(("p2p", current[0]))
            return real_batch(ops)
        dist.all_gather, dist.batch_isend_irecv = ag, batch
        P = M.parallel
        results, state, seq = [], [None, None], 0
        order = []
        for turn in range(6):                        # handles 0, 1, 0, 1, ...
            h = turn % 2
            if state[h] is not None:
                current[0] = state[h][1]
                order.append(("end", state[h][1]))
                results.append((state[h][1], state[h][2], P.encode_strips_end(state[h][0])))
            current[0] = seq
            dst = seq % world
            order.append(("begin", seq))
            state[h] = (P.encode_strips_begin(engs[h], rank, world, dist, dst=dst), seq, dst)
            seq += 1
        for h in (0, 1):
            current[0] = state[h][1]
            results.append((state[h][1], state[h][2], P.encode_strips_end(state[h][0])))
        ok = all((out == engs[k % 2].stream_bytes) if rank == dst else out is None for k, dst, out in results)
        # a sequence's sizes exchange is followed at once by its own strips' transfer (world 2: both ranks take part in it) - never by
        # another sequence's collectives
        tidy = all(i + 1 < len(log) and log[i + 1] == ("p2p", k) for i, (kind, k) in enumerate(log) if kind == "sizes")
        q.put((rank, bool(ok), bool(tidy), [x for x in log if x[0] == "sizes"], len(results)))
    finally:
        dist.destroy_process_group()


def test_two_strip_sequences_in_flight_from_one_thread_gloo():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_turns_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and tidy for _, ok, tidy, _, _ in res), res
    assert res[0][3] == res[1][3] == [("sizes", k) for k in range(6)]          # the sizes exchanges in sequence order, the same on every rank
    assert all(n == 6 for *_, n in res)


def test_partition_and_sequence_split():
    P = M.parallel
    for mbh in (4, 13, 72, 128):
        for world in (1, 2, 3, 4, 8):
            if world > mbh:
                with pytest.raises(ValueError):
                    P.partition_rows(mbh, world)
                continue
            rows = P.partition_rows(mbh, world)
            assert rows[0][0] == 0 and rows[-1][1] == mbh
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            sizes = [b - a for a, b in rows]
            assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    assert P.partition_rows(128, 8) == [(16 * r, 16 * r + 16) for r in range(8)]        # config c5: 16 rows per GPU
    got = sorted(s for r in range(4) for s in P.sequence_for_rank(10, r, 4))
    assert got == list(range(10))
