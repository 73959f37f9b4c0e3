"""fpga-mpeg2-encoder_amd — MI355X-native MPEG-2 I/P encoder, drop-in for RTL/mpeg2encoder.v.

This package is plumbing: it loads libm2v_mi355x.so (hand-written HIP for gfx950 behind the
C-ABI of include/m2v_mi355x.h) through ctypes and mirrors the module's port contract
(parameters XL/YL/VECTOR_LEVEL/Q_LEVEL; beats in; 32-byte stream words out).  There is no CPU
fallback: without the built library or without a GPU every entry point raises.
"""
import ctypes
import os
import sys
import threading

import numpy as np

from . import build as _build
from . import synth  # noqa: F401  (seeded synthetic clips, numpy only)
from . import parallel  # noqa: F401  (multi-GPU: independent sequences, macroblock-row strips)
from . import container  # noqa: F401  (CPU-side conveniences: stream scan, MPEG-PS / TS multiplexers)
from . import decoder  # noqa: F401  (analysis tool: MPEG-2 ES decoder written from ISO/IEC 13818-2, CPU, not on the encode path)

_HERE = os.path.dirname(os.path.abspath(__file__))
# M2V_LIB: development hook for same-box A/B timing of two builds (tools/ab.sh); never set otherwise
LIB_PATH = os.environ.get("M2V_LIB") or os.path.join(_HERE, "libm2v_mi355x.so")
# the -DM2V_DEBUG build (level dump, "keep_recon", "ablate"): stage-level parity tests and profiling scripts only
LIB_DBG_PATH = os.environ.get("M2V_LIB_DBG") or os.path.join(_HERE, "libm2v_mi355x_dbg.so")

_libs = {}

EXPORTS = [
    "m2v_version", "m2v_create", "m2v_destroy", "m2v_reset", "m2v_push_beats", "m2v_push_packed", "m2v_push_frames", "m2v_push_frames_pull",
    "m2v_sequence_stop", "m2v_busy", "m2v_pull", "m2v_geometry", "m2v_encode_resident", "m2v_encode_resident_begin",
    "m2v_encode_resident_end", "m2v_set_option",
    "m2v_kernel_stats", "m2v_debug_read", "m2v_last_error", "m2v_debug_table",
    "m2v_strip_begin", "m2v_strip_info", "m2v_strip_step", "m2v_strip_step_edges", "m2v_strip_step_interior", "m2v_strip_halo_in", "m2v_strip_finish", "m2v_strip_assemble",
    "m2v_strip_finish_async", "m2v_strip_offsets", "m2v_strip_encode", "m2v_strip_stats",
    "m2v_comm_unique_id", "m2v_comm_init_rccl", "m2v_comm_init_local", "m2v_comm_init_solo", "m2v_comm_destroy", "m2v_comm_last_error", "m2v_comm_selftest",
    "m2v_comm_init_solo_rccl", "m2v_comm_selftest_captured", "m2v_strip_graph_stats",
    "m2v_comm_init_callbacks", "m2v_comm_init_peer", "m2v_comm_peer_export", "m2v_comm_peer_connect", "m2v_comm_peer_connect_all",
    "m2v_comm_peer_stats", "m2v_comm_kind", "m2v_strip_last_form", "m2v_upload_wait", "m2v_device_pci_bus_id",
    "m2v_strip_encode_begin", "m2v_strip_encode_end",
]

PEER_DESC_BYTES = 128          # M2V_PEER_DESC_BYTES


class CommCallbacks(ctypes.Structure):
    """m2v_comm_callbacks (include/m2v_mi355x.h): the exchange supplied by the caller"""
    HALO = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.c_size_t, ctypes.c_void_p)
    ALLGATHER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
    GATHER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t),
                              ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p)
    _fields_ = [("halo", HALO), ("allgather_u64", ALLGATHER), ("gather", GATHER), ("user", ctypes.c_void_p)]


class M2VError(RuntimeError):
    pass


def lib(debug=False):
    """The C-ABI shared library; raises if it has not been built (no silent fallback).
    debug=True: the -DM2V_DEBUG build of the same sources (tests and profiling only)."""
    path = LIB_DBG_PATH if debug else LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise M2VError("%s is missing: run __graft_entry__.build() "
                           "(hipcc, gfx950); there is no CPU fallback" % os.path.basename(path))
        # One process must hold ONE HIP runtime.  PyTorch-ROCm (used here only for device memory and
        # streams) bundles its own libamdhip64; importing it first makes the loader resolve this
        # library's libamdhip64.so.7 dependency to the copy torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(path)
        vp, sz, u32, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int
        L.m2v_version.restype = ctypes.c_char_p
        L.m2v_create.restype = vp
        L.m2v_create.argtypes = [ci, ci, ci, ci, ci, ctypes.POINTER(ci)]
        L.m2v_destroy.argtypes = [vp]
        L.m2v_reset.argtypes = [vp]
        L.m2v_push_beats.argtypes = [vp, u32, u32, u32, vp, vp, vp, sz, ci]
        L.m2v_push_packed.argtypes = [vp, u32, u32, u32, vp, sz, ci, ci]
        L.m2v_push_frames.argtypes = [vp, u32, u32, u32, vp, sz]
        L.m2v_sequence_stop.argtypes = [vp]
        L.m2v_busy.argtypes = [vp]
        L.m2v_pull.restype = ctypes.c_longlong
        L.m2v_pull.argtypes = [vp, vp, sz, ctypes.POINTER(ci)]
        L.m2v_geometry.argtypes = [vp, u32, u32, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.m2v_encode_resident.argtypes = [vp, u32, u32, u32, vp, sz, vp, sz, ctypes.POINTER(sz), vp]
        L.m2v_encode_resident_begin.argtypes = [vp, u32, u32, u32, vp, sz, vp, sz, vp]
        L.m2v_encode_resident_end.argtypes = [vp, ctypes.POINTER(sz)]
        L.m2v_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_longlong]
        L.m2v_kernel_stats.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.m2v_debug_read.restype = ctypes.c_longlong
        L.m2v_debug_read.argtypes = [vp, ci, vp, sz]
        L.m2v_last_error.restype = ctypes.c_char_p
        L.m2v_last_error.argtypes = [vp]
        L.m2v_debug_table.argtypes = [ci, ci, ci]
        L.m2v_strip_begin.argtypes = [vp, u32, u32, u32, vp, sz, ci, ci, vp]
        L.m2v_strip_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(sz)]
        L.m2v_strip_step.argtypes = [vp, ci, vp, vp]
        L.m2v_strip_step_edges.argtypes = [vp, ci, vp, vp]
        L.m2v_strip_step_interior.argtypes = [vp, ci]
        L.m2v_strip_halo_in.argtypes = [vp, ci, vp, vp]
        L.m2v_strip_finish.argtypes = [vp, vp, sz, vp]
        L.m2v_strip_assemble.argtypes = [vp, u32, u32, u32, sz, ci, vp, vp, vp, sz, ctypes.POINTER(sz), vp]
        try:
            L.m2v_strip_finish_async.argtypes = [vp, vp, sz]
            L.m2v_strip_offsets.argtypes = [vp, vp]
            L.m2v_strip_encode.argtypes = [vp, vp, ci, ci, ci, u32, u32, u32, vp, sz, vp, sz, ctypes.POINTER(sz), vp]
            dp = ctypes.POINTER(ctypes.c_double)
            L.m2v_strip_stats.argtypes = [vp, dp, dp, dp, dp, dp]
            L.m2v_comm_init_solo.restype = vp
            L.m2v_comm_init_solo.argtypes = [ci, ctypes.POINTER(ci)]
            L.m2v_comm_unique_id.argtypes = [vp, sz]
            L.m2v_comm_init_rccl.restype = vp
            L.m2v_comm_init_rccl.argtypes = [vp, ci, ci, ci, ctypes.POINTER(ci)]
            L.m2v_comm_init_local.restype = vp
            L.m2v_comm_init_local.argtypes = [ci, ctypes.POINTER(ci)]
            L.m2v_comm_destroy.argtypes = [vp]
            L.m2v_comm_last_error.restype = ctypes.c_char_p
            L.m2v_comm_selftest.argtypes = [vp, ci, vp, vp, sz, vp]
            L.m2v_comm_init_solo_rccl.restype = vp
            L.m2v_comm_init_solo_rccl.argtypes = [ci, ctypes.POINTER(ci)]
            L.m2v_comm_selftest_captured.argtypes = [vp, ci, vp, vp, sz, vp, ci]
            ip = ctypes.POINTER(ci)
            L.m2v_strip_graph_stats.argtypes = [vp, ip, ip, ip]
            L.m2v_comm_init_callbacks.restype = vp
            L.m2v_comm_init_callbacks.argtypes = [ci, ctypes.POINTER(CommCallbacks), ip]
            L.m2v_comm_init_peer.restype = vp
            L.m2v_comm_init_peer.argtypes = [vp, ci, ci, sz, ip]
            L.m2v_comm_peer_export.argtypes = [vp, vp, sz]
            L.m2v_comm_peer_connect.argtypes = [vp, vp, vp]
            L.m2v_comm_peer_connect_all.argtypes = [vp]
            ullp = ctypes.POINTER(ctypes.c_ulonglong)
            L.m2v_comm_peer_stats.argtypes = [vp, ullp, ullp]
            L.m2v_comm_kind.restype = ctypes.c_char_p
            L.m2v_comm_kind.argtypes = [vp]
            L.m2v_strip_last_form.argtypes = [vp]
            L.m2v_upload_wait.argtypes = [vp]
            L.m2v_push_frames_pull.restype = ctypes.c_longlong
            L.m2v_push_frames_pull.argtypes = [vp, u32, u32, u32, vp, sz, vp, sz, ctypes.POINTER(ci)]
            L.m2v_device_pci_bus_id.argtypes = [ci, ctypes.c_char_p, sz]
            L.m2v_strip_encode_begin.argtypes = [vp, vp, ci, ci, ci, u32, u32, u32, vp, sz, vp, sz, vp]
            L.m2v_strip_encode_end.argtypes = [vp, ctypes.POINTER(sz)]
        except AttributeError:
            # an OLDER build handed in through M2V_LIB for a same-box A/B (tools/ab.sh) may lack the newer entry points; the library of
            # this tree must have every one of them (tests/test_abi.py)
            if not os.environ.get("M2V_LIB"):
                raise
        _libs[path] = L
    return _libs[path]


def build(force=False, verbose=False):
    return _build.build(force=force, verbose=verbose)


def device_pci_bus_id(device=0):
    """PCI address of HIP device `device` as sysfs spells it ("0000:c1:00.0"), or None (m2v_device_pci_bus_id)"""
    buf = ctypes.create_string_buffer(64)
    n = lib().m2v_device_pci_bus_id(int(device), buf, 64)
    return buf.value.decode() if n > 0 else None


def clamp_geometry(xsize16, ysize16, XL=7, YL=7):
    """Clamped (W, H) of RTL/mpeg2encoder.v:985-1006 (pure host arithmetic)."""
    def c(s, L):
        s &= (2 << L) - 1
        lim = 1 << L
        return lim - 1 if s > lim else 3 if s < 4 else s - 1
    return 16 * (c(xsize16, XL) + 1), 16 * (c(ysize16, YL) + 1)


class Mpeg2Encoder:
    """`mpeg2encoder #(XL, YL, VECTOR_LEVEL, Q_LEVEL)` on one MI355X (RTL/mpeg2encoder.v:10-38)."""

    def __init__(self, XL=6, YL=6, VECTOR_LEVEL=3, Q_LEVEL=2, device=0, debug=False):
        self.params = (XL, YL, VECTOR_LEVEL, Q_LEVEL)
        self._geom = {}
        err = ctypes.c_int(0)
        self._L = lib(debug)
        self._h = self._L.m2v_create(XL, YL, VECTOR_LEVEL, Q_LEVEL, device, ctypes.byref(err))
        if not self._h:
            raise M2VError("m2v_create failed with code %d (parameters %r, device %d): %s"
                           % (err.value, self.params, device, self._L.m2v_last_error(None).decode()))
        if os.environ.get("M2V_DCT_MFMA") in ("0", "1"):      # development hook: A/B runs of the DCT-as-GEMM variant of the kernel
            self.set_option("dct_mfma", int(os.environ["M2V_DCT_MFMA"]))

    def close(self):
        if getattr(self, "_h", None):
            self._L.m2v_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, r, what):
        if r < 0:
            raise M2VError("%s failed (%d): %s" % (what, r, self._L.m2v_last_error(self._h).decode()))
        return r

    def reset(self):
        """m2v_reset (`rstn` low, RTL:1028-1039): whatever is in flight on the handle is waited for and dropped, the handle is idle again"""
        self._chk(self._L.m2v_reset(self._h), "m2v_reset")

    def set_option(self, name, value):
        self._chk(self._L.m2v_set_option(self._h, name.encode(), int(value)), "m2v_set_option(%s)" % name)

    def geometry(self, xsize16, ysize16):
        # (a handle's clamps are fixed at creation: asked once per size - the per-call bindings below are on a caller's critical path)
        got = self._geom.get((xsize16, ysize16))
        if got is None:
            w, h = ctypes.c_int(), ctypes.c_int()
            self._chk(self._L.m2v_geometry(self._h, xsize16, ysize16, ctypes.byref(w), ctypes.byref(h)), "m2v_geometry")
            got = self._geom[(xsize16, ysize16)] = (w.value, h.value)
        return got

    @staticmethod
    def _flat_u8(a):
        """the array as contiguous bytes: itself when it already is (no copy, no new object beyond a view)"""
        if isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]:
            return a
        return np.ascontiguousarray(a, np.uint8)

    # ---- port-level interface ----
    def push_beats(self, xsize16, ysize16, pframes_count, y4, u4, v4, stop_with_last=False):
        y4 = np.ascontiguousarray(y4, np.uint8).reshape(-1)
        u4 = np.ascontiguousarray(u4, np.uint8).reshape(-1)
        v4 = np.ascontiguousarray(v4, np.uint8).reshape(-1)
        assert y4.size == u4.size == v4.size and y4.size % 4 == 0
        self._chk(self._L.m2v_push_beats(self._h, xsize16, ysize16, pframes_count, y4.ctypes.data, u4.ctypes.data,
                                         v4.ctypes.data, y4.size // 4, int(bool(stop_with_last))), "m2v_push_beats")

    PACKED = {"yuv24": (0, 3), "uyv24": (1, 3), "yuvx32": (2, 4), "ayuv32": (3, 4)}

    def push_packed(self, xsize16, ysize16, pframes_count, pixels, layout="yuv24", stop_with_last=False):
        """pixels: packed 4:4:4 samples in raster order ([..., bytes_per_pixel] uint8), a multiple of 4 pixels."""
        code, bpp = self.PACKED[layout]
        p = np.ascontiguousarray(pixels, np.uint8).reshape(-1)
        assert p.size % (4 * bpp) == 0
        self._chk(self._L.m2v_push_packed(self._h, xsize16, ysize16, pframes_count, p.ctypes.data, p.size // (4 * bpp),
                                          code, int(bool(stop_with_last))), "m2v_push_packed")

    def push_frames(self, xsize16, ysize16, pframes_count, frames444):
        W, H = self.geometry(xsize16, ysize16)
        f = self._flat_u8(frames444)
        assert f.size % (3 * W * H) == 0
        self._chk(self._L.m2v_push_frames(self._h, xsize16, ysize16, pframes_count, f.ctypes.data,
                                          f.size // (3 * W * H)), "m2v_push_frames")

    def push_frames_pull(self, xsize16, ysize16, pframes_count, frames444, dst, offset=0):
        """m2v_push_frames_pull: push_frames + pull_into(dst, offset) in one call, the stream bytes copied while the frames upload:
        -> (bytes written, last)"""
        W, H = self.geometry(xsize16, ysize16)
        f = self._flat_u8(frames444)
        assert f.size % (3 * W * H) == 0
        assert dst.dtype == np.uint8 and dst.flags["C_CONTIGUOUS"]
        last = ctypes.c_int(0)
        n = self._chk(self._L.m2v_push_frames_pull(self._h, xsize16, ysize16, pframes_count, f.ctypes.data, f.size // (3 * W * H),
                                                   dst.ctypes.data + offset, (dst.size - offset) & ~31, ctypes.byref(last)), "m2v_push_frames_pull")
        return n, bool(last.value)

    def upload_wait(self):
        """option direct_upload = 2: returns when every frame handed to push_frames so far has been read"""
        self._chk(self._L.m2v_upload_wait(self._h), "m2v_upload_wait")

    def sequence_stop(self):
        self._chk(self._L.m2v_sequence_stop(self._h), "m2v_sequence_stop")

    @property
    def busy(self):
        return bool(self._L.m2v_busy(self._h))

    def pull(self, max_bytes=1 << 20):
        """-> (bytes, last)"""
        # one buffer per handle, kept: a fresh 16 MB numpy array per call is a fresh mapping whose pages fault in one by one
        buf = getattr(self, "_pullbuf", None)
        if buf is None or buf.size < (max_bytes & ~31):
            buf = self._pullbuf = np.empty(max_bytes & ~31, np.uint8)
        last = ctypes.c_int(0)
        n = self._chk(self._L.m2v_pull(self._h, buf.ctypes.data, max_bytes & ~31, ctypes.byref(last)), "m2v_pull")
        return buf[:n].tobytes(), bool(last.value)

    def pull_into(self, dst, offset=0):
        """m2v_pull straight into the caller's uint8 array (from `offset` on, whole 32-byte words): -> (bytes written, last).  What a
        C caller does: no intermediate object."""
        assert dst.dtype == np.uint8 and dst.flags["C_CONTIGUOUS"]
        last = ctypes.c_int(0)
        n = self._chk(self._L.m2v_pull(self._h, dst.ctypes.data + offset, (dst.size - offset) & ~31, ctypes.byref(last)), "m2v_pull")
        return n, bool(last.value)

    def pull_all(self):
        out = []
        while True:
            b, last = self.pull()
            out.append(b)
            if last or not b:
                break
        return b"".join(out)

    def encode(self, frames444, xsize16, ysize16, pframes_count, nbeats=None):
        """One whole sequence from host memory through the beat interface; returns the stream bytes."""
        W, H = self.geometry(xsize16, ysize16)
        f = np.ascontiguousarray(frames444, np.uint8).reshape(-1, 3, H * W)
        bpf = W * H // 4
        total = f.shape[0] * bpf if nbeats is None else nbeats
        full = total // bpf
        if full:
            self.push_frames(xsize16, ysize16, pframes_count, f[:full])
        rem = total - full * bpf
        if rem:
            fr = f[full]
            self.push_beats(xsize16, ysize16, pframes_count, fr[0][:rem * 4], fr[1][:rem * 4], fr[2][:rem * 4])
        self.sequence_stop()
        return self.pull_all()

    # ---- HBM-resident interface (what bench.py times) ----
    def encode_resident(self, d_frames_ptr, nframes, d_out_ptr, cap, xsize16, ysize16, pframes_count, stream=0):
        n = ctypes.c_size_t(0)
        self._chk(self._L.m2v_encode_resident(self._h, xsize16, ysize16, pframes_count, d_frames_ptr, nframes,
                                              d_out_ptr, cap, ctypes.byref(n), stream), "m2v_encode_resident")
        return n.value

    def encode_resident_begin(self, d_frames_ptr, nframes, d_out_ptr, cap, xsize16, ysize16, pframes_count, stream=0):
        """enqueue a whole sequence and return; encode_resident_end() waits for it and returns the byte count"""
        self._chk(self._L.m2v_encode_resident_begin(self._h, xsize16, ysize16, pframes_count, d_frames_ptr, nframes, d_out_ptr, cap,
                                                    stream), "m2v_encode_resident_begin")

    def encode_resident_end(self):
        n = ctypes.c_size_t(0)
        self._chk(self._L.m2v_encode_resident_end(self._h, ctypes.byref(n)), "m2v_encode_resident_end")
        return n.value

    # ---- strip mode (config c5): see include/m2v_mi355x.h ----
    def strip_begin(self, d_frames_ptr, nframes, xsize16, ysize16, pframes_count, row0, row1, stream=0):
        self._chk(self._L.m2v_strip_begin(self._h, xsize16, ysize16, pframes_count, d_frames_ptr, nframes, row0, row1,
                                          stream), "m2v_strip_begin")
        steps, hb = ctypes.c_int(0), ctypes.c_size_t(0)
        self._chk(self._L.m2v_strip_info(self._h, ctypes.byref(steps), ctypes.byref(hb)), "m2v_strip_info")
        return steps.value, hb.value

    def strip_step(self, j, send_up_ptr, send_down_ptr):
        return self._chk(self._L.m2v_strip_step(self._h, j, send_up_ptr, send_down_ptr), "m2v_strip_step")

    def strip_step_edges(self, j, send_up_ptr, send_down_ptr):
        return self._chk(self._L.m2v_strip_step_edges(self._h, j, send_up_ptr, send_down_ptr), "m2v_strip_step_edges")

    def strip_step_interior(self, j):
        self._chk(self._L.m2v_strip_step_interior(self._h, j), "m2v_strip_step_interior")

    def strip_halo_in(self, j, from_up_ptr, from_down_ptr):
        self._chk(self._L.m2v_strip_halo_in(self._h, j, from_up_ptr, from_down_ptr), "m2v_strip_halo_in")

    def strip_finish(self, d_strip_ptr, cap, nframes):
        off = np.zeros(nframes + 1, np.uint64)
        self._chk(self._L.m2v_strip_finish(self._h, d_strip_ptr, cap, off.ctypes.data), "m2v_strip_finish")
        return off

    def strip_finish_async(self, d_strip_ptr, cap):
        self._chk(self._L.m2v_strip_finish_async(self._h, d_strip_ptr, cap), "m2v_strip_finish_async")

    def strip_offsets(self, nframes):
        off = np.zeros(nframes + 1, np.uint64)
        self._chk(self._L.m2v_strip_offsets(self._h, off.ctypes.data), "m2v_strip_offsets")
        return off

    def strip_encode(self, comm, rank, world, d_frames_ptr, nframes, xsize16, ysize16, pframes_count, d_out_ptr=None, cap=0, dst=0, stream=0):
        """m2v_strip_encode: this rank's strip of one sequence, exchange included, in one native call.  `comm`: a StripComm
        (None for world == 1).  Returns the stream's byte count on rank `dst`, 0 elsewhere."""
        n = ctypes.c_size_t(0)
        self._chk(self._L.m2v_strip_encode(self._h, comm.handle if comm is not None else None, rank, world, dst, xsize16, ysize16,
                                           pframes_count, d_frames_ptr, nframes, d_out_ptr, cap, ctypes.byref(n), stream), "m2v_strip_encode")
        return n.value

    def strip_encode_begin(self, comm, rank, world, d_frames_ptr, nframes, xsize16, ysize16, pframes_count, d_out_ptr=None, cap=0, dst=0, stream=0):
        """m2v_strip_encode_begin: the sequence's GOP steps and this strip's slices are enqueued, nothing is waited for.  Two handles
        (a peer communicator each, over ONE base communicator) taking turns from one thread keep two strip sequences in flight."""
        self._chk(self._L.m2v_strip_encode_begin(self._h, comm.handle if comm is not None else None, rank, world, dst, xsize16, ysize16,
                                                 pframes_count, d_frames_ptr, nframes, d_out_ptr, cap, stream), "m2v_strip_encode_begin")

    def strip_encode_end(self):
        """m2v_strip_encode_end: the sizes all-gather, the one host wait, the strips to the output rank, the assembly there.
        Returns the stream's byte count on the output rank, 0 elsewhere."""
        n = ctypes.c_size_t(0)
        self._chk(self._L.m2v_strip_encode_end(self._h, ctypes.byref(n)), "m2v_strip_encode_end")
        return n.value

    def strip_stats(self):
        """-> dict of the last strip_encode: steps, host_us_per_step, and (option profile) halo_total / halo_exposed / gather in ms"""
        v = [ctypes.c_double(0) for _ in range(5)]
        steps = self._chk(self._L.m2v_strip_stats(self._h, *[ctypes.byref(x) for x in v]), "m2v_strip_stats")
        return {"steps": steps, "halo_total": v[0].value, "halo_exposed": v[1].value, "gather": v[2].value, "host_us_per_step": v[3].value,
                "comm_us_per_step": v[4].value, "host_us_per_step_outside_comm": v[3].value - v[4].value}

    def strip_last_form(self):
        """how the last strip_encode ran its GOP steps: "calls", "graph" or "peer" (m2v_strip_last_form)"""
        return ("calls", "graph", "peer")[self._chk(self._L.m2v_strip_last_form(self._h), "m2v_strip_last_form")]

    def strip_graph_stats(self):
        """-> dict: was the last strip_encode launched as a recorded hipGraph, how many recordings / launches so far, and whether
        recording has failed on this handle (the sequence is then enqueued call by call)"""
        v = [ctypes.c_int(0) for _ in range(3)]
        broken = self._chk(self._L.m2v_strip_graph_stats(self._h, *[ctypes.byref(x) for x in v]), "m2v_strip_graph_stats")
        return {"last_call_was_graph": bool(v[0].value), "recordings": v[1].value, "launches": v[2].value, "broken": bool(broken)}

    def strip_assemble(self, strip_ptrs, frame_offs, nframes, d_out_ptr, cap, xsize16, ysize16, pframes_count, stream=0):
        n = len(strip_ptrs)
        ptrs = (ctypes.c_void_p * n)(*strip_ptrs)
        offs = [np.ascontiguousarray(o, np.uint64) for o in frame_offs]
        optrs = (ctypes.c_void_p * n)(*[o.ctypes.data for o in offs])
        out = ctypes.c_size_t(0)
        self._chk(self._L.m2v_strip_assemble(self._h, xsize16, ysize16, pframes_count, nframes, n, ptrs, optrs, d_out_ptr,
                                             cap, ctypes.byref(out), stream), "m2v_strip_assemble")
        return out.value

    def kernel_stats(self, kernel):
        ms, units = ctypes.c_double(0), ctypes.c_double(0)
        n = self._chk(self._L.m2v_kernel_stats(self._h, kernel, ctypes.byref(ms), ctypes.byref(units)), "m2v_kernel_stats")
        return n, ms.value, units.value

    def debug_read(self, what, nbytes, dtype):
        buf = np.zeros(nbytes, np.uint8)
        n = self._chk(self._L.m2v_debug_read(self._h, what, buf.ctypes.data, nbytes), "m2v_debug_read")
        return buf[:n].view(dtype)


class StripComm:
    """The exchange between the strips of config c5 (include/m2v_mi355x.h, csrc/m2v_comm.hpp): RCCL between processes, or
    mailboxes between the threads of one process."""

    def __init__(self, handle, kind, world):
        self.handle, self.kind, self.world = handle, kind, world

    @classmethod
    def rccl(cls, rank, world, device, dist=None, init_timeout=None):
        """Collective over `dist` (an initialised torch.distributed of any backend, used ONLY to hand rank 0's ncclUniqueId to
        the others); afterwards the data path talks to librccl directly.

        Rank 0 broadcasts (status, id) whatever happened: when librccl cannot produce an id there, EVERY rank raises here,
        together, and nobody is left waiting in the broadcast.  ncclCommInitRank itself blocks until all ranks have arrived; a rank
        that cannot get there (its own librccl missing) would leave the others inside it, so with `init_timeout` (seconds) a rank
        still inside after that long ends its process with exit code 70 - a launcher that watches its ranks (bench.py) then tears
        the job down instead of hanging."""
        L = lib()
        status, ident = 128, None
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            status = L.m2v_comm_unique_id(buf, 128)
            ident = buf.raw if status == 128 else L.m2v_comm_last_error().decode()
        if world > 1:
            box = [(status, ident)]
            dist.broadcast_object_list(box, src=0)
            status, ident = box[0]
        if status != 128:
            raise M2VError("m2v_comm_unique_id failed on rank 0 (%d): %s" % (status, ident))
        err = ctypes.c_int(0)
        done = threading.Event()
        if init_timeout:
            def watchdog():
                if not done.wait(init_timeout):
                    sys.stderr.write("m2v: rank %d still inside m2v_comm_init_rccl after %.0f s (another rank never arrived?): giving up\n"
                                     % (rank, init_timeout))
                    sys.stderr.flush()
                    os._exit(70)
            threading.Thread(target=watchdog, daemon=True).start()
        try:
            h = L.m2v_comm_init_rccl(ident, rank, world, device, ctypes.byref(err))
        finally:
            done.set()
        if not h:
            raise M2VError("m2v_comm_init_rccl failed (%d): %s" % (err.value, L.m2v_comm_last_error().decode()))
        return cls(h, "rccl", world)

    @classmethod
    def local(cls, world, debug=False):
        """debug=True: made by the -DM2V_DEBUG library (for handles of that library: a communicator and its users come from ONE library)"""
        L = lib(debug)
        err = ctypes.c_int(0)
        h = L.m2v_comm_init_local(world, ctypes.byref(err))
        if not h:
            raise M2VError("m2v_comm_init_local failed (%d): %s" % (err.value, L.m2v_comm_last_error().decode()))
        c = cls(h, "local", world)
        c._debug = debug
        return c

    @classmethod
    def solo(cls, world, rccl=False, debug=False):
        """timing aid (tools/strip_solo.py): one rank of `world` alone on its GPU; the output is NOT a valid stream.
        rccl=True: the rows travel through a 1-rank RCCL communicator (ncclSend / ncclRecv to itself) instead of device copies"""
        L = lib(debug)
        err = ctypes.c_int(0)
        h = (L.m2v_comm_init_solo_rccl if rccl else L.m2v_comm_init_solo)(world, ctypes.byref(err))
        if not h:
            raise M2VError("m2v_comm_init_solo%s failed (%d): %s" % ("_rccl" if rccl else "", err.value, L.m2v_comm_last_error().decode()))
        c = cls(h, "solo-rccl" if rccl else "solo", world)
        c._debug = debug
        return c

    @classmethod
    def callbacks(cls, world, halo, allgather_u64, gather, debug=False):
        """The exchange supplied by the caller (m2v_comm_init_callbacks).  The three Python callables get the C arguments of
        m2v_comm_callbacks without `user` - device addresses as ints, the HIP stream as an int - and return 0 for success; an exception
        is reported as failure.  The callback objects live as long as the communicator."""
        L = lib(debug)

        def guard(fn):
            def run(user, *a):
                try:
                    return int(fn(*a) or 0)
                except BaseException as ex:  # noqa: BLE001  (nothing may unwind into C)
                    sys.stderr.write("m2v: communicator callback failed: %r\n" % (ex,))
                    return 1
            return run
        cb = CommCallbacks(CommCallbacks.HALO(guard(halo)), CommCallbacks.ALLGATHER(guard(allgather_u64)), CommCallbacks.GATHER(guard(gather)), None)
        err = ctypes.c_int(0)
        h = L.m2v_comm_init_callbacks(world, ctypes.byref(cb), ctypes.byref(err))
        if not h:
            raise M2VError("m2v_comm_init_callbacks failed (%d): %s" % (err.value, L.m2v_comm_last_error().decode()))
        c = cls(h, "callbacks", world)
        c._keep, c._debug = cb, debug
        return c

    @classmethod
    def peer(cls, base, rank, device=0, halo_bytes=0, connect=True):
        """The peer transport on top of `base` (m2v_comm_init_peer): rows stored straight into the neighbours' landing blocks by the
        macroblock kernel.  connect=True: m2v_comm_peer_connect_all - collective over `base` (every rank of it makes this call).
        `base` stays alive as long as the result (close this one first)."""
        L = lib(getattr(base, "_debug", False))
        err = ctypes.c_int(0)
        h = L.m2v_comm_init_peer(base.handle, rank, device, halo_bytes, ctypes.byref(err))
        if not h:
            raise M2VError("m2v_comm_init_peer failed (%d): %s" % (err.value, L.m2v_comm_last_error().decode()))
        c = cls(h, L.m2v_comm_kind(h).decode(), base.world)
        c._base, c._debug = base, getattr(base, "_debug", False)
        if connect:
            r = L.m2v_comm_peer_connect_all(h)
            if r < 0:
                msg = L.m2v_comm_last_error().decode()
                c.close()
                raise M2VError("m2v_comm_peer_connect_all failed (%d): %s" % (r, msg))
        return c

    def peer_export(self):
        buf = ctypes.create_string_buffer(PEER_DESC_BYTES)
        r = lib(getattr(self, "_debug", False)).m2v_comm_peer_export(self.handle, buf, PEER_DESC_BYTES)
        if r < 0:
            raise M2VError("m2v_comm_peer_export failed (%d): %s" % (r, lib().m2v_comm_last_error().decode()))
        return buf.raw

    def peer_connect(self, desc_up, desc_down):
        r = lib(getattr(self, "_debug", False)).m2v_comm_peer_connect(self.handle, desc_up, desc_down)
        if r < 0:
            raise M2VError("m2v_comm_peer_connect failed (%d): %s" % (r, lib().m2v_comm_last_error().decode()))

    def peer_stats(self):
        """-> dict: sequences that ran in the peer form, waits that gave up, and whether the communicator has fallen back to its base for good"""
        a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        r = lib(getattr(self, "_debug", False)).m2v_comm_peer_stats(self.handle, ctypes.byref(a), ctypes.byref(b))
        if r < 0:
            raise M2VError("m2v_comm_peer_stats: not a peer communicator")
        return {"peer_sequences": a.value, "giveups": b.value, "fell_back": bool(r)}

    def selftest(self, rank, d_send_ptr, d_recv_ptr, nbytes, stream=0):
        r = lib().m2v_comm_selftest(self.handle, rank, d_send_ptr, d_recv_ptr, nbytes, stream)
        if r < 0:
            raise M2VError("m2v_comm_selftest failed (%d): %s" % (r, lib().m2v_comm_last_error().decode()))

    def selftest_captured(self, rank, d_send_ptr, d_recv_ptr, nbytes, stream=0, launches=2):
        """the same pair recorded into a hipGraph and launched `launches` times (can this transport be part of a recorded strip sequence?)"""
        r = lib().m2v_comm_selftest_captured(self.handle, rank, d_send_ptr, d_recv_ptr, nbytes, stream, launches)
        if r < 0:
            raise M2VError("m2v_comm_selftest_captured failed (%d): %s" % (r, lib().m2v_comm_last_error().decode()))

    def close(self):
        if getattr(self, "handle", None):
            lib(getattr(self, "_debug", False)).m2v_comm_destroy(self.handle)
            self.handle = None
