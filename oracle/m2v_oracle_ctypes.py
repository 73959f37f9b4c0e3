"""ctypes binding of the CPU oracle (oracle/libm2v_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libm2v_oracle.so")


class Params(ctypes.Structure):
    _fields_ = [("XL", ctypes.c_int), ("YL", ctypes.c_int),
                ("VECTOR_LEVEL", ctypes.c_int), ("Q_LEVEL", ctypes.c_int)]


class Dump(ctypes.Structure):
    _fields_ = [("mb_inter", ctypes.c_void_p), ("mb_mvx", ctypes.c_void_p), ("mb_mvy", ctypes.c_void_p),
                ("mb_cbp", ctypes.c_void_p), ("coef", ctypes.c_void_p), ("recon", ctypes.c_void_p),
                ("yuv420", ctypes.c_void_p), ("mb_bits", ctypes.c_void_p)]


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("m2v_oracle.c", "m2v_oracle.h", "m2v_tables.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-s", "-C", _HERE, "libm2v_oracle.so", "m2v_oracle_cli"])
    return _LIB


_libs = {}
_active = _LIB          # path of the library lib() talks to; only tests/test_oracle_mutants.py ever changes it


def _load(path):
    if path not in _libs:
        L = ctypes.CDLL(path)
        L.m2v_oracle_encode.restype = ctypes.c_size_t
        L.m2v_oracle_encode.argtypes = [ctypes.POINTER(Params), ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p]
        L.m2v_oracle_geometry.restype = ctypes.c_int
        L.m2v_oracle_geometry.argtypes = [ctypes.POINTER(Params), ctypes.c_uint, ctypes.c_uint,
                                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.m2v_oracle_frame_count.restype = ctypes.c_size_t
        L.m2v_oracle_frame_count.argtypes = [ctypes.POINTER(Params), ctypes.c_uint, ctypes.c_uint, ctypes.c_size_t]
        for name in ("fdct", "idct"):
            getattr(L, "m2v_oracle_" + name).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        for name in ("quant", "dequant"):
            getattr(L, "m2v_oracle_" + name).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.m2v_oracle_subsample.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _libs[path] = L
    return _libs[path]


class _Active:
    """What lib() hands out: every attribute is looked up in the library that is active at the time of the call, so
    modules that keep `L = orc.lib()` follow a switch made by use_library()."""

    def __getattr__(self, name):
        if _active == _LIB:
            build()
        return getattr(_load(_active), name)


_proxy = _Active()


def lib():
    return _proxy


def build_mutants():
    """oracle/_mut/libm2v_oracle_mut<k>.so: the oracle compiled with ONE deliberate mis-reading each (m2v_oracle.c,
    M2V_ORACLE_MUTANT).  Test-only.  -> {k: path}"""
    subprocess.check_call(["make", "-s", "-j4", "-C", _HERE, "mutants"])
    d = os.path.join(_HERE, "_mut")
    return {int(f[len("libm2v_oracle_mut"):-3]): os.path.join(d, f) for f in sorted(os.listdir(d)) if f.startswith("libm2v_oracle_mut")}


def use_library(path=None):
    """Point lib() / encode() at another build of the oracle (a mutant); None = back to the real one."""
    global _active
    _active = path or _LIB


def geometry(xsize16, ysize16, XL=7, YL=7, VL=3, Q=2):
    p = Params(XL, YL, VL, Q)
    w, h = ctypes.c_int(), ctypes.c_int()
    if lib().m2v_oracle_geometry(ctypes.byref(p), xsize16, ysize16, ctypes.byref(w), ctypes.byref(h)):
        raise ValueError("bad parameters")
    return w.value, h.value


def encode(frames444, xsize16, ysize16, pframes, XL=7, YL=7, VL=3, Q=2, nbeats=None, dump=False, conformant=False):
    """Encode a sequence.  `frames444`: uint8 array [nframes, 3, H, W] (or flat) in the CLAMPED geometry.

    Returns bytes, or (bytes, dict of numpy dumps) when dump=True.  conformant=True is NOT the reference's behaviour:
    ISO/IEC 13818-2 reconstruction loop, the checker of the GPU path's option "conformant" (see m2v_oracle.h).
    """
    p = Params(XL, YL, VL, Q)
    W, H = geometry(xsize16, ysize16, XL, YL, VL, Q)
    f = np.ascontiguousarray(frames444, dtype=np.uint8).reshape(-1)
    bpf = W * H // 4
    if nbeats is None:
        nbeats = (f.size // (3 * W * H)) * bpf
    nframes = (nbeats + bpf - 1) // bpf
    assert f.size >= nframes * 3 * W * H, "input shorter than the beats pushed"
    cap = max(4096, int(nframes) * W * H * 3 + 4096)
    out = np.zeros(cap, np.uint8)
    d = None
    arrs = {}
    if dump and nframes:
        mbs = (W // 16) * (H // 16)
        arrs = dict(mb_inter=np.zeros((nframes, mbs), np.int8), mb_mvx=np.zeros((nframes, mbs), np.int8),
                    mb_mvy=np.zeros((nframes, mbs), np.int8), mb_cbp=np.zeros((nframes, mbs), np.uint8),
                    coef=np.zeros((nframes, mbs, 6, 64), np.int16),
                    recon=np.zeros((nframes, W * H * 3 // 2), np.uint8),
                    yuv420=np.zeros((nframes, W * H * 3 // 2), np.uint8),
                    mb_bits=np.zeros((nframes, mbs), np.uint32))
        d = Dump(*[arrs[k].ctypes.data for k in ("mb_inter", "mb_mvx", "mb_mvy", "mb_cbp", "coef", "recon",
                                                  "yuv420", "mb_bits")])
    lib().m2v_oracle_set_conformant(1 if conformant else 0)
    try:
        n = lib().m2v_oracle_encode(ctypes.byref(p), xsize16, ysize16, pframes, f.ctypes.data, nbeats,
                                    out.ctypes.data, cap, ctypes.byref(d) if d is not None else None)
    finally:
        lib().m2v_oracle_set_conformant(0)
    if n == ctypes.c_size_t(-1).value:
        raise ValueError("oracle rejected the parameters")
    if n > cap:
        raise RuntimeError("oracle output exceeded the buffer")
    data = out[:n].tobytes()
    return (data, arrs) if dump else data
