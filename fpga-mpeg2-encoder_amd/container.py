"""ctypes binding of libm2v_container.so (include/m2v_container.h): elementary-stream scan and MPEG-2 PS / TS
multiplexers.  CPU-only conveniences around the encoder's output; nothing here touches the GPU
path or the encoded bits."""
import ctypes
import importlib
import os

_build = importlib.import_module(__package__ + ".build")     # the package also exports a function called build

_L = None


class StreamInfo(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("frame_rate_code", ctypes.c_uint32),
                ("aspect_ratio_code", ctypes.c_uint32), ("bit_rate_400", ctypes.c_uint32),
                ("pictures", ctypes.c_uint32), ("i_pictures", ctypes.c_uint32), ("p_pictures", ctypes.c_uint32),
                ("gops", ctypes.c_uint32), ("slices", ctypes.c_uint32), ("bytes", ctypes.c_uint64),
                ("padding_bytes", ctypes.c_uint64), ("has_sequence_end", ctypes.c_int)]


class Picture(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("bytes", ctypes.c_uint64), ("coding_type", ctypes.c_uint32),
                ("temporal_reference", ctypes.c_uint32), ("gop_start", ctypes.c_uint32), ("slices", ctypes.c_uint32)]


class ContainerError(RuntimeError):
    pass


def lib():
    global _L
    if _L is None:
        path = _build.CONTAINER_LIB
        if not os.path.exists(path):
            _build.build_container()
        L = ctypes.CDLL(path)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.m2vc_frame_rate.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        L.m2vc_scan.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(StreamInfo), vp, sz, ctypes.POINTER(sz)]
        for f in (L.m2vc_mux_ps, L.m2vc_mux_ts):
            f.argtypes = [ctypes.c_char_p, sz, vp, sz, ctypes.POINTER(sz)]
        _L = L
    return _L


def _chk(r, what):
    if r < 0:
        raise ContainerError("%s failed: %s" % (what, {-1: "bad parameter", -2: "not an elementary stream of this encoder",
                                                      -3: "buffer too small"}.get(r, r)))


def frame_rate(code):
    num, den = ctypes.c_uint32(), ctypes.c_uint32()
    _chk(lib().m2vc_frame_rate(code, ctypes.byref(num), ctypes.byref(den)), "m2vc_frame_rate")
    return num.value, den.value


def scan(es):
    """-> (StreamInfo, [Picture])"""
    es = bytes(es)
    info, n = StreamInfo(), ctypes.c_size_t()
    _chk(lib().m2vc_scan(es, len(es), ctypes.byref(info), None, 0, ctypes.byref(n)), "m2vc_scan")
    pics = (Picture * max(n.value, 1))()
    _chk(lib().m2vc_scan(es, len(es), ctypes.byref(info), pics, n.value, ctypes.byref(n)), "m2vc_scan")
    return info, list(pics[:n.value])


def _mux(fn, what, es):
    es = bytes(es)
    n = ctypes.c_size_t()
    _chk(fn(es, len(es), None, 0, ctypes.byref(n)), what)
    out = ctypes.create_string_buffer(n.value)
    _chk(fn(es, len(es), out, n.value, ctypes.byref(n)), what)
    return out.raw[:n.value]


def mux_ps(es):
    """MPEG-2 Program Stream around the elementary stream `es`."""
    return _mux(lib().m2vc_mux_ps, "m2vc_mux_ps", es)


def mux_ts(es):
    """MPEG-2 Transport Stream around the elementary stream `es`."""
    return _mux(lib().m2vc_mux_ts, "m2vc_mux_ts", es)
