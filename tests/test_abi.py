"""The C-ABI library: loads, exports every symbol include/m2v_mi355x.h declares, validates
parameters, and refuses to work without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

import m2v_load

M = m2v_load.load()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    M.build()
    return M.lib()


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "m2v_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(m2v_[a-z_]+)\s*\(", txt)))


def test_exports_every_declared_symbol(L):
    names = declared_functions()
    assert len(names) >= 15 and set(M.EXPORTS) <= set(names)
    for n in names:
        assert hasattr(L, n), "missing export " + n


def test_no_torch_or_hip_types_in_header():
    txt = open(os.path.join(ROOT, "include", "m2v_mi355x.h")).read()
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    assert "#include <hip" not in code and "torch" not in code and "hipStream_t" not in code
    assert 'extern "C"' in code


def test_parameter_validation_and_no_cpu_fallback(L):
    import torch
    err = ctypes.c_int(0)
    for bad in [(3, 6, 3, 2), (8, 6, 3, 2), (6, 3, 3, 2), (6, 6, 0, 2), (6, 6, 4, 2), (6, 6, 3, 0), (6, 6, 3, 5)]:
        assert not L.m2v_create(*bad, 0, ctypes.byref(err)) and err.value == -1
    if not torch.cuda.is_available():
        assert not L.m2v_create(6, 6, 3, 2, 0, ctypes.byref(err)) and err.value == -2     # M2V_E_NODEVICE
        with pytest.raises(M.M2VError):
            M.Mpeg2Encoder(6, 6, 3, 2)
    assert L.m2v_busy(None) == 0 and L.m2v_reset(None) == -1
    assert b"m2v_mi355x" in L.m2v_version()


def test_product_tables_match_oracle_tables(L):
    from oracle import m2v_oracle_ctypes as orc
    O = orc.lib()

    def pair(fn, *a):
        c, n = ctypes.c_int(), ctypes.c_int()
        fn(*a, ctypes.byref(c), ctypes.byref(n))
        return c.value, n.value
    for i in range(8):
        for j in range(8):
            assert L.m2v_debug_table(0, i, j) == O.m2v_oracle_tab_dct(i, j)
            assert L.m2v_debug_table(1, i, j) == O.m2v_oracle_tab_intra_w(i, j)
            assert L.m2v_debug_table(2, i, j) == O.m2v_oracle_tab_zigzag(i, j)
    for k in range(17):
        c, n = pair(O.m2v_oracle_tab_motion, k)
        assert L.m2v_debug_table(3, k, 0) == (n << 8) | c
    for k in range(64):
        c, n = pair(O.m2v_oracle_tab_cbp, k)
        assert L.m2v_debug_table(4, k, 0) == (n << 8) | c
    for ch in range(2):
        for k in range(12):
            c, n = pair(O.m2v_oracle_tab_dc, ch, k)
            assert L.m2v_debug_table(5, ch, k) == (n << 16) | c
    for run in range(32):
        for lvl in range(1, 41):
            c, n = pair(O.m2v_oracle_tab_ac, run, lvl)
            assert L.m2v_debug_table(6, run, lvl) == ((n << 8) | c if n else 0)


def test_block_permutation_is_a_bijection_for_every_grid_and_cu_pack(L):
    """k_mb finds its macroblock through xcd_remap(blockIdx, gridDim, option "cu_pack"): whatever the launch size and the option, every
    position 0 .. n-1 must be taken exactly once (a position left out would be a macroblock nobody encodes).  Walked on the host."""
    sizes = list(range(1, 260)) + [511, 512, 513, 1200, 2047, 2048, 2049, 2055, 4095, 4104, 8640, 16383, 16392, 20480]
    for p in range(9):
        for n in sizes:
            got = sorted(L.m2v_debug_table(16 + p, b, n) for b in range(n))
            assert got == list(range(n)), (p, n)
    for p in (0, 5, 8):                                    # one launch of the benchmark's P step
        n = 86400
        seen = bytearray(n)
        for b in range(n):
            seen[L.m2v_debug_table(16 + p, b, n)] += 1
        assert seen == bytearray([1]) * n, p
    assert L.m2v_debug_table(16 + 9, 0, 8) == -1 and L.m2v_debug_table(16, 8, 8) == -1


def test_communicator_constructors_validate_without_a_gpu(L):
    """m2v_comm_*: the in-process communicators are plain host objects (creating one needs no GPU); bad arguments give NULL, a code and
    a text; the RCCL one refuses a bad rank before it touches the device."""
    err = ctypes.c_int(0)
    for bad in (0, -1, 17):
        assert not L.m2v_comm_init_local(bad, ctypes.byref(err)) and err.value == -1
        assert b"1..16" in L.m2v_comm_last_error()
        assert not L.m2v_comm_init_solo(bad, ctypes.byref(err)) and err.value == -1
    for world in (1, 2, 8, 16):
        c = L.m2v_comm_init_local(world, ctypes.byref(err))
        assert c and err.value == 0
        L.m2v_comm_destroy(c)
        c = L.m2v_comm_init_solo(world, ctypes.byref(err))
        assert c and err.value == 0
        L.m2v_comm_destroy(c)
    ident = ctypes.create_string_buffer(128)
    for rank, world in ((2, 2), (-1, 2), (0, 0), (0, 17)):
        assert not L.m2v_comm_init_rccl(ident, rank, world, 0, ctypes.byref(err)) and err.value == -1
    assert not L.m2v_comm_init_rccl(None, 0, 1, 0, ctypes.byref(err)) and err.value == -1
    assert L.m2v_comm_unique_id(ident, 64) == -1                  # the buffer must hold a ncclUniqueId (128 bytes)
    assert L.m2v_comm_selftest(None, 0, None, None, 0, None) == -1
    assert L.m2v_strip_encode(None, None, 0, 1, 0, 4, 4, 0, None, 0, None, 0, None, None) == -1
    L.m2v_comm_destroy(None)                                      # like free(NULL)


def test_communicator_constructors_validate_their_arguments_without_a_gpu(L):
    """m2v_comm_init_callbacks / m2v_comm_init_peer and friends: what can be refused before anything touches HIP is refused there, with a text"""
    err = ctypes.c_int(0)
    assert not L.m2v_comm_init_callbacks(2, None, ctypes.byref(err)) and err.value == -1
    cb = M.CommCallbacks(M.CommCallbacks.HALO(lambda *a: 0), M.CommCallbacks.ALLGATHER(lambda *a: 0), M.CommCallbacks.GATHER(), None)
    assert not L.m2v_comm_init_callbacks(2, ctypes.byref(cb), ctypes.byref(err)) and err.value == -1        # a function is missing
    assert b"all three functions" in L.m2v_comm_last_error()
    cb = M.CommCallbacks(M.CommCallbacks.HALO(lambda *a: 0), M.CommCallbacks.ALLGATHER(lambda *a: 0), M.CommCallbacks.GATHER(lambda *a: 0), None)
    assert not L.m2v_comm_init_callbacks(17, ctypes.byref(cb), ctypes.byref(err)) and err.value == -1       # at most 16 strips
    c = L.m2v_comm_init_callbacks(3, ctypes.byref(cb), ctypes.byref(err))
    assert c and err.value == 0
    try:
        assert L.m2v_comm_kind(c) == b"callbacks" and L.m2v_comm_kind(None) == b""
        assert L.m2v_comm_peer_stats(c, None, None) == -1                                                   # not a peer communicator
        buf = ctypes.create_string_buffer(M.PEER_DESC_BYTES)
        assert L.m2v_comm_peer_export(c, buf, M.PEER_DESC_BYTES) == -1 and b"not a peer communicator" in L.m2v_comm_last_error()
        assert L.m2v_comm_peer_connect(c, None, None) == -1 and L.m2v_comm_peer_connect_all(c) == -1
        assert not L.m2v_comm_init_peer(None, 0, 0, 0, ctypes.byref(err)) and err.value == -1
        assert not L.m2v_comm_init_peer(c, 3, 0, 0, ctypes.byref(err)) and err.value == -1                  # rank outside the base communicator
        assert not L.m2v_comm_init_peer(c, -1, 0, 0, ctypes.byref(err)) and err.value == -1
        import torch
        if not torch.cuda.is_available():
            assert not L.m2v_comm_init_peer(c, 0, 0, 0, ctypes.byref(err)) and err.value == -2              # M2V_E_NODEVICE: no landing block without a GPU
    finally:
        L.m2v_comm_destroy(c)
    assert L.m2v_upload_wait(None) == -1 and L.m2v_strip_last_form(None) == -1
