"""Builds libm2v_mi355x.so (HIP kernels + C-ABI) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libm2v_mi355x.so")
LIB_DBG = os.path.join(HERE, "libm2v_mi355x_dbg.so")       # -DM2V_DEBUG: level dump, keep_recon, ablate (tests / profiling only)
TB = os.path.join(HERE, "m2v_tb")
CONTAINER_LIB = os.path.join(HERE, "libm2v_container.so")      # CPU-only conveniences (include/m2v_container.h)
SOURCES = ["m2v_mi355x.hip", "m2v_kernels.hpp", "m2v_tables.hpp", "m2v_comm.hpp"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fwrapv", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", "m2v_mi355x.h")]
    for target, extra in ((LIB, []), (LIB_DBG, ["-DM2V_DEBUG"])):
        if force or _stale(target, deps):
            tmp = "%s.%d.tmp" % (target, os.getpid())     # atomic replace: concurrent ranks never see a half-written library
            cmd = [hipcc()] + HIPCC_FLAGS + extra + ["-shared", "-o", tmp, os.path.join(CSRC, "m2v_mi355x.hip")]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, target)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
    build_container(force, verbose)
    tb_src = os.path.join(CSRC, "m2v_tb.cpp")
    if os.path.exists(tb_src) and (force or _stale(TB, [tb_src, LIB, CONTAINER_LIB])):
        tmp = "%s.%d.tmp" % (TB, os.getpid())
        cmd = [hipcc(), "-O2", "-std=c++17", "-o", tmp, tb_src, "-L" + HERE, "-lm2v_mi355x", "-lm2v_container",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, TB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB


def build_container(force=False, verbose=False):
    """libm2v_container.so: elementary-stream scan + PS/TS multiplexers, plain C++ (g++), no GPU involved."""
    src = os.path.join(CSRC, "m2v_container.cpp")
    hdr = os.path.join(HERE, "..", "include", "m2v_container.h")
    if force or _stale(CONTAINER_LIB, [src, hdr]):
        cxx = shutil.which("g++") or hipcc()
        tmp = "%s.%d.tmp" % (CONTAINER_LIB, os.getpid())
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-Wall", "-shared", "-o", tmp, src]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, CONTAINER_LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return CONTAINER_LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
