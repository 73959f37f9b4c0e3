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
# translation units of the library (csrc/m2v_host.hpp says what lives where); only m2v_launch.hip contains device code
UNITS = ["m2v_launch.hip", "m2v_core.hip", "m2v_port.hip", "m2v_resident.hip", "m2v_strips.hip"]
HEADERS = ["m2v_kernels.hpp", "m2v_tables.hpp", "m2v_types.hpp", "m2v_host.hpp", "m2v_comm.hpp"]
SOURCES = UNITS + HEADERS
OBJDIR = os.path.join(HERE, "build")
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
    jobs = []          # (object, command)
    links = []         # (library, objects)
    for target, extra, tag in ((LIB, [], "rel"), (LIB_DBG, ["-DM2V_DEBUG"], "dbg")):
        if force or _stale(target, deps):
            objs = []
            for u in UNITS:
                obj = os.path.join(OBJDIR, "%s.%s.%d.o" % (u[:-4], tag, os.getpid()))
                jobs.append((obj, [hipcc()] + HIPCC_FLAGS + extra + ["-c", "-o", obj, os.path.join(CSRC, u)]))
                objs.append(obj)
            links.append((target, objs))
    if jobs:
        os.makedirs(OBJDIR, exist_ok=True)
        try:
            # the units compile side by side (the device code of m2v_launch.hip is most of the time)
            from concurrent.futures import ThreadPoolExecutor
            def run(job):
                if verbose:
                    print(" ".join(job[1]))
                subprocess.check_call(job[1])
            with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
                list(pool.map(run, jobs))
            for target, objs in links:
                tmp = "%s.%d.tmp" % (target, os.getpid())     # atomic replace: concurrent ranks never see a half-written library
                cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", tmp] + objs
                if verbose:
                    print(" ".join(cmd))
                try:
                    subprocess.check_call(cmd)
                    os.replace(tmp, target)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            for obj, _ in jobs:
                if os.path.exists(obj):
                    os.remove(obj)
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
