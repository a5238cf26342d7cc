"""Builds the native parts in-tree: libseqkit_hip.so (HIP kernels + C ABI, gfx950) and the
C++ command-line driver.  hipcc cross-compiles, so this works without a GPU."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libseqkit_hip.so")
CLI = os.path.join(PKG, "seqkit-hip")

LIB_SRCS = [os.path.join(PKG, "csrc", f) for f in ("capi.hip", "kernels.hip", "parse.hip", "format.hip", "slices.hip", "gaps.hip", "multi.hip", "shares.hip")]
LIB_DEPS = LIB_SRCS + [os.path.join(PKG, "csrc", "kernels.h"), os.path.join(PKG, "csrc", "device_util.h"), os.path.join(PKG, "csrc", "ctx.h"), os.path.join(ROOT, "include", "seqkit_hip.h")]
CLI_SRCS = [os.path.join(PKG, "host", f) for f in ("main.cpp", "fastx_reader.cpp", "commands.cpp", "format.cpp")]
CLI_DEPS = CLI_SRCS + [os.path.join(PKG, "host", "host.h"), os.path.join(ROOT, "include", "seqkit_hip.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_lib(force=False, verbose=False):
    if not force and not _stale(LIB, LIB_DEPS):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "csrc"),
           "-Wall", "-Wno-unused-function", "-o", LIB] + LIB_SRCS + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


LIB_CHECKED = os.path.join(PKG, "libseqkit_hip_checked.so")


def build_lib_checked(force=False, verbose=False):
    """The same library with SKH_STAGE_CHECK: the LDS-staged reverse kernel compares every stage buffer with
    global memory after its hand-counted wait and the entry point reports mismatches.  Tests only."""
    if not force and not _stale(LIB_CHECKED, LIB_DEPS):
        return LIB_CHECKED
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSKH_STAGE_CHECK",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "csrc"),
           "-Wall", "-Wno-unused-function", "-o", LIB_CHECKED] + LIB_SRCS + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_CHECKED


def build_cli(force=False, verbose=False):
    if not all(os.path.exists(s) for s in CLI_SRCS):
        return None
    if not force and not _stale(CLI, CLI_DEPS + [LIB]):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", CLI] + CLI_SRCS + \
          ["-L", PKG, "-lseqkit_hip", "-Wl,-rpath,$ORIGIN", "-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


def build_all(force=False, verbose=False):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as pool:              # two independent hipcc runs
        jobs = [pool.submit(build_lib, force, verbose), pool.submit(build_lib_checked, force, verbose)]
        for j in jobs:
            j.result()
    build_cli(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
