"""Builds libvo_hip.so (hipcc, gfx950 only) in-tree under visual_odometry_ros_amd/lib/.

hipcc cross-compiles without a GPU. -ffp-contract=off keeps every float
operation individually rounded so that the kernels and the CPU oracle evaluate
the same expressions bit for bit.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"] + os.environ.get("VO_EXTRA_FLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hs.append(os.path.join(HERE, "..", "include", "vo_hip.h"))
    return hs


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    hdr_m = max(os.path.getmtime(h) for h in headers())
    relink = force or not os.path.exists(LIB)
    jobs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:  # one hipcc per translation unit
            list(ex.map(run, jobs))
        relink = True
    if relink:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
