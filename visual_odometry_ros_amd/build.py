"""Builds libvo_hip.so (hipcc, gfx950 only) in-tree under visual_odometry_ros_amd/lib/.

hipcc cross-compiles without a GPU. -ffp-contract=off keeps every float
operation individually rounded so that the kernels and the CPU oracle evaluate
the same expressions bit for bit.

An object is rebuilt when the SHA-256 of (its source, every header, the flags, the hipcc version) differs
from the stamp next to it — not by mtime, so a stale object can never be linked. `build(force=True)`,
`python -m visual_odometry_ros_amd.build --force` or VO_REBUILD=1 rebuild everything. Every call reports what
it compiled (stdout with verbose, and lib/build_log.json).
"""
import hashlib
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvo_hip.so")
LOG = os.path.join(LIBDIR, "build_log.json")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"] + os.environ.get("VO_EXTRA_FLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h")))
    hs.append(os.path.join(HERE, "..", "include", "vo_hip.h"))
    return hs


FRAME_KERNEL_SOURCES = ("frame_fused.hip", "klt_device.hpp", "ic_device.hpp", "vo_internal.hpp")
MONO_KERNEL_SOURCES = ("frame_mono.hip", "klt_device.hpp", "ic_device.hpp", "vo_internal.hpp")


def kernel_source_sha(names=FRAME_KERNEL_SOURCES):
    """Identity of the dominant kernel's code: what profiles/*_frame_pmc.json / *_sq_counters.json were measured on.
    bench.py reports their numbers only while this still matches (a stale counter file is dropped, not quoted)."""
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _tool_version():
    try:
        return subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    except OSError:
        return ""


def _stamp(src, common):
    h = hashlib.sha256(common)
    with open(src, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    force = force or os.environ.get("VO_REBUILD", "") not in ("", "0")
    os.makedirs(LIBDIR, exist_ok=True)
    common = hashlib.sha256()
    for h in headers():
        with open(h, "rb") as f:
            common.update(f.read())
    common.update(" ".join(FLAGS).encode())
    common.update(_tool_version().encode())
    common = common.digest()
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        want = _stamp(src, common)
        have = None
        if os.path.exists(obj) and os.path.exists(obj + ".sha"):
            have = open(obj + ".sha").read().strip()
        if force or have != want:
            jobs.append((src, obj, want))
    t0 = time.time()
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(job):
            src, obj, want = job
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(obj + ".sha", "w") as f:
                f.write(want)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:  # one hipcc per translation unit
            list(ex.map(run, jobs))
    relink = bool(jobs) or not os.path.exists(LIB)
    if relink:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    log = {"compiled": [os.path.basename(j[0]) for j in jobs], "translation_units": len(objs), "linked": relink,
           "forced": bool(force), "seconds": round(time.time() - t0, 1), "flags": FLAGS}
    with open(LOG, "w") as f:
        json.dump(log, f)
    print(f"[build] libvo_hip.so: compiled {len(jobs)}/{len(objs)} translation units"
          f"{' (forced)' if force else ''}{', linked' if relink else ', up to date'} in {log['seconds']} s", flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
