"""The hot kernels keep everything in registers: no scratch memory (a kernel-argument block that grows past what the compiler
keeps in SGPRs, or an array indexed at run time, ends up as a per-lane copy in scratch — measured once as +45 us per stereo
frame when the mono epilogue's arguments were referenced from the stereo BA kernel) and no VGPR spills. hipcc cross-compiles
for gfx950 without a GPU; device code only, four files in parallel."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from visual_odometry_ros_amd import build as B

HOT = {
    "gn_pose.hip": ("gn_pose_kernelILb1E", "gn_pose_kernelILb0E"),
    "frame_fused.hip": ("frame_track_kernelILi21E", "frame_replay_kernelILi21E"),
    "frame_mono.hip": ("mono_track_kernelILi15E", "mono_replay_kernel"),
    "sba.hip": ("sba_solve_reg_kernelILi42E", "sba_update_point_kernelILb1E", "sba_pose_schur_kernel"),
}


def _usage(src):
    flags = [f for f in B.FLAGS if f not in ("-Wall", "-Wno-unused-function")]
    cmd = [B.HIPCC] + flags + ["-I" + os.path.join(os.path.dirname(B.HERE), "include"), "-I" + B.CSRC, "--offload-device-only",
                               "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, src), "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
        for key in ("ScratchSize [bytes/lane]", "VGPRs Spill", "VGPRs"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name:
                out[name].setdefault(key, int(m.group(1)))
    return out


@pytest.mark.skipif(not os.path.exists(B.HIPCC), reason="no hipcc")
def test_hot_kernels_use_no_scratch_memory():
    with ThreadPoolExecutor(4) as ex:
        res = dict(zip(HOT, ex.map(_usage, HOT)))
    for src, kernels in HOT.items():
        for k in kernels:
            hit = [v for n, v in res[src].items() if k in n]
            assert hit, (src, k, sorted(res[src]))
            # (the register solve sits at its 256-VGPR ceiling — 512 lanes, two wavefronts per SIMD — with three dwords of the
            # pose update on the stack: as measured, 25 us; everything else: nothing)
            allowed = 16 if "sba_solve_reg" in k else 0
            for v in hit:
                assert v["ScratchSize [bytes/lane]"] <= allowed and v["VGPRs Spill"] == 0, (src, k, v)
