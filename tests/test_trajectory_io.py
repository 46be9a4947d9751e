"""The reference's trajectory dump format (stereo_vo.cpp:55-115: id + 12 floats, fixed, precision 4) from the Python
and the C++ mirror: identical bytes, and the bytes the reference's `ostream << float` with precision(4) / fixed gives."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trajectory_dump_format(tmp_path, vo):
    rng = np.random.default_rng(5)
    n = 7
    T = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    T[:, :3, :] = rng.normal(0, 30, (n, 3, 4)).astype(np.float32)
    T[0, 0, 0], T[0, 0, 1], T[0, 0, 2], T[0, 0, 3] = 1.0, -0.00004, 0.00006, 12345.678  # rounding, -0.0000, large
    T[1, 1, 1] = 0.12345  # a tie in decimal that is not one in binary
    ids = [0, 2, 4, 6, 8, 10, 12]  # left-frame ids of a stereo stream (the right frames take the odd ones)
    p_py = tmp_path / "py.txt"
    vo.write_trajectory(p_py, ids, T)
    lines = open(p_py).read().splitlines()
    assert len(lines) == n
    f0 = lines[0].split(" ")
    assert f0[0] == "0" and len(f0) == 13 and f0[1] == "1.0000" and f0[2] == "-0.0000" and f0[3] == "0.0001" and f0[4] == "12345.6777"
    assert not lines[0].endswith(" ")
    # the C++ mirror writes the same bytes
    exe = str(tmp_path / "trajectory_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, os.path.join(ROOT, "tests", "cpp", "trajectory_demo.cpp"), "-o", exe])
    inp, outp = tmp_path / "in.bin", tmp_path / "cpp.txt"
    with open(inp, "wb") as f:
        f.write(struct.pack("i", n))
        f.write(np.asarray(ids, np.int32).tobytes())
        f.write(np.ascontiguousarray(T, np.float32).tobytes())
    assert subprocess.call([exe, str(inp), str(outp)]) == 0  # (also: an unopenable path throws, as the reference does)
    assert open(outp, "rb").read() == open(p_py, "rb").read()
