"""CPU suite: the tile kernels of the ingestion chain (csrc/pyr_tile.hpp, csrc/orb_tile.hpp) run on CPU threads by the
emulation harness of tests/emu/ (g++ compiles the same kernel text; one OS thread per HIP thread, one workgroup at a time)
and compared with the oracle. What this can check without a GPU: every index of the kernels — regions and halos, the
REFLECT_101 mirror lists, ownership (each byte written exactly once), the launch plan, capacities; what it cannot: anything
the GPU does differently from C++ (memory model between workgroups, wavefront intrinsics) — the -m gpu tests cover the
same kernels on the device against the same oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
PAD = 40


def _compile(tmp_path_factory, name):
    out = tmp_path_factory.mktemp("emu") / name
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", os.path.join(EMU, name + ".cpp"), "-o", str(out)])
    return str(out)


@pytest.fixture(scope="module")
def emu_pyr(tmp_path_factory):
    return _compile(tmp_path_factory, "emu_pyr")


def _reflect(p, n):
    if n == 1:
        return 0
    while p < 0 or p >= n:
        p = -p if p < 0 else 2 * n - 2 - p
    return p


@pytest.mark.parametrize("w,h,top,nimg", [(333, 251, 4, 2), (64, 48, 6, 1), (47, 33, 6, 1), (1241, 376, 4, 1), (640, 240, 3, 1),
                                          (752, 480, 5, 1), (90, 70, 1, 1), (65, 65, 0, 1), (130, 40, 2, 1)])
def test_pyramid_tile_kernel_every_padded_byte(oracle, emu_pyr, tmp_path, w, h, top, nimg):
    """pyr_build_kernel: every byte of every padded level plane — the level itself and its REFLECT_101 border, 40 pixels wide,
    also where a level is narrower than the border (several reflections) and where the pyramid needs a second launch (more
    than four levels above the base) — against cv::pyrDown restated (oracle_klt.c)."""
    rng = np.random.default_rng(w * 7 + h)
    imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(nimg)]
    fin, fout = tmp_path / "in.raw", tmp_path / "out.bin"
    fin.write_bytes(b"".join(i.tobytes() for i in imgs))
    subprocess.check_call([emu_pyr, str(w), str(h), str(top), str(fin), str(fout), str(nimg)])
    raw = fout.read_bytes()
    nlv, off = int(np.frombuffer(raw, np.int32, 1)[0]), 4
    assert nlv >= 1
    for i in range(nimg):
        ref = [imgs[i]]
        for _ in range(1, nlv):
            ref.append(oracle.pyr_down(ref[-1]))
        for l in range(nlv):
            lw, lh, st = (int(v) for v in np.frombuffer(raw, np.int32, 3, off))
            off += 12
            plane = np.frombuffer(raw, np.uint8, st * (lh + 2 * PAD), off).reshape(lh + 2 * PAD, st)
            off += st * (lh + 2 * PAD)
            assert ref[l].shape == (lh, lw)
            ys = np.array([_reflect(p - PAD, lh) for p in range(lh + 2 * PAD)])
            xs = np.array([_reflect(p - PAD, lw) for p in range(lw + 2 * PAD)])
            assert np.array_equal(plane[:, :lw + 2 * PAD], ref[l][ys][:, xs]), (i, l)
