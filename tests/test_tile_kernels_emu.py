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


@pytest.mark.parametrize("w,h,top,nimg", [(333, 251, 4, 2), (64, 48, 6, 1), (47, 33, 6, 1), (1241, 376, 4, 1), (160, 120, 3, 1),
                                          (200, 150, 5, 1), (90, 70, 1, 1), (65, 65, 0, 1), (130, 40, 2, 1)])
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


@pytest.fixture(scope="module")
def plan_check(tmp_path_factory):
    return _compile(tmp_path_factory, "plan_check")


@pytest.mark.parametrize("case", [
    (1241, 376, 8, 1.2, 31, 48, 32, 64), (752, 480, 8, 1.2, 31, 48, 32, 64), (3840, 2160, 8, 1.2, 31, 48, 32, 64),
    (3840, 2160, 8, 1.2, 31, 56, 48, 96),  # the tile orb_prepare takes for images of more than 1024 small tiles
    (1920, 1080, 8, 1.2, 31, 56, 48, 96), (640, 240, 8, 1.2, 31, 40, 24, 64), (333, 251, 8, 1.2, 31, 48, 32, 64),
    (620, 188, 3, 1.5, 16, 48, 32, 64), (2000, 40 + 2 * 31 + 9, 2, 1.2, 31, 48, 32, 64),
])
def test_orb_tile_plan_invariants(plan_check, case):
    """csrc/orb_plan.hpp at sizes the emulated kernels are too slow for (3840 x 2160, both tile sizes): the owned intervals
    partition every level's detection rectangle, every staged region holds its owned pixels + the ring FAST / non-max / Harris
    read and the source footprint of the level above, the coefficient tables stay inside their source level, and the LDS
    areas (regions, table slices, score tiles, stash) neither overlap nor pass the limit."""
    w, h, nl, sf, edge, tw, th, kb = case
    r = subprocess.run([plan_check, str(w), str(h), str(nl), repr(sf), str(edge), str(tw), str(th), str(kb * 1024)], capture_output=True, text=True)
    assert r.returncode == 0, (case, r.returncode, r.stderr)


@pytest.fixture(scope="module")
def emu_orb(tmp_path_factory):
    return _compile(tmp_path_factory, "emu_orb")


def _frame(seed, w, h):
    from visual_odometry_ros_amd import synthetic as S
    st = S.StereoStream(width=w, height=h, n_u=8, n_v=4, n_new=8, seed=seed)
    return st.render_pair(st.poses(1)[0])[0]


def _noise(w, h):
    return np.random.default_rng(0).integers(0, 256, (h, w), dtype=np.uint8)


ORB_CASES = [  # image, FAST threshold, bins, ORB overrides, tile (one case at full size; the others smaller: the harness runs OS threads)
    (lambda: _frame(4, 1241, 376), 15, (60, 25), {}, None),
    (lambda: _frame(9, 620, 188), 7, (30, 12), dict(nfeatures=300), None),        # both cuts bite
    (lambda: _frame(4, 620, 188), 10, (20, 12), dict(nfeatures=0), None),
    (lambda: _frame(4, 620, 188), 12, (20, 12), dict(n_levels=3, scale_factor=1.5, edge_threshold=16), None),
    (lambda: _frame(5, 376, 240), 20, (20, 12), {}, None),
    (lambda: np.full((240, 376), 90, np.uint8), 20, (20, 12), {}, None),
    (lambda: _noise(600, 320), 20, (30, 16), {}, None),                          # levels beyond 16 384 candidates
    (lambda: _frame(3, 640, 240), 15, (20, 8), {}, (40, 24)),                     # another tile size
    (lambda: _noise(600, 320), 20, (30, 16), {}, (48, 32, 3, 16384, 5)),          # three finishing workgroups per level, five histogram copies
    (lambda: _frame(9, 620, 188), 7, (30, 12), dict(nfeatures=300), (48, 32, 2, 16384, 2)),
    (lambda: _frame(4, 620, 188), 10, (20, 12), {}, (48, 32, 2, 64)),             # more first-cut survivors than the list holds: radix path
    (lambda: _frame(6, 640, 240), 15, (20, 8), {}, (56, 48, 2, 16384, 3)),        # the large images' tile (80 KB of LDS)
]


@pytest.mark.parametrize("case", range(len(ORB_CASES)))
def test_orb_tile_kernels_table(oracle, emu_orb, tmp_path, case):
    """orb_tile_kernel + orb_finish_kernel (the per-bin candidate table of the closed step [10] in two launches): keypoint
    count, which bins hold a keypoint and the pixel of each bin's best keypoint against cv::ORB::detect restated
    (oracle_orb.c) + the arg-max per bin of extractORBwithBinning_fast; the counters and keys are left zeroed."""
    import struct
    make, thr, (nbu, nbv), orb, tile = ORB_CASES[case]
    img = make()
    h, w = img.shape
    prm = dict(nfeatures=10000, scale_factor=1.2, n_levels=8, edge_threshold=31)
    prm.update(orb)
    f = np.float32
    iu = f(1.0) / f(int(np.floor(f(w) / f(nbu))))  # FeatureExtractor::initParams (feature_extractor.cpp:30-57)
    iv = f(1.0) / f(int(np.floor(f(h) / f(nbv))))
    fin, fout = tmp_path / "in.raw", tmp_path / "out.bin"
    fin.write_bytes(img.tobytes())
    args = [emu_orb, str(w), str(h), str(prm["n_levels"]), repr(prm["scale_factor"]), str(prm["nfeatures"]), str(prm["edge_threshold"]),
            str(thr), str(nbu), str(nbv), "%08x" % iu.view(np.uint32), "%08x" % iv.view(np.uint32), str(fin), str(fout)]
    if tile:
        args += [str(v) for v in tile]  # (tile_w, tile_h[, finishing workgroups per level, capacity of a level's survivor list[, histogram copies]])
    subprocess.check_call(args)
    raw = fout.read_bytes()
    assert struct.unpack_from("i", raw, 0)[0] == 1  # the plan fits
    flags, n_det, dirty, nx, ny, lds, nb, cap = struct.unpack_from("8i", raw, 4)
    off = 36
    xy = np.frombuffer(raw, np.float32, 2 * nb, off).reshape(nb, 2)
    off += 8 * nb
    has = np.frombuffer(raw, np.uint8, nb, off)
    d = oracle.orb_detect(img, thr, nfeatures=prm["nfeatures"], scale_factor=prm["scale_factor"], n_levels=prm["n_levels"],
                          edge_threshold=prm["edge_threshold"], max_kp=400000)
    cand, _ = oracle.bucket_argmax(d["xy"], d["response"], iu, iv, nbu, nbv, np.ones(nbu * nbv, np.int32))
    assert (flags, dirty) == (0, 0) and lds <= (96 if tile and tile[0] > 48 else 64) * 1024
    assert n_det == d["xy"].shape[0]
    assert set(np.unique(has).tolist()) <= {0, 1} and int(has.sum()) == cand.shape[0]
    assert np.array_equal(xy[has == 1].view(np.uint32), cand.view(np.uint32)) and np.all(xy[has == 0] == 0)


# ---- orb_finish_kernel alone, on candidate lists of 4K size ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def emu_finish(tmp_path_factory):
    return _compile(tmp_path_factory, "emu_finish")


def _finish_reference(levels, nbu, nbv, iu, iv, max_out):
    """retainBest(2 n_l) on the score, retainBest(n_l) on the response (keypoint.cpp: everything >= the value of that rank
    stays), then per bin the first keypoint of largest response in (level, y, x) order — written with numpy sorts, nothing
    shared with the kernel."""
    f = np.float32
    best = {}
    total = 0
    surv = []
    for l, (x, y, sc, r, quota, scale) in enumerate(levels):
        n = len(sc)
        keep = np.ones(n, bool)
        if n > 2 * quota:
            keep = sc >= np.sort(sc)[::-1][2 * quota - 1] if quota else np.zeros(n, bool)
        if keep.sum() > quota:
            keep &= (r >= np.sort(r[keep])[::-1][quota - 1]) if quota else False
        surv.append(int(keep.sum()))
        total += surv[-1]
        for i in np.nonzero(keep)[0]:
            xs = f(x[i]) * f(scale) if l else f(x[i])
            ys = f(y[i]) * f(scale) if l else f(y[i])
            u, v = int(np.floor(xs * iu)), int(np.floor(ys * iv))
            if not (0 <= u < nbu and 0 <= v < nbv) or not (r[i] > -1.0):
                continue
            k = (float(r[i]), -l, -int(y[i]), -int(x[i]))
            b = v * nbu + u
            if b not in best or k > best[b][0]:
                best[b] = (k, xs, ys)
    xy = np.zeros((nbu * nbv, 2), f)
    has = np.zeros(nbu * nbv, np.uint8)
    for b, (_, xs, ys) in best.items():
        has[b] = 1
        xy[b] = (xs, ys)
    return xy, has, min(total, max_out), surv


FINISH_CASES = [  # (finishing workgroups per level, histogram copies, capacity of a level's survivor list)
    (26, 64, 16384),   # 3840 x 2160 as orb_prepare sets it up
    (4, 3, 16384),     # slices of more than one round of 8 192 scores per workgroup
    (1, 1, 16384),
    (3, 2, 1024),      # the first cut leaves more than the list holds: the radix fall-back over 171 000 candidates
]


@pytest.mark.parametrize("case", range(len(FINISH_CASES)))
def test_orb_finish_kernel_on_4k_candidate_lists(emu_finish, tmp_path, case):
    """orb_finish_kernel on levels of 171 000 / 60 000 / 20 000 / ... candidates (the 3840 x 2160 case: more than the emulated
    tile kernel can produce in this suite's time), with score ties at the first cut, response ties at the second, levels
    below their quota and an empty level: table, presence, keypoint count and per-level survivors against a numpy reference;
    histogram copies, counters, tickets and keys left zeroed."""
    import struct
    parts, copies, cidx_cap = FINISH_CASES[case]
    rng = np.random.default_rng(17)
    W, H, nbu, nbv, cand_cap, max_out = 3840, 2160, 100, 80, 518400, 14096
    ns = [171000, 60000, 20000, 5000, 2600, 300, 10, 0]
    quotas = [2172, 1810, 1508, 1257, 1047, 873, 727, 606]
    f = np.float32
    iu, iv = f(1.0) / f(W // nbu), f(1.0) / f(H // nbv)
    levels, blob = [], b""
    for l, (n, q) in enumerate(zip(ns, quotas)):
        scale = f(1.2 ** l)
        lw, lh = int(round(W / float(scale))), int(round(H / float(scale)))
        x = rng.integers(31, lw - 31, n).astype(np.int16)
        y = rng.integers(31, lh - 31, n).astype(np.int16)
        sc = np.clip(rng.normal(30, 6, n), 16, 255).astype(np.uint8)            # a few dozen distinct values: wide ties
        r = (rng.normal(1e-4, 5e-5, n)).astype(f)
        coarse = rng.random(n) < 0.3
        r[coarse] = np.round(r[coarse] * f(2e4)) / f(2e4)                        # coarse values: ties at the second cut
        r[r == 0] = f(1e-7)                                                       # (+-0 order differently as keys: not part of this test)
        if n > 5:
            r[:3] = f(-2.0)                                                       # never beats the initial max_score of -1
        levels.append((x, y, sc, r, q, scale))
        blob += struct.pack("iif", n, q, float(scale)) + x.tobytes() + y.tobytes() + sc.tobytes() + r.tobytes()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(blob)
    subprocess.check_call([emu_finish, str(len(ns)), str(parts), str(copies), str(cidx_cap), str(cand_cap), str(max_out), str(nbu), str(nbv),
                           "%08x" % iu.view(np.uint32), "%08x" % iv.view(np.uint32), str(fin), str(fout)])
    raw = fout.read_bytes()
    flags, n_det, dirty, nb = struct.unpack_from("4i", raw, 0)
    xy = np.frombuffer(raw, np.float32, 2 * nb, 16).reshape(nb, 2)
    has = np.frombuffer(raw, np.uint8, nb, 16 + 8 * nb)
    surv = np.frombuffer(raw, np.int32, len(ns), 16 + 9 * nb)
    xy_ref, has_ref, n_ref, surv_ref = _finish_reference(levels, nbu, nbv, iu, iv, max_out)
    assert (flags, dirty, nb) == (0, 0, nbu * nbv)
    assert surv.tolist() == surv_ref and n_det == n_ref
    assert np.array_equal(has, has_ref) and np.array_equal(xy.view(np.uint32), xy_ref.view(np.uint32))
