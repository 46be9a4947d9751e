"""Undistortion / stereo-rectification ingestion (camera.cpp:56-90, :166-183, :300-336, :364-546 + the
drivers' convertTo(CV_8UC1)): device maps and the remap fused into the pyramid's level 0, bit-exact
against the oracle; the rest of the pyramid must be what the plain path builds from the oracle's
rectified image."""
import ctypes as C

import numpy as np
import pytest

from visual_odometry_ros_amd.api import Camera, StereoCamera
from util import DeviceBuffer

pytestmark = pytest.mark.gpu

KL, DL = (458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
KR, DR = (457.587, 456.134, 379.999, 255.238), (-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05, 0.0)


def _T_lr():
    # a EuRoC-like rig: 11 cm baseline, ~1 degree of relative rotation
    w = np.array([0.012, -0.009, 0.004])
    th = np.linalg.norm(w)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    T[:3, 3] = [0.1100, -0.0002, 0.0008]
    return T.astype(np.float32)


def _image(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2)).astype(np.float64)
    img = np.kron(base, np.ones((4, 4)))[:h, :w]
    img += rng.normal(0, 6.0, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def rctx(vo):
    c = vo.Context(device=0, max_width=752, max_height=480, max_points=256, n_slots=3, max_level=4)
    yield c
    c.close()


def test_mono_undistort_map_and_image(rctx, vo, oracle):
    W, H = 752, 480
    cam = Camera(rctx, 0)
    cam.initParams(W, H, KL, DL)
    mu, mv = cam.maps()
    mu_o, mv_o = oracle.image_undistort_maps(W, H, KL, DL)
    assert np.array_equal(mu.view(np.uint32), mu_o.view(np.uint32))
    assert np.array_equal(mv.view(np.uint32), mv_o.view(np.uint32))
    raw = _image(H, W, 1)
    cam.undistortImage(raw, 0)
    ref = oracle.remap_linear_u8(raw, mu_o, mv_o)
    assert np.array_equal(rctx.get_level(0, 0), ref)
    assert (ref == 0).any() and (ref > 0).mean() > 0.9  # barrel distortion: the corners map outside the raw image
    # levels above 0 = the plain path's pyramid of the rectified image
    rctx.set_image(1, ref)
    for lvl in range(1, 5):
        assert np.array_equal(rctx.get_level(0, lvl), rctx.get_level(1, lvl))
    # a padded, strided source (what a ROS image message with row padding looks like)
    wide = np.zeros((H, W + 24), np.uint8)
    wide[:, :W] = raw
    cam.undistortImage(wide[:, :W], 2)
    assert np.array_equal(rctx.get_level(2, 0), ref)
    with pytest.raises(vo.VoError):
        cam.undistortImage(raw[:, :-1], 0)  # camera.cpp:168-169


def test_stereo_rectify_maps_and_images(rctx, vo, oracle):
    W, H = 752, 480
    sc = StereoCamera(rctx)
    with pytest.raises(vo.VoError):
        sc.getRectifiedCamera()  # camera.cpp:268-269
    sc.initParams(W, H, KL, DL, KR, DR)
    sc.setStereoPoseLeft2Right(_T_lr())
    sc.initStereoCameraToRectify()
    o = oracle.stereo_rectify_maps(W, H, KL, DL, KR, DR, _T_lr())
    (lu, lv), (ru, rv) = sc.maps()
    for a, b in ((lu, o["left"][0]), (lv, o["left"][1]), (ru, o["right"][0]), (rv, o["right"][1])):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(sc.getRectifiedCamera(), o["K_rect"])
    assert np.array_equal(sc.getRectifiedStereoPoseLeft2Right(), o["T_lr_rect"])
    T_rl = sc.getRectifiedStereoPoseRight2Left()
    assert np.array_equal(T_rl[:3, 3], -o["T_lr_rect"][:3, 3]) and np.array_equal(T_rl[:3, :3], np.eye(3))
    # geometry: rotation removed, baseline length kept
    assert abs(np.linalg.norm(o["T_lr_rect"][:3, 3]) - np.linalg.norm(_T_lr()[:3, 3])) < 1e-6
    assert abs(o["T_lr_rect"][1, 3]) < 1e-6 and abs(o["T_lr_rect"][2, 3]) < 1e-6
    L, R = _image(H, W, 2), _image(H, W, 3)
    sc.rectifyStereoImages(L, R, 0, 1)
    assert np.array_equal(rctx.get_level(0, 0), oracle.remap_linear_u8(L, *o["left"]))
    assert np.array_equal(rctx.get_level(1, 0), oracle.remap_linear_u8(R, *o["right"]))
    # device-pointer pair entry point (one launch chain for both images)
    dL, dR = DeviceBuffer(L), DeviceBuffer(R)
    rctx.set_stereo_pair_rectified_device(1, dL.data_ptr(), 2, dR.data_ptr(), W, H, W)
    rctx.synchronize()
    assert np.array_equal(rctx.get_level(1, 0), oracle.remap_linear_u8(L, *o["left"]))
    assert np.array_equal(rctx.get_level(2, 0), oracle.remap_linear_u8(R, *o["right"]))
    dL.free()
    dR.free()
    rctx.set_image(0, oracle.remap_linear_u8(R, *o["right"]))
    for lvl in range(1, 5):
        assert np.array_equal(rctx.get_level(2, lvl), rctx.get_level(0, lvl))


def test_caller_supplied_maps_edge_cases(rctx, vo, oracle):
    """cv::remap's corner cases: half-pixel ties of the 1/32 quantisation, taps straddling every border,
    coordinates far outside, NaN."""
    W, H = 64, 48
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, (H, W), dtype=np.uint8)
    mu = rng.uniform(-3, W + 2, (H, W)).astype(np.float32)
    mv = rng.uniform(-3, H + 2, (H, W)).astype(np.float32)
    mu[0, :32] = (np.arange(32) + 0.5) / 32.0 + 5.0          # exact halves of the 1/32 grid (cvRound ties)
    mv[0, :32] = 7.0 + 1.0 / 64.0
    mu[1, :8] = [-1.0, -0.5, -1.0 - 1 / 64, W - 1, W - 0.5, W, 1e9, -1e9]
    mv[1, :8] = [3.25] * 8
    mv[2, :8] = [-1.0, -0.5, -1.0 - 1 / 64, H - 1, H - 0.5, H, 3e9, -3e9]
    mu[2, :8] = [10.75] * 8
    mu[3, :2] = np.nan
    mv[3, 2:4] = np.nan
    # values that make the interpolated sum an exact .5 (round-half-to-even of the convertTo)
    raw[20, 20:22] = [10, 11]
    mu[4, 0], mv[4, 0] = 20.5, 20.0
    raw[21, 20:22] = [11, 12]
    mu[4, 1], mv[4, 1] = 20.5, 21.0
    fp = C.POINTER(C.c_float)
    rctx.check(rctx.lib.vo_rectify_set_maps(rctx.handle, 0, mu.ctypes.data_as(fp), mv.ctypes.data_as(fp), W, H))
    rctx.set_image_rectified(0, raw, 0)
    ref = oracle.remap_linear_u8(raw, mu, mv)
    got = rctx.get_level(0, 0)
    assert np.array_equal(got, ref)
    assert ref[4, 0] == 10 and ref[4, 1] == 12  # 10.5 -> 10, 11.5 -> 12
    assert ref[1, 6] == 0 and ref[3, 0] == 0
