"""Independent cross-checks of the oracle's THIRD-PARTY restatements with what this image does have (numpy / scipy).

Not a pin: the reference's own dependencies (OpenCV 4, Eigen 3) are absent here, so nothing can say "this is what cv:: /
Eigen computes" bit for bit (tests/test_opencv_pin.py and tests/test_eigen_pin.py would, where the libraries exist). What
these checks do is compare each restated piece of oracle/*.c with a SECOND formulation that shares no code with it and is
written from the published definition of the operation, on the committed fixtures and on seeded inputs:

  pyrDown       oracle_klt.c  vs  scipy.ndimage.correlate1d ([1 4 6 4 1], mirror = REFLECT_101), (x + 128) >> 8, every 2nd sample
                catches: kernel, border mode, rounding, decimation phase, output size. cannot catch: nothing cv::pyrDown-specific
                remains for 8-bit input (its fixed-point path is exactly this integer expression)
  Scharr/Sobel  oracle_klt.c / oracle_ic.c  vs  full 3 x 3 kernels through scipy.ndimage.correlate (mirror)
                catches: kernel coefficients, sign / axis conventions, border. cannot catch: the derivative SCALE OpenCV's
                LK applies afterwards (restated separately in the PyrLK code: W_BITS / FLT_SCALE)
  LDLT          oracle_gn.c (float, 6 x 6, Eigen's pivoting) / oracle_sba.c (double, n = 3, 12, 42)  vs  numpy.linalg.solve
                catches: wrong pivot bookkeeping, a wrong triangular sweep (errors of order 1). cannot catch: the ORDER of
                Eigen's operations (last-bit differences) — only the Eigen pin can
  JacobiSVD     oracle_vo.c (4 x 4, null vector used by triangulateDLT)  vs  numpy.linalg.svd
                catches: wrong rotation / sorting / sign handling. cannot catch: sweep order and threshold (last bits)
  PyrLK         oracle_klt.c (fixed-point bilinear weights, integer Scharr, OpenCV's iteration rules)  vs  a float64 pyramidal
                Lucas-Kanade written from the textbook equations (Bouguet's formulation, no fixed point, no rounding of the pyramid)
                on tests/golden/klt_small.npz
                catches: a wrong sign, a wrong window, a wrong level-to-level propagation, wrong convergence handling — anything
                that moves a converged point by more than a few hundredths of a pixel. cannot catch: OpenCV's fixed-point
                details (W_BITS = 14 weights, 2^-20 scaling, the min-eigenvalue normalisation) at the 1e-3 px level
"""
import os

import numpy as np
import pytest
from scipy import ndimage

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- pyrDown -------------------------------------------------------------------------------------------------------------------
def _pyr_down_scipy(img):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    f = img.astype(np.int64)
    s = ndimage.correlate1d(ndimage.correlate1d(f, k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
    return ((s[::2, ::2] + 128) >> 8).astype(np.uint8)


@pytest.mark.parametrize("shape", [(37, 53), (64, 80), (160, 208), (376, 1241), (5, 7)])
def test_pyr_down_vs_separable_mirror_convolution(oracle, shape):
    img = np.random.default_rng(shape[0] * 31 + shape[1]).integers(0, 256, shape, dtype=np.uint8)
    got, want = oracle.pyr_down(img), _pyr_down_scipy(img)
    assert got.shape == ((shape[0] + 1) // 2, (shape[1] + 1) // 2) == want.shape
    assert np.array_equal(got, want)


def test_pyramid_chain_on_the_committed_fixture(oracle):
    z = np.load(os.path.join(GOLD, "klt_small.npz"))
    lv = oracle.build_pyramid(z["img0"], 21, 3)
    ref = z["img0"]
    for l, g in enumerate(lv):
        assert np.array_equal(g, ref), l
        ref = _pyr_down_scipy(ref)


# ---- derivatives ----------------------------------------------------------------------------------------------------------------
def test_scharr_and_sobel_vs_full_3x3_kernels(oracle):
    img = np.random.default_rng(11).integers(0, 256, (61, 83), dtype=np.uint8)
    f = img.astype(np.int64)
    scharr_x = np.array([[-3, 0, 3], [-10, 0, 10], [-3, 0, 3]], np.int64)
    sobel_x = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int64)
    d = oracle.scharr(img)
    assert np.array_equal(d[..., 0], ndimage.correlate(f, scharr_x, mode="mirror"))
    assert np.array_equal(d[..., 1], ndimage.correlate(f, scharr_x.T, mode="mirror"))
    du, dv = oracle.sobel3(img)
    assert np.array_equal(du, ndimage.correlate(f, sobel_x, mode="mirror"))
    assert np.array_equal(dv, ndimage.correlate(f, sobel_x.T, mode="mirror"))


# ---- dense solves ------------------------------------------------------------------------------------------------------------------
def test_ldlt6_float_vs_numpy_solve(oracle):
    rng = np.random.default_rng(3)
    for _ in range(50):
        J = rng.normal(size=(40, 6))
        A = (J.T @ J + 1e-3 * np.eye(6)).astype(np.float32)
        b = rng.normal(size=6).astype(np.float32)
        x = oracle.ldlt6_solve(A, b)
        want = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
        assert np.linalg.norm(x - want) <= 1e-5 * max(1.0, np.linalg.cond(A.astype(np.float64))) * np.linalg.norm(want)


@pytest.mark.parametrize("n", [3, 12, 42])
def test_ldlt_double_vs_numpy_solve(oracle, n):
    rng = np.random.default_rng(n)
    for _ in range(10):
        J = rng.normal(size=(3 * n, n))
        A = J.T @ J + 1e-6 * np.eye(n)
        B = rng.normal(size=(n, 2))
        X = oracle.ldlt_solve_f64(A, B)
        want = np.linalg.solve(A, B)
        assert np.linalg.norm(X - want) <= 1e-10 * np.linalg.cond(A) * np.linalg.norm(want)


def test_jacobi_svd4_vs_numpy_svd(oracle):
    rng = np.random.default_rng(8)
    for k in range(40):
        M = rng.normal(size=(4, 4)).astype(np.float32)
        if k % 2:
            M[3] = 0.3 * M[0] - 0.7 * M[1]  # rank 3: the DLT's case, a clean null vector
        V, sv, _sweeps = oracle.jacobi_svd4(M)
        s_np = np.linalg.svd(M.astype(np.float64), compute_uv=False)
        assert np.allclose(np.sort(sv)[::-1], s_np, rtol=1e-5, atol=1e-5)
        _, _, Vt = np.linalg.svd(M.astype(np.float64))
        v_o, v_n = V[:, 3].astype(np.float64), Vt[3]
        if s_np[2] - s_np[3] > 1e-3:  # (the null direction is only defined when the last two singular values differ)
            assert abs(abs(v_o @ v_n) - 1.0) < 1e-4


# ---- PyrLK against a float64 textbook Lucas-Kanade ----------------------------------------------------------------------------------
def _bilinear(I, x, y):
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    a, b = x - x0, y - y0
    return ((1 - a) * (1 - b) * I[y0, x0] + a * (1 - b) * I[y0, x0 + 1] + (1 - a) * b * I[y0 + 1, x0] + a * b * I[y0 + 1, x0 + 1])


def _textbook_lk(img0, img1, pts0, win, levels, iters=30, eps=0.01):
    """Pyramidal Lucas-Kanade (Bouguet 2000): float64 Gaussian pyramid ([1 4 6 4 1] / 16, mirror border, no rounding), Scharr
    derivatives / 32 of the first image, G = sum of grad grad^T over the window, v += G^-1 sum (I - J(x + g + v)) grad.
    Returns (points, ok) — ok False where a window left the image at some level (the border handling is OpenCV's own)."""
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16.0

    def down(f):
        return ndimage.correlate1d(ndimage.correlate1d(f, k, axis=0, mode="mirror"), k, axis=1, mode="mirror")[::2, ::2]

    P0, P1 = [img0.astype(np.float64)], [img1.astype(np.float64)]
    for _ in range(levels):
        P0.append(down(P0[-1]))
        P1.append(down(P1[-1]))
    sx = np.array([[-3, 0, 3], [-10, 0, 10], [-3, 0, 3]], np.float64) / 32.0
    half = (win - 1) / 2.0
    oy, ox = np.mgrid[0:win, 0:win]
    out, ok = np.zeros_like(pts0, dtype=np.float64), np.ones(pts0.shape[0], bool)
    for i, p in enumerate(pts0.astype(np.float64)):
        g = np.zeros(2)
        for l in range(levels, -1, -1):
            I, J = P0[l], P1[l]
            Ix, Iy = ndimage.correlate(I, sx, mode="mirror"), ndimage.correlate(I, sx.T, mode="mirror")
            c = p / (2.0 ** l) - half
            xs, ys = c[0] + ox, c[1] + oy
            if xs.min() < 0 or ys.min() < 0 or xs.max() >= I.shape[1] - 1 or ys.max() >= I.shape[0] - 1:
                ok[i] = False
                break
            T, gx, gy = _bilinear(I, xs, ys), _bilinear(Ix, xs, ys), _bilinear(Iy, xs, ys)
            G = np.array([[np.sum(gx * gx), np.sum(gx * gy)], [np.sum(gx * gy), np.sum(gy * gy)]])
            v = np.zeros(2)
            for _ in range(iters):
                xq, yq = xs + g[0] + v[0], ys + g[1] + v[1]
                if xq.min() < 0 or yq.min() < 0 or xq.max() >= J.shape[1] - 1 or yq.max() >= J.shape[0] - 1:
                    ok[i] = False
                    break
                d = T - _bilinear(J, xq, yq)
                dv = np.linalg.solve(G, np.array([np.sum(d * gx), np.sum(d * gy)]))
                v += dv
                if dv @ dv < eps * eps:
                    break
            if not ok[i]:
                break
            g = (g + v) * (2.0 if l > 0 else 1.0)
        out[i] = p + g
    return out, ok


def test_pyr_lk_vs_float64_textbook_lucas_kanade(oracle):
    z = np.load(os.path.join(GOLD, "klt_small.npz"))
    lv, p1, st, err = oracle.calc_optical_flow_pyr_lk(z["img0"], z["img1"], z["pts0"], None, 21, 3)
    assert np.array_equal(p1.view(np.uint32), z["pts1"].view(np.uint32)) and np.array_equal(st, z["status"])  # (the fixture is the oracle's)
    ref, ok = _textbook_lk(z["img0"], z["img1"], z["pts0"], 21, lv)
    use = ok & (st > 0)
    assert use.sum() >= 40  # interior points of the 160 x 208 fixture whose 21 x 21 window stays inside on every level
    d = np.abs(p1[use].astype(np.float64) - ref[use])
    assert d.max() < 0.05, (d.max(), np.argmax(d.max(axis=1)))
    assert np.median(d) < 0.01
