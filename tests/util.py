"""Shared helpers for the test-suite: seeded synthetic image pairs."""
import numpy as np
from scipy import ndimage


def texture(h, w, seed=0, sigma=1.5, contrast=60.0):
    rng = np.random.default_rng(seed)
    a = ndimage.gaussian_filter(rng.standard_normal((h, w)), sigma)
    b = ndimage.gaussian_filter(rng.standard_normal((h, w)), sigma * 4)
    t = a / a.std() + 1.5 * b / b.std()
    return np.clip(128 + contrast * t / 1.8, 0, 255)


def warp(img_f, dx=0.0, dy=0.0, scale=1.0, angle=0.0):
    """Sample img at the inverse similarity so that a point p moves to c + s R (p - c) + d."""
    h, w = img_f.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    ca, sa = np.cos(angle), np.sin(angle)
    x = xx - cx - dx
    y = yy - cy - dy
    xs = (ca * x + sa * y) / scale + cx
    ys = (-sa * x + ca * y) / scale + cy
    return ndimage.map_coordinates(img_f, [ys, xs], order=3, mode="reflect")


def move_points(pts, shape, dx=0.0, dy=0.0, scale=1.0, angle=0.0):
    h, w = shape
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    ca, sa = np.cos(angle), np.sin(angle)
    x, y = pts[:, 0] - cx, pts[:, 1] - cy
    return np.stack([scale * (ca * x - sa * y) + cx + dx, scale * (sa * x + ca * y) + cy + dy], 1)


def image_pair(h=240, w=320, seed=0, **motion):
    base = texture(h, w, seed)
    img0 = np.rint(base).astype(np.uint8)
    img1 = np.rint(np.clip(warp(base, **motion), 0, 255)).astype(np.uint8)
    return img0, img1


def grid_points(h, w, step=17, margin=12, jitter_seed=1):
    rng = np.random.default_rng(jitter_seed)
    ys, xs = np.mgrid[margin:h - margin:step, margin:w - margin:step]
    pts = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float64)
    pts += rng.uniform(-0.5, 0.5, pts.shape)
    return pts.astype(np.float32)


class DeviceBuffer:
    """A raw HIP allocation holding a copy of a numpy array (through the HIP runtime libvo_hip.so is
    already linked to — no torch, whose bundled runtime must not be initialised after it)."""

    def __init__(self, arr):
        import ctypes as C
        self._C = C
        self.hip = C.CDLL("libamdhip64.so.7")
        a = np.ascontiguousarray(arr)
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(a.nbytes)) == 0
        assert self.hip.hipMemcpy(self.ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0  # H2D

    def data_ptr(self):
        return self.ptr.value

    def free(self):
        if self.ptr:
            self.hip.hipFree(self.ptr)
            self.ptr = self._C.c_void_p()
