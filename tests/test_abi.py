"""CPU checks of the C-ABI library: it loads, exports every symbol include/vo_hip.h
declares, and refuses to compute without a gfx950 device (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(vo):
    lib = vo.load()
    declared = _declared_functions()
    assert len(declared) >= 30
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    from visual_odometry_ros_amd import _capi
    assert sorted(_capi.SYMBOLS) == declared
    assert lib.vo_abi_version() == 1


def test_pyramid_level_rule_host_side(vo, oracle):
    lib = vo.load()
    for (w, h, win, ml) in [(1241, 376, 21, 6), (752, 480, 15, 5), (3840, 2160, 21, 4), (64, 48, 21, 3)]:
        assert lib.vo_pyramid_levels(w, h, win, ml) == oracle.pyramid_levels(w, h, win, ml)


def test_no_device_means_error_not_fallback(vo):
    lib = vo.load()
    if lib.vo_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(vo.VoError) as e:
        vo.Context()
    assert e.value.code == -3  # VO_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "visual_odometry_ros_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                # comments may mention the oracle; nothing may import, link or dlopen it
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "libvo_oracle" not in txt and "vo_oracle.h" not in txt and "vo_ref_" not in txt, f
