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
    assert lib.vo_abi_version() == 3


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


def test_every_entry_point_rejects_null_arguments(vo):
    """Every int-returning entry point called with a NULL context and NULL / zero arguments returns a negative
    status (no dereference before the argument check) — runs without a GPU."""
    lib = vo.load()
    src = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = re.findall(r"\bint\s+(vo_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)
    assert len(protos) >= 40
    for name, args in protos:
        if name in ("vo_abi_version", "vo_device_count", "vo_pyramid_levels", "vo_create"):
            continue
        call = []
        for a in (x.strip() for x in args.split(",")):
            if "*" in a or "[" in a:
                call.append(None)
            elif a.startswith("double"):
                call.append(C.c_double(0.0))
            elif a.startswith("float"):
                call.append(C.c_float(0.0))
            else:
                call.append(0)
        f = getattr(lib, name)
        saved = f.argtypes
        f.restype, f.argtypes = C.c_int, None
        try:
            assert f(*call) < 0, name
        finally:
            f.argtypes = saved


def test_build_reports_what_it_compiled():
    """__graft_entry__.build() leaves a log of the translation units it compiled (hash-stamped objects)."""
    import json
    from visual_odometry_ros_amd import build as B
    B.build()
    log = json.load(open(B.LOG))
    assert log["translation_units"] == len(B.sources()) and isinstance(log["compiled"], list)
    for src in B.sources():
        obj = os.path.join(B.LIBDIR, os.path.basename(src)[:-4] + ".o")
        assert os.path.exists(obj + ".sha")
