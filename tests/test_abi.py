"""The C-ABI library loads without a GPU and exports exactly what
include/rt_amd.h declares.  No compute entry point is called here."""
import ctypes
import os
import re

import rtamd

HEADER = os.path.join(rtamd.REPO_DIR, "include", "rt_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([a-z_0-9]+)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(names))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ["gpu_render", "gpu_render_ids", "rt_scene_create", "rt_scene_destroy",
                 "rt_render_device", "gpu_check_intersection", "rt_last_error",
                 "rt_camera_default"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(rtamd.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in rt_amd.h but not exported"


def test_struct_layouts_match_header():
    # sizes the C compiler gives the public structs (LP64)
    assert ctypes.sizeof(rtamd.Material) == 16
    assert ctypes.sizeof(rtamd.Camera) == 20
    assert ctypes.sizeof(rtamd.SceneDesc) == 88
    assert ctypes.sizeof(rtamd.Frame) == 56
    assert rtamd.Frame.stream.offset == 48 and rtamd.Frame.flags.offset == 40


def test_camera_default_is_the_reference_camera():
    cam = rtamd.Camera()
    rtamd.lib().rt_camera_default(ctypes.byref(cam))
    ref = rtamd.default_camera()
    assert cam.fov_deg == ref.fov_deg == 90          # main.cpp:88
    assert list(cam.origin) == list(ref.origin) == [0.0, 0.0, 1.0]  # renderer.cpp:39
    assert cam.plane_z == ref.plane_z == -1.0        # renderer.cpp:38


def test_band_plan_covers_every_row_exactly_once():
    """rt_band_plan (host arithmetic, no GPU): the shares of any split are a
    partition of the rows, bands are dealt round-robin (renderer.cpp:27-28
    parallelises rows; here bands of rows)."""
    for h, band_rows in [(2160, 8), (1440, 16), (1081, 8), (7, 8), (100, 24)]:
        for n in (1, 2, 3, 4, 8):
            seen = [0] * h
            for r in range(n):
                plan = rtamd.band_plan(h, band_rows, r, n)
                for k, (b, e) in enumerate(plan):
                    assert b == (r + k * n) * band_rows and b < e <= h
                    assert e - b == band_rows or e == h
                    for i in range(b, e):
                        seen[i] += 1
            assert seen == [1] * h, (h, band_rows, n)
    # a share beyond the last band owns nothing
    assert rtamd.band_plan(16, 8, 5, 8) == []


def test_band_plan_rejects_bad_arguments():
    import pytest
    for args in [(0, 8, 0, 1), (100, 12, 0, 1), (100, 8, 2, 2), (100, 8, -1, 2), (100, 8, 0, 0)]:
        with pytest.raises(rtamd.RtError):
            rtamd.band_plan(*args)


def test_default_mode_round_trip():
    """rt_set_screen only records the mode that scenes created later start with."""
    L = rtamd.lib()
    try:
        for name, code in rtamd.SCREEN_MODES.items():
            rtamd.set_screen(name)
            assert L.rt_get_screen() == code
        L.rt_set_screen(17)          # out of range: back to the default
        assert L.rt_get_screen() == 1
    finally:
        rtamd.set_screen("auto")
