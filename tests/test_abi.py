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
    assert ctypes.sizeof(rtamd.Frame) == 48


def test_camera_default_is_the_reference_camera():
    cam = rtamd.Camera()
    rtamd.lib().rt_camera_default(ctypes.byref(cam))
    ref = rtamd.default_camera()
    assert cam.fov_deg == ref.fov_deg == 90          # main.cpp:88
    assert list(cam.origin) == list(ref.origin) == [0.0, 0.0, 1.0]  # renderer.cpp:39
    assert cam.plane_z == ref.plane_z == -1.0        # renderer.cpp:38
