"""The sphere lists by screen cell (rays from the eye) and by direction from a light
(the primary hits' shadow rays) against the oracle and against the hierarchy walks
that they replace.  Needs an MI355X."""
import numpy as np
import pytest

import golden_scenes
import rtamd
from test_gpu_parity import _wild_scene, assert_frame_parity

pytestmark = pytest.mark.gpu


def _cloud(seed, n, n_lights=1, spread=40.0, rmax=3.0, light_at=None):
    rng = np.random.RandomState(seed)
    c = (rng.uniform(-spread, spread, (n, 3)) + [0, 0, -90]).astype(np.float32)
    r = rng.uniform(0.2, rmax, n).astype(np.float32)
    col = rng.randint(0, 256, (n, 3)).astype(np.uint8)
    tris = np.float32([[-200, -60, -160, 200, -60, -160, 0, 200, -160]])
    lights = rng.uniform(-60, 60, (n_lights, 3)).astype(np.float32)
    if light_at is not None:
        lights[0] = light_at
    return rtamd.Scene(tris, np.uint8([[90, 90, 200]]), c, r, col, lights)


def _frame_in_modes(sc, w, h, oracle, cam=None, **kw):
    fo, io, no = oracle.render(sc, w, h, cam=cam)
    for mode in ("auto", "bvh"):
        try:
            rtamd.set_screen(mode)
            fb, ids, n = rtamd.render(sc, w, h, cam=cam)
        finally:
            rtamd.set_screen("auto")
        assert_frame_parity(fb, ids, fo, io, f"mode {mode}")
        assert n == no


def test_lists_exist_and_are_used_on_the_bench_scene():
    sc = rtamd.grid_scene(100, 100)
    import torch
    fb = torch.zeros(720, 1280, 4, dtype=torch.uint8, device="cuda")
    with rtamd.DeviceScene(sc) as ds:
        ds.render_device(fb.data_ptr(), None, 1280, 720)
        d = ds.describe()
    assert d["lights_with_lists"] == 1
    assert d["screen_cells_with_list"] == (1280 // 16) * (720 // 16)
    assert d["light_cells_without_list"] == 0
    with rtamd.DeviceScene(rtamd.grid_scene(5, 5)) as ds:      # 25 spheres: no lists
        ds.render_device(fb.data_ptr(), None, 1280, 720)
        d = ds.describe()
    assert d["lights_with_lists"] == 0 and d["screen_cells_with_list"] == -1


@pytest.mark.parametrize("w,h", [(250, 130), (1000, 562), (16, 16), (33, 47)])
def test_frame_sizes_that_leave_partial_cells(oracle, w, h):
    _frame_in_modes(_cloud(1, 400), w, h, oracle)


def test_cameras(oracle):
    sc = _cloud(2, 600)
    for fov, origin, pz in [(60, (0, 0, 1), -1.0), (90, (15.0, -5.0, 3.0), -1.0),
                            (120, (0, 2, 0), -2.0), (90, (0, 0, -90), -91.0),   # eye inside the cloud
                            (30, (0, 0, 1), -1.0), (170, (0, 0, 1), -1.0)]:
        cam = rtamd.default_camera(fov)
        cam.origin[0], cam.origin[1], cam.origin[2] = origin
        cam.plane_z = pz
        _frame_in_modes(sc, 400, 250, oracle, cam=cam)


def test_lights_inside_outside_far_and_more_than_four(oracle):
    _frame_in_modes(_cloud(3, 500, n_lights=1, light_at=(0, 0, -90)), 300, 200, oracle)   # inside
    _frame_in_modes(_cloud(4, 500, n_lights=1, light_at=(0, 1e5, 0)), 300, 200, oracle)   # far away
    _frame_in_modes(_cloud(5, 500, n_lights=6), 300, 200, oracle)                         # 4 with lists + 2
    sc = _cloud(6, 500, n_lights=2)
    sc.lights[1] = sc.spheres[7]                                                          # at a centre
    _frame_in_modes(sc, 300, 200, oracle)


def test_lists_that_overflow_fall_back_to_the_walk(oracle):
    # a thousand tiny spheres inside one screen cell and one light cell, and a sphere that
    # fills the view
    rng = np.random.RandomState(7)
    c = (rng.normal(scale=0.05, size=(1000, 3)) + [0.3, 0.2, -30]).astype(np.float32)
    r = np.full(1000, 0.01, np.float32)
    c = np.concatenate([c, np.float32([[0, 0, -400]])])
    r = np.concatenate([r, np.float32([300])])
    col = rng.randint(0, 256, (1001, 3)).astype(np.uint8)
    sc = rtamd.Scene(np.zeros((0, 9), np.float32), np.zeros((0, 3), np.uint8), c, r, col,
                     np.float32([[0, 5, 1]]))
    _frame_in_modes(sc, 320, 200, oracle)
    import torch
    fb = torch.zeros(200, 320, 4, dtype=torch.uint8, device="cuda")
    with rtamd.DeviceScene(sc) as ds:
        ds.render_device(fb.data_ptr(), None, 320, 200)
        d = ds.describe()
    assert 0 < d["screen_cells_with_list"] < (320 // 16) * ((200 + 15) // 16)
    assert d["light_cells_without_list"] > 0


def test_wild_scene_and_band_sharding(oracle):
    import torch
    sc = _wild_scene(105, 1500)
    cam = rtamd.default_camera(100)
    cam.origin[0], cam.origin[1], cam.origin[2] = 3.0, -2.0, 60.0
    w, h = 480, 296
    fo, io, no = oracle.render(sc, w, h, cam=cam)
    fb = torch.zeros(h, w, 4, dtype=torch.uint8, device="cuda")
    ids = torch.zeros(h, w, dtype=torch.int32, device="cuda")
    with rtamd.DeviceScene(sc) as ds:
        for k in range(3):      # three ranks' bands into one frame
            ds.render_device(fb.data_ptr(), ids.data_ptr(), w, h, cam=cam, band_rows=8, band_index=k,
                             band_count=3)
        torch.cuda.synchronize()
    assert_frame_parity(fb.cpu().numpy(), ids.cpu().numpy(), fo, io, "wild scene in three shares")


def test_camera_and_size_changes_on_a_resident_scene(oracle):
    """The screen lists belong to one view: a resident scene makes them anew when the camera or
    the frame size changes, and goes back and forth between views."""
    import torch
    sc = _cloud(8, 800)
    cams = []
    for origin in [(0, 0, 1), (10, 3, 5), (0, 0, 1)]:
        cam = rtamd.default_camera(80)
        cam.origin[0], cam.origin[1], cam.origin[2] = origin
        cams.append(cam)
    with rtamd.DeviceScene(sc) as ds:
        for cam, (w, h) in zip(cams + cams, [(320, 200)] * 3 + [(200, 120), (200, 120), (320, 200)]):
            fb = torch.zeros(h, w, 4, dtype=torch.uint8, device="cuda")
            ids = torch.zeros(h, w, dtype=torch.int32, device="cuda")
            ds.render_device(fb.data_ptr(), ids.data_ptr(), w, h, cam=cam)
            torch.cuda.synchronize()
            fo, io, _ = oracle.render(sc, w, h, cam=cam)
            assert_frame_parity(fb.cpu().numpy(), ids.cpu().numpy(), fo, io, f"{w}x{h}")


def test_bounce_frames_with_lists(oracle):
    sc = _cloud(9, 300, n_lights=2)
    sc.mat_sphere = rtamd.seeded_materials(sc.s_size, 3)
    for depth in (2, 5):
        fo, io, no = oracle.render(sc, 240, 160, max_depth=depth)
        fb, ids, n = rtamd.render(sc, 240, 160, max_depth=depth)
        assert_frame_parity(fb, ids, fo, io, f"depth {depth}")
        assert n == no


def test_no_list_lacks_a_sphere_a_ray_could_hit():
    """rt_verify_sphere_lists: every sphere against every pixel ray and primary shadow ray; a
    sphere that passes the reference's discriminant test must be in the ray's list."""
    for sc, (w, h), cam in [(rtamd.grid_scene(100, 100), (640, 360), None),
                            (_wild_scene(106, 700), (320, 200), "moved"),
                            (_cloud(10, 900, n_lights=3), (400, 250), None),
                            (_cloud(11, 500, n_lights=1, light_at=(0, 0, -90)), (300, 200), "wide")]:
        c = rtamd.default_camera(140 if cam == "wide" else 90)
        if cam == "moved":
            c.origin[0], c.origin[1], c.origin[2] = 3.0, -2.0, 60.0
        with rtamd.DeviceScene(sc) as ds:
            served, missing, s_served, s_missing = ds.verify_sphere_lists(w, h, c)
        assert missing == 0 and s_missing == 0, (served, missing, s_served, s_missing)
        if cam is None:        # (large spheres and a 140-degree view leave many cells without a list)
            assert served > 0.5 * w * h and s_served > 0
