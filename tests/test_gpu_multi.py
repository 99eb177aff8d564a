"""gpu_render_multi: one frame on several GPUs of one process (the offload
contract of renderer.cpp:13-30).  A one-GPU box can only run n_devices = 1; the
band bookkeeping for larger splits is covered on CPU (tests/test_abi.py) and by
rendering every share of a split on this one GPU."""
import numpy as np
import pytest

import golden_scenes
import rtamd

pytestmark = pytest.mark.gpu


def test_multi_with_one_device_equals_the_single_call(oracle):
    sc = golden_scenes.mixed_scene(seed=71, n_spheres=120, n_tris=10, n_lights=2)
    sc.mat_sphere = rtamd.seeded_materials(sc.s_size)
    for depth in (1, 5):
        fb, ids, n = rtamd.render_multi(sc, 400, 232, max_depth=depth, n_devices=1, band_rows=16)
        f1, i1, n1 = rtamd.render(sc, 400, 232, max_depth=depth)
        assert np.array_equal(fb, f1) and np.array_equal(ids, i1) and n == n1
        fo, io, no = oracle.render(sc, 400, 232, max_depth=depth)
        assert np.array_equal(fb, fo) and np.array_equal(ids, io) and n == no


def test_multi_rejects_what_it_cannot_do():
    import torch
    sc = golden_scenes.mixed_scene()
    with pytest.raises(rtamd.RtError):
        rtamd.render_multi(sc, 64, 64, n_devices=torch.cuda.device_count() + 1)
    with pytest.raises(rtamd.RtError):
        rtamd.render_multi(sc, 64, 64, n_devices=0)
    with pytest.raises(rtamd.RtError):
        rtamd.render_multi(sc, 64, 64, n_devices=1, band_rows=12)


def test_all_visible_devices(oracle):
    """Uses every GPU the process sees (1 on the test box, more on a node)."""
    import torch
    n = min(torch.cuda.device_count(), 8)
    sc = rtamd.material_grid_scene(20, 20)
    fb, ids, rays = rtamd.render_multi(sc, 640, 360, max_depth=4, n_devices=n, band_rows=8)
    fo, io, no = oracle.render(sc, 640, 360, max_depth=4)
    assert np.array_equal(fb, fo) and np.array_equal(ids, io) and rays == no
