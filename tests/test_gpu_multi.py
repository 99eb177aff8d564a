"""gpu_render_multi: one frame on several GPUs of one process (the offload
contract of renderer.cpp:13-30).  A one-GPU box has one real device; with
RT_AMD_LOGICAL_DEVICES=n the call takes up to n slots mapped onto the GPUs
present, each with its own kept scene, frame buffers and stream, so that the band
copies into slot 0's frame, the per-slot overflow retry and the ray-count sum run
as they would between n GPUs (what a rehearsal cannot show: peer access between
different devices and the xGMI links themselves).  The band bookkeeping is also
covered on CPU (tests/test_abi.py)."""
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


@pytest.fixture
def eight_slots(monkeypatch):
    monkeypatch.setenv("RT_AMD_LOGICAL_DEVICES", "8")
    rtamd.lib().rt_cache_clear()
    yield
    rtamd.lib().rt_cache_clear()


def test_rehearsal_of_2_4_8_devices_equals_the_single_call(eight_slots):
    """Byte for byte against gpu_render_ids at depth 1 and 5; 203 rows in bands of
    8 and 16 leave a ragged last band, 40 rows in bands of 8 leave three of eight
    devices without any band."""
    sc = golden_scenes.mixed_scene(seed=72, n_spheres=150, n_tris=10, n_lights=2)
    sc.mat_sphere = rtamd.seeded_materials(sc.s_size)
    b0 = int(rtamd.lib().rt_cache_builds())
    for w, h, band_rows in [(400, 232, 16), (320, 203, 8), (320, 203, 16), (256, 40, 8)]:
        for depth in (1, 5):
            f1, i1, n1 = rtamd.render(sc, w, h, max_depth=depth)
            for n in (2, 4, 8):
                fb, ids, rays = rtamd.render_multi(sc, w, h, max_depth=depth, n_devices=n,
                                                   band_rows=band_rows)
                assert np.array_equal(ids, i1), (w, h, band_rows, depth, n)
                assert np.array_equal(fb, f1), (w, h, band_rows, depth, n)
                assert rays == n1, (w, h, band_rows, depth, n)
    # one scene per slot, built once however many frames followed
    assert int(rtamd.lib().rt_cache_builds()) == b0 + 8


def test_rehearsal_overflow_retry_runs_on_every_device(eight_slots, oracle):
    """Half mirror, half transparent, depth 9: the default queues of the
    launch-per-level path (a scene with a triangle hierarchy) overflow on every
    slot; all slots must enlarge theirs and render again, and the sum of their ray
    counts must be the frame's."""
    sc = golden_scenes.mixed_scene(seed=77, n_spheres=80, n_tris=20, n_lights=1)
    sc.mat_sphere = np.tile(np.array([0.5, 0.5, 1.0, 0], np.float32), (sc.s_size, 1))
    sc.mat_tri = np.tile(np.array([0.5, 0.5, 1.0, 0], np.float32), (sc.t_size, 1))
    w, h, depth = 192, 104, 9
    fo, io, no = oracle.render(sc, w, h, max_depth=depth)
    assert no > 20 * w * h
    fb, ids, rays = rtamd.render_multi(sc, w, h, max_depth=depth, n_devices=4, band_rows=8)
    assert np.array_equal(ids, io) and np.array_equal(fb, fo) and rays == no


def test_rehearsal_headline_frame_split_eight_ways(eight_slots):
    """The bench frame (10 000 spheres, 3840x2160, depth 8) through eight slots
    against the single call."""
    sc = rtamd.material_grid_scene(100, 100)
    f1, i1, n1 = rtamd.render(sc, 3840, 2160, max_depth=8)
    fb, ids, rays = rtamd.render_multi(sc, 3840, 2160, max_depth=8, n_devices=8, band_rows=8)
    assert rays == n1 and np.array_equal(ids, i1) and np.array_equal(fb, f1)


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
    """Uses every GPU the process sees: ONE on the test box, where this is the
    single-device call again; only on a node with more does it cross devices."""
    import torch
    n = min(torch.cuda.device_count(), 8)
    if n == 1:
        print("one visible device: peer access and peer copies between devices not exercised")
    sc = rtamd.material_grid_scene(20, 20)
    fb, ids, rays = rtamd.render_multi(sc, 640, 360, max_depth=4, n_devices=n, band_rows=8)
    fo, io, no = oracle.render(sc, 640, 360, max_depth=4)
    assert np.array_equal(fb, fo) and np.array_equal(ids, io) and rays == no
