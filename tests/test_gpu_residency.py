"""Scene residency of the drop-in calls.  The reference uploads its scene once
per render() (`map(to:)`, renderer.cpp:13-21) and its harness pattern is one
check_intersection per pixel with the same arrays (renderer.hpp:21-23): the
one-call entry points keep the device copy of the last scene and rebuild it
only when pointers, sizes or bytes change."""
import ctypes as C

import numpy as np
import pytest

import golden_scenes
import rtamd

pytestmark = pytest.mark.gpu


def builds():
    return int(rtamd.lib().rt_cache_builds())


def test_second_render_of_the_same_arrays_builds_nothing(oracle):
    sc = golden_scenes.mixed_scene(seed=61, n_spheres=60, n_tris=20)
    rtamd.lib().rt_cache_clear()
    b0 = builds()
    fb1, ids1, n1 = rtamd.render(sc, 320, 200)
    assert builds() == b0 + 1
    fb2, ids2, n2 = rtamd.render(sc, 320, 200)
    fb3 = rtamd.render_plain(sc, 200, 120)          # other size, same scene
    assert builds() == b0 + 1
    assert np.array_equal(fb1, fb2) and np.array_equal(ids1, ids2) and n1 == n2
    fo, io, no = oracle.render(sc, 320, 200)
    assert np.array_equal(fb1, fo) and np.array_equal(ids1, io) and n1 == no
    assert np.array_equal(fb3, oracle.render(sc, 200, 120)[0])


def test_a_changed_array_invalidates_the_kept_scene(oracle):
    sc = golden_scenes.mixed_scene(seed=62, n_spheres=50, n_tris=10)
    rtamd.lib().rt_cache_clear()
    rtamd.render(sc, 256, 144)
    b0 = builds()
    # same pointers and sizes, one float changed in place
    sc.spheres[11, 2] += np.float32(3.0)
    fb, ids, n = rtamd.render(sc, 256, 144)
    assert builds() == b0 + 1
    fo, io, no = oracle.render(sc, 256, 144)
    assert np.array_equal(fb, fo) and np.array_equal(ids, io) and n == no
    # a colour byte, a light, a material
    sc.color_sphere[0, 1] ^= 0x55
    assert np.array_equal(rtamd.render(sc, 256, 144)[0], oracle.render(sc, 256, 144)[0])
    sc.lights[0, 0] += np.float32(1.0)
    assert np.array_equal(rtamd.render(sc, 256, 144)[0], oracle.render(sc, 256, 144)[0])
    assert builds() == b0 + 3
    sc.mat_sphere = rtamd.seeded_materials(sc.s_size)
    f4 = rtamd.render(sc, 256, 144, max_depth=4)[0]
    assert builds() == b0 + 4
    assert np.array_equal(f4, oracle.render(sc, 256, 144, max_depth=4)[0])
    rtamd.lib().rt_cache_clear()


def test_shim_check_intersection_per_ray_keeps_the_scene(oracle):
    """The reference's mangled check_intersection symbol, called once per ray with
    the same arrays: one build, every answer equal to the oracle's."""
    import os
    shim = C.CDLL(os.path.join(rtamd.PKG_DIR, "host", "librt_shim.so"))
    fn = getattr(shim, "_Z18check_intersectionPfiS_S_iS_PiS_S_")
    fn.restype = C.c_int
    sc = golden_scenes.mixed_scene(seed=63, n_spheres=80, n_tris=24)
    o, d = golden_scenes.random_rays(300, seed=5)
    ko, io, Po = oracle.check_intersection(sc, o, d)
    rtamd.lib().rt_cache_clear()
    b0 = builds()
    vp = C.c_void_p
    for r in range(len(o)):
        P = np.zeros(3, np.float32)
        idx = C.c_int(-7)
        k = fn(vp(sc.tris.ctypes.data), sc.t_size, vp(sc.spheres.ctypes.data),
               vp(sc.radius.ctypes.data), sc.s_size, vp(P.ctypes.data), C.byref(idx),
               vp(o[r].ctypes.data), vp(d[r].ctypes.data))
        assert k == ko[r]
        if k:
            assert idx.value == io[r] and np.array_equal(P.view(np.uint32), Po[r].view(np.uint32))
        else:
            assert idx.value == -7          # untouched on a miss, renderer.cpp:179-180
    assert builds() == b0 + 1
    rtamd.lib().rt_cache_clear()


def test_alternating_render_and_check_intersection_builds_once(oracle):
    """The reference's two symbols called in turn by one host with the same arrays
    (render() passes colours and lights, check_intersection() cannot): the scene
    that render() left resident serves the geometric queries, ten rounds, one
    build."""
    import os
    shim = C.CDLL(os.path.join(rtamd.PKG_DIR, "host", "librt_shim.so"))
    chk = getattr(shim, "_Z18check_intersectionPfiS_S_iS_PiS_S_")
    chk.restype = C.c_int
    ren = getattr(shim, "_Z6renderPhiPfS_iS0_S0_S_iS0_i")
    ren.restype = None
    sc = golden_scenes.mixed_scene(seed=64, n_spheres=70, n_tris=9)
    o, d = golden_scenes.random_rays(10, seed=6)
    ko, io, Po = oracle.check_intersection(sc, o, d)
    fo = oracle.render(sc, 2560, 1440)[0]        # the shim's compiled-in frame size
    fb = np.zeros((1440, 2560, 4), np.uint8)
    vp = C.c_void_p
    rtamd.lib().rt_cache_clear()
    b0 = builds()
    for r in range(10):
        ren(vp(fb.ctypes.data), 90, vp(sc.tris.ctypes.data), vp(sc.color_tri.ctypes.data),
            sc.t_size, vp(sc.spheres.ctypes.data), vp(sc.radius.ctypes.data),
            vp(sc.color_sphere.ctypes.data), sc.s_size, vp(sc.lights.ctypes.data), sc.l_size)
        if r in (0, 9):
            assert np.array_equal(fb, fo)
        P = np.zeros(3, np.float32)
        idx = C.c_int(-7)
        k = chk(vp(sc.tris.ctypes.data), sc.t_size, vp(sc.spheres.ctypes.data),
                vp(sc.radius.ctypes.data), sc.s_size, vp(P.ctypes.data), C.byref(idx),
                vp(o[r].ctypes.data), vp(d[r].ctypes.data))
        assert k == ko[r]
        if k:
            assert idx.value == io[r] and np.array_equal(P.view(np.uint32), Po[r].view(np.uint32))
    assert builds() == b0 + 1
    # the other way round: queries first build a geometry-only scene, the first
    # render() replaces it (it needs colours and lights), then nothing more
    rtamd.lib().rt_cache_clear()
    b0 = builds()
    for r in range(3):
        P = np.zeros(3, np.float32)
        idx = C.c_int(-7)
        chk(vp(sc.tris.ctypes.data), sc.t_size, vp(sc.spheres.ctypes.data),
            vp(sc.radius.ctypes.data), sc.s_size, vp(P.ctypes.data), C.byref(idx),
            vp(o[r].ctypes.data), vp(d[r].ctypes.data))
        ren(vp(fb.ctypes.data), 90, vp(sc.tris.ctypes.data), vp(sc.color_tri.ctypes.data),
            sc.t_size, vp(sc.spheres.ctypes.data), vp(sc.radius.ctypes.data),
            vp(sc.color_sphere.ctypes.data), sc.s_size, vp(sc.lights.ctypes.data), sc.l_size)
    assert builds() == b0 + 2
    assert np.array_equal(fb, fo)
    rtamd.lib().rt_cache_clear()


def test_trusted_pointers_skip_the_hash_and_locked_buffers_receive_the_frame(oracle):
    """rt_cache_trust_pointers: same pointers and sizes are the same scene, edits in
    place go unnoticed (that is the contract).  rt_alloc_host_buffer: a page-locked
    frame buffer receives the same bytes."""
    sc = golden_scenes.mixed_scene(seed=65, n_spheres=40, n_tris=8)
    L = rtamd.lib()
    L.rt_cache_clear()
    fo = oracle.render(sc, 320, 200)[0]
    L.rt_cache_trust_pointers(1)
    try:
        b0 = builds()
        with rtamd.HostBuffer(320 * 200 * 4) as hb:
            fb = hb.array.reshape(200, 320, 4)
            d = sc.desc()
            cam = rtamd.default_camera()
            for _ in range(3):
                fb[:] = 0
                rtamd._check(L.gpu_render(C.byref(d), C.byref(cam), fb.ctypes.data, 320, 200, 1))
                assert np.array_equal(fb, fo)
            assert builds() == b0 + 1
            sc.spheres[5, 2] += np.float32(2.0)  # in place: not noticed while trusted
            rtamd._check(L.gpu_render(C.byref(d), C.byref(cam), fb.ctypes.data, 320, 200, 1))
            assert builds() == b0 + 1 and np.array_equal(fb, fo)
            del fb
    finally:
        L.rt_cache_trust_pointers(0)
    fb2 = rtamd.render_plain(sc, 320, 200)   # hashed again: the edit is seen
    assert np.array_equal(fb2, oracle.render(sc, 320, 200)[0])
    assert not np.array_equal(fb2, fo)
    L.rt_cache_clear()
