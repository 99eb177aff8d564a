"""CPU-only checks of the oracle's bounce extension (no reference counterpart:
parity unpinned beyond depth 1).  These pin the properties the GPU tests lean on."""
import numpy as np

import golden_scenes
import rtamd


def test_zero_materials_equal_depth_one(oracle):
    sc = golden_scenes.mixed_scene(seed=61)
    fb1, ids1, n1 = oracle.render(sc, 160, 90, max_depth=1)
    sc.mat_sphere = np.zeros((sc.s_size, 4), np.float32)
    sc.mat_tri = np.zeros((sc.t_size, 4), np.float32)
    fb8, ids8, n8 = oracle.render(sc, 160, 90, max_depth=8)
    assert np.array_equal(fb1, fb8) and np.array_equal(ids1, ids8) and n1 == n8


def test_mirror_adds_rays_and_light(oracle):
    sc = golden_scenes.mixed_scene(seed=62)
    sc.mat_sphere = np.zeros((sc.s_size, 4), np.float32)
    sc.mat_sphere[:, 0] = 0.5
    fb2, _, n2 = oracle.render(sc, 160, 90, max_depth=2)
    fb3, _, n3 = oracle.render(sc, 160, 90, max_depth=3)
    _, _, n1 = oracle.render(sc, 160, 90, max_depth=1)
    assert n3 > n2 > n1
    # a deeper limit only ever adds light: the shallower terms are identical and
    # every bounce adds a non-negative integer
    assert (fb3.astype(int) >= fb2.astype(int)).all()


def test_weight_cutoff_terminates(oracle):
    sc = golden_scenes.mixed_scene(seed=63)
    sc.mat_sphere = np.zeros((sc.s_size, 4), np.float32)
    sc.mat_sphere[:, 0] = 0.5   # 0.5^8 = 2^-8: the 9th bounce is never spawned
    sc.mat_tri = np.zeros((sc.t_size, 4), np.float32)
    sc.mat_tri[:, 0] = 0.5
    _, _, n20 = oracle.render(sc, 96, 54, max_depth=20)
    _, _, n60 = oracle.render(sc, 96, 54, max_depth=60)
    assert n20 == n60


def test_refract_straight_through_with_ior_one(oracle):
    """ior = 1 does not bend: refract(d, n, 1) is d up to rounding."""
    import ctypes as C
    d = np.array([0.3, -0.2, -0.9327379], np.float32)
    n = np.array([0.0, 0.0, 1.0], np.float32)
    out = np.zeros(3, np.float32)
    oracle.L.oracle_refract(d.ctypes.data, n.ctypes.data, C.c_float(1.0), out.ctypes.data)
    assert np.allclose(out, d, atol=1e-6)
