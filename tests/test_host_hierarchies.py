"""The hierarchy builders are host code (csrc/rt_bvh.cpp, csrc/rt_tribvh.cpp).
rt_host_check_hierarchies builds both trees for a scene without touching a GPU
and counts violated invariants: every sphere inside all its ancestor boxes, every
triangle corner inside all its ancestor spheres, every triangle normal inside
its ancestors' normal cones, distance bounds that are true lower bounds, every
primitive in exactly one place."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import golden_scenes
import rtamd

sys.path.insert(0, os.path.join(rtamd.REPO_DIR, "tools"))
import fuzz_parity  # noqa: E402


def check(sc):
    out = (C.c_int64 * 8)()
    d = sc.desc()
    fn = rtamd.lib().rt_host_check_hierarchies
    fn.argtypes = [C.POINTER(rtamd.SceneDesc), C.POINTER(C.c_int64)]
    assert fn(C.byref(d), out) == 0
    return list(out)


def assert_sound(o):
    assert o[1] == 0 and o[2] == 0 and o[3] == 0 and o[6] == 0 and o[7] == 0, o


def test_grids_and_mixed_scenes():
    for sc in (rtamd.grid_scene(10, 10), rtamd.grid_scene(30, 30), rtamd.grid_scene(100, 100),
               golden_scenes.mixed_scene(), golden_scenes.mixed_scene(seed=9, n_spheres=500, n_tris=40)):
        o = check(sc)
        assert_sound(o)
        assert o[0] > 0
    o = check(rtamd.grid_scene(100, 100))
    # 10 000 single-sphere leaves: at least ceil(10 000 / 4) + ... + 1 = 3 334 nodes; two rounds of
    # halving per node make 4^0 + ... + 4^6 = 5 461, the cuts by box area fewer
    assert 3334 <= o[0] < 5461


def test_sphere_tree_is_as_shallow_as_four_way_allows():
    """The checker counts a tree that is deeper than ceil(log4 n) levels into out[3]: sizes around
    the powers of four, clustered and coincident centres (cuts by area alone would make chains)."""
    rng = np.random.RandomState(5)
    for n in (1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 1024, 1025, 4096, 4097):
        for kind in range(4):
            if kind == 0:      # uniform cloud
                c = rng.uniform(-50, 50, (n, 3))
            elif kind == 1:    # a few tight clusters and far outliers
                c = rng.normal(size=(n, 3)) * 0.01 + rng.randint(0, 3, (n, 1)) * 1000.0
                c[:: max(1, n // 7)] += 1e6
            elif kind == 2:    # all centres equal
                c = np.zeros((n, 3))
            else:              # a line with growing gaps
                c = np.zeros((n, 3))
                c[:, 0] = np.cumsum(np.arange(n, dtype=np.float64) ** 2)
            c = c.astype(np.float32) + np.float32([0, 0, -80])
            rad = rng.uniform(0.1, 2.0, n).astype(np.float32)
            sc = rtamd.Scene(np.zeros((0, 9), np.float32), np.zeros((0, 3), np.uint8), c, rad,
                             np.full((n, 3), 200, np.uint8), np.float32([[0, 50, 0]]))
            o = check(sc)
            assert_sound(o)
            assert o[0] >= 1


@pytest.mark.parametrize("rough", [0.0, 1.5, 6.0])
def test_meshes(rough):
    sc = rtamd.mesh_scene(80, 60, grid=6, roughness=rough)
    o = check(sc)
    assert_sound(o)
    assert o[4] > sc.t_size // 4 and o[5] == 0            # all 9 600 triangles in the tree


def test_awkward_triangles_stay_outside_the_tree():
    rng = np.random.RandomState(3)
    base = rtamd.mesh_scene(10, 10, grid=3)
    extra = []
    for k in range(30):   # slivers
        p = rng.uniform(-30, 30, 3) + [0, 0, -60]
        extra.append(np.concatenate([p, p + rng.normal(size=3) * 20, p + rng.normal(size=3) * 1e-4]))
    extra.append(np.zeros(9))
    extra.append(np.array([0, 0, -40, 1e20, 0, -40, 0, 1e20, -40]))
    extra.append(np.array([np.nan, 0, -40, 3, 0, -40, 0, 5, -40]))
    tris = np.concatenate([base.tris, np.array(extra, np.float32)])
    col = np.concatenate([base.color_tri, np.zeros((len(extra), 3), np.uint8)])
    sc = rtamd.Scene(tris, col, base.spheres, base.radius, base.color_sphere, base.lights)
    o = check(sc)
    assert_sound(o)
    assert o[5] >= 25        # the slivers and the huge one are tested by every ray instead


def test_triangle_tree_is_as_shallow_as_four_way_allows():
    """Triangle soups whose cuts by size alone would make chains: coincident copies, a cluster with
    far outliers, sizes around the powers of four."""
    rng = np.random.RandomState(11)
    one = np.float32([0, 0, -40, 3, 0, -40, 0, 5, -41])
    for n in (1, 2, 4, 5, 16, 17, 64, 65, 257, 1025):
        for kind in range(3):
            if kind == 0:
                t = np.tile(one, (n, 1))
            elif kind == 1:
                t = np.tile(one, (n, 1)) + np.tile(rng.normal(size=(n, 3)) * 0.05, (1, 3)).astype(np.float32)
                t[:: max(1, n // 5)] += np.float32([1e4, 0, 0] * 3)
            else:
                t = (np.tile(one, (n, 1)) + np.tile(rng.uniform(-80, 80, (n, 3)), (1, 3))).astype(np.float32)
            sc = rtamd.Scene(t.astype(np.float32), np.full((n, 3), 90, np.uint8), np.zeros((0, 3), np.float32),
                             np.zeros(0, np.float32), np.zeros((0, 3), np.uint8), np.float32([[0, 50, 0]]))
            o = check(sc)
            assert_sound(o)
            assert o[4] >= 1 and o[5] == 0, (n, kind, o)


def test_random_scenes():
    for seed in range(300, 340):
        sc, _, _ = fuzz_parity.scene(seed)
        assert_sound(check(sc))
