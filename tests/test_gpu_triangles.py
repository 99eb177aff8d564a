"""The triangle hierarchy (csrc/rt_tribvh.h): rays walk it instead of looping
over every triangle (renderer.cpp:156-166) once a scene has 16 or more.  What
is tested changes, the result must not: every test here demands the same
kind / index / P bits, hit ids, pixels and ray counts as the loop — taken from
the CPU oracle where it finishes in seconds, and from the library's own plain
loop (rt_set_screen(0)) at sizes only a GPU can brute-force.
"""
import numpy as np
import pytest

import rtamd

pytestmark = pytest.mark.gpu


def check_rays(sc, o, d, oracle):
    with rtamd.DeviceScene(sc) as ds:
        k1, i1, p1 = ds.check_intersection(o, d)
    k0, i0, p0 = oracle.check_intersection(sc, o, d)
    assert np.array_equal(k0, k1), f"{int((k0 != k1).sum())} kinds differ"
    assert np.array_equal(i0[k0 != 0], i1[k0 != 0])
    m = k0 != 0
    assert np.array_equal(p0.view(np.uint32)[m], p1.view(np.uint32)[m])
    return k0, i0, p0


def rays_at_mesh(sc, m, seed, origin_scale):
    """Rays from a cloud around the world origin that the reference's triangle
    test reports as hitting chosen triangles: maths.cpp:77-78 computes
    t = (N.orig + N.p1) / N.dir, which is the textbook test seen from the origin
    mirrored in the plane through the world origin with normal N, so the rays are
    aimed from there."""
    rng = np.random.RandomState(seed)
    o = rng.normal(scale=origin_scale, size=(m, 3)).astype(np.float32)
    with np.errstate(all="ignore"):
        t = sc.tris.reshape(-1, 3, 3)[rng.randint(0, sc.t_size, m)].astype(np.float64)
        n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        n = np.nan_to_num(n)
        mirrored = o - 2 * (n * o).sum(1, keepdims=True) * n
        w = rng.dirichlet((1, 1, 1), m)
        tgt = np.nan_to_num((t * w[:, :, None]).sum(1), posinf=0, neginf=0)
        d = (tgt - mirrored).astype(np.float32)
    d[: m // 2] /= np.linalg.norm(d[: m // 2], axis=1, keepdims=True)
    return o, d


@pytest.mark.parametrize("rough", [0.0, 1.5, 6.0])
def test_random_rays_against_mesh(oracle, rough):
    """check_intersection on a 7 200-triangle mesh + spheres: origins near the
    eye (tight bounds), tens of units out and hundreds of units out (the
    mirrored-origin spread dominates), unit and unnormalised directions; then
    second-generation rays from the reported hit points."""
    sc = rtamd.mesh_scene(60, 60, grid=8, roughness=rough)
    for scale, seed in ((1.0, 1), (40.0, 2), (400.0, 3)):
        o, d = rays_at_mesh(sc, 6000, seed, scale)
        k0, i0, p0 = check_rays(sc, o, d, oracle)
        assert (k0 == 1).mean() > 0.05, "rays were meant to reach the mesh"
        hit = k0 != 0
        rng = np.random.RandomState(seed + 10)
        to_light = sc.lights[0] - p0[hit]
        to_light /= np.linalg.norm(to_light, axis=1, keepdims=True)
        o2 = np.concatenate([p0[hit], p0[hit]])
        d2 = np.concatenate([to_light, rng.normal(size=to_light.shape)]).astype(np.float32)
        check_rays(sc, o2, d2, oracle)


def test_awkward_triangles_and_rays(oracle):
    """Triangles the hierarchy refuses (slivers, needles, zero area, huge or
    non-finite corners) are still tested by every ray; rays with zero, huge,
    tiny or non-finite numbers get the loop's answer."""
    rng = np.random.RandomState(7)
    base = rtamd.mesh_scene(12, 10, grid=4)
    extra = []
    for k in range(40):   # slivers: two corners a hair apart
        p = rng.uniform(-30, 30, 3) + [0, 0, -60]
        q = p + rng.normal(size=3) * 20
        extra.append(np.concatenate([p, q, p + rng.normal(size=3) * 1e-4]))
    for k in range(20):   # needles with a sharp corner
        p = rng.uniform(-30, 30, 3) + [0, 0, -60]
        e = rng.normal(size=3) * 15
        extra.append(np.concatenate([p, p + e, p + e * 0.5 + rng.normal(size=3) * 0.02]))
    extra.append(np.zeros(9))                                     # zero area
    extra.append(np.concatenate([[1, 2, -50]] * 3))               # a point
    extra.append(np.array([0, 0, -40, 1e20, 0, -40, 0, 1e20, -40]))
    extra.append(np.array([0, 0, -40, np.inf, 0, -40, 0, 5, -40]))
    extra.append(np.array([np.nan, 0, -40, 3, 0, -40, 0, 5, -40]))
    extra.append(np.array([-5e3, -5e3, -70, 5e3, -5e3, -70, 0, 8e3, -70]))  # big but sane
    tris = np.concatenate([base.tris, np.array(extra, np.float32)])
    col = np.concatenate([base.color_tri, rng.randint(0, 256, (len(extra), 3)).astype(np.uint8)])
    sc = rtamd.Scene(tris, col, base.spheres, base.radius, base.color_sphere, base.lights)
    o, d = rays_at_mesh(sc, 4000, 5, 30.0)
    # aim a share of the rays at the awkward ones
    odd = rtamd.Scene(tris[-len(extra):], col[-len(extra):], base.spheres, base.radius,
                      base.color_sphere, base.lights)
    o[:1500], d[:1500] = rays_at_mesh(odd, 1500, 6, 30.0)
    special = [
        ([0, 0, 1], [0, 0, 0]), ([0, 0, 1], [0, 0, -1e-25]), ([0, 0, 1], [0, 0, -1e25]),
        ([0, 0, 1], [np.nan, 0, -1]), ([np.inf, 0, 1], [0, 0, -1]), ([1e25, 0, 1], [-1, 0, -1e-3]),
        ([0, 0, 0], [0, 0, -1]), ([0, 0, -75], [0, 0, -1]), ([0, 0, -75], [0, 0, 1]),
        ([3e19, 0, 0], [-1, 0, 0]),
    ]
    o = np.concatenate([o, np.array([s[0] for s in special], np.float32)])
    d = np.concatenate([d, np.array([s[1] for s in special], np.float32)])
    with np.errstate(all="ignore"):
        check_rays(sc, o, d, oracle)


@pytest.mark.parametrize("rough", [0.0, 3.0])
def test_frames_equal_the_plain_loop(rough):
    """20 000 triangles + 100 spheres, 640x360, depth 1 and 5: the frame through
    the hierarchy against the same library looping over every primitive."""
    sc = rtamd.mesh_scene(100, 100, grid=10, roughness=rough)
    for depth in (1, 5):
        try:
            rtamd.set_screen("off")
            f0, i0, n0 = rtamd.render(sc, 640, 360, max_depth=depth)
        finally:
            rtamd.set_screen("auto")
        f1, i1, n1 = rtamd.render(sc, 640, 360, max_depth=depth)
        assert np.array_equal(i0, i1)
        assert np.array_equal(f0, f1)
        assert n0 == n1
        assert ((i1 >= 0) & (i1 < sc.t_size)).mean() > 0.3


def test_mesh_rows_against_oracle(oracle):
    """80 000 triangles + 1 024 spheres at 1280x720, depth 4; the oracle checks
    every 90th row."""
    sc = rtamd.mesh_scene(200, 200, grid=32)
    rows = (45, 720, 90)
    fb, ids, n = rtamd.render(sc, 1280, 720, max_depth=4)
    fo, io, _ = oracle.render(sc, 1280, 720, max_depth=4, rows=rows)
    sel = slice(*rows)
    assert np.array_equal(ids[sel], io[sel])
    assert np.array_equal(fb[sel], fo[sel])


def test_million_triangle_frame_properties():
    """The 1 M-triangle configuration at 1920x1080, depth 8, on one GPU (the
    oracle cannot finish it; rows of it are pinned by the reduced sizes above).
    Properties that hold at any size: the frame is deterministic, the image
    equals the union of two row-band shards, and rays = pixels + one shadow ray
    per hit and light at depth 1."""
    import torch
    sc = rtamd.mesh_scene(1000, 500, grid=100)
    assert sc.t_size == 1_000_000
    w, h = 1920, 1080
    dev = torch.device("cuda", 0)
    with rtamd.DeviceScene(sc) as ds:
        def frame(depth, band_index=0, band_count=1, ids=False):
            fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
            idb = torch.full((h, w), -7, dtype=torch.int32, device=dev) if ids else None
            cnt = torch.zeros(1, dtype=torch.int64, device=dev)
            ds.render_device(fb.data_ptr(), idb.data_ptr() if ids else None, w, h, max_depth=depth,
                             band_rows=8, band_index=band_index, band_count=band_count,
                             dev_ray_count=cnt.data_ptr())
            torch.cuda.synchronize()
            assert not ds.queue_overflowed()
            return fb.cpu().numpy(), (idb.cpu().numpy() if ids else None), int(cnt.item())
        f1, i1, n1 = frame(1, ids=True)
        assert n1 == w * h + int((i1 >= 0).sum()) * sc.l_size
        assert ((i1 >= 0) & (i1 < sc.t_size)).mean() > 0.1   # the sphere block hides most of it
        fa, _, na = frame(8)
        fb_, _, nb = frame(8)
        assert np.array_equal(fa, fb_) and na == nb and na > 2 * n1
        s0, _, m0 = frame(8, 0, 2)
        s1, _, m1 = frame(8, 1, 2)
        assert m0 + m1 == na
        rows = np.arange(h) // 8 % 2
        assert np.array_equal(s0[rows == 0], fa[rows == 0])
        assert np.array_equal(s1[rows == 1], fa[rows == 1])


def test_million_triangles_8k_depth8_pixel_sample(oracle):
    """The triangle-heavy configuration at its full size (1 M triangles + 10 000
    spheres, 7680x4320, depth 8) on one GPU; the oracle renders 1 024 random
    pixels and four whole 8x8 tiles of it (about 9 000 rays against every
    primitive) and must agree on ids and bytes."""
    import torch
    sc = rtamd.mesh_scene(1000, 500, grid=100)
    w, h = 7680, 4320
    dev = torch.device("cuda", 0)
    fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
    ids = torch.zeros(h, w, dtype=torch.int32, device=dev)
    with rtamd.DeviceScene(sc) as ds:
        ds.render_device(fb.data_ptr(), ids.data_ptr(), w, h, max_depth=8)
        torch.cuda.synchronize()
        assert not ds.queue_overflowed()
    rng = np.random.RandomState(99)
    px = [np.stack([rng.randint(0, h, 1024), rng.randint(0, w, 1024)], 1)]
    for ti, tj in ((0, 0), (270, 480), (100, 900), (539, 959)):
        ii, jj = np.meshgrid(np.arange(8) + 8 * ti, np.arange(8) + 8 * tj, indexing="ij")
        px.append(np.stack([ii.ravel(), jj.ravel()], 1))
    px = np.concatenate(px)
    want, want_ids, _ = oracle.render_pixels(sc, w, h, 8, px)
    got = fb.cpu().numpy()[px[:, 0], px[:, 1]]
    got_ids = ids.cpu().numpy()[px[:, 0], px[:, 1]]
    assert np.array_equal(got_ids, want_ids)
    assert np.array_equal(got, want)
    assert ((want_ids >= 0) & (want_ids < sc.t_size)).sum() > 50


def test_mesh_scene_against_the_reference():
    """The same 2 000-triangle scene the real reference rendered at 2560x1440
    (tests/golden/golden_mesh_v1.npz): the frame through the triangle hierarchy
    must have the reference's row CRCs, framebuffer and hit ids, and
    check_intersection the reference's kind / index / P bits on 20 000 rays."""
    import os
    import sys
    import zlib
    sys.path.insert(0, os.path.join(rtamd.REPO_DIR, "oracle"))
    import gen_golden_mesh
    g = np.load(os.path.join(rtamd.REPO_DIR, "tests", "golden", "golden_mesh_v1.npz"))
    sc = gen_golden_mesh.mesh()
    assert sc.t_size == 2000
    fb, ids, n = rtamd.render(sc, 2560, 1440)
    crc = lambda a: np.array([zlib.crc32(np.ascontiguousarray(a[i]).tobytes()) for i in range(a.shape[0])], np.uint32)
    assert np.array_equal(crc(ids), g["mesh_idcrc"])
    assert np.array_equal(crc(fb), g["mesh_fbcrc"])
    o, d = gen_golden_mesh.mesh_rays(sc)
    with rtamd.DeviceScene(sc) as ds:
        kind, index, P = ds.check_intersection(o, d)
    assert np.array_equal(kind, g["mesh_rays_kind"].astype(np.int32))
    m = kind != 0
    assert np.array_equal(index[m], g["mesh_rays_index"][m])
    assert np.array_equal(P.view(np.uint32)[m], g["mesh_rays_P"][m])
