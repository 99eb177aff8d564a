"""max_depth > 1: material bounces.  The reference has no such thing
(renderer.cpp traces one primary and one shadow ray per light), so the only
comparator is the CPU oracle's definition of the extension: PARITY UNPINNED by
the reference beyond depth 1.  What IS pinned: with all-zero materials, or
max_depth = 1, the result equals the reference's golden output bit for bit.

GPU and oracle must agree exactly: framebuffer, ids and ray counts.
"""
import zlib

import numpy as np
import pytest

import golden_scenes
import rtamd

pytestmark = pytest.mark.gpu


def row_crc(a):
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(a.shape[0])], np.uint32)


def random_materials(n, seed):
    rng = np.random.RandomState(seed)
    m = np.zeros((n, 4), np.float32)
    cls = rng.randint(0, 4, n)
    m[cls == 1, 0] = rng.uniform(0.2, 0.9, (cls == 1).sum())            # mirrors
    m[cls == 2, 0] = 0.1                                                # glass
    m[cls == 2, 1] = rng.uniform(0.5, 0.85, (cls == 2).sum())
    m[cls == 2, 2] = rng.uniform(1.1, 2.0, (cls == 2).sum())
    m[cls == 3, 0] = 0.5                                                # both, ior 1
    m[cls == 3, 1] = 0.5
    m[cls == 3, 2] = 1.0
    return m


def material_scene(seed, **kw):
    sc = golden_scenes.mixed_scene(seed=seed, **kw)
    sc.mat_sphere = random_materials(sc.s_size, seed + 100)
    sc.mat_tri = random_materials(sc.t_size, seed + 200)
    return sc


def compare(sc, w, h, depth, oracle, rows=None):
    fb, ids, n = rtamd.render(sc, w, h, max_depth=depth)
    fo, io, no = oracle.render(sc, w, h, max_depth=depth, rows=rows)
    if rows is None:
        sel = slice(None)
    else:
        sel = slice(rows[0], rows[1], rows[2])
    assert np.array_equal(ids[sel], io[sel])
    bad = (fb[sel] != fo[sel]).any(-1)
    assert not bad.any(), f"{int(bad.sum())} pixels differ at depth {depth}; first " \
        f"{np.argwhere(bad)[:3].tolist()}"
    if rows is None:
        assert n == no
    return fb, n


@pytest.mark.parametrize("depth", [2, 3, 8])
def test_mixed_scene_depths(oracle, depth):
    sc = material_scene(51, n_spheres=60, n_tris=10, n_lights=2)
    compare(sc, 480, 270, depth, oracle)


def test_ray_count_grows_with_depth_then_saturates(oracle):
    sc = material_scene(52)
    counts = [rtamd.render(sc, 320, 180, max_depth=d, want_ids=False)[2] for d in (1, 2, 4, 8, 16)]
    assert counts[0] < counts[1] <= counts[2] <= counts[3] <= counts[4]
    # weights shrink below 2^-8 eventually: deeper limits add nothing
    assert rtamd.render(sc, 320, 180, max_depth=40, want_ids=False)[2] == \
        rtamd.render(sc, 320, 180, max_depth=63, want_ids=False)[2]


def test_zero_materials_any_depth_is_the_reference(golden):
    sc = golden_scenes.full_frame_scenes()["mixed"]
    sc.mat_sphere = np.zeros((sc.s_size, 4), np.float32)
    sc.mat_tri = np.zeros((sc.t_size, 4), np.float32)
    fb, ids, n = rtamd.render(sc, 2560, 1440, max_depth=8)
    assert np.array_equal(row_crc(fb), golden["mixed_fbcrc"])
    assert np.array_equal(row_crc(ids), golden["mixed_idcrc"])


def test_materials_are_ignored_at_depth_1(oracle):
    """max_depth = 1 is the reference's renderer whatever the materials say."""
    sc = material_scene(53)
    fb, n = compare(sc, 400, 240, 1, oracle)
    plain = golden_scenes.mixed_scene(seed=53)
    fb0, _, n0 = rtamd.render(plain, 400, 240, max_depth=1)
    assert np.array_equal(fb, fb0) and n == n0


def test_1k_grid_1080p_depth4(oracle):
    """The 1k-sphere configuration: 32x32 grid, 1920x1080, depth 4, seeded
    materials (one third diffuse / mirror / glass)."""
    sc = rtamd.material_grid_scene(32, 32)
    compare(sc, 1920, 1080, 4, oracle)


def test_10k_grid_4k_depth8_row_sample(oracle):
    """The 10k-sphere configuration at full size on the GPU; the oracle checks
    every 72nd row (30 rows) of it."""
    sc = rtamd.material_grid_scene(100, 100)
    compare(sc, 3840, 2160, 8, oracle, rows=(36, 2160, 72))


def test_screen_off_gives_the_same_bounces(oracle):
    sc = material_scene(54, n_spheres=120)
    out = {}
    for mode in ("off", "beam", "bvh", "auto"):
        try:
            rtamd.set_screen(mode)
            out[mode] = rtamd.render(sc, 320, 200, max_depth=6)
        finally:
            rtamd.set_screen("auto")
    fo, _, no = oracle.render(sc, 320, 200, max_depth=6, want_ids=False)
    for mode, (fb, _, n) in out.items():
        assert np.array_equal(fb, fo) and n == no, mode


def test_side_stream_and_mode_of_a_resident_scene(oracle):
    """rt_scene_set_side_stream moves the shadow walks to a stream of their own,
    rt_scene_set_screen switches the work-avoidance mode of a resident scene;
    pixels, ids and ray counts must not depend on either."""
    import torch
    sc = material_scene(55, n_spheres=150, n_tris=6, n_lights=2)
    w, h = 384, 216
    fo, io, no = oracle.render(sc, w, h, max_depth=6)
    dev = torch.device("cuda", 0)
    fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
    ids = torch.zeros(h, w, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    with rtamd.DeviceScene(sc) as ds:
        for side, one in ((0, 1), (0, 0), (1, 0)):
            for mode in ("auto", "off", "beam", "bvh"):
                ds.set_side_stream(side)
                ds.set_one_launch(one)
                ds.set_screen(mode)
                fb.zero_(), ids.zero_(), cnt.zero_()
                ds.render_device(fb.data_ptr(), ids.data_ptr(), w, h, max_depth=6,
                                 dev_ray_count=cnt.data_ptr())
                torch.cuda.synchronize()
                assert not ds.queue_overflowed()
                assert np.array_equal(ids.cpu().numpy(), io), (side, one, mode)
                assert np.array_equal(fb.cpu().numpy(), fo), (side, one, mode)
                assert int(cnt.item()) == no, (side, one, mode)
        with pytest.raises(rtamd.RtError):
            rtamd._check(rtamd.lib().rt_scene_set_screen(ds.handle, 9))


def test_bounce_frames_beyond_2_pow_26_pixels_are_refused():
    """A bounce ray carries its pixel in 26 bits of its tag: larger frames with
    bounces must be refused, not rendered wrongly.  (The check precedes every
    allocation and launch, so no such frame is ever allocated here.)"""
    import torch
    sc = material_scene(56)
    with rtamd.DeviceScene(sc) as ds:
        with pytest.raises(rtamd.RtError) as e:
            ds.render_device(1 << 12, None, 16384, 4104, max_depth=2)   # 67.2 M pixels
        assert e.value.code == rtamd.RT_ERR_UNSUPPORTED
        # without bounces (depth 1) the same scene object renders as usual
        fb = torch.zeros(64, 64, 4, dtype=torch.uint8, device="cuda")
        ds.render_device(fb.data_ptr(), None, 64, 64, max_depth=1)
        torch.cuda.synchronize()
        assert int(fb[..., 3].min()) == 255


def test_queue_capacity_grows_when_rays_multiply(oracle):
    """Half mirror, half transparent everywhere: every hit doubles the rays, up
    to 256 per pixel at depth 9 (the weight cut-off 2^-8), far beyond the default
    queue of 4 per pixel.  rt_render_device flags the overflow; with a larger queue
    (rt_scene_set_rays_per_pixel) the frame is complete; gpu_render_ids grows the
    queue by itself.  Complete frames equal the oracle's."""
    import torch
    sc = golden_scenes.mixed_scene(seed=77, n_spheres=80, n_tris=6, n_lights=1)
    sc.mat_sphere = np.tile(np.array([0.5, 0.5, 1.0, 0], np.float32), (sc.s_size, 1))
    sc.mat_tri = np.tile(np.array([0.5, 0.5, 1.0, 0], np.float32), (sc.t_size, 1))
    w, h, depth = 256, 144, 9
    fo, io, no = oracle.render(sc, w, h, max_depth=depth)
    assert no > 30 * w * h          # the rays really multiply
    fb, ids, n = rtamd.render(sc, w, h, max_depth=depth)
    assert np.array_equal(ids, io) and np.array_equal(fb, fo) and n == no
    dev = torch.device("cuda", 0)
    out = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
    with rtamd.DeviceScene(sc) as ds:
        # all levels in one launch: a wave works depth-first on a stack of its own,
        # which this many rays per pixel does not fill
        ds.render_device(out.data_ptr(), None, w, h, max_depth=depth)
        torch.cuda.synchronize()
        assert not ds.queue_overflowed()
        assert np.array_equal(out.cpu().numpy(), fo)
        # a launch per stage and level queues every level of the frame whole
        ds.set_one_launch(False)
        out.zero_()
        ds.render_device(out.data_ptr(), None, w, h, max_depth=depth)
        torch.cuda.synchronize()
        assert ds.queue_overflowed()
        ds.set_rays_per_pixel(256)
        ds.render_device(out.data_ptr(), None, w, h, max_depth=depth)
        torch.cuda.synchronize()
        assert not ds.queue_overflowed()
        assert np.array_equal(out.cpu().numpy(), fo)
        with pytest.raises(rtamd.RtError):
            ds.set_rays_per_pixel(257)


def test_bounce_machinery_that_does_not_fit_leaves_a_usable_scene(oracle):
    """A frame whose bounce machinery exceeds the scene's limit (or the device's
    memory: same code) fails with RT_ERR_CAPACITY and leaves the scene without
    any machinery; a smaller frame afterwards starts afresh and is complete."""
    import torch
    sc = material_scene(57)
    w, h = 256, 144
    fo, io, no = oracle.render(sc, w, h, max_depth=4)
    dev = torch.device("cuda", 0)
    fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def small():
        fb.zero_()
        cnt.zero_()
        ds.render_device(fb.data_ptr(), None, w, h, max_depth=4, dev_ray_count=cnt.data_ptr())
        torch.cuda.synchronize()
        assert not ds.queue_overflowed()
        assert np.array_equal(fb.cpu().numpy(), fo) and int(cnt.item()) == no

    with rtamd.DeviceScene(sc) as ds:
        assert ds.describe()["one_launch"]
        small()
        ds.set_arena_limit(1 << 30)
        small()                               # what it has already fits
        with pytest.raises(rtamd.RtError) as e:   # per-wave stacks for 63 levels: > 1 GiB
            ds.render_device(fb.data_ptr(), None, w, h, max_depth=63)
        assert e.value.code == rtamd.RT_ERR_CAPACITY
        assert not ds.queue_overflowed()      # reads no freed flag
        small()
        small()
        ds.set_arena_limit(0)
        ds.render_device(fb.data_ptr(), None, w, h, max_depth=63)
        torch.cuda.synchronize()
        assert not ds.queue_overflowed()
        assert np.array_equal(fb.cpu().numpy(), oracle.render(sc, w, h, max_depth=63)[0])


def test_invalid_materials_rejected():
    sc = golden_scenes.mixed_scene()
    sc.mat_sphere = np.zeros((sc.s_size, 4), np.float32)
    sc.mat_sphere[3] = [0.7, 0.6, 1.5, 0]
    with pytest.raises(rtamd.RtError):
        rtamd.render(sc, 32, 32, max_depth=3)
    with pytest.raises(rtamd.RtError):
        rtamd.render(golden_scenes.mixed_scene(), 32, 32, max_depth=64)


def test_triangle_mesh_configuration_reduced(oracle):
    """The triangle-heavy configuration (height-field mesh + spheres, materials
    on triangles too) at a size the oracle finishes: 4000 triangles, 256x144,
    depth 1 and depth 4.  The rays walk the triangle hierarchy; the second mesh
    has noisy vertex heights, so neighbouring normals disagree and the
    hierarchy's bounds are at their loosest."""
    for rough in (0.0, 6.0):
        sc = rtamd.mesh_scene(50, 40, roughness=rough)
        assert sc.t_size == 4000
        compare(sc, 256, 144, 1, oracle)
        compare(sc, 256, 144, 4, oracle)


def _device_frame(ds, w, h, depth, mode):
    import torch
    dev = torch.device("cuda", 0)
    fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
    ids = torch.zeros(h, w, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ds.set_screen(mode)
    ds.render_device(fb.data_ptr(), ids.data_ptr(), w, h, max_depth=depth,
                     dev_ray_count=cnt.data_ptr())
    torch.cuda.synchronize()
    assert not ds.queue_overflowed()
    return fb, ids, int(cnt.item())


def test_10k_grid_4k_depth8_every_row_equals_the_plain_loop():
    """The headline frame (10 000 spheres, 3840x2160, depth 8), all 2160 rows:
    the default mode against the library's own every-ray-every-primitive loop,
    which test_10k_grid_4k_depth8_row_sample pins to the oracle on sampled rows."""
    import torch
    sc = rtamd.material_grid_scene(100, 100)
    with rtamd.DeviceScene(sc) as ds:
        f0, i0, n0 = _device_frame(ds, 3840, 2160, 8, "off")
        for one in (1, 0):   # all levels in one launch / a launch per stage and level
            ds.set_one_launch(one)
            f1, i1, n1 = _device_frame(ds, 3840, 2160, 8, "auto")
            assert n0 == n1, one
            assert torch.equal(i0, i1), one
            assert torch.equal(f0, f1), one


def test_mesh_80k_720p_depth4_every_row_equals_the_plain_loop():
    """80 000 triangles + 1 024 spheres, 1280x720, depth 4, every row: triangle
    and sphere hierarchies against the plain loop."""
    import torch
    sc = rtamd.mesh_scene(200, 200, grid=32)
    with rtamd.DeviceScene(sc) as ds:
        f0, i0, n0 = _device_frame(ds, 1280, 720, 4, "off")
        f1, i1, n1 = _device_frame(ds, 1280, 720, 4, "auto")
        assert n0 == n1 and torch.equal(i0, i1) and torch.equal(f0, f1)


def test_which_machinery_a_scene_uses():
    """rt_scene_describe: the bench scene walks with LDS stacks and runs its bounce
    levels as one launch; a tree of more than 65 535 nodes does neither."""
    with rtamd.DeviceScene(rtamd.material_grid_scene(100, 100)) as ds:
        d = ds.describe()
        assert d["sphere_nodes"] > 3000 and d["sphere_levels"] <= 8
        assert d["lds_stacks"] and d["one_launch"] and d["triangle_nodes"] == 0
        ds.set_one_launch(False)
        assert not ds.describe()["one_launch"]
    with rtamd.DeviceScene(rtamd.material_grid_scene(500, 500)) as ds:
        d = ds.describe()
        assert d["sphere_nodes"] > 65535 and not d["lds_stacks"] and not d["one_launch"]


def test_tree_too_large_for_the_lds_stacks(oracle):
    """250 000 spheres: more nodes than a 16-bit stack entry can name, so the walks
    use their private stacks and the bounce levels run as a launch per stage.  Every
    row equals the plain loop's frame, sampled rows equal the oracle's."""
    sc = rtamd.material_grid_scene(500, 500)
    w, h, depth = 640, 360, 4
    try:
        fb, ids, n = rtamd.render(sc, w, h, max_depth=depth)
        rtamd.set_screen("off")
        fb0, ids0, n0 = rtamd.render(sc, w, h, max_depth=depth)
    finally:
        rtamd.set_screen("auto")
    assert n == n0 and np.array_equal(ids, ids0) and np.array_equal(fb, fb0)
    rows = (7, h, 60)
    fo, io, _ = oracle.render(sc, w, h, max_depth=depth, rows=rows)
    sel = slice(*rows)
    assert np.array_equal(ids[sel], io[sel]) and np.array_equal(fb[sel], fo[sel])


def test_frames_repeat_bit_for_bit_in_both_bounce_modes():
    """Which lane fetches which ray, and which idle lanes take over parts of whose
    walk at the end of a launch, differs from run to run; the frame must not.
    1 000 spheres at 1920x1080, depth 4, and 20 000 triangles + 100 spheres at
    640x360, depth 5: four frames with a launch per stage and level and four with one
    launch for all levels (the mesh scene always takes the former)."""
    import torch
    for sc, w, h, depth in ((rtamd.material_grid_scene(32, 32), 1920, 1080, 4),
                            (rtamd.mesh_scene(100, 100, grid=10, roughness=3.0), 640, 360, 5)):
        with rtamd.DeviceScene(sc) as ds:
            first = None
            for one in (0, 1):
                ds.set_one_launch(one)
                for _ in range(4):
                    f, i, n = _device_frame(ds, w, h, depth, "auto")
                    if first is None:
                        first = (f, i, n)
                    assert n == first[2] and torch.equal(i, first[1]) and torch.equal(f, first[0]), one


def test_million_triangle_1080p_frame_repeats_16_times_per_mode():
    """The frame on which walks in progress are shared between lanes the most: the
    1 M-triangle height field + 10 000 spheres at 1920x1080, depth 8.  Once a launch's
    queue is used up, idle lanes take over parts of other lanes' walks, and which lane
    helps whom differs from run to run (rt_walk.h); the order-free nearest-hit rule
    (renderer.cpp:159,170,181) must make that invisible.  Sixteen frames with the shadow
    walks on the scene's side stream and sixteen with every launch on one stream:
    identical bytes, ids and ray counts."""
    import torch
    sc = rtamd.mesh_scene(1000, 500, grid=100)
    with rtamd.DeviceScene(sc) as ds:
        first = None
        for side in (1, 0):
            ds.set_side_stream(side)
            for k in range(16):
                f, i, n = _device_frame(ds, 1920, 1080, 8, "auto")
                if first is None:
                    first = (f, i, n)
                assert n == first[2], (side, k)
                assert torch.equal(i, first[1]), (side, k)
                assert torch.equal(f, first[0]), (side, k)


@pytest.mark.parametrize("grid,levels", [(150, 8), (100, 7), (60, 6), (12, 4)])
def test_one_launch_kernel_for_every_depth_of_tree(oracle, grid, levels):
    """The one-launch kernel exists per depth of the sphere tree (a lane's LDS column is
    split between its stack of nodes and its parked leaves by what the tree needs:
    rt_trace.h): grids whose trees have 8, 7, 6 and 4 levels of inner nodes, all bounce
    levels in one launch against the launch-per-level path on the whole frame and
    against the oracle on every 16th row."""
    import torch
    sc = rtamd.material_grid_scene(grid, grid)
    w, h, depth = 960, 544, 5
    with rtamd.DeviceScene(sc) as ds:
        d = ds.describe()
        assert d["sphere_levels"] == levels and d["lds_stacks"] and d["one_launch"]
        f1, i1, n1 = _device_frame(ds, w, h, depth, "auto")
        ds.set_one_launch(0)
        f0, i0, n0 = _device_frame(ds, w, h, depth, "auto")
        assert n0 == n1 and torch.equal(i0, i1) and torch.equal(f0, f1)
    fo, io, _ = oracle.render(sc, w, h, max_depth=depth, rows=(0, h, 16))
    assert np.array_equal(i1.cpu().numpy()[::16], io[::16])
    assert np.array_equal(f1.cpu().numpy()[::16], fo[::16])
