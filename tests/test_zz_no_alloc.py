"""rt_render_device_noalloc: the frame of rt_render_device from the library's
fixed array, with no allocation.  It has no counterpart in the reference; its
comparator is rt_render_device (itself pinned to the oracle by the other tests)
to the same bar: framebuffer, ids and ray count bit-exact.  Two calls in a row,
the second with a smaller input, so that what the first call left in the shared
array cannot leak into the second."""
import numpy as np
import pytest

import golden_scenes
import rtamd

pytestmark = pytest.mark.gpu


def _scene(seed, n_spheres):
    sc = golden_scenes.mixed_scene(seed=seed, n_spheres=n_spheres, n_tris=8, n_lights=2)
    sc.mat_sphere = rtamd.seeded_materials(sc.s_size)
    sc.mat_tri = rtamd.seeded_materials(sc.t_size, seed=7)
    return sc


def _frame(ds, w, h, depth, noalloc):
    import torch
    dev = torch.device("cuda", 0)
    fb = torch.full((h, w, 4), 7, dtype=torch.uint8, device=dev)
    ids = torch.full((h, w), -5, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    over = torch.full((1,), 9, dtype=torch.int32, device=dev)
    if noalloc:
        ds.render_device_noalloc(fb.data_ptr(), ids.data_ptr(), w, h, max_depth=depth,
                                 dev_ray_count=cnt.data_ptr(), dev_overflow=over.data_ptr())
    else:
        ds.render_device(fb.data_ptr(), ids.data_ptr(), w, h, max_depth=depth,
                         dev_ray_count=cnt.data_ptr())
    torch.cuda.synchronize()
    if noalloc:
        assert int(over.item()) == 0
    else:
        assert not ds.queue_overflowed()
    return fb.cpu().numpy(), ids.cpu().numpy(), int(cnt.item())


def test_noalloc_equals_render_device_twice_second_call_smaller(oracle):
    import torch
    big, small = _scene(91, 160), _scene(92, 40)
    with rtamd.DeviceScene(big) as d_big, rtamd.DeviceScene(small) as d_small:
        assert 0 < d_big.noalloc_scratch_needed(224, 224, 8) <= 64 << 20
        want_big = _frame(d_big, 224, 224, 8, noalloc=False)
        want_small = _frame(d_small, 112, 72, 5, noalloc=False)
        got_big = _frame(d_big, 224, 224, 8, noalloc=True)
        free_before = torch.cuda.mem_get_info()[0]
        got_small = _frame(d_small, 112, 72, 5, noalloc=True)   # smaller scene, frame, depth
        got_big2 = _frame(d_big, 224, 224, 8, noalloc=True)
        assert torch.cuda.mem_get_info()[0] == free_before      # nothing was allocated
        for got, want in ((got_big, want_big), (got_small, want_small), (got_big2, want_big)):
            assert np.array_equal(got[1], want[1])
            assert np.array_equal(got[0], want[0])
            assert got[2] == want[2]
    # and the comparator is the oracle's frame
    fo, io, no = oracle.render(small, 112, 72, max_depth=5)
    assert np.array_equal(want_small[0], fo) and np.array_equal(want_small[1], io)
    assert want_small[2] == no


def test_noalloc_refuses_what_does_not_fit_and_serves_frames_without_bounces():
    import torch
    sc = _scene(93, 60)
    with rtamd.DeviceScene(sc) as ds:
        assert ds.noalloc_scratch_needed(1920, 1080, 8) > 64 << 20
        fb = torch.zeros(1080, 1920, 4, dtype=torch.uint8, device="cuda")
        with pytest.raises(rtamd.RtError) as e:
            ds.render_device_noalloc(fb.data_ptr(), None, 1920, 1080, max_depth=8)
        assert e.value.code == rtamd.RT_ERR_CAPACITY
        assert int(fb.max()) == 0                      # nothing was rendered
        # depth 1 needs none of the array, whatever the size
        assert ds.noalloc_scratch_needed(1920, 1080, 1) == 0
        a = _frame(ds, 1920, 1080, 1, noalloc=True)
        b = _frame(ds, 1920, 1080, 1, noalloc=False)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
