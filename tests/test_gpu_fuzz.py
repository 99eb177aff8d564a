"""A slice of the randomised parity campaign (tools/fuzz_parity.py): random
sphere clouds with radii over three decades, triangle soups of 0..300 triangles
(both the looped and the hierarchy path), random materials, lights, cameras,
sizes and depths.  Frame (ids, bytes, ray count) and a batch of random rays
(kind, index, P bits) must equal the CPU oracle's."""
import os
import sys

import numpy as np
import pytest

import rtamd

sys.path.insert(0, os.path.join(rtamd.REPO_DIR, "tools"))
import fuzz_parity  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [100, 120])
def test_random_scenes(oracle, first):
    for seed in range(first, first + 20):
        sc, cam, rng = fuzz_parity.scene(seed)
        w, h = int(rng.choice([64, 100, 192])), int(rng.choice([40, 72, 108]))
        depth = int(rng.choice([1, 2, 4, 9]))
        fb, ids, n = rtamd.render(sc, w, h, max_depth=depth, cam=cam)
        fo, io, no = oracle.render(sc, w, h, max_depth=depth, cam=cam)
        assert np.array_equal(ids, io), seed
        assert np.array_equal(fb, fo), seed
        assert n == no, seed
        ro = rng.normal(scale=rng.choice([2.0, 30.0, 300.0]), size=(2000, 3)).astype(np.float32)
        rd = rng.normal(size=(2000, 3)).astype(np.float32)
        rd[::7, rng.randint(0, 3)] = 0
        with rtamd.DeviceScene(sc) as ds:
            k1, i1, p1 = ds.check_intersection(ro, rd)
        k0, i0, p0 = oracle.check_intersection(sc, ro, rd)
        hit = k0 != 0
        assert np.array_equal(k0, k1), seed
        assert np.array_equal(i0[hit], i1[hit]), seed
        assert np.array_equal(p0.view(np.uint32)[hit], p1.view(np.uint32)[hit]), seed
