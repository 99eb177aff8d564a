#!/usr/bin/env python3
"""Generate tests/golden/golden_mesh_v1.npz from the REAL reference: a scene
with enough triangles (2 000) for the library's triangle hierarchy to be in
use, rendered by the reference at its fixed 2560x1440, plus check_intersection
on rays aimed at the mesh from near and far origins.

Run in the build container only (needs oracle/_ref/libref.so):

    make -C oracle && python oracle/gen_golden_mesh.py

Written: per-row CRC32 of framebuffer and hit-id buffer, kind / index / P bits
of 20 000 rays, and the row CRCs of the reference's render of its own largest
benchmark scene, the 100 x 100 sphere grid.  Inputs are seeded (rtamd.mesh_scene, the ray generator
below); no reference source text is stored."""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from oracle_py import Ref, rtamd  # noqa: E402


def mesh():
    return rtamd.mesh_scene(40, 25, grid=6, materials=False)


def mesh_rays(sc, m=20000, seed=5):
    """Half of the rays are aimed so that the reference's triangle test reports
    a chosen triangle (from the origin mirrored in the plane through the world
    origin with the triangle's normal: maths.cpp:77-78), half are random."""
    rng = np.random.RandomState(seed)
    scale = rng.choice([1.0, 40.0, 300.0], m)[:, None]
    o = (rng.normal(size=(m, 3)) * scale).astype(np.float32)
    t = sc.tris.reshape(-1, 3, 3)[rng.randint(0, sc.t_size, m)].astype(np.float64)
    n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    mirrored = o - 2 * (n * o).sum(1, keepdims=True) * n
    w = rng.dirichlet((1, 1, 1), m)
    d = ((t * w[:, :, None]).sum(1) - mirrored).astype(np.float32)
    d[m // 2:] = rng.normal(size=(m - m // 2, 3)).astype(np.float32)
    d[::4] /= np.linalg.norm(d[::4], axis=1, keepdims=True)
    return o, d


def row_crc(a):
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(a.shape[0])], np.uint32)


def main():
    ref = Ref()
    sc = mesh()
    fb = ref.render(sc)
    ids = ref.hit_ids(sc)
    out = {"mesh_fbcrc": row_crc(fb), "mesh_idcrc": row_crc(ids)}
    print("mesh:", sc.t_size, "triangles,", sc.s_size, "spheres; triangle pixels %.3f, sphere pixels %.3f"
          % (((ids >= 0) & (ids < sc.t_size)).mean(), (ids >= sc.t_size).mean()))
    o, d = mesh_rays(sc)
    kind, index, P = ref.check_intersection(sc, o, d)
    out["mesh_rays_kind"] = kind.astype(np.int8)
    out["mesh_rays_index"] = index.astype(np.int32)
    out["mesh_rays_P"] = P.view(np.uint32)
    print("rays: triangle hits %d, sphere hits %d of %d" % ((kind == 1).sum(), (kind == 2).sum(), len(kind)))
    # the reference's own largest benchmark scene (BENCHMAX, main.cpp:372-375):
    # the 100 x 100 sphere grid, which is also the bench scene of this repository
    g100 = rtamd.grid_scene(100, 100)
    fb = ref.render(g100)
    ids = ref.hit_ids(g100)
    out["g100_fbcrc"], out["g100_idcrc"] = row_crc(fb), row_crc(ids)
    print("g100: sphere pixels %.3f" % (ids >= g100.t_size).mean())
    path = os.path.join(HERE, "..", "tests", "golden", "golden_mesh_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", os.path.normpath(path), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
