"""Seeded inputs shared by oracle/gen_golden.py and the tests.  Data only.

numpy.random.RandomState streams are stable across numpy versions, so the
inputs regenerate bit-identically wherever the tests run.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "ray-tracer_amd", "python"))
import rtamd  # noqa: E402


def mixed_scene(seed=3, n_spheres=40, n_tris=12, n_lights=3):
    """Random spheres and triangles in front of the camera plus the wall of
    scene.txt, several lights (exercises the per-light clamp), one duplicated
    sphere (lowest index must win the tie) and one degenerate triangle."""
    rng = np.random.RandomState(seed)
    c = np.stack([rng.uniform(-14, 14, n_spheres), rng.uniform(-8, 8, n_spheres),
                  rng.uniform(-40, -6, n_spheres)], -1).astype(np.float32)
    r = rng.uniform(0.4, 2.5, n_spheres).astype(np.float32)
    c[7], r[7] = c[3], r[3]  # exact duplicate: index 3 must be reported
    col = rng.randint(0, 256, (n_spheres, 3)).astype(np.uint8)
    base = np.stack([rng.uniform(-12, 12, n_tris), rng.uniform(-7, 7, n_tris),
                     rng.uniform(-35, -8, n_tris)], -1)
    tri = (base[:, None, :] + rng.uniform(-3, 3, (n_tris, 3, 3))).astype(np.float32)
    tri = tri.reshape(n_tris, 9)
    tri[5] = 0  # degenerate: N = 0, never hit (maths.cpp:30-31,73)
    tcol = rng.randint(0, 256, (n_tris, 3)).astype(np.uint8)
    tris = np.concatenate([rtamd.WALL_TRIS, tri])
    tcols = np.concatenate([rtamd.WALL_COLOR, tcol])
    lights = np.array([[0, 5, 1], [6, 9, -3], [-9, 2, 4]], np.float32)[:n_lights]
    return rtamd.Scene(tris, tcols, c, r, col, lights)


def full_frame_scenes():
    return {
        # what the reference binary renders by default: scene.txt through its
        # loader (2 wall triangles + 6 blank-line phantoms) and the 10x10 grid
        "shipped": rtamd.grid_scene(10, 10, phantom_tris=6),
        "g30": rtamd.grid_scene(30, 30),
        "mixed": mixed_scene(),
    }


def specular_inputs():
    """(lang, colour) pairs for the specular term of renderer.cpp:106-110:
    specular = (unsigned char)(colour * pow(lang, 20)).  For every colour c and
    every integer k <= c the float nearest to (k/c)^(1/20) and its three
    neighbours on either side (there colour * pow sits within a few ulp of the
    integer k, where a last-place difference in pow would change the byte), then
    values on a regular grid, the ends of the range, and seeded random values
    with random colours."""
    lang, col = [], []
    for c in range(1, 256):
        k = np.arange(1, c + 1, dtype=np.float64)
        centre = ((k / c) ** (1.0 / 20.0)).astype(np.float32)
        bits = centre.view(np.int32)
        for d in range(-3, 4):
            v = (bits + d).view(np.float32)
            lang.append(v)
            col.append(np.full(len(v), c, np.uint8))
    grid = np.linspace(0.0, 1.0, 4097, dtype=np.float32)
    lang.append(grid)
    col.append(np.full(len(grid), 255, np.uint8))
    ends = np.array([0.0, 1e-30, 1e-10, 0.5, 0.999999, 1.0, 1.0000001, 1.0000002], np.float32)
    lang.append(ends)
    col.append(np.full(len(ends), 200, np.uint8))
    rng = np.random.RandomState(20)
    r = rng.uniform(0.6, 1.0, 60000).astype(np.float32)   # below 0.6 every byte is 0
    lang.append(r)
    col.append(rng.randint(1, 256, len(r)).astype(np.uint8))
    return np.concatenate(lang).astype(np.float32), np.concatenate(col).astype(np.uint8)


def highlight_scene():
    """A frame whose image is mostly specular highlight: a few large spheres and
    the wall, every channel of every colour different, three lights close to the
    line of sight so that lang (renderer.cpp:104) is near 1 over wide areas and
    the byte (unsigned char)(colour * pow(lang, 20)) takes every value."""
    c = np.array([[0, 0, -30], [-16, 6, -34], [15, -5, -26], [6, 9, -22], [-7, -8, -20]],
                 np.float32)
    r = np.array([9, 7, 6, 4, 3.5], np.float32)
    col = np.array([[251, 193, 127], [89, 241, 167], [173, 101, 255], [233, 233, 61],
                    [47, 149, 211]], np.uint8)
    tcol = np.array([[255, 125, 125], [199, 251, 83]], np.uint8)
    lights = np.array([[0.5, 1.0, 1.5], [-3, -1, 2], [2, 4, 0]], np.float32)
    return rtamd.Scene(rtamd.WALL_TRIS.copy(), tcol, c, r, col, lights)


def random_rays(n, seed):
    rng = np.random.RandomState(seed)
    o = np.stack([rng.uniform(-10, 10, n), rng.uniform(-6, 6, n), rng.uniform(-30, 3, n)],
                 -1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    half = n // 2
    d[:half] /= np.linalg.norm(d[:half], axis=1, keepdims=True).astype(np.float32)
    d[half:] *= rng.uniform(0.1, 4.0, (n - half, 1)).astype(np.float32)  # unnormalised
    return o, d


def kat_inputs(seed=5, n=3000):
    rng = np.random.RandomState(seed)
    f = np.float32
    # ---- ray / sphere ----
    so = rng.uniform(-5, 5, (n, 3)).astype(f)
    sd = rng.normal(size=(n, 3)).astype(f)
    sc = so + (sd * rng.uniform(-6, 12, (n, 1))).astype(f) + rng.normal(scale=1.0, size=(n, 3)).astype(f)
    sr = rng.uniform(0.2, 3.0, n).astype(f)
    special = [
        # tangent: discr == 0 exactly (maths.cpp:135-136)
        ((0, 0, 0), (0, 0, -1), (1, 0, -5), 1.0),
        ((0, 0, 0), (0, 0, -2), (1, 0, -5), 1.0),
        # origin inside the sphere: t0 <= eps, t1 used (maths.cpp:147-148)
        ((0, 0, 0), (0, 0, -1), (0, 0, 0.5), 2.0),
        # sphere behind the ray
        ((0, 0, 0), (0, 0, -1), (0, 0, 5), 1.0),
        # origin on the surface (a shadow ray leaving its own sphere)
        ((0, 0, -4), (0, 0, 1), (0, 0, -5), 1.0),
        ((0, 0, -4), (0, 0, -1), (0, 0, -5), 1.0),
        # b in (0, 0.001]: the other branch of maths.cpp:139
        ((0, 0, 0), (1, 0, 0), (-0.0004, 0, 0), 1.0),
        ((0, 0, 0), (1, 0, 0), (-0.0005, 0, 0), 0.0002),
        # zero direction, zero radius
        ((0, 0, 0), (0, 0, 0), (0, 0, -5), 1.0),
        ((0, 0, 0), (0, 0, -1), (0, 0, -5), 0.0),
        # the reference camera ray straight at a grid sphere
        ((0, 0, 1), (0, 0, -1), (0, 0, -50), 1.0),
    ]
    k = len(special)
    so[:k] = [s[0] for s in special]
    sd[:k] = [s[1] for s in special]
    sc[:k] = [s[2] for s in special]
    sr[:k] = [s[3] for s in special]
    # ---- ray / triangle ----
    to = rng.uniform(-4, 4, (n, 3)).astype(f)
    td = rng.normal(size=(n, 3)).astype(f)
    centre = to + (td * rng.uniform(-3, 9, (n, 1))).astype(f)
    tt = (centre[:, None, :] + rng.uniform(-2.5, 2.5, (n, 3, 3))).astype(f).reshape(n, 9)
    tt[0] = 0                                    # degenerate
    tt[1] = [0, 0, -5, 1, 0, -5, 2, 0, -5]       # collinear
    to[2], td[2] = (0, 0, 0), (1, 0, 0)          # parallel to the plane z=-5
    tt[2] = [-1, -1, -5, 1, -1, -5, 0, 1, -5]
    to[3], td[3] = (0, 0, 1), (0, 0, -1)         # the wall, head on
    tt[3] = rtamd.WALL_TRIS[0]
    to[4], td[4] = (0, 0, 1), (0.3, 0.2, -1)
    tt[4] = rtamd.WALL_TRIS[1]
    # ---- vectors ----
    v = rng.normal(scale=3.0, size=(n, 3)).astype(f)
    v2 = rng.normal(size=(n, 3)).astype(f)
    v[0] = 0
    v[1] = [1e-30, 0, 0]
    v[2] = [1e25, 1e25, 0]      # squares overflow
    v[3] = [3, 4, 0]
    v[4] = [1e-23, 1e-23, 1e-23]  # squares underflow to denormals
    return dict(sph_orig=so, sph_dir=sd, sph_centre=sc, sph_radius=sr,
                tri_orig=to, tri_dir=td, tri_tri=tt, vec=v, vec2=v2)
