#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz from the REAL reference.

Run in the build container only (needs oracle/_ref/libref.so, which
oracle/Makefile compiles from the sources under /root/reference):

    make -C oracle && python oracle/gen_golden.py

Everything written is data: inputs we made up (seeded) and the outputs the
reference produced for them.  No reference source text is stored.

Contents (all arrays; numpy.load(allow_pickle=False) reads them):
  <scene>_fbcrc / <scene>_idcrc   per-row CRC32 of the 2560x1440 framebuffer /
                                  hit-id buffer rendered by the reference
  <scene>_px / _pxfb / _pxid      a few hundred pixel coordinates (i,j) on
                                  silhouette and shadow edges, with their bytes
  kat_*                           known-answer vectors of the math functions
  rays_*                          check_intersection on seeded random rays
  grid_*                          init_spheres outputs
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from oracle_py import Ref, rtamd  # noqa: E402
import golden_scenes  # noqa: E402


def row_crc(a):
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(a.shape[0])], np.uint32)


def edge_pixels(ids, fb, n, seed):
    """Pixels whose right or lower neighbour sees a different primitive or
    whose colour changes sharply (shadow edges), sampled with a fixed seed."""
    d_id = (ids[:-1, :-1] != ids[1:, :-1]) | (ids[:-1, :-1] != ids[:-1, 1:])
    g = fb[..., :3].astype(np.int32).sum(-1)
    d_fb = (np.abs(g[:-1, :-1] - g[1:, :-1]) > 24) | (np.abs(g[:-1, :-1] - g[:-1, 1:]) > 24)
    cand = np.argwhere(d_id | d_fb)
    rng = np.random.RandomState(seed)
    pick = cand[rng.choice(len(cand), size=min(n, len(cand)), replace=False)]
    return pick.astype(np.int32)


def main():
    ref = Ref()
    out = {}
    for name, sc in golden_scenes.full_frame_scenes().items():
        fb = ref.render(sc)
        ids = ref.hit_ids(sc)
        out[f"{name}_fbcrc"] = row_crc(fb)
        out[f"{name}_idcrc"] = row_crc(ids)
        px = edge_pixels(ids, fb, 400, seed=7)
        out[f"{name}_px"] = px
        out[f"{name}_pxfb"] = fb[px[:, 0], px[:, 1]]
        out[f"{name}_pxid"] = ids[px[:, 0], px[:, 1]]
        print(name, "rows", fb.shape[0], "edge px", len(px),
              "hit frac %.3f" % (ids >= 0).mean())

    k = golden_scenes.kat_inputs()
    hit, P = ref.ray_sphere(k["sph_orig"], k["sph_dir"], k["sph_centre"], k["sph_radius"])
    out["kat_sph_hit"], out["kat_sph_P"] = hit, P.view(np.uint32)
    hit, P = ref.ray_triangle(k["tri_orig"], k["tri_dir"], k["tri_tri"])
    out["kat_tri_hit"], out["kat_tri_P"] = hit, P.view(np.uint32)
    out["kat_normalize"] = ref.normalize(k["vec"]).view(np.uint32)
    out["kat_length"] = ref.length(k["vec"]).view(np.uint32)
    out["kat_normal"] = ref.normal(k["tri_tri"]).view(np.uint32)
    out["kat_reflect"] = ref.reflect(k["vec"], k["vec2"]).view(np.uint32)
    print("kat sphere hits", int(out["kat_sph_hit"].sum()), "/", len(hit),
          "tri hits", int(out["kat_tri_hit"].sum()))

    sc = golden_scenes.mixed_scene()
    o, d = golden_scenes.random_rays(20000, seed=11)
    kind, index, P = ref.check_intersection(sc, o, d)
    out["rays_kind"], out["rays_index"] = kind.astype(np.int8), index.astype(np.int16)
    Pm = P.view(np.uint32).copy()
    Pm[kind == 0] = 0  # P is scratch on a miss
    out["rays_P"] = Pm
    print("random rays: kinds", np.bincount(kind, minlength=3))

    for row, col in [(10, 10), (3, 7), (32, 32)]:
        s, r, c = ref.init_spheres(row, col)
        out[f"grid_{row}x{col}_s"] = s
        out[f"grid_{row}x{col}_r"] = r
        out[f"grid_{row}x{col}_c"] = c

    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
