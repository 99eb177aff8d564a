#!/usr/bin/env python3
"""Randomised parity campaign: for each seed a random scene (sphere cloud with
radii over three decades, triangle soup of 0..300 triangles so that both the
looped and the hierarchy path occur, random materials, 1..17 lights, a random
camera), one frame at a random depth through the C ABI against the CPU oracle
(ids, bytes, ray count), and a batch of random rays through check_intersection.
usage: fuzz_parity.py [first_seed] [n_seeds]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np  # noqa: E402
import rtamd  # noqa: E402
from oracle_py import Oracle  # noqa: E402


def scene(seed):
    rng = np.random.RandomState(seed)
    ns = int(rng.choice([0, 1, 3, 17, 64, 200, 700]))
    nt = int(rng.choice([0, 1, 2, 15, 16, 17, 60, 300]))
    spread = float(rng.choice([3, 15, 60]))
    c = np.stack([rng.uniform(-spread, spread, ns), rng.uniform(-spread * 0.6, spread * 0.6, ns),
                  rng.uniform(-20 - 2 * spread, -4, ns)], -1).astype(np.float32)
    r = (10 ** rng.uniform(-2, 0.7, ns)).astype(np.float32)
    if ns > 4:
        c[2], r[2] = c[1], r[1]                      # duplicate: lowest index wins
        c[3] = c[0]; r[3] = r[0] * np.float32(1.5)   # concentric
    col = rng.randint(0, 256, (ns, 3)).astype(np.uint8)
    base = np.stack([rng.uniform(-spread, spread, nt), rng.uniform(-spread * 0.6, spread * 0.6, nt),
                     rng.uniform(-20 - 2 * spread, -5, nt)], -1)
    size = 10 ** rng.uniform(-1, 1.2, (nt, 1, 1))
    tri = (base[:, None, :] + rng.uniform(-1, 1, (nt, 3, 3)) * size).astype(np.float32).reshape(nt, 9)
    if nt > 3:
        tri[1] = 0                                   # degenerate
        tri[2, 3:6] = tri[2, 0:3]                    # two equal corners
    tcol = rng.randint(0, 256, (nt, 3)).astype(np.uint8)
    if rng.rand() < 0.5:
        tri = np.concatenate([rtamd.WALL_TRIS, tri]); tcol = np.concatenate([rtamd.WALL_COLOR, tcol])
    nl = int(rng.choice([1, 1, 2, 3, 5, 9, 17]))   # up to five groups of four lights
    lights = np.stack([rng.uniform(-20, 20, nl), rng.uniform(-10, 20, nl), rng.uniform(-30, 10, nl)], -1).astype(np.float32)
    sc = rtamd.Scene(tri, tcol, c, r, col, lights)
    if rng.rand() < 0.8:
        def mats(n):
            m = np.zeros((n, 4), np.float32)
            k = rng.randint(0, 4, n)
            m[k == 1, 0] = rng.uniform(0.2, 0.9, (k == 1).sum())
            m[k == 2, 0] = 0.1; m[k == 2, 1] = rng.uniform(0.4, 0.85, (k == 2).sum()); m[k == 2, 2] = rng.uniform(1.05, 2.2, (k == 2).sum())
            m[k == 3, 0] = 0.5; m[k == 3, 1] = 0.5; m[k == 3, 2] = 1.0
            return m
        sc.mat_sphere = mats(sc.s_size)
        sc.mat_tri = mats(sc.t_size)
    cam = rtamd.default_camera()
    if rng.rand() < 0.5:
        cam.fov_deg = int(rng.choice([30, 60, 90, 120]))
        cam.origin[0], cam.origin[1], cam.origin[2] = [float(x) for x in rng.uniform(-3, 3, 3)]
    return sc, cam, rng


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    o = Oracle()
    bad = 0
    for seed in range(first, first + count):
        sc, cam, rng = scene(seed)
        w, h = int(rng.choice([64, 100, 192])), int(rng.choice([40, 72, 108]))
        if os.environ.get("FUZZ_BIG"):   # frames large enough for every block of the persistent grids
            w, h = 8 * w, 8 * h
        depth = int(rng.choice([1, 2, 4, 9]))
        fb, ids, n = rtamd.render(sc, w, h, max_depth=depth, cam=cam)
        fo, io, no = o.render(sc, w, h, max_depth=depth, cam=cam)
        ok = np.array_equal(ids, io) and np.array_equal(fb, fo) and n == no
        m = 3000
        ro = rng.normal(scale=rng.choice([2.0, 30.0, 300.0]), size=(m, 3)).astype(np.float32)
        rd = rng.normal(size=(m, 3)).astype(np.float32)
        rd[::7, rng.randint(0, 3)] = 0
        with rtamd.DeviceScene(sc) as ds:
            k1, i1, p1 = ds.check_intersection(ro, rd)
        k0, i0, p0 = o.check_intersection(sc, ro, rd)
        hit = k0 != 0
        ok2 = np.array_equal(k0, k1) and np.array_equal(i0[hit], i1[hit]) and \
            np.array_equal(p0.view(np.uint32)[hit], p1.view(np.uint32)[hit])
        print(f"seed {seed}: {sc.s_size} spheres {sc.t_size} triangles {sc.l_size} lights {w}x{h} depth {depth} "
              f"rays {n}: frame {'ok' if ok else 'MISMATCH'}, rays {'ok' if ok2 else 'MISMATCH'} ({int(hit.sum())} hits)", flush=True)
        bad += (not ok) + (not ok2)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
