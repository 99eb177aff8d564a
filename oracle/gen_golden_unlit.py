#!/usr/bin/env python3
"""Generate tests/golden/golden_unlit_v1.npz from the REAL reference built with
-DUNLIT=1 (renderer.cpp:78,130-137; oracle/Makefile target _ref/libref_unlit.so).

Run in the build container only:

    make -C oracle && python oracle/gen_golden_unlit.py

Data only: per-row CRC32 of the 2560x1440 framebuffer the reference produced for
the seeded scenes of golden_scenes.full_frame_scenes().
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from oracle_py import Ref  # noqa: E402
import golden_scenes  # noqa: E402


def row_crc(a):
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(a.shape[0])], np.uint32)


def main():
    ref = Ref(unlit=True)
    out = {}
    for name, sc in golden_scenes.full_frame_scenes().items():
        fb = ref.render(sc)
        out[f"{name}_fbcrc"] = row_crc(fb)
        print(name, "distinct colours", len(np.unique(fb.reshape(-1, 4), axis=0)))
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "golden_unlit_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
