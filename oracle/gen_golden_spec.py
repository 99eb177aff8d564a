#!/usr/bin/env python3
"""Generate tests/golden/golden_spec_v1.npz: the specular term of
renderer.cpp:106-110, whose libm pow(lang, 20) the device replaces by five double
multiplies.

Run in the build container only (needs oracle/_ref/libref.so):

    make -C oracle && python oracle/gen_golden_spec.py

Two kinds of data, nothing else:
  spec_strength / spec_byte   for the (lang, colour) pairs of
                              golden_scenes.specular_inputs(): the float and the
                              byte that the two statements give when compiled
                              beside the reference (ref_harness.cpp,
                              ref_specular_batch: same compiler, flags and libm)
  highlight_fbcrc / _idcrc    per-row CRC32 of the frame and the hit ids the
                              reference's own render() produces for
                              golden_scenes.highlight_scene(), a frame that is
                              mostly specular highlight
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
    ref = Ref()
    out = {}
    lang, col = golden_scenes.specular_inputs()
    strength, spec = ref.specular(lang, col)
    out["spec_strength"] = strength.view(np.uint32)
    out["spec_byte"] = spec
    print(len(lang), "specular pairs; bytes used:", len(np.unique(spec)))
    sc = golden_scenes.highlight_scene()
    fb = ref.render(sc)
    ids = ref.hit_ids(sc)
    out["highlight_fbcrc"] = row_crc(fb)
    out["highlight_idcrc"] = row_crc(ids)
    print("highlight frame: distinct colours", len(np.unique(fb.reshape(-1, 4), axis=0)),
          "saturated channel bytes %.3f" % (fb[..., :3] == 255).mean())
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "golden_spec_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
