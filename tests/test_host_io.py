"""Host-side helpers (include/rt_host.h): scene-file reader, grid generator and
PPM writer, pinned to the reference's own (main.cpp:215-307, 358-395, 155-173)."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

import golden_scenes
import rtamd

HOSTLIB = os.path.join(rtamd.PKG_DIR, "host", "librt_host.so")
DRIVER = os.path.join(rtamd.PKG_DIR, "host", "raytracer_amd")
# md5 of the image.ppm the unmodified reference binary writes for
# `./raytracer -n -f scene.txt` (2 wall triangles, 10x10 grid, one light)
REFERENCE_PPM_MD5 = "479b149776b8421ea7fde6f0fca36d54"


class HostScene(C.Structure):
    _fields_ = [("tris", C.POINTER(C.c_float)), ("color_tri", C.POINTER(C.c_uint8)),
                ("t_size", C.c_int32), ("spheres", C.POINTER(C.c_float)),
                ("radius", C.POINTER(C.c_float)), ("color_sphere", C.POINTER(C.c_uint8)),
                ("s_size", C.c_int32), ("lights", C.POINTER(C.c_float)), ("l_size", C.c_int32)]


@pytest.fixture(scope="module")
def host():
    L = C.CDLL(HOSTLIB)
    L.rt_host_load_scene.argtypes = [C.c_char_p, C.c_int, C.POINTER(HostScene), C.POINTER(C.c_int)]
    L.rt_host_free_scene.argtypes = [C.POINTER(HostScene)]
    L.rt_host_sphere_grid.argtypes = [C.POINTER(HostScene), C.c_int, C.c_int]
    L.rt_host_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    return L


def load(host, path, mode=0):
    hs, bad = HostScene(), C.c_int(0)
    rc = host.rt_host_load_scene(str(path).encode(), mode, C.byref(hs), C.byref(bad))
    if rc:
        return rc, bad.value
    a = lambda p, n, t: np.ctypeslib.as_array(p, (max(n, 1),))[:n].astype(t).copy()
    sc = rtamd.Scene(a(hs.tris, 9 * hs.t_size, np.float32), a(hs.color_tri, 3 * hs.t_size, np.uint8),
                     a(hs.spheres, 3 * hs.s_size, np.float32), a(hs.radius, hs.s_size, np.float32),
                     a(hs.color_sphere, 3 * hs.s_size, np.uint8),
                     a(hs.lights, 3 * hs.l_size, np.float32))
    host.rt_host_free_scene(C.byref(hs))
    return 0, sc


SHIPPED_LIKE = """// sphere:      s radius point color
// triangle:    t point point point color
// lights       l point


    s 1.0 -18.0 -4.0 -50.0  0 255 0
    s 2.5  -16.0 4.25 -50.0  10 20 30

t  1000.0 -100.0 -100.0   1000.0 100.0 -100.0  -1000.0  100.0 -100.0   255 125 125
t  1000.0 -100.0 -100.0  -1000.0 100.0 -100.0  -1000.0 -100.0 -100.0  255 125 125


// Pyramids
//t -6 -1 -5  -4  1 -5  -4 -1 -5  255 255 255

l 0.0 5.0 1.0
//l 0.0 3.0 3.0"""


def test_loader_defined_behaviour(host, tmp_path):
    p = tmp_path / "scene.txt"
    p.write_text(SHIPPED_LIKE)
    rc, sc = load(host, p)
    assert rc == 0
    assert np.array_equal(sc.tris, rtamd.WALL_TRIS) and np.array_equal(sc.color_tri, rtamd.WALL_COLOR)
    assert np.array_equal(sc.lights, rtamd.LIGHT)
    assert np.array_equal(sc.radius, np.array([1.0, 2.5], np.float32))          # radius comes first
    assert np.array_equal(sc.spheres, np.array([[-18, -4, -50], [-16, 4.25, -50]], np.float32))
    assert np.array_equal(sc.color_sphere, np.array([[0, 255, 0], [10, 20, 30]], np.uint8))


def test_loader_phantom_mode_counts(host, tmp_path):
    """Blank lines after an `s`/`t` line append all-zero primitives of that type,
    which is what the reference's loader usually does (main.cpp:228-264)."""
    p = tmp_path / "scene.txt"
    p.write_text(SHIPPED_LIKE)
    rc, sc = load(host, p, mode=1)
    assert rc == 0
    assert sc.s_size == 2 + 1 and sc.t_size == 2 + 3 and sc.l_size == 1
    assert (sc.tris[2:] == 0).all() and (sc.spheres[2:] == 0).all()
    assert np.array_equal(sc.tris[:2], rtamd.WALL_TRIS)


def test_loader_errors(host, tmp_path):
    assert load(host, tmp_path / "missing.txt")[0] == -1
    p = tmp_path / "bad.txt"
    p.write_text("l 0 5 1\nt 1 2 3\n")
    rc, line = load(host, p)
    assert rc == -2 and line == 2


def test_loader_matches_reference_on_a_file_without_blank_lines(host, ref, tmp_path):
    rng = np.random.RandomState(3)
    lines = ["// generated"]
    for _ in range(25):
        v = rng.uniform(-50, 50, 4)
        c = rng.randint(0, 256, 3)
        lines.append("  s %.6g %.6g %.6g %.6g  %d %d %d" % (abs(v[0]), v[1], v[2], v[3], *c))
    for _ in range(11):
        v = rng.uniform(-200, 200, 9)
        c = rng.randint(0, 256, 3)
        lines.append("t " + " ".join("%.7g" % x for x in v) + "  %d %d %d" % tuple(c))
        lines.append("//t commented out")
    lines += ["l 0.0 5.0 1.0", "l -3.5 2 7"]
    p = tmp_path / "gen.txt"
    p.write_text("\n".join(lines))          # no trailing newline, like the shipped file
    rc, mine = load(host, p)
    theirs = ref.read_scene(str(p))
    assert rc == 0
    for f in ("tris", "color_tri", "spheres", "radius", "color_sphere", "lights"):
        assert np.array_equal(getattr(mine, f), getattr(theirs, f)), f


def test_loader_phantom_after_a_trailing_newline(host, ref, tmp_path):
    """The reference loops `while (!file.eof())` (main.cpp:228): a file that ends in
    a newline makes it read one more, empty, line and append one more primitive of
    the last line's type (here a light).  Phantom mode does the same; the counts are
    checked against the reference's own reader."""
    p = tmp_path / "nl.txt"
    p.write_text("s 1.0 0 0 -5  1 2 3\nl 0.0 5.0 1.0\n")
    rc, mine = load(host, p, mode=1)
    assert rc == 0
    assert (mine.s_size, mine.t_size, mine.l_size) == (1, 0, 2)
    assert np.array_equal(mine.lights[0], np.array([0, 5, 1], np.float32))
    assert (mine.lights[1] == 0).all()
    theirs = ref.read_scene(str(p))
    assert (theirs.s_size, theirs.t_size, theirs.l_size) == (mine.s_size, mine.t_size, mine.l_size)
    assert np.array_equal(theirs.lights[0], mine.lights[0])
    # without the newline neither appends anything
    p.write_text("s 1.0 0 0 -5  1 2 3\nl 0.0 5.0 1.0")
    rc, mine = load(host, p, mode=1)
    assert rc == 0 and (mine.s_size, mine.t_size, mine.l_size) == (1, 0, 1)
    # the defined behaviour ignores the empty last line
    p.write_text("s 1.0 0 0 -5  1 2 3\nl 0.0 5.0 1.0\n")
    rc, mine = load(host, p, mode=0)
    assert rc == 0 and (mine.s_size, mine.t_size, mine.l_size) == (1, 0, 1)


def test_grid_generator(host, golden):
    for row, col in [(10, 10), (3, 7), (32, 32)]:
        hs = HostScene()
        assert host.rt_host_sphere_grid(C.byref(hs), row, col) == 0
        n = row * col
        assert hs.s_size == n
        assert np.array_equal(np.ctypeslib.as_array(hs.spheres, (3 * n,)).reshape(n, 3),
                              golden[f"grid_{row}x{col}_s"])
        assert np.array_equal(np.ctypeslib.as_array(hs.radius, (n,)), golden[f"grid_{row}x{col}_r"])
        assert np.array_equal(np.ctypeslib.as_array(hs.color_sphere, (3 * n,)).reshape(n, 3),
                              golden[f"grid_{row}x{col}_c"])
        host.rt_host_free_scene(C.byref(hs))


def test_ppm_writer_format(host, tmp_path):
    fb = np.arange(2 * 3 * 4, dtype=np.uint8).reshape(2, 3, 4)
    p = tmp_path / "x.ppm"
    assert host.rt_host_write_ppm(str(p).encode(), fb.ctypes.data, 3, 2) == 0
    assert p.read_text() == "P3\n3 2 255\n2 1 0 6 5 4 10 9 8 \n14 13 12 18 17 16 22 21 20 \n"


def test_ppm_of_the_shipped_scene_has_the_reference_md5(host, oracle, tmp_path):
    """Oracle frame of the shipped configuration through the PPM writer equals,
    byte for byte, the image.ppm of the unmodified reference binary."""
    sc = golden_scenes.full_frame_scenes()["shipped"]
    fb, _, _ = oracle.render(sc, 2560, 1440, want_ids=False)
    p = tmp_path / "image.ppm"
    assert host.rt_host_write_ppm(str(p).encode(), fb.ctypes.data, 2560, 1440) == 0
    assert hashlib.md5(p.read_bytes()).hexdigest() == REFERENCE_PPM_MD5


@pytest.mark.gpu
def test_headless_driver_end_to_end(tmp_path):
    """raytracer_amd -n -f scene.txt (the reference's command line) on the GPU
    writes the same image.ppm as the reference binary."""
    scene = tmp_path / "scene.txt"
    scene.write_text(SHIPPED_LIKE)
    out = tmp_path / "image.ppm"
    for extra in ([], ["-phantoms"]):
        subprocess.check_call([DRIVER, "-n", "-f", str(scene), "-o", str(out)] + extra)
        assert hashlib.md5(out.read_bytes()).hexdigest() == REFERENCE_PPM_MD5
    subprocess.check_call([DRIVER, "-h"])
