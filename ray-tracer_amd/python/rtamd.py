"""ctypes binding of the C ABI in include/rt_amd.h, plus scene containers.

This is plumbing for tests and bench.py: it loads ``librt_amd.so`` (the HIP
library) and nothing else.  There is no CPU fallback: if the library is missing
or a call fails, the error is raised.

Scene arrays use the reference's flat layouts (main.cpp:270-305, 377-392).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
REPO_DIR = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "csrc", "librt_amd.so")

RT_OK = 0
RT_ERR_BAD_ARG = -1
RT_ERR_NO_DEVICE = -2
RT_ERR_HIP = -3
RT_ERR_CAPACITY = -4
RT_ERR_UNSUPPORTED = -5


class Material(C.Structure):
    _fields_ = [("reflectivity", C.c_float), ("transparency", C.c_float),
                ("ior", C.c_float), ("reserved", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("tris", C.c_void_p), ("color_tri", C.c_void_p), ("t_size", C.c_int32),
        ("spheres", C.c_void_p), ("radius", C.c_void_p),
        ("color_sphere", C.c_void_p), ("s_size", C.c_int32),
        ("lights", C.c_void_p), ("l_size", C.c_int32),
        ("mat_tri", C.c_void_p), ("mat_sphere", C.c_void_p),
    ]


class Camera(C.Structure):
    _fields_ = [("fov_deg", C.c_int32), ("origin", C.c_float * 3),
                ("plane_z", C.c_float)]


class Frame(C.Structure):
    _fields_ = [
        ("dev_fb", C.c_void_p), ("dev_ids", C.c_void_p),
        ("w", C.c_int32), ("h", C.c_int32), ("max_depth", C.c_int32),
        ("band_rows", C.c_int32), ("band_index", C.c_int32),
        ("band_count", C.c_int32), ("flags", C.c_int32), ("stream", C.c_void_p),
    ]


RT_FRAME_UNLIT = 1


def default_camera(fov_deg: int = 90) -> Camera:
    """Eye (0,0,1), plane z=-1, fov 90: renderer.cpp:38-39, main.cpp:88."""
    cam = Camera()
    cam.fov_deg = fov_deg
    cam.origin[0], cam.origin[1], cam.origin[2] = 0.0, 0.0, 1.0
    cam.plane_z = -1.0
    return cam


@dataclass
class Scene:
    """Host-side scene in the reference's flat-array form."""
    tris: np.ndarray            # (t, 9) float32
    color_tri: np.ndarray       # (t, 3) uint8
    spheres: np.ndarray         # (s, 3) float32
    radius: np.ndarray          # (s,)   float32
    color_sphere: np.ndarray    # (s, 3) uint8
    lights: np.ndarray          # (l, 3) float32
    mat_tri: Optional[np.ndarray] = None     # (t, 4) float32 or None
    mat_sphere: Optional[np.ndarray] = None  # (s, 4) float32 or None
    _desc: Optional[SceneDesc] = field(default=None, repr=False)

    def __post_init__(self):
        self.tris = np.ascontiguousarray(self.tris, np.float32).reshape(-1, 9)
        self.color_tri = np.ascontiguousarray(self.color_tri, np.uint8).reshape(-1, 3)
        self.spheres = np.ascontiguousarray(self.spheres, np.float32).reshape(-1, 3)
        self.radius = np.ascontiguousarray(self.radius, np.float32).reshape(-1)
        self.color_sphere = np.ascontiguousarray(self.color_sphere, np.uint8).reshape(-1, 3)
        self.lights = np.ascontiguousarray(self.lights, np.float32).reshape(-1, 3)
        if self.mat_tri is not None:
            self.mat_tri = np.ascontiguousarray(self.mat_tri, np.float32).reshape(-1, 4)
            assert len(self.mat_tri) == len(self.tris)
        if self.mat_sphere is not None:
            self.mat_sphere = np.ascontiguousarray(self.mat_sphere, np.float32).reshape(-1, 4)
            assert len(self.mat_sphere) == len(self.spheres)
        assert len(self.color_tri) == len(self.tris)
        assert len(self.radius) == len(self.spheres) == len(self.color_sphere)

    @property
    def t_size(self) -> int:
        return len(self.tris)

    @property
    def s_size(self) -> int:
        return len(self.spheres)

    @property
    def l_size(self) -> int:
        return len(self.lights)

    def desc(self) -> SceneDesc:
        """A scene_desc whose pointers borrow this object's arrays."""
        d = SceneDesc()
        ptr = lambda a: a.ctypes.data if a is not None and a.size else None
        d.tris, d.color_tri, d.t_size = ptr(self.tris), ptr(self.color_tri), self.t_size
        d.spheres, d.radius = ptr(self.spheres), ptr(self.radius)
        d.color_sphere, d.s_size = ptr(self.color_sphere), self.s_size
        d.lights, d.l_size = ptr(self.lights), self.l_size
        d.mat_tri, d.mat_sphere = ptr(self.mat_tri), ptr(self.mat_sphere)
        self._desc = d
        return d


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt_amd error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load librt_amd.so once; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime under the same soname.  A process may
    # hold only one: if torch is going to be used (device buffers, RCCL) it has
    # to be the one that loads it, so give it the chance before dlopen.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is not built (run python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(LIB_PATH)
    L.rt_last_error.restype = C.c_char_p
    L.rt_version.restype = C.c_char_p
    L.rt_camera_default.argtypes = [C.POINTER(Camera)]
    L.rt_camera_default.restype = None
    L.gpu_render.argtypes = [C.POINTER(SceneDesc), C.POINTER(Camera), C.c_void_p,
                             C.c_int, C.c_int, C.c_int]
    L.gpu_render_ids.argtypes = [C.POINTER(SceneDesc), C.POINTER(Camera), C.c_void_p,
                                 C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int,
                                 C.c_int]
    L.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
    L.rt_scene_destroy.argtypes = [C.c_void_p]
    L.rt_scene_destroy.restype = None
    L.rt_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame),
                                   C.c_void_p]
    L.rt_render_device_selfcheck.argtypes = [C.c_void_p, C.POINTER(Camera),
                                             C.POINTER(Frame), C.c_void_p, C.c_void_p]
    L.rt_set_screen.argtypes = [C.c_int]
    L.rt_set_screen.restype = None
    L.rt_scene_set_screen.argtypes = [C.c_void_p, C.c_int]
    L.rt_scene_set_side_stream.argtypes = [C.c_void_p, C.c_int]
    L.rt_scene_set_one_launch.argtypes = [C.c_void_p, C.c_int]
    L.rt_scene_describe.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.rt_verify_sphere_lists.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.c_int,
                                         C.POINTER(C.c_uint64)]
    L.rt_scene_time_kernels.argtypes = [C.c_void_p, C.c_int]
    L.rt_scene_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.rt_scene_set_rays_per_pixel.argtypes = [C.c_void_p, C.c_int]
    L.rt_scene_queue_overflowed.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.gpu_render_unlit.argtypes = [C.POINTER(SceneDesc), C.POINTER(Camera), C.c_void_p,
                                   C.c_int, C.c_int]
    L.gpu_render_multi.argtypes = [C.POINTER(SceneDesc), C.POINTER(Camera), C.c_void_p,
                                   C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int]
    L.gpu_check_intersection_arrays.argtypes = [C.POINTER(SceneDesc), C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_band_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_int]
    L.rt_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame),
                                           C.c_void_p, C.c_void_p]
    L.rt_noalloc_scratch_needed.argtypes = [C.c_void_p, C.POINTER(Frame),
                                            C.POINTER(C.c_uint64)]
    L.rt_cache_clear.argtypes = []
    L.rt_cache_clear.restype = None
    L.rt_cache_builds.argtypes = []
    L.rt_cache_builds.restype = C.c_uint64
    L.gpu_check_intersection.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_cache_trust_pointers.argtypes = [C.c_int]
    L.rt_cache_trust_pointers.restype = None
    L.rt_alloc_host_buffer.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.rt_free_host_buffer.argtypes = [C.c_void_p]
    L.rt_free_host_buffer.restype = None
    L.rt_scene_set_arena_limit.argtypes = [C.c_void_p, C.c_uint64]
    L.gpu_specular_terms.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def _check(code: int):
    if code != RT_OK:
        raise RtError(code, lib().rt_last_error().decode())


SCREEN_MODES = {"off": 0, "auto": 1, "beam": 2, "bvh": 3}


def _mode(mode) -> int:
    if isinstance(mode, bool):
        mode = "auto" if mode else "off"
    return SCREEN_MODES[mode]


def set_screen(mode):
    """rt_set_screen: the mode of scenes created from now on (rtamd.render creates
    its scene inside the call): "off" (plain every-ray-every-sphere loop), "auto"
    (default), "beam" or "bvh"; True/False mean "auto"/"off"."""
    lib().rt_set_screen(_mode(mode))


def specular_terms(lang: np.ndarray, color: np.ndarray):
    """gpu_specular_terms: (strength float32, specular uint8) of renderer.cpp:106-110 as
    the device computes them."""
    lang = np.ascontiguousarray(lang, np.float32).reshape(-1)
    color = np.ascontiguousarray(color, np.uint8).reshape(-1)
    assert len(lang) == len(color)
    strength = np.empty(len(lang), np.float32)
    spec = np.empty(len(lang), np.uint8)
    _check(lib().gpu_specular_terms(len(lang), lang.ctypes.data, color.ctypes.data,
                                    strength.ctypes.data, spec.ctypes.data))
    return strength, spec


class HostBuffer:
    """rt_alloc_host_buffer / rt_free_host_buffer: page-locked host memory as a
    numpy uint8 array (``.array``) that stays valid until close()."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        _check(lib().rt_alloc_host_buffer(nbytes, C.byref(self.ptr)))
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self.ptr.value))

    def close(self):
        if self.ptr:
            self.array = None
            lib().rt_free_host_buffer(self.ptr)
            self.ptr = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def render_unlit(scene: "Scene", w: int, h: int, cam: Optional[Camera] = None) -> np.ndarray:
    """gpu_render_unlit: the reference built with -DUNLIT=1."""
    cam = cam or default_camera()
    fb = np.empty((h, w, 4), np.uint8)
    d = scene.desc()
    _check(lib().gpu_render_unlit(C.byref(d), C.byref(cam), fb.ctypes.data, w, h))
    return fb


def render_multi(scene: "Scene", w: int, h: int, max_depth: int = 1, n_devices: int = 1,
                 band_rows: int = 8, cam: Optional[Camera] = None, want_ids: bool = True):
    """gpu_render_multi: one frame on n_devices GPUs of this process."""
    cam = cam or default_camera()
    fb = np.empty((h, w, 4), np.uint8)
    ids = np.empty((h, w), np.int32) if want_ids else None
    n = C.c_uint64(0)
    d = scene.desc()
    _check(lib().gpu_render_multi(C.byref(d), C.byref(cam), fb.ctypes.data,
                                  ids.ctypes.data if want_ids else None, C.byref(n),
                                  w, h, max_depth, n_devices, band_rows))
    return fb, ids, int(n.value)


def band_plan(h: int, band_rows: int, band_index: int, band_count: int):
    """rt_band_plan: list of (row_begin, row_end) of the bands a share owns."""
    n = lib().rt_band_plan(h, band_rows, band_index, band_count, None, None, 0)
    if n < 0:
        raise RtError(n, lib().rt_last_error().decode())
    b = np.zeros(max(n, 1), np.int32)
    e = np.zeros(max(n, 1), np.int32)
    n2 = lib().rt_band_plan(h, band_rows, band_index, band_count, b.ctypes.data,
                            e.ctypes.data, n)
    assert n2 == n
    return [(int(b[k]), int(e[k])) for k in range(n)]


def render(scene: Scene, w: int, h: int, max_depth: int = 1,
           cam: Optional[Camera] = None, want_ids: bool = True):
    """gpu_render_ids: returns (fb[h,w,4] u8, ids[h,w] i32 or None, n_rays)."""
    cam = cam or default_camera()
    fb = np.empty((h, w, 4), np.uint8)
    ids = np.empty((h, w), np.int32) if want_ids else None
    n = C.c_uint64(0)
    d = scene.desc()
    _check(lib().gpu_render_ids(C.byref(d), C.byref(cam), fb.ctypes.data,
                                ids.ctypes.data if want_ids else None, C.byref(n),
                                w, h, max_depth))
    return fb, ids, int(n.value)


def render_plain(scene: Scene, w: int, h: int, max_depth: int = 1,
                 cam: Optional[Camera] = None) -> np.ndarray:
    """gpu_render: the one-call drop-in entry point."""
    cam = cam or default_camera()
    fb = np.empty((h, w, 4), np.uint8)
    d = scene.desc()
    _check(lib().gpu_render(C.byref(d), C.byref(cam), fb.ctypes.data, w, h, max_depth))
    return fb


class DeviceScene:
    """rt_scene_create / rt_scene_destroy as a context manager."""

    def __init__(self, scene: Scene):
        self.scene = scene
        self.handle = C.c_void_p()
        d = scene.desc()
        _check(lib().rt_scene_create(C.byref(d), C.byref(self.handle)))

    def close(self):
        if self.handle:
            lib().rt_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_screen(self, mode):
        """rt_scene_set_screen: the work-avoidance mode of this scene's later frames."""
        _check(lib().rt_scene_set_screen(self.handle, _mode(mode)))

    def describe(self):
        """rt_scene_describe as a dict."""
        info = (C.c_int64 * 8)()
        _check(lib().rt_scene_describe(self.handle, info))
        return {"sphere_nodes": info[0], "sphere_levels": info[1], "lds_stacks": bool(info[2]),
                "one_launch": bool(info[3]), "triangle_nodes": info[4],
                "lights_with_lists": info[5], "screen_cells_with_list": info[6],
                "light_cells_without_list": info[7]}

    def verify_sphere_lists(self, w: int, h: int, cam: Optional[Camera] = None):
        """rt_verify_sphere_lists: (pixel rays served by a list, passing spheres missing from
        those lists, shadow rays served, passing spheres missing from theirs)."""
        cam = cam or default_camera()
        out = (C.c_uint64 * 4)()
        _check(lib().rt_verify_sphere_lists(self.handle, C.byref(cam), w, h, out))
        return tuple(int(x) for x in out)

    def time_kernels(self, on: bool = True):
        """rt_scene_time_kernels: record events around the stages of later frames."""
        _check(lib().rt_scene_time_kernels(self.handle, int(bool(on))))

    def kernel_times(self):
        """rt_scene_kernel_times: (primary pass, bounce levels, final sum) of the last
        frame in ms."""
        ms = (C.c_float * 3)()
        _check(lib().rt_scene_kernel_times(self.handle, ms))
        return float(ms[0]), float(ms[1]), float(ms[2])

    def set_one_launch(self, on: bool):
        """rt_scene_set_one_launch: all bounce levels in one launch (default) or a
        launch per stage and level."""
        _check(lib().rt_scene_set_one_launch(self.handle, int(bool(on))))

    def set_side_stream(self, on: bool):
        """rt_scene_set_side_stream."""
        _check(lib().rt_scene_set_side_stream(self.handle, int(bool(on))))

    def render_device(self, dev_fb: int, dev_ids: Optional[int], w: int, h: int,
                      max_depth: int = 1, cam: Optional[Camera] = None,
                      band_rows: int = 8, band_index: int = 0, band_count: int = 1,
                      stream: Optional[int] = None, dev_ray_count: Optional[int] = None,
                      dev_selfcheck_stats: Optional[int] = None, flags: int = 0):
        cam = cam or default_camera()
        f = Frame(dev_fb, dev_ids, w, h, max_depth, band_rows, band_index, band_count,
                  flags, stream)
        if dev_selfcheck_stats is not None:
            _check(lib().rt_render_device_selfcheck(self.handle, C.byref(cam), C.byref(f),
                                                    dev_ray_count, dev_selfcheck_stats))
        else:
            _check(lib().rt_render_device(self.handle, C.byref(cam), C.byref(f),
                                          dev_ray_count))

    def render_device_noalloc(self, dev_fb: int, dev_ids: Optional[int], w: int, h: int,
                              max_depth: int = 1, cam: Optional[Camera] = None,
                              stream: Optional[int] = None,
                              dev_ray_count: Optional[int] = None,
                              dev_overflow: Optional[int] = None, flags: int = 0):
        """rt_render_device_noalloc: the same frame from the library's fixed array."""
        cam = cam or default_camera()
        f = Frame(dev_fb, dev_ids, w, h, max_depth, 8, 0, 1, flags, stream)
        _check(lib().rt_render_device_noalloc(self.handle, C.byref(cam), C.byref(f),
                                              dev_ray_count, dev_overflow))

    def noalloc_scratch_needed(self, w: int, h: int, max_depth: int) -> int:
        f = Frame(None, None, w, h, max_depth, 8, 0, 1, 0, None)   # sizes only: no framebuffer
        n = C.c_uint64(0)
        _check(lib().rt_noalloc_scratch_needed(self.handle, C.byref(f), C.byref(n)))
        return int(n.value)

    def set_arena_limit(self, max_bytes: int):
        """rt_scene_set_arena_limit: cap on the device memory of the bounce machinery."""
        _check(lib().rt_scene_set_arena_limit(self.handle, max_bytes))

    def set_rays_per_pixel(self, n: int):
        """rt_scene_set_rays_per_pixel: bounce-queue capacity of later frames."""
        _check(lib().rt_scene_set_rays_per_pixel(self.handle, n))

    def queue_overflowed(self) -> bool:
        """rt_scene_queue_overflowed: did the last frame drop bounce rays?"""
        flag = C.c_int(0)
        _check(lib().rt_scene_queue_overflowed(self.handle, C.byref(flag)))
        return bool(flag.value)

    def check_intersection(self, orig: np.ndarray, dir: np.ndarray):
        orig = np.ascontiguousarray(orig, np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, np.float32).reshape(-1, 3)
        n = len(orig)
        kind = np.empty(n, np.int32)
        index = np.empty(n, np.int32)
        P = np.empty((n, 3), np.float32)
        _check(lib().gpu_check_intersection(self.handle, n, orig.ctypes.data,
                                            dir.ctypes.data, kind.ctypes.data,
                                            index.ctypes.data, P.ctypes.data))
        return kind, index, P


# ---------------------------------------------------------------------------
# Scene builders (host side).  These restate the reference's scene sources so
# tests and the bench can build the configurations without the reference.
# ---------------------------------------------------------------------------

WALL_TRIS = np.array([  # scene.txt:115-116
    [1000, -100, -100, 1000, 100, -100, -1000, 100, -100],
    [1000, -100, -100, -1000, 100, -100, -1000, -100, -100]], np.float32)
WALL_COLOR = np.array([[255, 125, 125], [255, 125, 125]], np.uint8)
LIGHT = np.array([[0.0, 5.0, 1.0]], np.float32)  # scene.txt:155


def sphere_grid(row: int, col: int):
    """init_spheres (main.cpp:358-395): radius 1, centre ((2j-col),(2i-row),-50)."""
    i, j = np.meshgrid(np.arange(row), np.arange(col), indexing="ij")
    spheres = np.stack([(j * 2 - col) * np.float32(1.0), (i * 2 - row) * np.float32(1.0),
                        np.full_like(i, -50)], axis=-1).astype(np.float32).reshape(-1, 3)
    radius = np.ones(row * col, np.float32)
    colors = np.tile(np.array([0, 255, 0], np.uint8), (row * col, 1))
    return spheres, radius, colors


def grid_scene(row: int, col: int, phantom_tris: int = 0) -> Scene:
    """Wall + light of scene.txt with a row x col grid (main.cpp:98-101).

    phantom_tris appends that many all-zero triangles, which is what the
    reference's loader produces for blank lines after a `t` line
    (main.cpp:228-264); they are degenerate and never hit."""
    s, r, c = sphere_grid(row, col)
    tris = np.concatenate([WALL_TRIS, np.zeros((phantom_tris, 9), np.float32)])
    ctri = np.concatenate([WALL_COLOR, np.zeros((phantom_tris, 3), np.uint8)])
    return Scene(tris, ctri, s, r, c, LIGHT.copy())


def seeded_materials(n: int, seed: int = 0x5EED) -> np.ndarray:
    """One third diffuse, one third mirror, one third glass, fixed seed."""
    rng = np.random.RandomState(seed)
    cls = rng.randint(0, 3, size=n)
    m = np.zeros((n, 4), np.float32)
    m[cls == 1] = [0.6, 0.0, 1.0, 0.0]
    m[cls == 2] = [0.1, 0.8, 1.5, 0.0]
    return m


def material_grid_scene(row: int, col: int, seed: int = 0x5EED) -> Scene:
    sc = grid_scene(row, col)
    sc.mat_sphere = seeded_materials(sc.s_size, seed)
    return sc


def mesh_scene(nx: int, ny: int, grid: int = 10, seed: int = 0x5EED, materials: bool = True,
               roughness: float = 0.0) -> Scene:
    """The triangle-heavy configuration at any size: a smooth height field over
    x in [-80,80], y in [-45,45], z in [-90,-60] (four seeded sine modes, the same
    surface at every tessellation) cut into nx*ny quads (2*nx*ny triangles;
    1000 x 500 gives the 1 M-triangle scene), plus a grid x grid block of the
    reference's unit spheres and its light.  roughness > 0 adds that much uniform
    noise to every vertex height (a deliberately hostile mesh for the triangle
    hierarchy: neighbouring normals then disagree).  Fixed seed."""
    rng = np.random.RandomState(seed)
    xs = np.linspace(-80, 80, nx + 1, dtype=np.float32)
    ys = np.linspace(-45, 45, ny + 1, dtype=np.float32)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    amp = (10.0, 3.0, 1.0, 0.3)
    freq = (0.025, 0.06, 0.1, 0.18)
    ph = rng.uniform(0, 2 * np.pi, (4, 2))
    Z = np.full(X.shape, -75.0)
    for k in range(4):
        Z += amp[k] * np.sin(freq[k] * X + ph[k, 0]) * np.cos(freq[k] * 1.3 * Y + ph[k, 1])
    noise = rng.uniform(-1.0, 1.0, X.shape)
    if roughness > 0:
        Z += roughness * noise
    Z = Z.astype(np.float32)
    P = np.stack([X, Y, Z], -1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.concatenate([a, b, c], -1).reshape(-1, 9)
    t2 = np.concatenate([a, c, d], -1).reshape(-1, 9)
    tris = np.concatenate([t1, t2]).astype(np.float32)
    tcol = rng.randint(40, 256, (len(tris), 3)).astype(np.uint8)
    s, r, col = sphere_grid(grid, grid)
    sc = Scene(tris, tcol, s, r, col, LIGHT.copy())
    if materials:
        sc.mat_sphere = seeded_materials(sc.s_size, seed)
        sc.mat_tri = seeded_materials(sc.t_size, seed + 1)
    return sc
