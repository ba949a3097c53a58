"""ctypes binding of the CPU oracle (oracle/libluma_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under raytracer_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libluma_oracle.so")
LITERAL_SO = os.path.join(ORACLE_DIR, "libluma_oracle_literal.so")   # -DLO_LITERAL build: sensitivity study only

FLAG_NOISE_OFF = 1
FLAG_ORIGIN_FROM_HIT = 2
FLAG_CLAMP_PACK = 4
FLAG_PACK_RGBA8 = 16
FLAG_NO_SHOW_DEPTH = 8

MODE_RAYTRACE = 0
MODE_PATHTRACE = 1


class Sphere(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("radius", C.c_float), ("diffuse", C.c_float * 3),
                ("albedo", C.c_float), ("metallic", C.c_float), ("roughness", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("proj_inv", C.c_float * 16), ("view_inv", C.c_float * 16),
                ("focus_depth", C.c_float), ("show_depth", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32),
                ("bounces", C.c_uint32), ("paths", C.c_uint32), ("mode", C.c_uint32),
                ("spp", C.c_uint32), ("flags", C.c_uint32), ("seed", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("sphere_tests", C.c_uint64)]


def build(force=False):
    """Compile the oracle with oracle/Makefile if the .so is missing or stale."""
    src = os.path.join(ORACLE_DIR, "luma_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "luma_oracle.h")
    stale = any(not os.path.exists(so)
                or (os.path.exists(src) and os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
                for so in (ORACLE_SO, LITERAL_SO) if so.startswith(ORACLE_DIR))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR] + (["-B"] if force else []))


_lib = None
_literal = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(ORACLE_SO))
    return _lib


def literal_lib():
    """The -DLO_LITERAL build of the oracle (no fusion, double DOF term, libm expf).  NOT the parity
    checker: it exists to measure how far frames move when the arithmetic contract's choices flip."""
    global _literal
    if _literal is None:
        build()
        _literal = _bind(C.CDLL(LITERAL_SO))
    return _literal


def _bind(L):
    f32p = C.POINTER(C.c_float)
    u32p = C.POINTER(C.c_uint32)
    L.lo_expf.restype = C.c_float
    L.lo_expf.argtypes = [C.c_float]
    L.lo_pcg3d.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p]
    L.lo_seed_keys.argtypes = [C.c_uint64, u32p]
    L.lo_rng_vec3.argtypes = [u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, f32p]
    L.lo_rng_vec3_sphere.argtypes = [u32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p]
    L.lo_focus.restype = C.c_float
    L.lo_focus.argtypes = [C.c_float, C.c_float, f32p]
    L.lo_tonemap.argtypes = [f32p, f32p]
    L.lo_pack_rgb.restype = C.c_uint32
    L.lo_pack_rgb.argtypes = [f32p, C.c_uint32]
    L.lo_mat4_inverse.argtypes = [f32p, f32p]
    L.lo_camera_init.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32]
    L.lo_camera_pose.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                 f32p, C.c_float, C.c_float]
    L.lo_camera_matrices.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                     f32p, f32p, f32p, f32p, f32p, f32p]
    L.lo_camera_rays.argtypes = [C.POINTER(Camera), C.c_uint32, C.c_uint32, f32p]
    L.lo_default_scene.argtypes = [C.POINTER(Sphere)]
    L.lo_synthetic_scene.argtypes = [C.POINTER(Sphere), C.c_uint32, C.c_uint64]
    L.lo_render.argtypes = [C.POINTER(Sphere), C.c_uint32, C.POINTER(Camera), C.POINTER(Params), C.c_uint32,
                            C.c_uint32, C.c_uint32, f32p, u32p, C.POINTER(Stats)]
    L.lo_export_ppm.restype = C.c_int
    L.lo_export_ppm.argtypes = [u32p, C.c_int32, C.c_int32, C.c_char_p]
    L.lo_export_bmp.restype = C.c_int
    L.lo_export_bmp.argtypes = [u32p, C.c_int32, C.c_int32, C.c_char_p]
    L.lo_dof_prepass.argtypes = [C.POINTER(Sphere), C.c_uint32, C.POINTER(Camera), C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_uint32, f32p, C.POINTER(C.c_uint8)]
    return L


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def default_scene():
    arr = (Sphere * 5)()
    lib().lo_default_scene(arr)
    return arr


def synthetic_scene(n, seed=0x4C554D41):
    arr = (Sphere * n)()
    lib().lo_synthetic_scene(arr, n, seed)
    return arr


def camera(width, height, fov=70.0, near=0.1, far=100.0):
    cam = Camera()
    lib().lo_camera_init(C.byref(cam), fov, near, far, width, height)
    return cam


def camera_pose(width, height, pos, pitch, yaw, fov=70.0, near=0.1, far=100.0):
    cam = Camera()
    p = np.asarray(pos, dtype=np.float32)
    lib().lo_camera_pose(C.byref(cam), fov, near, far, width, height, _f32p(p), pitch, yaw)
    return cam


def camera_rays(cam, width, height):
    rays = np.empty((height, width, 3), dtype=np.float32)
    lib().lo_camera_rays(C.byref(cam), width, height, _f32p(rays))
    return rays


def make_params(width, height, spp=1, bounces=1, paths=1, samples=1, mode=MODE_RAYTRACE, flags=0,
                seed=0x4C554D41):
    return Params(width, height, samples, bounces, paths, mode, spp, flags, seed)


class Renderer:
    """Holds the reference's accumulated_data / frame_count across render calls
    (renderer.h:52-56) and renders with the oracle."""

    def __init__(self, spheres, cam, width, height, L=None):
        self.L = L or lib()
        self.spheres = spheres
        self.cam = cam
        self.width, self.height = width, height
        self.reset()

    def reset(self):
        self.accum = np.zeros((self.height, self.width, 3), dtype=np.float32)
        self.target = np.zeros((self.height, self.width), dtype=np.uint32)
        self.frame_index = 0
        self.stats = Stats(0, 0)

    def render(self, params, rows=None):
        y0, y1 = rows if rows is not None else (0, self.height)
        self.L.lo_render(self.spheres, len(self.spheres), C.byref(self.cam), C.byref(params), self.frame_index,
                        y0, y1, _f32p(self.accum), _u32p(self.target), C.byref(self.stats))
        if rows is None:
            self.frame_index += params.spp
        return self.target


def render(spheres, cam, params, rows=None):
    r = Renderer(spheres, cam, params.width, params.height)
    r.render(params, rows)
    return r.target, r.accum, r.stats


def dof_prepass(spheres, cam, width, height, rows=None, L=None):
    """(blur float32 [H,W], in_focus uint8 [H,W]) of the DOF pre-pass for the rows."""
    y0, y1 = rows if rows is not None else (0, height)
    blur = np.zeros((height, width), dtype=np.float32)
    flag = np.zeros((height, width), dtype=np.uint8)
    (L or lib()).lo_dof_prepass(spheres, len(spheres), C.byref(cam), width, height, y0, y1, _f32p(blur),
                                flag.ctypes.data_as(C.POINTER(C.c_uint8)))
    return blur, flag


def export_ppm(target, path):
    h, w = target.shape
    t = np.ascontiguousarray(target, dtype=np.uint32)
    return lib().lo_export_ppm(_u32p(t), w, h, path.encode())


def export_bmp(target, path):
    h, w = target.shape
    t = np.ascontiguousarray(target, dtype=np.uint32)
    return lib().lo_export_bmp(_u32p(t), w, h, path.encode())
