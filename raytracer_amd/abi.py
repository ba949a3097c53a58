"""ctypes binding of libluma_hip.so (include/luma_hip.h), one Python symbol per C entry point.

This is plumbing for the tests and bench.py; the product is the shared library.  There is
no CPU fallback: if the library is missing, loading fails loudly.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# A/B runs of two builds name the other one in LUMA_HIP_LIB; it is honoured only together with LUMA_AB=1, so
# that a stray variable can never make tests or the bench load something other than the in-tree library
_AB = os.environ.get("LUMA_HIP_LIB") if os.environ.get("LUMA_AB") == "1" else None
LIB_PATH = _AB or os.path.join(_PKG, "libluma_hip.so")

OK, E_ARG, E_HIP, E_NOMEM, E_STATE, E_RCCL = 0, -1, -2, -3, -4, -5
MODE_RAYTRACE, MODE_PATHTRACE = 0, 1
NOALLOC_SCRATCH_BYTES = 64 << 20
FLAG_NOISE_OFF, FLAG_ORIGIN_FROM_HIT, FLAG_CLAMP_PACK, FLAG_NO_SHOW_DEPTH, FLAG_PACK_RGBA8, FLAG_GRID = 1, 2, 4, 8, 16, 32

# every symbol include/luma_hip.h declares (checked by tests/test_abi_symbols.py)
SYMBOLS = [
    "luma_hip_create", "luma_hip_destroy", "luma_hip_set_scene", "luma_hip_set_camera",
    "luma_hip_set_frame", "luma_hip_set_frame_striped", "luma_hip_reset", "luma_hip_render", "luma_hip_render_device",
    "luma_hip_target_device", "luma_hip_read_accum", "luma_hip_frame_index",
    "luma_hip_generate_rays", "luma_camera_default", "luma_camera_from_pose",
    "luma_scene_default", "luma_scene_synthetic", "luma_hip_strerror", "luma_hip_last_error",
    "luma_hip_abi_version",
    "luma_hip_render_frame", "luma_hip_device_count", "luma_hip_group_create", "luma_hip_group_destroy",
    "luma_hip_group_size", "luma_hip_group_context", "luma_hip_group_set_scene", "luma_hip_group_set_camera",
    "luma_hip_group_set_frame", "luma_hip_group_reset", "luma_hip_group_render", "luma_hip_group_read_accum",
    "luma_hip_host_alloc", "luma_hip_host_free", "luma_grid_describe", "luma_hip_read_target",
    "luma_hip_render_device_noalloc",
    "luma_hip_group_render_device", "luma_hip_group_frame_device", "luma_hip_group_read_frame",
    "luma_hip_reserve", "luma_hip_alloc_count", "luma_hip_set_park_limit", "luma_hip_group_reserve", "luma_view_grid_describe",
    "luma_split_plan_describe", "luma_gather_plan_describe",
]


class SplitPlan(C.Structure):
    _fields_ = [("lanes", C.c_uint32), ("fine_lanes", C.c_uint32), ("fine_row0", C.c_uint32), ("park_row0", C.c_uint32),
                ("park_bytes", C.c_uint64), ("refused", C.c_uint32), ("park_depth", C.c_uint32)]


class GatherPlan(C.Structure):
    _fields_ = [("n_members", C.c_uint32), ("n_ranks", C.c_uint32), ("striped", C.c_uint32), ("stripe_rows", C.c_uint32)]


class GatherMember(C.Structure):
    _fields_ = [("device", C.c_int32), ("comm_rank", C.c_int32), ("row_begin", C.c_uint32), ("rows", C.c_uint32),
                ("pixels", C.c_uint64), ("recv_offset", C.c_uint64)]


class Sphere(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("radius", C.c_float),
                ("dr", C.c_float), ("dg", C.c_float), ("db", C.c_float),
                ("albedo", C.c_float), ("metallic", C.c_float), ("roughness", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("proj_inv", C.c_float * 16), ("view_inv", C.c_float * 16),
                ("focus_depth", C.c_float), ("show_depth", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32),
                ("bounces", C.c_uint32), ("paths", C.c_uint32), ("mode", C.c_uint32),
                ("spp", C.c_uint32), ("flags", C.c_uint32), ("seed", C.c_uint64)]


class GridInfo(C.Structure):
    _fields_ = [("dims", C.c_uint32 * 3), ("org", C.c_float * 3), ("cell", C.c_float),
                ("n_items", C.c_uint32), ("n_large", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("gather_ms", C.c_double), ("total_ms", C.c_double),
                ("rays", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("lanes_per_pixel", C.c_uint32), ("split_refused", C.c_uint32)]


class LumaError(RuntimeError):
    def __init__(self, code, where, detail):
        super().__init__(f"{where}: {detail} (code {code})")
        self.code = code


_lib = None


def lib():
    """Load libluma_hip.so.  Import torch BEFORE calling this in a process that uses torch,
    so both share torch's HIP runtime (same libamdhip64 soname)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    ctx, ctxp = C.c_void_p, C.POINTER(C.c_void_p)
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    sig = {
        "luma_hip_create": (C.c_int, [ctxp, C.c_int]),
        "luma_hip_destroy": (None, [ctx]),
        "luma_hip_set_scene": (C.c_int, [ctx, C.POINTER(Sphere), C.c_uint32]),
        "luma_hip_set_camera": (C.c_int, [ctx, C.POINTER(Camera)]),
        "luma_hip_set_frame": (C.c_int, [ctx, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
        "luma_hip_set_frame_striped": (C.c_int, [ctx, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
        "luma_hip_reset": (C.c_int, [ctx]),
        "luma_hip_reserve": (C.c_int, [ctx, C.c_uint32]),
        "luma_hip_alloc_count": (C.c_int, [ctx, C.POINTER(C.c_uint64)]),
        "luma_hip_set_park_limit": (C.c_int, [ctx, C.c_uint64]),
        "luma_hip_render": (C.c_int, [ctx, C.POINTER(Params), u32p, C.POINTER(Stats)]),
        "luma_hip_render_device": (C.c_int, [ctx, C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
        "luma_hip_render_device_noalloc": (C.c_int, [ctx, C.POINTER(Params), C.c_void_p, C.c_void_p]),
        "luma_hip_target_device": (C.c_int, [ctx, C.POINTER(C.c_void_p)]),
        "luma_hip_read_accum": (C.c_int, [ctx, f32p]),
        "luma_hip_read_target": (C.c_int, [ctx, C.c_void_p, u32p]),
        "luma_hip_frame_index": (C.c_int, [ctx, u32p]),
        "luma_hip_generate_rays": (C.c_int, [ctx, f32p]),
        "luma_camera_default": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.POINTER(Camera)]),
        "luma_camera_from_pose": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, f32p,
                                            C.c_float, C.c_float, C.POINTER(Camera)]),
        "luma_scene_default": (C.c_int, [C.POINTER(Sphere)]),
        "luma_scene_synthetic": (C.c_int, [C.POINTER(Sphere), C.c_uint32, C.c_uint64]),
        "luma_hip_render_frame": (C.c_int, [ctx, C.POINTER(Params), u32p, C.POINTER(Stats)]),
        "luma_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "luma_hip_group_create": (C.c_int, [ctxp, C.c_int, C.POINTER(C.c_int)]),
        "luma_hip_group_destroy": (None, [ctx]),
        "luma_hip_group_size": (C.c_int, [ctx, C.POINTER(C.c_int)]),
        "luma_hip_group_context": (C.c_int, [ctx, C.c_int, ctxp]),
        "luma_hip_group_set_scene": (C.c_int, [ctx, C.POINTER(Sphere), C.c_uint32]),
        "luma_hip_group_set_camera": (C.c_int, [ctx, C.POINTER(Camera)]),
        "luma_hip_group_set_frame": (C.c_int, [ctx, C.c_uint32, C.c_uint32]),
        "luma_hip_group_reset": (C.c_int, [ctx]),
        "luma_hip_group_reserve": (C.c_int, [ctx, C.c_uint32]),
        "luma_hip_group_render": (C.c_int, [ctx, C.POINTER(Params), u32p, C.POINTER(Stats)]),
        "luma_hip_group_read_accum": (C.c_int, [ctx, f32p]),
        "luma_hip_group_render_device": (C.c_int, [ctx, C.POINTER(Params), C.c_void_p, C.POINTER(Stats)]),
        "luma_hip_group_frame_device": (C.c_int, [ctx, C.POINTER(C.c_void_p)]),
        "luma_hip_group_read_frame": (C.c_int, [ctx, u32p]),
        "luma_hip_host_alloc": (C.c_int, [ctxp, C.c_size_t]),
        "luma_hip_host_free": (None, [C.c_void_p]),
        "luma_grid_describe": (C.c_int, [C.POINTER(Sphere), C.c_uint32, C.POINTER(GridInfo), u32p, u32p, u32p]),
        "luma_view_grid_describe": (C.c_int, [C.POINTER(Sphere), C.c_uint32, f32p, u32p, u32p, u32p, u32p, u32p, u32p]),
        "luma_split_plan_describe": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(SplitPlan)]),
        "luma_gather_plan_describe": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.POINTER(GatherPlan),
                                                C.POINTER(GatherMember), C.POINTER(C.c_int)]),
        "luma_hip_strerror": (C.c_char_p, [C.c_int]),
        "luma_hip_last_error": (C.c_char_p, []),
        "luma_hip_abi_version": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(code, where):
    if code != OK:
        L = lib()
        raise LumaError(code, where, (L.luma_hip_last_error() or b"").decode() or L.luma_hip_strerror(code).decode())


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def default_scene():
    arr = (Sphere * 5)()
    check(lib().luma_scene_default(arr), "luma_scene_default")
    return arr


def synthetic_scene(n, seed=0x4C554D41):
    arr = (Sphere * n)()
    check(lib().luma_scene_synthetic(arr, n, seed), "luma_scene_synthetic")
    return arr


def default_camera(width, height, fov=70.0, near=0.1, far=100.0):
    cam = Camera()
    check(lib().luma_camera_default(fov, near, far, width, height, C.byref(cam)), "luma_camera_default")
    return cam


def camera_from_pose(width, height, pos, pitch, yaw, fov=70.0, near=0.1, far=100.0):
    cam = Camera()
    p = np.asarray(pos, dtype=np.float32)
    check(lib().luma_camera_from_pose(fov, near, far, width, height, _f32p(p), pitch, yaw, C.byref(cam)),
          "luma_camera_from_pose")
    return cam


def make_params(width, height, spp=1, bounces=1, paths=1, samples=1, mode=MODE_RAYTRACE, flags=0,
                seed=0x4C554D41):
    return Params(width, height, samples, bounces, paths, mode, spp, flags, seed)


class HostFrame:
    """A page-locked [H, W] uint32 frame from luma_hip_host_alloc, as a numpy view."""

    def __init__(self, height, width):
        self._p = C.c_void_p()
        check(lib().luma_hip_host_alloc(C.byref(self._p), height * width * 4), "luma_hip_host_alloc")
        buf = (C.c_uint32 * (height * width)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=np.uint32).reshape(height, width)

    def close(self):
        if self._p:
            self.array = None
            lib().luma_hip_host_free(self._p)
            self._p = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def describe_grid(spheres):
    """luma_grid_describe: (GridInfo, cell_start, items, large) as numpy arrays."""
    n = len(spheres)
    info = GridInfo()
    check(lib().luma_grid_describe(spheres if n else None, n, C.byref(info), None, None, None), "luma_grid_describe")
    cells = int(info.dims[0]) * int(info.dims[1]) * int(info.dims[2])
    cell_start = np.zeros(cells + 1, dtype=np.uint32)
    items = np.zeros(max(info.n_items, 1), dtype=np.uint32)
    large = np.zeros(max(info.n_large, 1), dtype=np.uint32)
    check(lib().luma_grid_describe(spheres if n else None, n, C.byref(info), _u32p(cell_start), _u32p(items), _u32p(large)),
          "luma_grid_describe")
    return info, cell_start, items[:info.n_items], large[:info.n_large]


def describe_view_grid(spheres, origin):
    """luma_view_grid_describe: (res, cell_start, items, large) as numpy arrays."""
    n = len(spheres)
    o = np.asarray(origin, dtype=np.float32)
    res, n_items, n_large = C.c_uint32(), C.c_uint32(), C.c_uint32()
    check(lib().luma_view_grid_describe(spheres if n else None, n, _f32p(o), C.byref(res), C.byref(n_items), C.byref(n_large),
                                        None, None, None), "luma_view_grid_describe")
    cell_start = np.zeros(6 * res.value * res.value + 1, dtype=np.uint32)
    items = np.zeros(max(n_items.value, 1), dtype=np.uint32)
    large = np.zeros(max(n_large.value, 1), dtype=np.uint32)
    check(lib().luma_view_grid_describe(spheres if n else None, n, _f32p(o), C.byref(res), C.byref(n_items), C.byref(n_large),
                                        _u32p(cell_start), _u32p(items), _u32p(large)), "luma_view_grid_describe")
    return res.value, cell_start, items[:n_items.value], large[:n_large.value]


def split_plan(width, rows, spp, n_spheres, flags=0, park_budget=2 << 30):
    """luma_split_plan_describe: how a render call would spread a pixel's frames over lanes (host-only)."""
    plan = SplitPlan()
    check(lib().luma_split_plan_describe(width, rows, spp, n_spheres, flags, park_budget, C.byref(plan)), "luma_split_plan_describe")
    return plan


def gather_plan(device_ids, width, height):
    """luma_gather_plan_describe: (GatherPlan, [GatherMember], [device of each communicator rank]) -- host-only."""
    n = len(device_ids)
    ids = (C.c_int * max(n, 1))(*device_ids)
    plan, members, ranks = GatherPlan(), (GatherMember * max(n, 1))(), (C.c_int * max(n, 1))()
    check(lib().luma_gather_plan_describe(n, ids, width, height, C.byref(plan), members, ranks), "luma_gather_plan_describe")
    return plan, list(members[:plan.n_members]), list(ranks[:plan.n_ranks])


def device_count():
    n = C.c_int()
    check(lib().luma_hip_device_count(C.byref(n)), "luma_hip_device_count")
    return n.value


class Context:
    """Owns one luma_ctx (one device, one scanline band)."""

    def __init__(self, device_id=-1, _borrowed=None):
        self._owned = _borrowed is None
        self._h = C.c_void_p() if self._owned else _borrowed
        if self._owned:
            check(lib().luma_hip_create(C.byref(self._h), device_id), "luma_hip_create")
        self.width = self.height = self.row_begin = self.row_end = 0

    def close(self):
        if self._h and self._owned:
            lib().luma_hip_destroy(self._h)
        self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_frame(self, params, frame, want_stats=False):
        """luma_hip_render_frame: writes this context's rows into `frame` ([H, W] uint32, in place)."""
        st = Stats()
        check(lib().luma_hip_render_frame(self._h, C.byref(params), _u32p(frame), C.byref(st) if want_stats else None),
              "luma_hip_render_frame")
        return st

    def set_scene(self, spheres):
        n = len(spheres)
        check(lib().luma_hip_set_scene(self._h, spheres if n else None, n), "luma_hip_set_scene")

    def set_camera(self, cam):
        check(lib().luma_hip_set_camera(self._h, C.byref(cam)), "luma_hip_set_camera")

    def set_frame(self, width, height, row_begin=0, row_end=None):
        row_end = height if row_end is None else row_end
        check(lib().luma_hip_set_frame(self._h, width, height, row_begin, row_end), "luma_hip_set_frame")
        self.width, self.height, self.row_begin, self.row_end = width, height, row_begin, row_end

    def set_frame_striped(self, width, height, stripe_rows, rank, n_ranks):
        check(lib().luma_hip_set_frame_striped(self._h, width, height, stripe_rows, rank, n_ranks),
              "luma_hip_set_frame_striped")
        self.width, self.height = width, height
        self.row_begin, self.row_end = 0, height // n_ranks      # rows == row_end - row_begin

    def reset(self):
        check(lib().luma_hip_reset(self._h), "luma_hip_reset")

    def reserve(self, max_spp):
        check(lib().luma_hip_reserve(self._h, max_spp), "luma_hip_reserve")

    def set_park_limit(self, nbytes):
        check(lib().luma_hip_set_park_limit(self._h, nbytes), "luma_hip_set_park_limit")

    def alloc_count(self):
        n = C.c_uint64()
        check(lib().luma_hip_alloc_count(self._h, C.byref(n)), "luma_hip_alloc_count")
        return n.value

    @property
    def rows(self):
        return self.row_end - self.row_begin

    def render(self, params, want_stats=True, out=None):
        """luma_hip_render: returns (band uint32 [rows, W], Stats).  `out`: a frame to fill in place."""
        out = np.empty((self.rows, self.width), dtype=np.uint32) if out is None else out
        st = Stats()
        check(lib().luma_hip_render(self._h, C.byref(params), _u32p(out), C.byref(st) if want_stats else None),
              "luma_hip_render")
        return out, st

    def render_device(self, params, target_ptr=None, stream=None, want_stats=False):
        """luma_hip_render_device: band stays in HBM.  Returns Stats (zeros unless want_stats)."""
        st = Stats()
        check(lib().luma_hip_render_device(self._h, C.byref(params), target_ptr, stream,
                                           C.byref(st) if want_stats else None), "luma_hip_render_device")
        return st

    def render_device_noalloc(self, params, target_ptr=None, stream=None):
        """luma_hip_render_device_noalloc: as render_device, scratch in the library's static array."""
        check(lib().luma_hip_render_device_noalloc(self._h, C.byref(params), target_ptr, stream),
              "luma_hip_render_device_noalloc")

    def target_device(self):
        p = C.c_void_p()
        check(lib().luma_hip_target_device(self._h, C.byref(p)), "luma_hip_target_device")
        return p.value

    def read_accum(self):
        out = np.empty((self.rows, self.width, 3), dtype=np.float32)
        check(lib().luma_hip_read_accum(self._h, _f32p(out)), "luma_hip_read_accum")
        return out

    def read_target(self, target_ptr=None):
        """luma_hip_read_target: the band left in device memory by render_device, as [rows, W] uint32."""
        out = np.empty((self.rows, self.width), dtype=np.uint32)
        check(lib().luma_hip_read_target(self._h, target_ptr, _u32p(out)), "luma_hip_read_target")
        return out

    def frame_index(self):
        v = C.c_uint32()
        check(lib().luma_hip_frame_index(self._h, C.byref(v)), "luma_hip_frame_index")
        return v.value

    def generate_rays(self):
        out = np.empty((self.rows, self.width, 3), dtype=np.float32)
        check(lib().luma_hip_generate_rays(self._h, _f32p(out)), "luma_hip_generate_rays")
        return out


class Group:
    """Owns one luma_group: n contexts (one per device id) that render one host frame together."""

    def __init__(self, device_ids):
        ids = (C.c_int * len(device_ids))(*device_ids)
        self._h = C.c_void_p()
        check(lib().luma_hip_group_create(C.byref(self._h), len(device_ids), ids), "luma_hip_group_create")
        self.width = self.height = 0

    def close(self):
        if self._h:
            lib().luma_hip_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self):
        n = C.c_int()
        check(lib().luma_hip_group_size(self._h, C.byref(n)), "luma_hip_group_size")
        return n.value

    def context(self, i, rows):
        """Borrowed Context of member i; `rows` = its row count (for the read-back shapes)."""
        h = C.c_void_p()
        check(lib().luma_hip_group_context(self._h, i, C.byref(h)), "luma_hip_group_context")
        c = Context(_borrowed=h)
        c.width, c.height, c.row_begin, c.row_end = self.width, self.height, 0, rows
        return c

    def set_scene(self, spheres):
        n = len(spheres)
        check(lib().luma_hip_group_set_scene(self._h, spheres if n else None, n), "luma_hip_group_set_scene")

    def set_camera(self, cam):
        check(lib().luma_hip_group_set_camera(self._h, C.byref(cam)), "luma_hip_group_set_camera")

    def set_frame(self, width, height):
        check(lib().luma_hip_group_set_frame(self._h, width, height), "luma_hip_group_set_frame")
        self.width, self.height = width, height

    def reset(self):
        check(lib().luma_hip_group_reset(self._h), "luma_hip_group_reset")

    def reserve(self, max_spp):
        check(lib().luma_hip_group_reserve(self._h, max_spp), "luma_hip_group_reserve")

    def read_accum(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        check(lib().luma_hip_group_read_accum(self._h, _f32p(out)), "luma_hip_group_read_accum")
        return out

    def render_device(self, params, frame_ptr=None, want_stats=True):
        """luma_hip_group_render_device: bands gathered on the root GPU over RCCL.  Returns Stats."""
        st = Stats()
        check(lib().luma_hip_group_render_device(self._h, C.byref(params), frame_ptr, C.byref(st) if want_stats else None),
              "luma_hip_group_render_device")
        return st

    def frame_device(self):
        p = C.c_void_p()
        check(lib().luma_hip_group_frame_device(self._h, C.byref(p)), "luma_hip_group_frame_device")
        return p.value

    def read_frame(self):
        out = np.empty((self.height, self.width), dtype=np.uint32)
        check(lib().luma_hip_group_read_frame(self._h, _u32p(out)), "luma_hip_group_read_frame")
        return out

    def render(self, params, want_stats=True, out=None):
        out = np.empty((self.height, self.width), dtype=np.uint32) if out is None else out
        st = Stats()
        check(lib().luma_hip_group_render(self._h, C.byref(params), _u32p(out), C.byref(st) if want_stats else None),
              "luma_hip_group_render")
        return out, st
