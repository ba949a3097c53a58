"""CPU tests of the C-ABI library: it loads, exports every symbol include/luma_hip.h declares,
its host-side helpers (camera matrices, scenes) are bit-identical to the oracle's, and with no
GPU present it fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "luma_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(luma_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(abi):
    names = declared_symbols()
    assert len(names) >= 19
    L = abi.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/luma_hip.h but not exported"
    assert sorted(abi.SYMBOLS) == names
    assert L.luma_hip_abi_version() == 3
    assert L.luma_hip_strerror(0) == b"ok" and L.luma_hip_strerror(-1) == b"invalid argument"


def test_no_oracle_in_product():
    """The product must not reach into oracle/ (or tests/): grep the package sources."""
    for base, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".txt")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(base, f)


def test_host_camera_matches_oracle(abi, oracle):
    for (W, H) in [(1920, 1080), (1280, 720), (64, 36), (500, 500), (1584, 396)]:
        a, o = abi.default_camera(W, H), oracle.camera(W, H)
        assert bytes(a) == bytes(o)
    for pos, pitch, yaw in [((1.5, -0.25, 3.0), 0.3, -0.7), ((0, 0, 0), 0.0, 0.0), ((-4, 1, 9), -2.0, 3.0)]:
        a, o = abi.camera_from_pose(96, 54, pos, pitch, yaw), oracle.camera_pose(96, 54, pos, pitch, yaw)
        assert bytes(a) == bytes(o)


def test_host_scenes_match_oracle(abi, oracle):
    assert bytes(abi.default_scene()) == bytes(oracle.default_scene())
    for n in (1, 2, 300, 4096):
        assert bytes(abi.synthetic_scene(n)) == bytes(oracle.synthetic_scene(n))
    assert bytes(abi.synthetic_scene(64, seed=5)) != bytes(abi.synthetic_scene(64, seed=6))


def test_argument_errors_without_gpu(abi):
    L = abi.lib()
    assert L.luma_hip_create(None, 0) == abi.E_ARG
    assert L.luma_camera_default(70.0, 0.1, 100.0, 0, 10, None) == abi.E_ARG
    assert L.luma_scene_default(None) == abi.E_ARG
    assert b"luma_scene_default" in L.luma_hip_last_error()


def test_create_fails_loudly_without_gpu(abi):
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    rc = abi.lib().luma_hip_create(C.byref(h), 0)
    assert rc in (abi.E_HIP, abi.E_ARG) and not h.value
    assert abi.lib().luma_hip_last_error() != b""
