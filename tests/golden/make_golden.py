#!/usr/bin/env python3
"""Regenerates the committed golden vectors under tests/golden/.

Two kinds:
  * ref_*  -- data taken from files the reference itself holds (/root/reference/output.ppm,
    luma.ppm): the only known-answer material the reference has for this path.
  * oracle_* -- outputs of the CPU oracle (oracle/luma_oracle.c) on fixed inputs.  They pin the
    oracle against drift and give the GPU tests frames that do not depend on rebuilding it.
Run from the repo root in the build container:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402

REF = "/root/reference"
SEED = 0x4C554D41


def ref_fixtures():
    # output.ppm: 50x50, every pixel "115 128 156" -- tonemap + pack + PPM text KAT
    shutil.copyfile(os.path.join(REF, "output.ppm"), os.path.join(HERE, "ref_output.ppm"))
    # luma.ppm: only its layout is usable (older shading model); keep the facts, not the 800 KB file
    with open(os.path.join(REF, "luma.ppm")) as f:
        lines = f.read().split("\n")
    body = [ln for ln in lines[3:] if ln.strip()]
    facts = {"header": lines[:3], "body_lines": len(body),
             "triples_per_line": sorted({len(ln.split()) // 3 for ln in body}),
             "first_triple": body[0].split()[:3],
             "separator_two_spaces": all("  " in ln for ln in body[:4])}
    with open(os.path.join(HERE, "ref_luma_ppm_layout.json"), "w") as f:
        json.dump(facts, f, indent=1)


def oracle_frames():
    out = {}
    cases = {
        "primary_64x36": dict(W=64, H=36, spp=1, bounces=1, mode=0, flags=ob.FLAG_NOISE_OFF),
        "primary_160x90": dict(W=160, H=90, spp=1, bounces=1, mode=0, flags=ob.FLAG_NOISE_OFF),
        "direct_noise_64x36": dict(W=64, H=36, spp=1, bounces=1, mode=0, flags=0),
        "path_64x36_4spp": dict(W=64, H=36, spp=4, bounces=5, mode=1, flags=0),
        "path_160x90_4spp": dict(W=160, H=90, spp=4, bounces=5, mode=1, flags=0),
        "path_fromhit_64x36": dict(W=64, H=36, spp=2, bounces=5, mode=1, flags=ob.FLAG_ORIGIN_FROM_HIT, paths=2),
    }
    for name, c in cases.items():
        W, H = c.pop("W"), c.pop("H")
        p = ob.make_params(W, H, seed=SEED, **c)
        t, a, st = ob.render(ob.default_scene(), ob.camera(W, H), p)
        out[name + "_target"] = t
        out[name + "_accum"] = a
        out[name + "_rays"] = np.array([st.rays, st.sphere_tests], dtype=np.uint64)
    W, H = 48, 27
    p = ob.make_params(W, H, spp=2, bounces=5, mode=1, seed=SEED)
    t, a, st = ob.render(ob.synthetic_scene(300), ob.camera(W, H), p)
    out["synthetic300_48x27_target"], out["synthetic300_48x27_accum"] = t, a
    np.savez_compressed(os.path.join(HERE, "oracle_frames.npz"), **out)


def oracle_kats():
    L = ob.lib()
    import ctypes as C
    k = {}
    xs = np.concatenate([np.linspace(-90, 2, 2001), -np.logspace(-8, 1.9, 500)]).astype(np.float32)
    k["expf_x"] = xs
    k["expf_y"] = np.array([L.lo_expf(float(x)) for x in xs], dtype=np.float32)
    ctr = np.array([[0, 0, 0], [1, 2, 3], [0xFFFFFFFF, 0, 7], [123456, 63, 14], [2073599, 4095, 11]], dtype=np.uint32)
    outs = np.zeros_like(ctr)
    for i, (x, y, z) in enumerate(ctr):
        o = (C.c_uint32 * 3)()
        L.lo_pcg3d(int(x), int(y), int(z), o)
        outs[i] = list(o)
    k["pcg3d_in"], k["pcg3d_out"] = ctr, outs
    keys = (C.c_uint32 * 3)()
    L.lo_seed_keys(SEED, keys)
    k["seed_keys"] = np.array(list(keys), dtype=np.uint32)
    v = np.zeros((8, 3), dtype=np.float32)
    for i in range(8):
        o = (C.c_float * 3)()
        L.lo_rng_vec3(keys, 1000 + i, i, 2 * i, -0.5, 0.5, o)
        v[i] = list(o)
    k["rng_vec3"] = v
    for i in range(8):
        o = (C.c_float * 3)()
        L.lo_rng_vec3_sphere(keys, 1000 + i, i, 2 * i, o)
        v[i] = list(o)
    k["rng_vec3_sphere"] = v.copy()
    depths = np.array([3.4028235e38, 9.0, 9.5, 10.0, 10.4, 10.964, 11.0, 11.5, 13.0, 90.0], dtype=np.float32)
    fb = np.zeros((len(depths), 2), dtype=np.float32)
    for i, d in enumerate(depths):
        b = C.c_float()
        fb[i, 0] = L.lo_focus(float(d), 10.0, C.byref(b))
        fb[i, 1] = b.value
    k["focus_depth"], k["focus_blur"] = depths, fb
    cam = ob.camera(1920, 1080)
    k["cam1080_proj_inv"] = np.array(cam.proj_inv[:], dtype=np.float32)
    k["cam1080_view_inv"] = np.array(cam.view_inv[:], dtype=np.float32)
    rays = ob.camera_rays(ob.camera(64, 36), 64, 36)
    k["rays_64x36"] = rays
    posed = ob.camera_pose(96, 54, (1.5, -0.25, 3.0), 0.3, -0.7)
    k["campose_proj_inv"] = np.array(posed.proj_inv[:], dtype=np.float32)
    k["campose_view_inv"] = np.array(posed.view_inv[:], dtype=np.float32)
    sc = ob.synthetic_scene(4096)
    k["synthetic4096_sha256"] = np.frombuffer(hashlib.sha256(bytes(sc)).digest(), dtype=np.uint8)
    k["synthetic_first8"] = np.frombuffer(bytes(sc)[:8 * 40], dtype=np.float32).reshape(8, 10)
    np.savez_compressed(os.path.join(HERE, "oracle_kats.npz"), **k)


if __name__ == "__main__":
    ob.build()
    if os.path.isdir(REF):
        ref_fixtures()
    oracle_frames()
    oracle_kats()
    print("golden vectors written to", HERE)
