#!/usr/bin/env python3
"""How far do frames move when the arithmetic contract's choices flip?

north_star asks for "a stated MAE/PSNR bound" against the reference CPU renderer.  The HIP path is
bit-identical to the contract oracle (oracle/libluma_oracle.so), but the contract fixes choices the
reference leaves open (MSVC /fp:fast, raytracer.vcxproj:187-190): fused multiply-adds, normalize
by reciprocal, a float exp, a float DOF term.  This script renders BASELINE configs[1] (1080p,
1 spp, primary rays, noise off) and configs[2] (1080p, 64 spp, 5 bounces, pathtrace) with the
contract oracle and with its -DLO_LITERAL build (no fusion anywhere, divide-by-sqrt normalize, libm
expf, the DOF term in double exactly as Renderer.cpp:427-432) and records the 8-bit differences in
tests/golden/contract_sensitivity.json.  Both are restatements (the reference itself cannot be
built here), so this bounds the effect of the arithmetic choices, not of the unknown flux library.

Run:  python tests/golden/make_contract_sensitivity.py        (about a minute on 8 cores)
"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402

W, H, SEED = 1920, 1080, 0x4C554D41

# The bound DESIGN.md section 2 states and tests/test_contract_sensitivity.py asserts (shares are of the
# pixels compared).  Measured values sit well inside: see contract_sensitivity.json "results".
STATED_BOUND = {
    # primary rays, noise off: every pixel within 1 LSB except silhouette flips (d > 0 / t > 0, Renderer.cpp:253,:258)
    "config2": {"mae_lsb": 1e-3, "share_over_1_lsb": 1e-5},
    # 64 spp, 5 bounces, the reference's un-clamped pack: the outliers are channels wrapping at 256 (Renderer.cpp:44-55, :452)
    "config3": {"mae_lsb": 0.05, "psnr_db": 40.0, "share_over_2_lsb": 2e-3},
    # the same frames with LUMA_FLAG_CLAMP_PACK (no wrap-around)
    "config3_clamped_pack": {"mae_lsb": 1e-3, "psnr_db": 60.0, "share_over_2_lsb": 1e-4},
    # pixels whose `focus < .5f` decision (Renderer.cpp:450) differs
    "share_focus_lt_half_flipped": 1e-5,
}


def within_bound(res, bound=STATED_BOUND):
    """List of violations (empty == inside the stated bound) for a results dict of study()."""
    bad = []
    r2, r3, r3c = res["config2_primary_noise_off"], res["config3_64spp_5_bounces"], res["config3_64spp_5_bounces_clamped_pack"]
    if r2["mae_lsb"] > bound["config2"]["mae_lsb"]: bad.append(("config2 mae", r2["mae_lsb"]))
    if r2["pixels_over_1_lsb"] > bound["config2"]["share_over_1_lsb"] * r2["pixels"] + 1: bad.append(("config2 over 1 LSB", r2["pixels_over_1_lsb"]))
    for name, r in (("config3", r3), ("config3_clamped_pack", r3c)):
        b = bound[name]
        if r["mae_lsb"] > b["mae_lsb"]: bad.append((name + " mae", r["mae_lsb"]))
        if r["psnr_db"] < b["psnr_db"]: bad.append((name + " psnr", r["psnr_db"]))
        if r["pixels_over_2_lsb"] > b["share_over_2_lsb"] * r["pixels"] + 1: bad.append((name + " over 2 LSB", r["pixels_over_2_lsb"]))
    d = res["dof_prepass_full_frame"]
    if d["focus_lt_half_flipped"] > bound["share_focus_lt_half_flipped"] * d["pixels"] + 1: bad.append(("focus flips", d["focus_lt_half_flipped"]))
    return bad


def channels(img):
    return np.stack([(img >> 16) & 0xFF, (img >> 8) & 0xFF, img & 0xFF], axis=-1).astype(np.int32)


def compare(a, b):
    d = np.abs(channels(a) - channels(b))
    mse = float((d.astype(np.float64) ** 2).mean())
    return {
        "pixels": int(a.size),
        "pixels_differing": int((a != b).sum()),
        "max_lsb": int(d.max()),
        "mae_lsb": float(d.mean()),
        "psnr_db": float("inf") if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse)),
        "pixels_over_1_lsb": int((d.max(axis=-1) > 1).sum()),
        "pixels_over_2_lsb": int((d.max(axis=-1) > 2).sum()),
    }


def render_rows(L, groups, **kw):
    """The given 4-row groups of the frame, one oracle call per group (threads: rows are independent)."""
    scene, cam = ob.default_scene(), ob.camera(W, H)
    p = ob.make_params(W, H, seed=SEED, **kw)

    def one(y0):
        r = ob.Renderer(scene, cam, W, H, L=L)
        r.render(p, rows=(y0, y0 + 4))
        return y0, r.target[y0:y0 + 4].copy()

    out = np.zeros((4 * len(groups), W), dtype=np.uint32)
    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as pool:
        for i, (y0, rows) in enumerate(sorted(pool.map(one, groups))):
            out[4 * i:4 * i + 4] = rows
    return out


def study(groups2, groups3):
    C, Lit = ob.lib(), ob.literal_lib()
    scene, cam = ob.default_scene(), ob.camera(W, H)
    res = {}
    kw2 = dict(spp=1, bounces=1, paths=1, samples=1, mode=0, flags=ob.FLAG_NOISE_OFF)
    res["config2_primary_noise_off"] = dict(
        compare(render_rows(C, groups2, **kw2), render_rows(Lit, groups2, **kw2)), rows=4 * len(groups2), **{"spp": 1})
    kw3 = dict(spp=64, bounces=5, paths=1, samples=1, mode=1, flags=0)
    res["config3_64spp_5_bounces"] = dict(
        compare(render_rows(C, groups3, **kw3), render_rows(Lit, groups3, **kw3)), rows=4 * len(groups3), **{"spp": 64})
    # the same frames packed with LUMA_FLAG_CLAMP_PACK: the reference's pack has no clamp (Renderer.cpp:44-55),
    # so an in-focus x1.2 channel (:452) that crosses 256/255 wraps into its neighbour and a 1e-7 change shows as
    # up to 255 LSB; clamped, what is left is silhouette / noise-threshold flips
    kw3c = dict(kw3, flags=ob.FLAG_CLAMP_PACK)
    res["config3_64spp_5_bounces_clamped_pack"] = dict(
        compare(render_rows(C, groups3, **kw3c), render_rows(Lit, groups3, **kw3c)), rows=4 * len(groups3), **{"spp": 64})
    # the hard threshold of Renderer.cpp:450 and the blur handed to render_pixel, over the WHOLE frame
    blur_c, flag_c = ob.dof_prepass(scene, cam, W, H, L=C)
    blur_l, flag_l = ob.dof_prepass(scene, cam, W, H, L=Lit)
    res["dof_prepass_full_frame"] = {
        "pixels": W * H,
        "focus_lt_half_flipped": int((flag_c != flag_l).sum()),
        "max_abs_blur_delta": float(np.abs(blur_c.astype(np.float64) - blur_l).max()),
    }
    return res


def main():
    groups2 = list(range(0, H, 4))            # config #2: the whole frame
    groups3 = list(range(0, H, 16))           # config #3: 4-row groups every 16 rows = 272 rows, all 64 spp
    res = study(groups2, groups3)
    doc = {
        "what": "contract oracle vs -DLO_LITERAL oracle, default scene 1920x1080, seed 0x4C554D41 "
                "(tests/golden/make_contract_sensitivity.py)",
        "literal_build": "no fused multiply-add, normalize = v / sqrtf(dot), tonemap by libm expf, DOF term "
                         "1 - exp(-pow(dd, 10)) and lerp(.5, 5, focus) in double, focus < .5f compared in double",
        "results": res,
        # the bound DESIGN.md states and tests/test_contract_sensitivity.py asserts (on a smaller sample)
        "stated_bound": STATED_BOUND,
    }
    with open(os.path.join(HERE, "contract_sensitivity.json"), "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print(json.dumps(res, indent=1))
    print("violations of the stated bound:", within_bound(res))


if __name__ == "__main__":
    main()
