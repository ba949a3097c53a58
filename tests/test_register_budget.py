"""The headline kernels' register budgets, checked where the library is built (no GPU needed).  The config #3
instances of the ray-per-lane kernel run seven waves per SIMD with at most 72 VGPRs and 96 SGPRs and no spills; an
eighth wave (64 VGPRs) was worth ~1 % to the one-class instance (profiles/r02/ab_lazy_args.txt), one
wave fewer costs 10 %.  The two-class instance that renders the whole config #3 frame (FINE = 4) was measured at seven
waves per SIMD (72 VGPRs, 96 SGPRs) and must not spill: the no-allocation entry point launches it too.
Compiles the kernel instance groups (luma_inst_*.hip) to assembly once (about a minute) and reads the kernels' metadata."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernel_metadata(tmp_path_factory):
    """Every group of kernel instances (luma_inst_*.hip) compiled to assembly, in parallel like the Makefile does."""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    tmp = tmp_path_factory.mktemp("asm")
    flags = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]   # raytracer_amd/csrc/Makefile

    def one(src):
        out = tmp / (os.path.basename(src) + ".s")
        subprocess.check_call([HIPCC] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--offload-arch=gfx950",
                                                 "--cuda-device-only", "-S", "-o", str(out), src],
                              stderr=subprocess.DEVNULL, timeout=900)
        return out.read_text()

    sources = sorted(glob.glob(os.path.join(CSRC, "luma_inst_*.hip")))
    assert len(sources) >= 13
    meta = {}
    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as pool:
        for text in pool.map(one, sources):
            for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
                meta[m.group(1)] = {k: int(v) for k, v in re.findall(r"\.(\w+_count|private_segment_fixed_size):\s+(\d+)", m.group(2))}
    return meta


def test_makefile_flags_are_the_ones_checked():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    for flag in ("-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "--offload-arch=$(ARCH)", "ARCH     ?= gfx950"):
        assert flag in mk, flag


def test_config3_one_class_kernel_keeps_seven_waves_per_simd(kernel_metadata):
    # render_kernel<TILE_W=32, WAVES=4, SCENE_LDS_ALL, NOISE, !FROM_HIT, SPLIT=1, !GRID, NS=5, FINE=0>: 64 VGPRs and eight
    # waves until round 3, whose select-based candidate handling (-27 % scalar and branch instructions, -1.4 % time on the
    # two-class instance that renders config #3) needs 65; eight waves were worth ~1 % to this instance, config #4 measured
    # the same before and after (profiles/r03/ab_select_scan.txt)
    # (and its TAKES twin, whose lanes take frames of their wave's other pixels: the last template argument)
    names = [n for n in kernel_metadata if re.search(r"render_kernelILi32ELi4ELi2ELb1ELb0ELi1ELb0ELi5ELi0ELb[01]E", n)]
    assert len(names) == 2, names
    for name in names:
        k = kernel_metadata[name]
        assert k["vgpr_count"] <= 72, (name, k)
        assert k["sgpr_count"] <= 96, (name, k)
        assert k["vgpr_spill_count"] == 0 and k["sgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (name, k)


def test_config3_two_class_kernel_keeps_seven_waves_per_simd(kernel_metadata):
    # render_kernel<TILE_W=32, WAVES=4, SCENE_LDS_ALL, NOISE, !FROM_HIT, SPLIT=1 or 2, !GRID, NS=5, FINE=4> and <..., SPLIT=4, ..., FINE=8>
    # (SPLIT = 1, FINE = 4 twice: the TAKES instance renders the whole config #3 frame, its twin launches of more than
    #  kStealSppMax frames per pixel)
    names = [n for n in kernel_metadata if re.search(r"render_kernelILi32ELi4ELi2ELb1ELb0ELi([12]ELb0ELi5ELi4|4ELb0ELi5ELi8)ELb[01]E", n)]
    assert len(names) == 4, names
    for name in names:
        k = kernel_metadata[name]
        assert k["vgpr_count"] <= 72 and k["sgpr_count"] <= 96, (name, k)
        assert k["vgpr_spill_count"] == 0 and k["sgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (name, k)


def test_kernels_behind_the_no_alloc_entry_point_use_no_private_memory(kernel_metadata):
    # luma_hip_render_device_noalloc refuses LUMA_FLAG_GRID only: every other render kernel must run without the
    # scratch memory the HIP runtime would allocate at dispatch (include/luma_hip.h)
    checked = 0
    for name, k in kernel_metadata.items():
        m = re.search(r"render_kernelILi\d+ELi\d+ELi\d+ELb[01]ELb[01]ELi\d+ELb([01])E", name)
        sp = re.search(r"render_kernel_spILi\d+ELi\d+ELb[01]ELb[01]ELb([01])E", name)
        grid = (m and m.group(1) == "1") or (sp and sp.group(1) == "1")
        if (m or sp) and not grid:
            assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (name, k)
            checked += 1
    assert checked >= 200
