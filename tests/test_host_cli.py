"""The headless C++ host (raytracer_amd/host/): luma::Arguments / Renderer / Camera / Image over
the C-ABI.  CPU tests cover the flag parser's log-and-continue behaviour (arguments.cpp:105-117,
:227-230); the GPU test renders through `luma_headless --backend hip` and compares the PPM it
writes, byte for byte, with the oracle's frame written by the oracle's PPM writer."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytracer_amd", "luma_headless")


def run(*args, timeout=120):
    return subprocess.run([EXE, *args], capture_output=True, text=True, timeout=timeout)


def test_parser_logs_and_continues():
    # --context=directx makes main() return 2 right after parsing, before any GPU call
    r = run("--context=directx", "--width=abc", "--bogus=1", "stray", "--mode=fancy", "--paths=3", "--height")
    assert r.returncode == 2
    out = r.stdout
    assert "[INFORMATION]: unrecognized image width `abc`" in out
    assert "[INFORMATION]: unrecognized option `bogus`" in out
    assert "[INFORMATION]: invalid argument `stray`" in out
    assert "[INFORMATION]: unrecognized mode `fancy`" in out
    assert "[WARNING]: path count `3` will be ignored if render mode is not set to \"pathtrace\"" in out
    assert "[INFORMATION]: unrecognized option format `--height`" in out
    assert "only --context=headless" in out


def test_parser_space_and_equals_forms():
    # without a GPU the run ends at the first HIP call, after the options were echoed
    r = run("--backend", "hip", "--gpus", "4", "--width=64", "--height", "36", "--context=directx")
    assert r.returncode == 2
    r = run("--devices=0,0,1", "--width=0")
    assert r.returncode == 2 and "unrecognized" not in r.stdout
    r = run("--devices=0,,1", "--context=directx")
    assert "unrecognized device list `0,,1`" in r.stdout
    r = run("--pos=1,2", "--pos=1,2,3,4", "--pitch=abc", "--pos=1.5,-2e0,3", "--yaw=0.5", "--context=directx")
    assert "unrecognized camera position `1,2`" in r.stdout and "unrecognized camera position `1,2,3,4`" in r.stdout
    assert "unrecognized camera pitch `abc`" in r.stdout
    assert "`1.5,-2e0,3`" not in r.stdout and "camera yaw" not in r.stdout
    r = run("--gpus=9", "--context=directx")
    assert "unrecognized gpu count `9`" in r.stdout
    r = run("--width=0", "--height=10")
    assert r.returncode == 2 and "must be positive" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("args,okw", [
    (["--width=96", "--height=54", "--mode=raytrace", "--bounces=1", "--samples=1", "--frames=1", "--flags=1"],
     dict(spp=1, bounces=1, mode=0, flags=1)),
    (["--width=96", "--height=54", "--mode=pathtrace", "--bounces", "5", "--samples=1", "--paths=1", "--frames=6",
      "--seed=0x4C554D41"], dict(spp=6, bounces=5, mode=1, flags=0)),
])
def test_headless_ppm_matches_oracle(tmp_path, oracle, args, okw):
    out = tmp_path / "frame.ppm"
    r = run("--backend", "hip", *args, f"--output={out}")
    assert r.returncode == 0, r.stdout + r.stderr
    frames = [ln for ln in r.stdout.splitlines() if ln.startswith("[DELIVERED FRAME]:")]
    assert len(frames) == okw["spp"] and frames[0].startswith("[DELIVERED FRAME]:0:")
    W, H = 96, 54
    want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H),
                               oracle.make_params(W, H, seed=0x4C554D41, **okw))
    ref = tmp_path / "oracle.ppm"
    assert oracle.export_ppm(want, str(ref)) == 0
    assert out.read_bytes() == ref.read_bytes()


@pytest.mark.gpu
@pytest.mark.parametrize("sel,H", [("--devices=0,0", 60), ("--devices=0,0,0", 50), ("--gpus=1", 36)])
def test_headless_several_devices_one_frame(tmp_path, oracle, sel, H):
    """--gpus N / --devices: the devices of one process share the frame (luma_hip_group_*); here the
    same GPU stands in for all of them.  The PPM equals the oracle's byte for byte."""
    W = 96
    out = tmp_path / "frame.ppm"
    r = run("--backend", "hip", sel, f"--width={W}", f"--height={H}", "--mode=pathtrace", "--bounces=5", "--samples=1",
            "--paths=1", "--frames=4", f"--output={out}")
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"on {sel.count(',') + 1} device(s)" in r.stdout
    want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H),
                               oracle.make_params(W, H, seed=0x4C554D41, spp=4, bounces=5, mode=1, flags=0))
    ref = tmp_path / "oracle.ppm"
    assert oracle.export_ppm(want, str(ref)) == 0
    assert out.read_bytes() == ref.read_bytes()


@pytest.mark.gpu
def test_headless_config1_writes_the_full_size_ppm(tmp_path, oracle):
    """BASELINE configs[0] end to end: default scene, 1280x720, 1 spp, direct shading (raytrace, one
    bounce), written as output.ppm by Image::export_to_ppm (image.cpp:30-64) -- every byte of the
    22 MB P3 file equals the oracle's frame through the oracle's writer."""
    W, H = 1280, 720
    out = tmp_path / "output.ppm"
    r = run("--backend", "hip", f"--width={W}", f"--height={H}", "--mode=raytrace", "--samples=1", "--bounces=1",
            "--frames=1", f"--output={out}")
    assert r.returncode == 0, r.stdout + r.stderr
    want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H),
                               oracle.make_params(W, H, seed=0x4C554D41, spp=1, bounces=1, mode=0, flags=0))
    ref = tmp_path / "oracle.ppm"
    assert oracle.export_ppm(want, str(ref)) == 0
    got, exp = out.read_bytes(), ref.read_bytes()
    assert got[:16] == b"P3\n1280 720\n255\n"
    assert got == exp


@pytest.mark.gpu
@pytest.mark.parametrize("sel,H", [("--devices=0", 54), ("--devices=0,0", 60), ("--devices=0,0,0", 54)])
def test_headless_gather_over_rccl(tmp_path, oracle, sel, H):
    """--gather=rccl: the bands are gathered on the first device over RCCL
    (luma_hip_group_render_device) and the frame comes to the host in one copy; same PPM."""
    W = 96
    out = tmp_path / "frame.ppm"
    r = run("--backend", "hip", sel, "--gather=rccl", f"--width={W}", f"--height={H}", "--mode=pathtrace", "--bounces=5",
            "--samples=1", "--paths=1", "--frames=3", f"--output={out}")
    assert r.returncode == 0, r.stdout + r.stderr
    want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H),
                               oracle.make_params(W, H, seed=0x4C554D41, spp=3, bounces=5, mode=1, flags=0))
    ref = tmp_path / "oracle.ppm"
    assert oracle.export_ppm(want, str(ref)) == 0
    assert out.read_bytes() == ref.read_bytes()


@pytest.mark.gpu
def test_headless_posed_camera(tmp_path, oracle):
    """--pos/--pitch/--yaw go through Camera::move_to -> update -> the camera.moved branch
    (Renderer.cpp:377-390) before the first frame."""
    W, H = 96, 54
    out = tmp_path / "frame.ppm"
    r = run("--backend=hip", f"--width={W}", f"--height={H}", "--mode=pathtrace", "--bounces=5", "--frames=3",
            "--pos=2,-1,4", "--pitch=0.2", "--yaw", "0.35", f"--output={out}")
    assert r.returncode == 0, r.stdout + r.stderr
    want, _, _ = oracle.render(oracle.default_scene(), oracle.camera_pose(W, H, (2.0, -1.0, 4.0), 0.2, 0.35),
                               oracle.make_params(W, H, seed=0x4C554D41, spp=3, bounces=5, mode=1, flags=0))
    ref = tmp_path / "oracle.ppm"
    assert oracle.export_ppm(want, str(ref)) == 0
    assert out.read_bytes() == ref.read_bytes()


@pytest.mark.gpu
def test_headless_too_many_gpus_is_refused():
    r = run("--backend", "hip", "--gpus=8", "--width=32", "--height=32")
    if "on 8 device(s)" in r.stdout and r.returncode == 0:
        pytest.skip("this box really has 8 GPUs")
    assert r.returncode != 0 and "HIP device(s) are visible" in r.stdout


@pytest.mark.gpu
def test_headless_bmp(tmp_path, oracle):
    out = tmp_path / "frame.bmp"
    r = run("--backend=hip", "--width=33", "--height=17", "--mode=raytrace", "--bounces=1", "--frames=1", "--flags=1",
            f"--output={out}")
    assert r.returncode == 0, r.stdout + r.stderr
    want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(33, 17),
                               oracle.make_params(33, 17, spp=1, bounces=1, mode=0, flags=1))
    ref = tmp_path / "oracle.bmp"
    assert oracle.export_bmp(want, str(ref)) == 0
    assert out.read_bytes() == ref.read_bytes()


@pytest.mark.gpu
def test_headless_binary_pnm(tmp_path, oracle):
    """--output=*.pnm: binary P6 in raster order; the P3 text stays the default."""
    import numpy as np
    out = tmp_path / "frame.pnm"
    r = run("--backend=hip", "--width=33", "--height=17", "--mode=raytrace", "--bounces=1", "--frames=1", "--flags=5",
            f"--output={out}")
    assert r.returncode == 0, r.stdout + r.stderr
    want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(33, 17),
                               oracle.make_params(33, 17, spp=1, bounces=1, mode=0, flags=5))
    blob = out.read_bytes()
    head = b"P6\n33 17\n255\n"
    assert blob.startswith(head) and len(blob) == len(head) + 33 * 17 * 3
    rgb = np.frombuffer(blob[len(head):], dtype=np.uint8).reshape(17, 33, 3).astype(np.uint32)
    assert np.array_equal((rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2], want)


@pytest.mark.gpu
def test_headless_shared_memory_frame(tmp_path, oracle):
    """--shm publishes every frame in a POSIX shared-memory block, the Linux form of the
    reference's LumaFramebuffer mapping (main.cpp:249-256, :274)."""
    name = f"luma_test_fb_{os.getpid()}"
    path = os.path.join("/dev/shm", name)
    out = tmp_path / "f.ppm"
    try:
        r = run("--backend=hip", "--width=48", "--height=20", "--mode=raytrace", "--bounces=1", "--frames=2", "--flags=1",
                f"--shm={name}", f"--output={out}")
        assert r.returncode == 0, r.stdout + r.stderr
        want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(48, 20),
                                   oracle.make_params(48, 20, spp=2, bounces=1, mode=0, flags=1))
        got = np.fromfile(path, dtype=np.uint32).reshape(20, 48)
        assert np.array_equal(got, want)
    finally:
        if os.path.exists(path):
            os.remove(path)
