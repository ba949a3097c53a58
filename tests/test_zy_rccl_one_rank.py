"""One-rank RCCL rehearsal of the band gather.  The development box has one GPU, so
the multi-rank exchange itself is covered on CPU (tests/test_bands.py, gloo, 2 ranks); here the
same torch.distributed calls run on the "nccl" (= RCCL) backend with world_size 1, which checks
the stream ordering between the render launch and the collective and the device-side layout.
(Named to be collected last: a failure here must not hide the parity tests under pytest -x.)"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_rccl_one_rank_gathers_match_direct_render():
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_one_rank_worker.py")],
                       capture_output=True, text=True, timeout=480)   # a cold first `import torch` can take two minutes
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_bench_single_process_mode_times_the_library_gather():
    """bench.py --single-process: one process, one luma_group, the step is luma_hip_group_render_device (the library's
    own RCCL gather).  On the one-GPU box device 0 is repeated (4 bands, striped, one RCCL rank sending to itself);
    the frame it assembles must be the one-band frame: equal checksums."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    sums = {}
    for n in (1, 4):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--single-process", "--repeat-devices",
                            "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["n_gpus"] == n and d["ranks"] == 1 and d["value"] > 0 and d["gather_ms"] > 0
        assert "luma_hip_group_render_device" in d["config"]["parallelism"] and d["roofline"]["scope"].startswith("per GPU")
        assert ("stripes" in d["config"]["parallelism"]) == (n == 4)
        sums[n] = d["frame_checksum"]
    assert sums[1] == sums[4]
