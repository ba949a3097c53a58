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
