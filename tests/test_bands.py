"""Host logic of the multi-GPU partition (raytracer_amd/bands.py), on CPU: band arithmetic and a
world_size-2 gloo run of the gather.  The per-rank renderer here is the oracle, standing in for a
GPU rank; what is under test is the partition + gather, which is device independent."""
import os
import socket

import numpy as np
import pytest

from raytracer_amd import bands

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_rows_partition():
    for H in (1080, 2160, 4320, 104, 7):
        for N in (1, 2, 3, 4, 7, 8):
            if H < N:
                continue
            rows = [bands.band_rows(H, N, r) for r in range(N)]
            assert rows[0][0] == 0 and rows[-1][1] == H
            assert all(rows[i][1] == rows[i + 1][0] for i in range(N - 1))
            sizes = [e - b for b, e in rows]
            assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    assert bands.band_rows(1080, 8, 3) == (405, 540)      # 135-row bands
    assert bands.band_rows(2160, 8, 7) == (1890, 2160)
    with pytest.raises(ValueError):
        bands.band_rows(4, 8, 0)
    with pytest.raises(ValueError):
        bands.band_rows(100, 2, 2)


def test_striped_rows_partition():
    for H, N, S in [(1080, 8, 15), (1080, 4, 15), (2160, 8, 15), (120, 2, 15), (64, 4, 4)]:
        rows = [bands.striped_rows(H, N, r, S) for r in range(N)]
        assert sorted(y for rr in rows for y in rr) == list(range(H))
        assert all(len(rr) == H // N for rr in rows)
        assert rows[1][:S] == list(range(S, 2 * S)) and rows[0][S:2 * S] == list(range(N * S, N * S + S))
    assert bands.stripe_rows_for(1080, 8) == 15 and bands.stripe_rows_for(1080, 1) == 0
    assert bands.stripe_rows_for(1000, 8) == 0
    with pytest.raises(ValueError):
        bands.striped_rows(100, 8, 0, 15)


def _striped_worker(rank, world, port, H, W, stripe, q):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = bands.striped_rows(H, world, rank, stripe)
        p = ob.make_params(W, H, spp=2, bounces=5, mode=1, seed=0x4C554D41)
        r = ob.Renderer(ob.default_scene(), ob.camera(W, H), W, H)
        for y0 in rows[::stripe]:
            r.render(p, rows=(y0, y0 + stripe))
        band = torch.from_numpy(r.target[rows].astype(np.int64).astype(np.int32).copy())
        frame = bands.gather_striped(band, H, W, rank, world, stripe)
        if rank == 0:
            q.put(frame.numpy().astype(np.uint32))
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


def _pipelined_worker(rank, world, port, H, W, stripe, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = bands.striped_rows(H, world, rank, stripe) if stripe else list(range(*bands.band_rows(H, world, rank)))
        pg = bands.PipelinedGather(H, W, rank, world, stripe, torch.device("cpu"))
        for i in range(5):                       # frame i holds the value 1000*i + row
            buf = pg.acquire(i)
            buf.copy_(torch.tensor([[1000 * i + y] * W for y in rows], dtype=torch.int32))
            pg.submit(i)
        frame = pg.finish()
        if rank == 0:
            q.put(frame.numpy().copy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("stripe", [0, 3])
def test_pipelined_gather_two_ranks(stripe):
    import torch.multiprocessing as mp
    W, H = 8, 24
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, H, W, stripe, q)) for r in range(2)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(frame, np.array([[4000 + y] * W for y in range(H)], dtype=np.int32))


def test_two_rank_striped_gather_equals_full_frame():
    import torch.multiprocessing as mp
    import oracle_binding as ob
    W, H, stripe = 40, 24, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_striped_worker, args=(r, 2, port, H, W, stripe, q)) for r in range(2)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full, _, _ = ob.render(ob.default_scene(), ob.camera(W, H), ob.make_params(W, H, spp=2, bounces=5, mode=1, seed=0x4C554D41))
    assert np.array_equal(frame, full)


def _worker(rank, world, port, H, W, q):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        y0, y1 = bands.band_rows(H, world, rank)
        p = ob.make_params(W, H, spp=2, bounces=5, mode=1, seed=0x4C554D41)
        r = ob.Renderer(ob.default_scene(), ob.camera(W, H), W, H)
        r.render(p, rows=(y0, y1))
        band = torch.from_numpy(r.target[y0:y1].astype(np.int64).astype(np.int32).copy())
        frame = bands.gather_frame(band, H, W, rank, world)
        if rank == 0:
            q.put(frame.numpy().astype(np.uint32))
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H", [36, 37])          # equal bands (gather) and ragged bands (send/recv)
def test_two_rank_gather_equals_full_frame(H):
    import torch.multiprocessing as mp
    import oracle_binding as ob
    W = 48
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, H, W, q)) for r in range(2)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full, _, _ = ob.render(ob.default_scene(), ob.camera(W, H), ob.make_params(W, H, spp=2, bounces=5, mode=1, seed=0x4C554D41))
    assert np.array_equal(frame, full)


# ---- bench.py --gpus N from a plain shell: the launcher path (no GPU needed) ----------

def _run_bench(*args, timeout=120):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_starts_its_own_ranks():
    """WORLD_SIZE unset and --gpus 2: bench.py spawns two rank processes itself (before any torch
    import in the parent), they rendezvous on 127.0.0.1, and rank 0 alone prints the JSON line."""
    import json
    r = _run_bench("--gpus", "2", "--selftest-spawn")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"spawn_selftest": {"world": 2, "rank_sum": 1}}


def test_bench_launcher_fails_when_a_rank_fails():
    """A rank that dies must not leave the others waiting in a collective: the parent stops them
    (by PID) and exits non-zero."""
    r = _run_bench("--gpus", "3", "--selftest-spawn", "--selftest-fail-rank", "1", timeout=90)
    assert r.returncode == 3
    assert "spawn_selftest" not in r.stdout


def test_bench_launcher_stops_ranks_that_hang():
    """A rank stuck in a rendezvous or a collective (the usual first-contact failure on a new node) must not burn the
    driver's whole time limit: after --rank-deadline seconds the parent stops the ranks, names them, exits 124."""
    r = _run_bench("--gpus", "2", "--selftest-spawn", "--selftest-hang-rank", "1", "--rank-deadline", "10", timeout=90)
    assert r.returncode == 124
    assert "still running after 10 s" in r.stderr and "spawn_selftest" not in r.stdout
