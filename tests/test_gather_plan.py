"""Host logic of luma_hip_group_render_device's band exchange (luma_host.cpp plan_gather, the plan the library itself
runs, shown by luma_gather_plan_describe; no GPU): which communicator rank sends which band, where the root receives it,
and that the receives (plus the de-interleave of striped frames) put every row of the frame in its place exactly once.
The exchange is replayed in numpy: a "band" is the list of frame rows its member renders."""
import numpy as np
import pytest


def member_rows(plan, m, i, height):
    """Frame rows of member i, in the order they lie in its band buffer (luma_kernels.hip global_row)."""
    if not plan.striped:
        return list(range(m.row_begin, m.row_begin + m.rows))
    s, n = plan.stripe_rows, plan.n_members
    return [q * s * n + i * s + r for q in range(height // (s * n)) for r in range(s)]


def replay(abi, ids, W, H):
    plan, members, ranks = abi.gather_plan(ids, W, H)
    assert plan.n_members == len(ids) and len(members) == len(ids)
    # communicator: one rank per distinct device, first appearance first, the root (member 0's device) is rank 0
    seen = []
    for d in ids:
        if d not in seen:
            seen.append(d)
    assert ranks == seen and plan.n_ranks == len(seen) and members[0].comm_rank == 0
    for d, m in zip(ids, members):
        assert m.device == d and ranks[m.comm_rank] == d
    # sends and receives pair up: between one (sender rank -> root) pair RCCL matches them in posting order, and
    # the library posts send i and receive i together, in member order, with the same element count
    buf = np.full(W * H, -1, dtype=np.int64)           # the frame (contiguous bands) or the staging buffer (striped)
    for i, m in enumerate(members):
        rows = member_rows(plan, m, i, H)
        assert len(rows) == m.rows and m.pixels == m.rows * W
        band = np.repeat(np.array(rows, dtype=np.int64), W)        # what member i sends: its rows' numbers
        dst = buf[m.recv_offset:m.recv_offset + m.pixels]
        assert dst.size == m.pixels, "receive runs past the buffer"
        assert (dst == -1).all(), "two receives overlap"
        dst[:] = band
    assert (buf >= 0).all(), "a part of the buffer receives nothing"
    if plan.striped:
        # deinterleave_kernel (luma_kernels.hip): frame row y comes from rank (y / stripe) % n, row (y / (stripe n)) * stripe + y % stripe
        s, n = plan.stripe_rows, plan.n_members
        band_rows = H // n
        frame = np.empty(W * H, dtype=np.int64)
        for y in range(H):
            chunk = y // s
            src = ((chunk % n) * band_rows + (chunk // n) * s + y % s) * W
            frame[y * W:(y + 1) * W] = buf[src:src + W]
    else:
        frame = buf
    assert np.array_equal(frame, np.repeat(np.arange(H, dtype=np.int64), W)), "frame rows out of place"
    return plan, members, ranks


@pytest.mark.parametrize("ids", [[0], list(range(8)), [0, 0, 1, 1], [3, 1], [0, 1, 0, 1, 2], [5, 5, 5], list(range(7, -1, -1))])
@pytest.mark.parametrize("W,H", [(1920, 1080), (64, 120), (7, 53), (33, 240)])
def test_bands_and_offsets_tile_the_frame(abi, ids, W, H):
    plan, members, ranks = replay(abi, ids, W, H)
    n = len(ids)
    assert plan.striped == (1 if n > 1 and H % (15 * n) == 0 else 0)
    sizes = [m.rows for m in members]
    assert sum(sizes) == H and max(sizes) - min(sizes) <= 1 and min(sizes) >= 1


def test_config_frames_on_eight_gpus(abi):
    """BASELINE configs #3-#5 over 8 devices: 15-row stripes, 135 / 270 / 540 rows and 1.04 / 4.15 / 16.6 MB per GPU."""
    for (W, H), rows in (((1920, 1080), 135), ((3840, 2160), 270), ((7680, 4320), 540)):
        plan, members, ranks = abi.gather_plan(list(range(8)), W, H)
        assert (plan.striped, plan.stripe_rows, plan.n_ranks) == (1, 15, 8) and ranks == list(range(8))
        for i, m in enumerate(members):
            assert (m.comm_rank, m.row_begin, m.rows, m.pixels, m.recv_offset) == (i, 15 * i, rows, rows * W, i * rows * W)


def test_repeated_and_unordered_devices(abi):
    plan, members, ranks = abi.gather_plan([3, 1, 3, 1], 40, 120)          # 120 = 15 * 4 * 2: striped
    assert ranks == [3, 1] and [m.comm_rank for m in members] == [0, 1, 0, 1] and plan.striped == 1
    plan, members, ranks = abi.gather_plan([2, 2, 0], 40, 50)               # ragged contiguous bands
    assert ranks == [2, 0] and [m.comm_rank for m in members] == [0, 0, 1] and plan.striped == 0
    assert [(m.row_begin, m.rows) for m in members] == [(0, 16), (16, 17), (33, 17)]
    assert [m.recv_offset for m in members] == [0, 16 * 40, 33 * 40]


def test_bad_groups_are_refused(abi):
    for ids, W, H in (([], 8, 8), ([0, -1], 8, 8), ([0, 1, 2], 8, 2), ([0], 0, 8), (list(range(65)), 8, 128)):
        with pytest.raises(abi.LumaError):
            abi.gather_plan(ids, W, H)
