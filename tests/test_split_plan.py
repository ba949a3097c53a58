"""Host logic of the ray-per-lane kernel's launch plan (luma_kernels.hip plan_split, through luma_split_plan_describe;
no GPU): lanes per pixel by band size, the fine class (the last rows of a launch run more lanes per pixel and are
dispatched last), and what the parking budget does to both."""
import pytest

W = 1920


def geometry_ok(p, rows):
    assert p.fine_row0 <= rows
    if p.fine_lanes:
        assert p.fine_row0 < rows and p.fine_row0 % 4 == 0       # whole coarse workgroups (4 rows) above the fine rows
        assert p.fine_lanes > p.lanes or p.lanes == 4 and p.fine_lanes == 8
    else:
        assert p.fine_row0 == rows
    assert p.park_row0 == (0 if p.lanes > 1 else p.fine_row0)


@pytest.mark.parametrize("rows,lanes,fine,share", [(1080, 1, 4, 0.06), (540, 1, 4, 0.12), (360, 1, 4, 0.255), (270, 1, 4, 0.37),
                                                   (216, 1, 4, 0.40), (180, 4, 8, 0.05), (135, 4, 8, 0.05)])
def test_config3_frame_and_its_bands(abi, rows, lanes, fine, share):
    # 64 frames per call: the lanes of one-lane-per-pixel waves take frames of each other's pixels, so one lane per pixel
    # serves down to the bands that fill the GPU once (~380k pixels); below that four lanes per pixel and eight in the class
    p = abi.split_plan(W, rows, 64, 5)
    assert (p.lanes, p.fine_lanes, p.refused) == (lanes, fine, 0)
    geometry_ok(p, rows)
    fine_rows = rows - p.fine_row0
    assert abs(fine_rows - share * rows) <= 6, (fine_rows, rows)
    # only frames claimed from the back are parked: (L-1)/L of them plus a margin, for the most lanes per pixel of the launch
    assert p.park_depth == {4: 52, 8: 58}[fine]
    assert p.park_bytes == p.park_depth * 12 * W * (rows - p.park_row0)


def test_lanes_per_pixel_where_no_frames_are_taken(abi):
    # more than kStealSppMax frames per call, or fewer than 8: the round-2 rule (1 lane from 1.5M pixels, 2 from 750k, else 4)
    assert [abi.split_plan(W, rows, 130, 5).lanes for rows in (1080, 540, 270)] == [1, 2, 4]
    assert [abi.split_plan(W, rows, 4, 5).lanes for rows in (1080, 540, 270)] == [1, 2, 4]
    # hit-point origins have no fine class: one lane per pixel from 600k pixels
    assert [abi.split_plan(W, rows, 64, 5, flags=abi.FLAG_ORIGIN_FROM_HIT).lanes for rows in (1080, 540, 360, 270)] == [1, 1, 1, 4]
    # more than 16 spheres (run-time sphere counts): one lane from 1.5M pixels, four below; the two-lane instances neither
    # defer hits nor take frames and lose to both
    assert [abi.split_plan(W, rows, 64, 64).lanes for rows in (1080, 540, 270)] == [1, 4, 4]


def test_no_fine_class_below_eight_frames_or_with_many_rounds(abi):
    for spp in (1, 7):
        p = abi.split_plan(W, 1080, spp, 5)
        assert (p.lanes, p.fine_lanes, p.park_bytes) == (1, 0, 0)
    # config #4's whole frame: 18 rounds of workgroups, the drain is too small a part of the launch
    p = abi.split_plan(3840, 2160, 256, 5)
    assert (p.lanes, p.fine_lanes, p.park_bytes) == (1, 0, 0)
    geometry_ok(p, 2160)


def test_other_kernels_have_one_class(abi):
    assert abi.split_plan(W, 1080, 64, 5, flags=abi.FLAG_ORIGIN_FROM_HIT).fine_lanes == 0     # general instances
    assert abi.split_plan(W, 1080, 64, 64).fine_lanes == 0                                     # more than 8 spheres
    assert abi.split_plan(W, 1080, 64, 5, flags=abi.FLAG_GRID).fine_lanes == 0
    p = abi.split_plan(W, 135, 64, 4096)                                                      # sphere-parallel kernel: persistent waves
    assert (p.lanes, p.fine_lanes, p.park_bytes) == (1, 0, 0)


def test_budget_clamps_the_fine_rows_and_refuses_whole_band_parking(abi):
    # one lane per pixel above: only the fine rows park, as many as the budget holds
    budget = 64 << 20
    p = abi.split_plan(W, 1080, 64, 5, park_budget=budget)
    assert (p.lanes, p.fine_lanes, p.refused) == (1, 4, 0)
    assert 0 < p.park_bytes <= budget and 1080 - p.fine_row0 < 60
    geometry_ok(p, 1080)
    assert abi.split_plan(W, 1080, 64, 5, park_budget=1000).fine_lanes == 0
    # four lanes per pixel want the whole band parked: refused when that does not fit, one lane per pixel instead
    p = abi.split_plan(3840, 270, 256, 5, park_budget=2 << 30)
    assert (p.lanes, p.refused) == (1, 1)
    assert p.park_bytes <= 2 << 30
    geometry_ok(p, 270)


@pytest.mark.parametrize("rows", [1, 2, 3, 5, 33, 77, 1079])
def test_awkward_row_counts(abi, rows):
    for spp in (8, 64):
        for width in (1, 31, 1920, 7680):
            p = abi.split_plan(width, rows, spp, 5)
            geometry_ok(p, rows)
            most = max(p.lanes, p.fine_lanes)
            assert p.park_depth == (min(spp - 1, -(-spp * (most - 1) // most) + max(2, spp // (4 * most))) if most > 1 else 0)
            assert p.park_bytes == (p.park_depth * 12 * width * (rows - p.park_row0) if most > 1 else 0)
