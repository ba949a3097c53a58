"""CPU tests of the uniform grid behind LUMA_FLAG_GRID (raytracer_amd/csrc/luma_grid.cpp, seen
through luma_grid_describe): whatever the scene, every sphere is either on the always-tested list
or listed in every cell its bounding box touches, so a ray that walks the cells it crosses cannot
miss a sphere it hits.  (That the frames equal the linear scan's is checked on the GPU.)"""
import numpy as np
import pytest


def scene_from(abi, pos, rad):
    n = len(rad)
    arr = (abi.Sphere * n)()
    for i in range(n):
        arr[i] = abi.Sphere(float(pos[i][0]), float(pos[i][1]), float(pos[i][2]), float(rad[i]), .5, .5, .5, 1.0, .1, .3)
    return arr


def check_invariants(abi, scene):
    n = len(scene)
    info, cell_start, items, large = abi.describe_grid(scene)
    dx, dy, dz = (int(v) for v in info.dims)
    assert 1 <= dx <= 258 and 1 <= dy <= 258 and 1 <= dz <= 258 and dx * dy * dz <= (4 << 20) + 3 * 258 * 258
    assert cell_start[0] == 0 and cell_start[-1] == len(items) and (np.diff(cell_start.astype(np.int64)) >= 0).all()
    assert info.cell > 0 and np.isfinite(info.cell)
    listed = set(int(i) for i in items)
    assert listed.isdisjoint(int(i) for i in large)
    assert listed | set(int(i) for i in large) == set(range(n))
    assert list(large) == sorted(large)
    org, h = np.array(info.org[:], np.float64), float(info.cell)
    lists = {}
    for c in range(dx * dy * dz):
        seg = items[cell_start[c]:cell_start[c + 1]]
        if len(seg):
            assert (np.diff(seg.astype(np.int64)) > 0).all(), "indices ascend inside a cell"
            lists[c] = set(int(i) for i in seg)
    for i in listed:
        s = scene[i]
        c = np.array([s.cx, s.cy, s.cz], np.float64)
        r = abs(float(s.radius))
        lo = np.floor((c - r - org) / h).astype(int)
        hi = np.floor((c + r - org) / h).astype(int)
        assert (lo >= 0).all() and (hi < [dx, dy, dz]).all(), "the box of a listed sphere lies inside the grid"
        for z in range(lo[2], hi[2] + 1):
            for y in range(lo[1], hi[1] + 1):
                for x in range(lo[0], hi[0] + 1):
                    assert i in lists.get((z * dy + y) * dx + x, ()), f"sphere {i} missing from cell {(x, y, z)}"
    return info, len(large)


def test_grid_of_the_benchmark_scenes(abi):
    info, n_large = check_invariants(abi, abi.default_scene())
    assert n_large >= 1                                   # the 100-unit ground sphere and the 10-unit one are not binned
    info, n_large = check_invariants(abi, abi.synthetic_scene(4096))
    assert n_large == 1                                   # only the ground
    assert info.n_items < 8 * 4096                        # a sphere sits in a handful of cells


@pytest.mark.parametrize("seed", range(6))
def test_grid_invariants_on_random_scenes(abi, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 17, 200, 1500]))
    scale = float(rng.choice([1e-3, 1.0, 1e4]))
    pos = rng.normal(0, 10, size=(n, 3)) * scale
    rad = np.abs(rng.normal(0.5, 0.4, size=n)) * scale
    rad[rng.integers(0, n)] *= 50                         # one giant
    if n > 2:
        pos[1] = pos[0]                                   # a coincident pair
    check_invariants(abi, scene_from(abi, pos, rad))


def test_grid_degenerate_inputs(abi):
    check_invariants(abi, (abi.Sphere * 0)())
    inf, nan = float("inf"), float("nan")
    odd = scene_from(abi, [[0, 0, -5], [inf, 0, 0], [0, nan, 0], [1, 1, 1], [2, 2, 2]], [1.0, 1.0, 1.0, inf, 0.0])
    info, cell_start, items, large = abi.describe_grid(odd)
    assert {1, 2, 3} <= set(int(i) for i in large)        # non-finite spheres are tested against every ray
    check_invariants(abi, scene_from(abi, [[3, 3, 3]] * 50, [0.0] * 50))       # zero radius, one point
    check_invariants(abi, scene_from(abi, [[i, 0, 0] for i in range(300)], [1e-9] * 300))


# ---- the direction grid around camera.pos (camera-origin rays) -------------------------------

def bin_of(res, d):
    """The kernel's bin of a direction (luma_kernels.hip scan_view), in float64."""
    ax = np.abs(d)
    m = 0 if (ax[0] >= ax[1] and ax[0] >= ax[2]) else (1 if ax[1] >= ax[2] else 2)
    a, b = [i for i in range(3) if i != m]
    face = 2 * m + (1 if d[m] < 0 else 0)
    iu = int(min(max(np.floor((d[a] / ax[m] + 1.0) * 0.5 * res), 0), res - 1))
    iv = int(min(max(np.floor((d[b] / ax[m] + 1.0) * 0.5 * res), 0), res - 1))
    return (face * res + iv) * res + iu


def check_view_grid(abi, scene, origin, rng, n_dirs=4000):
    n = len(scene)
    res, cell_start, items, large = abi.describe_view_grid(scene, origin)
    assert 16 <= res <= 256 and len(cell_start) == 6 * res * res + 1
    assert cell_start[0] == 0 and cell_start[-1] == len(items) and (np.diff(cell_start.astype(np.int64)) >= 0).all()
    assert set(int(i) for i in items).isdisjoint(int(i) for i in large)
    assert set(int(i) for i in items) | set(int(i) for i in large) <= set(range(n))
    c = np.array([[s.cx, s.cy, s.cz] for s in scene], np.float64).reshape(n, 3)
    r = np.abs(np.array([s.radius for s in scene], np.float64))
    o = np.asarray(origin, np.float64)
    always = set(int(i) for i in large)
    # directions: random ones, and ones aimed at sphere limbs and at cube-map edges and corners
    dirs = list(rng.normal(size=(n_dirs, 3)))
    for i in rng.integers(0, max(n, 1), size=min(n_dirs, 40 * max(n, 1))) if n else []:
        if np.isfinite(c[i]).all() and np.isfinite(r[i]):
            t = c[i] - o + rng.normal(size=3) * r[i] * 0.999 / np.sqrt(3)
            dirs.append(t)
    dirs += [np.array(v, np.float64) + rng.normal(size=3) * 1e-7 for v in
             ([1, 1, 0], [1, -1, 0], [-1, 0, 1], [0, 1, 1], [1, 1, 1], [-1, 1, -1], [0, -1, 1e-9], [1e-9, 0, -1])]
    for d in dirs:
        if not np.isfinite(d).all() or not np.abs(d).max() > 0:
            continue
        seg = cell_start[bin_of(res, d)]
        listed = set(int(i) for i in items[seg:cell_start[bin_of(res, d) + 1]])
        # the reference's candidate test (Renderer.cpp:244-258) in float64: discriminant > 0 and the near root ahead
        diff = o - c
        a = d @ d
        b = diff @ d
        disc = b * b - a * ((diff * diff).sum(axis=1) - r * r)
        with np.errstate(invalid="ignore"):
            hit = np.where((disc > 0) & (-b - np.sqrt(np.maximum(disc, 0)) > 0))[0]
        for i in hit:
            assert int(i) in listed or int(i) in always, f"sphere {i} hit by direction {d} is not in its bin"
    return res, len(items), len(large)


def test_view_grid_of_the_benchmark_scenes(abi):
    rng = np.random.default_rng(7)
    res, n_items, n_large = check_view_grid(abi, abi.default_scene(), [0, 0, 0], rng)
    assert res == 16                                      # five spheres: the coarsest grid (the ground fills half its bins)
    res, n_items, n_large = check_view_grid(abi, abi.synthetic_scene(4096), [0, 0, 0], rng)
    assert res == 256 and n_large == 1 and n_items < 60 * 4096   # 4 sqrt(n) bins per face edge, at most 256


@pytest.mark.parametrize("seed", range(6))
def test_view_grid_never_loses_a_hit(abi, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([1, 3, 40, 300, 1500]))
    scale = float(rng.choice([1e-3, 1.0, 1e4]))
    pos = rng.normal(0, 10, size=(n, 3)) * scale
    rad = np.abs(rng.normal(0.5, 0.4, size=n)) * scale
    rad[rng.integers(0, n)] *= 30                         # one giant
    origin = rng.normal(0, 6, size=3) * scale             # often inside the cloud, sometimes inside a sphere
    if seed % 2:
        origin = pos[0] + np.array([rad[0] * 0.5, 0, 0])  # the origin inside sphere 0
    check_view_grid(abi, scene_from(abi, pos, rad), origin.astype(np.float32), rng, n_dirs=1500)


def test_view_grid_degenerate_inputs(abi):
    rng = np.random.default_rng(3)
    check_view_grid(abi, (abi.Sphere * 0)(), [0, 0, 0], rng, n_dirs=10)
    inf, nan = float("inf"), float("nan")
    odd = scene_from(abi, [[0, 0, -5], [inf, 0, 0], [0, nan, 0], [1, 1, 1], [2, 2, 2]], [1.0, 1.0, 1.0, inf, 0.0])
    res, cell_start, items, large = abi.describe_view_grid(odd, [0, 0, 0])
    assert {1, 2, 3} <= set(int(i) for i in large)
    check_view_grid(abi, scene_from(abi, [[3, 3, 3]] * 50, [0.0] * 50), [0, 0, 0], rng, n_dirs=200)
    check_view_grid(abi, scene_from(abi, [[0, 0, -5]], [1.0]), [0, 0, -5], rng, n_dirs=200)   # the origin at a centre
