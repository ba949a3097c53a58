#!/usr/bin/env python3
"""bench.py -- Mrays/s and frame ms of the Luma render loop on MI355X.

Workload (BASELINE.json configs[2], the 1920x1080 config the metric is quoted on):
reference default 5-sphere scene, 1920x1080, 64 spp accumulated per step, 5 bounces,
pathtrace, paths=1, counter RNG seed 0x4C554D41.  One "step" = one luma_hip_render_device
call (64 frames of Renderer::render_to fused in one launch) with the accumulation planes
and the packed frame resident in HBM.  With N ranks the same frame is split into N scanline
bands (strong scaling) and the bands are gathered on rank 0 over RCCL inside the timed step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, BOUNCES, PATHS, SAMPLES, SEED = 1920, 1080, 64, 5, 1, 1, 0x4C554D41
# --workload: the default is the bench line; the others are the remaining GPU configs of
# BASELINE.json, for profiling (same JSON shape, `config.workload` says which).
WORKLOADS = {
    "config3": dict(w=1920, h=1080, spp=64, bounces=5, mode=1, flags=0, scene=0,
                    name="default 5-sphere scene 1920x1080, 64 spp/step, 5 bounces, pathtrace paths=1 (BASELINE configs[2])"),
    "config2": dict(w=1920, h=1080, spp=1, bounces=1, mode=0, flags=1, scene=0,
                    name="default 5-sphere scene 1920x1080, 1 spp, primary rays only, NOISE_OFF (BASELINE configs[1])"),
    "config4": dict(w=3840, h=2160, spp=256, bounces=5, mode=1, flags=0, scene=0,
                    name="default 5-sphere scene 3840x2160, 256 spp/step, 5 bounces, pathtrace (BASELINE configs[3])"),
    "config5": dict(w=7680, h=4320, spp=16, bounces=5, mode=1, flags=0, scene=4096,
                    name="synthetic 4096-sphere scene 7680x4320, 16 spp/step, 5 bounces, pathtrace (BASELINE configs[4])"),
}
FLOP_PER_TEST, FLOP_PER_RAY = 17.0, 64.0          # SURVEY.md 8d / BASELINE.md section 5
PEAK_FP32_VALU_TFLOPS = 157.3                     # MI355X_MICROARCH.md, "Peak FP32 (vector)"


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r01/traffic.json: FETCH_SIZE x2 for gfx950 + WRITE_SIZE), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "traffic.json")) as f:
            return json.load(f).get(workload, {}).get("corrected_bytes")
    except (OSError, ValueError):
        return None


def cpu_baseline(seconds_target=10.0, threads_all=None):
    """The CPU oracle (a port: the reference itself cannot be built) on a bounded sample of the
    same workload: 4-row groups spread evenly over the 1080 rows, all 64 spp.  The headline figure
    is ONE host core, the reference's execution model (Renderer.cpp:392); `all_cores` repeats it
    with one oracle thread per core of this process's CPU share (rows are independent)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cam, scene = ob.camera(WIDTH, HEIGHT), ob.default_scene()
    p = ob.make_params(WIDTH, HEIGHT, spp=SPP, bounces=BOUNCES, paths=PATHS, samples=SAMPLES, mode=1, seed=SEED)

    def one_group(y0):
        r = ob.Renderer(scene, cam, WIDTH, HEIGHT)   # own buffers and counters per call
        r.render(p, rows=(y0, y0 + 4))
        return r.stats.rays

    groups = list(range(0, HEIGHT, 8))             # 135 groups of 4 rows
    t0 = time.perf_counter()
    rows, rays = 0, 0
    for y0 in groups:
        rays += one_group(y0)
        rows += 4
        if time.perf_counter() - t0 > seconds_target:
            break
    dt = time.perf_counter() - t0
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": f"{rows} of {HEIGHT} rows (4-row groups every 8 rows, time-capped) x {WIDTH} px x {SPP} spp, "
                     f"{rays} rays in {dt:.2f} s",
           "msamples_per_s": rows * WIDTH * SPP / dt / 1e6}
    try:
        ncores = threads_all or min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")))
    except (AttributeError, ValueError):
        ncores = 1
    if ncores > 1:
        sample = groups[1::2][:ncores * 6]          # up to 6 four-row groups per thread, a second or so of wall time
        t0 = time.perf_counter()
        with ThreadPoolExecutor(ncores) as pool:
            rays_mt = sum(pool.map(one_group, sample))
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": rays_mt / dt / 1e6, "unit": "Mrays/s", "cores": ncores,
                            "sample": f"{4 * len(sample)} rows over {ncores} threads, {rays_mt} rays in {dt:.2f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config3")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="profiling aid: render only one band of this many bands on one GPU")
    ap.add_argument("--emulate-rank", type=int, default=0, help="which band --emulate-world renders")
    ap.add_argument("--grid", action="store_true",
                    help="LUMA_FLAG_GRID: closest-hit tests through the uniform grid (same frame, not the reference's linear scan)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: init the process group and run the gather collectives even with one rank")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks "
                         "(bands are gathered through host memory; not a measurement)")
    a = ap.parse_args()
    global WIDTH, HEIGHT, SPP, BOUNCES
    wl = WORKLOADS[a.workload]
    WIDTH, HEIGHT, SPP, BOUNCES = wl["w"], wl["h"], wl["spp"], wl["bounces"]

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL between processes needs dmabuf IPC on this driver
    import torch                                   # before abi.lib(): one shared HIP runtime
    import torch.distributed as dist
    from raytracer_amd import abi, bands

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    rehearsal = a.dist_backend == "gloo"
    if rehearsal:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.force_dist
    if a.force_dist:
        bands.FORCE_COLLECTIVE = True
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if use_dist:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    red_dev = torch.device("cpu") if rehearsal else dev

    nb, rb = (a.emulate_world or world), (a.emulate_rank if a.emulate_world else rank)
    stripe = bands.stripe_rows_for(HEIGHT, nb)    # interleaved stripes even out sky / ground rows
    y0, y1 = bands.band_rows(HEIGHT, nb, rb)
    ctx = abi.Context(local)
    ctx.set_scene(abi.synthetic_scene(wl["scene"], SEED) if wl["scene"] else abi.default_scene())
    ctx.set_camera(abi.default_camera(WIDTH, HEIGHT))
    if stripe:
        ctx.set_frame_striped(WIDTH, HEIGHT, stripe, rb, nb)
    else:
        ctx.set_frame(WIDTH, HEIGHT, y0, y1)
    params = abi.make_params(WIDTH, HEIGHT, spp=SPP, bounces=BOUNCES, paths=PATHS, samples=SAMPLES,
                             mode=wl["mode"], flags=wl["flags"] | (abi.FLAG_GRID if a.grid else 0), seed=SEED)
    stream = torch.cuda.Stream(device=dev)          # a real (non-null) stream: kernels, gather and events share it
    torch.cuda.set_stream(stream)
    pipelined = use_dist and (stripe or HEIGHT % world == 0) and not rehearsal
    pg = bands.PipelinedGather(HEIGHT, WIDTH, rank, world, stripe, dev, rows=y1 - y0) if pipelined or world == 1 else None
    band = pg.bufs[0] if pg else torch.empty((y1 - y0, WIDTH), dtype=torch.int32, device=dev)
    nstep = [0]

    def step():
        # one frame: render this rank's band, hand it to the gather.  With several ranks the
        # gather of frame i overlaps the render of frame i+1 (two band buffers).
        if pg:
            buf = pg.acquire(nstep[0])
            ctx.render_device(params, buf.data_ptr(), stream.cuda_stream)
            pg.submit(nstep[0])
            nstep[0] += 1
            return None
        ctx.render_device(params, band.data_ptr(), stream.cuda_stream)
        send = band.cpu() if rehearsal else band
        if stripe:
            return bands.gather_striped(send, HEIGHT, WIDTH, rank, world, stripe)
        return bands.gather_frame(send, HEIGHT, WIDTH, rank, world)

    def fence():
        if pg:
            pg.finish()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    ctx.render_device(params, band.data_ptr(), stream.cuda_stream, want_stats=True)   # drain the ray counter
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0

    # kernel-only leg: K back-to-back launches between two HIP events on the launch stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.render_device(params, band.data_ptr(), stream.cuda_stream, want_stats=True)   # counts timed steps
    fence()
    ev0.record(stream)
    for _ in range(a.steps):
        ctx.render_device(params, band.data_ptr(), stream.cuda_stream)
    ev1.record(stream)
    st = ctx.render_device(params, band.data_ptr(), stream.cuda_stream, want_stats=True)
    torch.cuda.synchronize()
    kernel_ms = ev0.elapsed_time(ev1) / a.steps
    rays_per_launch = st.rays / (a.steps + 1)      # a.steps timed launches + the stats launch itself
    tests_per_launch = st.sphere_tests / (a.steps + 1)

    t = torch.tensor([dt, rays_per_launch, tests_per_launch, kernel_ms], dtype=torch.float64, device=red_dev)
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, kernel_ms = float(tmax[0]), float(tmax[3])
        rays_per_step, tests_per_step = float(tsum[1]), float(tsum[2])
    else:
        rays_per_step, tests_per_step = rays_per_launch, tests_per_launch

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = rays_per_step * a.steps / dt / 1e6
        flop = FLOP_PER_TEST * tests_per_launch + FLOP_PER_RAY * rays_per_launch   # rank 0's launch
        achieved = flop / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (gloo rehearsal, not a measurement)" if rehearsal and world > 1 else ""),
            "config": {"workload": wl["name"] + f", seed 0x{SEED:08X}" + (", uniform grid (LUMA_FLAG_GRID)" if a.grid else "")
                                   + (f", band {a.emulate_rank} of {a.emulate_world} only" if a.emulate_world else ""),
                       "parallelism": ("one GPU, whole frame" if world == 1 and not a.emulate_world
                                       else f"{world} ranks, interleaved {stripe}-row stripes, RCCL gather to rank 0" if world > 1 and stripe
                                       else f"{world} scanline band(s), RCCL gather to rank 0")},
            "frame_ms": ms_per_step,
            "msamples_per_s": WIDTH * (HEIGHT // (a.emulate_world or 1)) * SPP * a.steps / dt / 1e6,
            "rays_per_step": rays_per_step,
            # the reference traces the noise-free DOF ray once per frame (Renderer.cpp:416); the kernels
            # trace it once per call and reuse it, so `value` counts (spp-1)*W*H fewer rays per step than
            # the reference's own loop (and the oracle in cpu_baseline) would for the same frames
            "reference_equivalent_mrays_per_s": (rays_per_step + (SPP - 1) * WIDTH * (HEIGHT if not a.emulate_world else (y1 - y0)))
                                                * a.steps / dt / 1e6,
            "kernel_ms": kernel_ms,
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "traffic": measured_traffic(a.workload) if world == 1 and not a.emulate_world else None,
                         "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/r01/traffic.json)",
                         "algorithmic_bytes": 28 * WIDTH * (y1 - y0),
                         "kernel": "luma::render_kernel", "flop_model": "17*sphere_tests + 64*rays (SURVEY 8d)"},
        }
        if not a.no_cpu_baseline and world == 1 and a.workload == "config3" and not a.emulate_world:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)

    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
