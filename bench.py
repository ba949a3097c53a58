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
import socket
import subprocess
import sys
import tempfile
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
# "confirm by FMA microbenchmark on the box and use the measured figure if lower".
# tools/ubench_valu.hip (v_pk_fma_f32, 8 waves/SIMD, every CU) on MI355X: profiles/r01/ubench_valu.txt.
PEAK_FP32_VALU_MEASURED_TFLOPS = 121.0
PEAK_MEASURED_SOURCE = "tools/ubench_valu.hip, profiles/r01/ubench_valu.txt (commit 8fe0062)"


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x2 for
    gfx950 + WRITE_SIZE), newest round first, with the commit the passes were taken at: a
    committed measurement, not a counter of this run.  (bytes, source) or (None, None)."""
    for rnd in ("r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "traffic.json")) as f:
                doc = json.load(f)
        except (OSError, ValueError):
            continue
        entry = doc.get(workload, {})
        if entry.get("corrected_bytes") is not None:
            src = f"profiles/{rnd}/traffic.json @ {doc.get('commit', 'round-1 kernels (30d299c)')}"
            if entry.get("note"):
                src += "; " + entry["note"]
            return entry["corrected_bytes"], src
    return None, None


def spawn_ranks(n, argv, deadline_s=420.0):
    """`bench.py --gpus N` from a plain shell: start the N rank processes ourselves.  This parent
    has made no GPU call and imported no torch; each rank is a fresh child (never an exec of a
    process that touched the GPU).  Rank 0's child prints the JSON line; the exit code is non-zero
    if any rank fails, and the others are then stopped by their own PIDs.  Ranks still running
    `deadline_s` after the start (stuck in a rendezvous or a collective: the usual first-contact
    failure) are stopped the same way, named on stderr, and the exit code is 124."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    t0 = time.monotonic()
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:                     # a rank died: the rest would wait in a collective forever
                    q.terminate()
        if live and rc == 0 and time.monotonic() - t0 > deadline_s:
            alive = sorted(procs.index(q) for q in live)
            print(f"bench.py: ranks {alive} of {n} still running after {deadline_s:.0f} s "
                  f"(exited cleanly: {sorted(set(range(n)) - set(alive))}); stopping them", file=sys.stderr, flush=True)
            rc = 124
            for q in live:
                q.terminate()
            t_kill = time.monotonic() + 10.0       # a rank stuck inside a GPU call may ignore SIGTERM
            while any(q.poll() is None for q in live) and time.monotonic() < t_kill:
                time.sleep(0.2)
            for q in live:
                if q.poll() is None:
                    q.kill()
            break
    return rc if rc >= 0 else 1


def single_process(a, wl):
    """`--single-process`: the product's own multi-GPU path (north_star: C++ host -> C-ABI -> RCCL).  ONE process,
    one luma_group over devices 0..N-1; a step is one luma_hip_group_render_device call: every device renders its
    band, one grouped ncclSend/ncclRecv exchange over the library's own communicator (ncclCommInitAll) puts the
    frame together in the root GPU's memory (de-interleaved when striped), and the call returns when the frame is
    complete -- nothing is hidden behind the next frame.  No torch in this process.  Prints one JSON line."""
    from raytracer_amd import abi
    n = a.gpus
    seen = abi.device_count()
    ids = list(range(n)) if seen >= n or not a.repeat_devices else [i % seen for i in range(n)]
    if seen < n and not a.repeat_devices:
        raise SystemExit(f"bench.py --single-process: --gpus {n} but {seen} device(s) visible (rehearsal on fewer: --repeat-devices)")
    plan, members, ranks = abi.gather_plan(ids, WIDTH, HEIGHT)
    params = abi.make_params(WIDTH, HEIGHT, spp=SPP, bounces=BOUNCES, paths=PATHS, samples=SAMPLES,
                             mode=wl["mode"], flags=wl["flags"] | (abi.FLAG_GRID if a.grid else 0), seed=SEED)
    with abi.Group(ids) as g:
        g.set_scene(abi.synthetic_scene(wl["scene"], SEED) if wl["scene"] else abi.default_scene())
        g.set_camera(abi.default_camera(WIDTH, HEIGHT))
        g.set_frame(WIDTH, HEIGHT)
        g.reserve(SPP)
        g.frame_device()                            # the group's frame on the root: allocated before the timed region
        for _ in range(max(a.warmup, 1)):           # the first call also builds the communicator
            g.render_device(params, want_stats=False)
        g.render_device(params, want_stats=True)    # drains the ray counters
        t0 = time.perf_counter()
        for _ in range(a.steps):
            g.render_device(params, want_stats=False)     # synchronous: frame complete on return
        dt = time.perf_counter() - t0
        st = g.render_device(params, want_stats=True)
        rays_per_step = st.rays / (a.steps + 1)
        tests_per_step = st.sphere_tests / (a.steps + 1)
        n_frame = max(1, min(a.steps, 20))
        t1 = time.perf_counter()
        for _ in range(n_frame):
            g.render_device(params, want_stats=False)
            frame = g.read_frame()                  # + one D2H copy of the assembled frame
        frame_ms = (time.perf_counter() - t1) / n_frame * 1e3
        checksum = int(frame.astype("uint64").sum())
    ms = dt / a.steps * 1e3
    flop = FLOP_PER_TEST * tests_per_step + FLOP_PER_RAY * rays_per_step
    achieved = flop / (st.kernel_ms * 1e-3) / 1e12 / len(ids)         # per GPU, over the slowest device's kernel
    out = {
        "metric": "Mrays/s", "value": rays_per_step * a.steps / dt / 1e6, "unit": "Mrays/s", "n_gpus": n, "ranks": plan.n_ranks,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["name"] + f", seed 0x{SEED:08X}" + (", uniform grid (LUMA_FLAG_GRID)" if a.grid else ""),
                   "parallelism": (f"single process, luma_hip_group_render_device over devices {ids}: {plan.n_ranks} RCCL rank(s) "
                                   f"(ncclCommInitAll in the library), "
                                   + (f"interleaved {plan.stripe_rows}-row stripes, grouped ncclSend/ncclRecv into a staging buffer + de-interleave on the root"
                                      if plan.striped else "contiguous bands, grouped ncclSend/ncclRecv straight into the frame on the root")),
                   "step": "render of every band + gather, synchronous: ms_per_step is a frame latency with the frame left in the root GPU's HBM"},
        "frame_ms": frame_ms,
        "frame_ms_definition": "wall of luma_hip_group_render_device + luma_hip_group_read_frame (one D2H copy of the assembled frame)",
        "kernel_ms": st.kernel_ms, "gather_ms": st.gather_ms, "rays_per_step": rays_per_step, "frame_checksum": checksum,
        "roofline": {"bound": "valu_fp32", "scope": "per GPU, slowest device's kernel", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": None,
                     "kernel": "luma::render_kernel", "flop_model": "17*sphere_tests + 64*rays (SURVEY 8d)"},
    }
    print(json.dumps(out), flush=True)


def single_process_leg(a, timeout_s=150.0):
    """Rank 0, after its own measurement: the same workload through the library's single-process path, in a fresh child
    (no torch, no exec of this process), bounded in time.  Its JSON line, or what went wrong."""
    argv = [sys.executable, os.path.abspath(__file__), "--gpus", str(a.gpus), "--single-process", "--workload", a.workload,
            "--steps", str(min(a.steps, 30)), "--warmup", str(min(a.warmup, 3))] + (["--grid"] if a.grid else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                               "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    try:
        r = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {timeout_s:.0f} s (stopped)"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"exit code {r.returncode}", "stderr_tail": r.stderr[-600:]}
    d = json.loads(lines[-1])
    keep = ("value", "unit", "ms_per_step", "frame_ms", "kernel_ms", "gather_ms", "ranks", "steps", "frame_checksum")
    return dict({k: d[k] for k in keep}, parallelism=d["config"]["parallelism"], step=d["config"]["step"],
                roofline_frac=d["roofline"]["frac"])


def cpu_baseline(seconds_target=10.0, threads_all=None):
    """The CPU oracle (a port: the reference itself cannot be built) on a bounded sample of the
    same workload: 4-row groups spread evenly over the 1080 rows, all 64 spp.  The headline figure
    is ONE host core, the reference's execution model (Renderer.cpp:392); `all_cores` repeats it
    with one oracle thread per core of this process's CPU share (rows are independent)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    # the baseline leg is built for THIS host (-O3 -march=native); the
    # in-tree library (oracle/Makefile, -O2 -march=x86-64-v3, portable) is the fallback.  The
    # arithmetic contract (-ffp-contract=off, explicit fmaf) is the same, so are the frames.
    flags = "-O2 -std=c99 -ffp-contract=off -fno-fast-math -march=x86-64-v3 (oracle/Makefile, prebuilt)"
    tmp = tempfile.TemporaryDirectory(prefix="luma_oracle_")   # removed when this function returns (the library stays mapped)
    try:
        native = os.path.join(tmp.name, "libluma_oracle_native.so")
        opt = ["-O3", "-std=c99", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-march=native"]
        subprocess.check_call(["gcc"] + opt + ["-shared", "-o", native, os.path.join(ob.ORACLE_DIR, "luma_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        ob.ORACLE_SO = native
        flags = "gcc " + " ".join(opt) + " (built on this host for this leg)"
    except (OSError, subprocess.SubprocessError):
        pass
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
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "flags": flags,
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
    ap.add_argument("--no-frame-leg", action="store_true",
                    help="skip the host-delivery leg (frame_ms = null): profiler runs then see only the device-target launches the roofline is quoted on")
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
    ap.add_argument("--single-process", action="store_true",
                    help="one process, one luma_group over devices 0..N-1: the timed step is luma_hip_group_render_device "
                         "(kernels + the library's own RCCL gather on the root); no torch, no rank processes")
    ap.add_argument("--repeat-devices", action="store_true",
                    help="with --single-process on a box with fewer GPUs than --gpus: reuse device ids (rehearsal, not a measurement)")
    ap.add_argument("--no-single-process-leg", action="store_true",
                    help="skip the `single_process` object of the line (rank 0 runs --single-process in a child after its own measurement)")
    ap.add_argument("--rank-deadline", type=float, default=420.0,
                    help="seconds after which the self-started ranks of --gpus N are stopped and named (exit 124)")
    ap.add_argument("--selftest-hang-rank", type=int, default=-1, help="with --selftest-spawn: this rank never finishes")
    ap.add_argument("--selftest-spawn", action="store_true",
                    help="no GPU work: the ranks rendezvous (gloo), all-reduce their rank numbers, rank 0 prints "
                         "the sum; checks the launcher path of --gpus N on a box without GPUs")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help="with --selftest-spawn: this rank exits 3 instead")
    a = ap.parse_args()
    global WIDTH, HEIGHT, SPP, BOUNCES
    wl = WORKLOADS[a.workload]
    WIDTH, HEIGHT, SPP, BOUNCES = wl["w"], wl["h"], wl["spp"], wl["bounces"]
    if a.single_process:
        if int(os.environ.get("RANK", "0")) != 0:
            return                                                 # under a rank launcher: one process does it all
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return single_process(a, wl)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:], a.rank_deadline))   # before any torch import or GPU call

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL between processes needs dmabuf IPC on this driver
    import torch                                   # before abi.lib(): one shared HIP runtime
    import torch.distributed as dist
    from raytracer_amd import abi, bands

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    if a.selftest_spawn:
        if rank == a.selftest_fail_rank:
            raise SystemExit(3)
        if rank == a.selftest_hang_rank:
            time.sleep(3600)
        dist.init_process_group("gloo")
        t = torch.tensor([rank], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"spawn_selftest": {"world": world, "rank_sum": int(t[0])}}), flush=True)
        dist.destroy_process_group()
        return
    if torch.cuda.device_count() == 0:
        raise SystemExit("bench.py: no GPU visible (the render loop has no CPU fallback)")
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
    ctx.reserve(SPP)                               # set-up: no allocation inside the timed render calls
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

    # frame leg (SURVEY 8d "frame ms", what the reference itself times: one whole render_to,
    # main.cpp:261-276): wall clock of one call that ends with the finished frame in HOST memory.
    # One GPU: luma_hip_render into a page-locked frame (the kernel writes it over PCIe as it goes).  N ranks: render, gather of
    # the bands on rank 0 (not pipelined: the frame is waited for), D2H of the assembled frame.
    n_frame = max(1, min(a.steps, 20))
    if a.no_frame_leg:
        frame_ms = 0.0
    elif world == 1 and not a.force_dist:
        with abi.HostFrame(y1 - y0, WIDTH) as hf:
            ctx.render(params, want_stats=False, out=hf.array)    # first touch of the frame
            t1 = time.perf_counter()
            for _ in range(n_frame):
                ctx.render(params, want_stats=False, out=hf.array)
            frame_ms = (time.perf_counter() - t1) / n_frame * 1e3
    else:
        host = torch.empty((HEIGHT, WIDTH), dtype=torch.int32).pin_memory() if rank == 0 and not rehearsal else None

        def whole_frame():
            ctx.render_device(params, band.data_ptr(), stream.cuda_stream)
            send = band.cpu() if rehearsal else band
            if a.emulate_world:      # one band of an emulated split: the gather moves just that band
                full = bands.gather_frame(send, (y1 - y0) * world, WIDTH, rank, world)
            else:
                full = (bands.gather_striped(send, HEIGHT, WIDTH, rank, world, stripe) if stripe
                        else bands.gather_frame(send, HEIGHT, WIDTH, rank, world))
            if host is not None and full is not None and full.numel() == host.numel():
                host.copy_(full.reshape(HEIGHT, WIDTH), non_blocking=True)
            torch.cuda.synchronize()

        whole_frame()
        fence()
        t1 = time.perf_counter()
        for _ in range(n_frame):
            whole_frame()
        fence()
        frame_ms = (time.perf_counter() - t1) / n_frame * 1e3

    t = torch.tensor([dt, rays_per_launch, tests_per_launch, kernel_ms, frame_ms], dtype=torch.float64, device=red_dev)
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, kernel_ms, frame_ms = float(tmax[0]), float(tmax[3]), float(tmax[4])
        rays_per_step, tests_per_step = float(tsum[1]), float(tsum[2])
    else:
        rays_per_step, tests_per_step = rays_per_launch, tests_per_launch

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = rays_per_step * a.steps / dt / 1e6
        flop = FLOP_PER_TEST * tests_per_launch + FLOP_PER_RAY * rays_per_launch   # rank 0's launch
        achieved = flop / (kernel_ms * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic(a.workload)
        out = {
            "metric": "Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (gloo rehearsal, not a measurement)" if rehearsal and world > 1 else ""),
            "config": {"workload": wl["name"] + f", seed 0x{SEED:08X}" + (", uniform grid (LUMA_FLAG_GRID)" if a.grid else "")
                                   + (f", band {a.emulate_rank} of {a.emulate_world} only" if a.emulate_world else ""),
                       "parallelism": ("one GPU, whole frame" if world == 1 and not a.emulate_world
                                       else f"{world} rank processes, interleaved {stripe}-row stripes, torch.distributed (RCCL) gather to rank 0" if world > 1 and stripe
                                       else f"{world} scanline band(s), torch.distributed (RCCL) gather to rank 0"),
                       "step": ("one luma_hip_render_device call, frame left in HBM" if world == 1 and not a.force_dist
                                else "render of this rank's band + hand-over to the gather; the gather of frame i overlaps the render of frame i+1 "
                                     "(two band buffers), so ms_per_step is a THROUGHPUT figure -- frame_ms is the latency of one frame")},
            # ms_per_step (and value) time the step with the frame left in HBM, as the bench contract
            # asks; frame_ms is the host-target frame: kernel + (gather) + D2H into page-locked memory
            "frame_ms": None if a.no_frame_leg else frame_ms,
            "frame_ms_definition": ("wall of luma_hip_render into a page-locked host frame (written by the kernel over PCIe, no copy after it)" if world == 1 and not a.force_dist
                                    else "wall of render + gather of the bands on rank 0 + D2H of the frame, not pipelined, max over ranks"),
            "msamples_per_s": WIDTH * (HEIGHT // (a.emulate_world or 1)) * SPP * a.steps / dt / 1e6,
            "rays_per_step": rays_per_step,
            # the reference traces the noise-free DOF ray once per frame (Renderer.cpp:416); the kernels
            # trace it once per call and reuse it, so `value` counts (spp-1)*W*H fewer rays per step than
            # the reference's own loop (and the oracle in cpu_baseline) would for the same frames
            "reference_equivalent_mrays_per_s": (rays_per_step + (SPP - 1) * WIDTH * (HEIGHT if not a.emulate_world else (y1 - y0)))
                                                * a.steps / dt / 1e6,
            "kernel_ms": kernel_ms,
            "roofline": {"bound": "valu_fp32", "scope": ("one GPU" if world == 1 else "per GPU: rank 0's launch over the slowest rank's kernel time"),
                         "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "peak_measured": PEAK_FP32_VALU_MEASURED_TFLOPS, "peak_measured_source": PEAK_MEASURED_SOURCE,
                         "frac_of_measured": achieved / PEAK_FP32_VALU_MEASURED_TFLOPS,
                         "traffic": traffic if world == 1 and not a.emulate_world else None,
                         "traffic_unit": "bytes/launch, rocprofv3 PMC passes committed at " + str(traffic_src),
                         "algorithmic_bytes": 28 * WIDTH * (y1 - y0),
                         "kernel": "luma::render_kernel", "flop_model": "17*sphere_tests + 64*rays (SURVEY 8d)"},
        }
        if not a.no_cpu_baseline and world == 1 and a.workload == "config3" and not a.emulate_world:
            out["cpu_baseline"] = cpu_baseline()

    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        if not a.no_single_process_leg and not a.emulate_world and not rehearsal and not a.force_dist:
            # the product's own multi-GPU path beside the rank processes' figure: same frame, same devices, through
            # luma_hip_group_render_device; the other ranks have let go of their GPUs' contexts by now
            out["single_process"] = single_process_leg(a)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
