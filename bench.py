"""Headline benchmark: Mrays/s of the ray-tracing hot path on the reference's bunny frame.

One step = one whole frame through the hot path (primary rays, nearest hit, light-visibility rays, colour replay, 8-bit
conversion) with the scene resident in HBM and the frame left in HBM.  With N > 1 GPUs the frame is sharded by
interleaved row bands (scene replicated per GPU, no exchange while tracing) and the finished 8-bit rows are
all-gathered over RCCL each step, overlapped with the next frame.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bunny_lambert_1080p]

`--gpus N` with N > 1 starts N ranks itself (one per GPU, torch.distributed.run on 127.0.0.1) unless it is already
running as a rank of such a job.  Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E
SIMDS = 1024                 # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9
ISSUE_CYCLES = 2.0           # a SIMD issues one wave64 vector instruction per 2 cycles at best
BAND_ROWS = 8


def _workloads():
    from raytracer_amd import scenes
    return {
        # BASELINE.json configs[0] / the target of its north_star: the shipped Stanford-bunny scene, Lambert, shadow
        # rays, -spp 2 (4 samples per pixel), -spr 3, 3 bounces, 1920x1080 (reference README.md:26-40)
        "bunny_lambert_1080p": (lambda: scenes.bunny(), dict(width=1920, height=1080, hfov=60, spp=2, spr=3, max_bounce=3,
                                                              eps=1e-4, ray_eps=0.1)),
        # the same with a second directional light, as BASELINE.json words configs[0] (the shipped scene file has one
        # light; the second one is this build's: direction (0.6, 0.7, -0.2) normalised, colour 0.6)
        "bunny_lambert_2lights_1080p": (lambda: scenes.bunny(second_light=True),
                                        dict(width=1920, height=1080, hfov=60, spp=2, spr=3, max_bounce=3, eps=1e-4, ray_eps=0.1)),
        # not a BASELINE configuration: the first two with rt_options.skip_unlit - the light rays of hits whose colour the
        # light cannot change (surfaces facing away from it, no ambient colour) are left out; the frames are the same bit
        # for bit, fewer rays are traced and counted
        "bunny_lambert_1080p_skip_unlit": (lambda: scenes.bunny(), dict(width=1920, height=1080, hfov=60, spp=2, spr=3, max_bounce=3,
                                                                         eps=1e-4, ray_eps=0.1, skip_unlit=1)),
        "bunny_lambert_2lights_1080p_skip_unlit": (lambda: scenes.bunny(second_light=True),
                                                   dict(width=1920, height=1080, hfov=60, spp=2, spr=3, max_bounce=3, eps=1e-4,
                                                        ray_eps=0.1, skip_unlit=1)),
        # configs[1]: ground and dome Phong-reflective, 2 bounces (README.md:42-56)
        "bunny_phong_1080p": (lambda: scenes.bunny(phong_surroundings=True),
                              dict(width=1920, height=1080, hfov=60, spp=2, spr=3, max_bounce=2, eps=1e-3, ray_eps=0.1)),
        # configs[2]
        "mirror_room": (lambda: scenes.mirror_room(), dict(width=1827, height=1077, hfov=60, spp=2, spr=3, max_bounce=3,
                                                            eps=1e-3, ray_eps=0.01)),
        # configs[3]
        "glass_balls": (lambda: scenes.glass_balls(), dict(width=1920, height=1080, hfov=54.5, spp=2, spr=4, max_bounce=3,
                                                            max_refrac_bounce=5, eps=1e-3, ray_eps=0.1)),
        # configs[4]: ~10 M triangles (2013 bunnies, mixed materials), 3840x2160 at 64 spp; one step takes seconds
        "bunny_field_4k": (lambda: scenes.bunny_field(), dict(width=3840, height=2160, hfov=54.5, spp=8, spr=3, max_bounce=3,
                                                               max_refrac_bounce=3, eps=1e-3, ray_eps=0.1)),
    }


WORKLOADS = _workloads()
WORKLOAD_NAMES = ["bunny_lambert_1080p", "bunny_lambert_2lights_1080p", "bunny_phong_1080p", "mirror_room", "glass_balls",
                  "bunny_field_4k", "bunny_lambert_1080p_skip_unlit", "bunny_lambert_2lights_1080p_skip_unlit"]


def kernel_source_sha():
    """Identifies the kernels a set of hardware counters was collected on: hash of every source the library is built from."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """Starts args.gpus ranks of this script as child processes (this process never touches a GPU) and relays rank 0's
    JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in p.stdout.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
        else:
            print(l, file=sys.stderr)
    if p.returncode != 0 or line is None:
        print("bench.py: the %d-rank job failed (exit %d)" % (args.gpus, p.returncode), file=sys.stderr)
        return p.returncode or 1
    print(line, flush=True)
    return 0


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bunny_lambert_1080p", choices=WORKLOAD_NAMES)
    ap.add_argument("--gather", default="u8", choices=["u8", "f32"],
                    help="what the ranks exchange per frame: the 8-bit rows (what the reference's save_jpg consumes) or the float rows")
    ap.add_argument("--streams", type=int, default=0,
                    help="scene handles / HIP streams that consecutive frames alternate over (0: 1 on one GPU, 3 when the "
                         "frame is shared by several ranks - a rank's share is too small to keep its GPU busy by itself)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the per-configuration frame times of the other workloads")
    ap.add_argument("--no-roofline", action="store_true", help="skip the roofline block (profiling runs that only want the frames)")
    ap.add_argument("--scale", type=float, default=1.0,
                    help="render the workload at this fraction of its width and height (rehearsals of the heavy workloads; a "
                         "scaled run is not the benchmark and says so in config.scale)")
    ap.add_argument("--box-mode", type=int, default=0, choices=[0, 1, 2],
                    help="rt_options.exact_boxes of the timed frames: 0 the pruning traversal whose answers are checked against the "
                         "reference's leaf boxes (the default, what the benchmark is), 1 the reference's arithmetic on its own tree, "
                         "2 the pruning traversal unchecked (config.box_mode says which)")
    ap.add_argument("--verify", action="store_true",
                    help="check that the gathered frame equals the same frame rendered whole on this rank")
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, argv)

    import numpy as np
    import torch
    import torch.distributed as dist
    from raytracer_amd import api
    from raytracer_amd._abi import Options, RT_JITTER_HASH
    from raytracer_amd.distributed import FrameGather, local_to_image_rows, run_pipelined

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RT_BENCH_WATCHDOG"):          # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RT_BENCH_WATCHDOG"]), exit=True)
    # RT_BENCH_REHEARSE_GATHER=1: one rank goes through the process group, the pipelined loop and the all-gather all the
    # same (what several ranks do, on a box with one GPU - RCCL runs a one-rank collective on its own stream)
    rehearse = world == 1 and os.environ.get("RT_BENCH_REHEARSE_GATHER") == "1"
    if rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
    if world > 1 or rehearse:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RT_BENCH_BACKEND=gloo lets several ranks rehearse the sharded path on ONE GPU (RCCL refuses that)
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(os.environ.get("RT_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    api.init(-1)

    def barrier():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[dev.index])       # names the device instead of leaving RCCL to guess it
        else:
            dist.barrier()

    make, kw = WORKLOADS[args.workload]
    scene = make()
    if args.scale != 1.0:
        kw = dict(kw, width=max(8, int(kw["width"] * args.scale)), height=max(8, int(kw["height"] * args.scale)))
    opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=args.box_mode, **kw)
    W, H = opt.width, opt.height
    # A rank's share of a millisecond frame is too small to keep its GPU busy (ramp-up and tail of small grids), so
    # consecutive frames alternate over three scene handles and streams.  Scenes whose materials spawn rays get ONE: their
    # shares run for many milliseconds, and every handle keeps level queues of its own (up to a quarter of the free memory
    # each, held until the handle is destroyed).
    spawns = any(scene.materials[m.material].kind in (2, 3) for m in scene.meshes)
    n_streams = args.streams if args.streams > 0 else (1 if ((world == 1 and not rehearse) or spawns) else 3)
    tracers = [api.TracerScene(scene) for _ in range(n_streams)]      # calls on one handle serialise: a handle per stream
    tracer = tracers[0]
    stream = torch.cuda.current_stream().cuda_stream
    side_streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else None
    n_sets = max(2, n_streams)
    # this rank's share: every world-th band of BAND_ROWS rows, rendered by ONE call per frame
    n_local = len(local_to_image_rows(H, world, rank, BAND_ROWS)) if world > 1 else H
    if world > 1:
        opt.row_stride, opt.row_band, opt.row_phase = world, BAND_ROWS, rank
    u8 = args.gather == "u8"
    gather = FrameGather(H, W, world, BAND_ROWS, dev, dtype=torch.uint8 if u8 else torch.float32, flipped=u8,
                         collective_on_one_rank=rehearse) if (world > 1 or rehearse) else None
    # n_sets sets of buffers: the gather of frame k overlaps the rendering of the next frames
    rgb = [torch.empty((max(n_local, 1), W, 3), dtype=torch.float32, device=dev) for _ in range(n_sets)]
    bytes_ = [(gather.new_local() if (gather and u8) else torch.empty((max(n_local, 1), W, 3), dtype=torch.uint8, device=dev)) for _ in range(n_sets)]
    locals_ = bytes_ if u8 else ([gather.new_local() for _ in range(n_sets)] if gather else rgb)
    if gather and not u8:
        rgb = locals_
    stacks = [gather.new_stack() if gather else None for _ in range(n_sets)]

    def render(k, stats=False):
        b = k % n_sets
        cur = torch.cuda.current_stream().cuda_stream
        st = tracers[b % n_streams].render_rect(opt, 0, 0, W, n_local, out=rgb[b][:n_local], stats=stats, stream=cur)[1]
        # the reference's save path converts the frame to bytes (src/main.cpp:157-167); every rank converts its rows
        api.resolve_u8(rgb[b][:n_local], out=bytes_[b][:n_local], stream=cur)
        return st

    # rays per frame (the same every step: the seed is fixed)
    st = render(0, stats=True)
    rays = torch.tensor([float(st.rays_closest + st.rays_any)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(rays)
    rays_frame = float(rays.item())

    def run(steps):
        if not gather:
            for k in range(steps):
                if side_streams:
                    with torch.cuda.stream(side_streams[k % n_streams]):
                        render(k)
                else:
                    render(k)
            return bytes_[(steps - 1) % n_sets]
        return run_pipelined(steps, lambda k, buf: render(k), gather, locals_, stacks, side_streams)

    # Spin-up, before the W warm-up steps and outside every timed region: a GPU that has been idle runs its first ten or
    # so milliseconds of work measurably slower (kernel trace: the light-ray kernel takes 0.93 ms in the first frame after
    # an idle period, 0.80 ms from about the twelfth on).  Frames until 40 ms have passed, at most 40 of them.
    spinup = 0
    torch.cuda.synchronize()
    if world == 1:
        t_spin = time.perf_counter()
        while spinup < 40 and (spinup < 2 or time.perf_counter() - t_spin < 0.04):
            run(2)
            torch.cuda.synchronize()
            spinup += 2
    else:
        # every rank must run the same number of frames (each one is a collective): decided from the slowest rank's
        # first frame, not from this rank's clock
        slowest = torch.tensor([float(st.ms_total)], dtype=torch.float64, device=dev)
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        spinup = 36 if float(slowest.item()) < 5.0 else 2
        run(spinup)
        torch.cuda.synchronize()
    run(args.warmup)
    if world > 1:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frame = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = rays_frame / (dt / args.steps) / 1e6
    if gather:       # restore whole-frame options for the single-GPU legs below
        opt.row_stride, opt.row_band, opt.row_phase = 0, 0, 0
    if args.verify:
        whole = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        tracer.render_rect(opt, 0, 0, W, H, out=whole, stats=False, stream=stream)
        if u8 or not gather:
            want = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
            api.resolve_u8(whole, out=want, stream=stream)
            torch.cuda.synchronize()
            same = want == frame
        else:
            torch.cuda.synchronize()
            same = (whole.view(torch.int32) == frame.view(torch.int32)) | (torch.isnan(whole) & torch.isnan(frame))
        if not bool(same.all()):
            raise SystemExit("rank %d: gathered frame differs from the whole frame in %d values" % (rank, int((~same).sum())))

    result = {
        "metric": "Mrays/s (primary+secondary), whole 1080p 4spp frame through the hot path",
        "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "reference scene fixture (input_files/defaultScene/stanford_bunny.fbx baked to .rtscene), "
                                "counter-hash jitter seed 1",
        "config": {"workload": args.workload, "width": W, "height": H, "spp": opt.spp * opt.spp, "spr": opt.spr + 1,
                   "max_bounce": opt.max_bounce, "triangles": scene.n_tris, "lights": len(scene.lights),
                   "rays_per_frame": int(rays_frame),
                   "sharding": ("%d-row bands interleaved over ranks, one all_gather of the %s rows per frame, overlapped with the "
                                "next frames; consecutive frames alternate over %d streams per rank"
                                % (BAND_ROWS, "8-bit" if u8 else "float", n_streams)) if world > 1 else "none",
                   "streams": n_streams, "gather_rehearsal": bool(rehearse), "spinup_frames": spinup, "scale": args.scale, "box_mode": args.box_mode,
                   "verified": bool(args.verify)},
        "ms_per_frame": round(ms_per_step, 4),
    }

    if rank == 0:
        prof_all, stale = load_profile()
        if not args.no_roofline:
            result["roofline"] = roofline(tracer, opt, W, H, dev, stream, args.workload, prof_all.get(args.workload, {}), stale)
        if world == 1 and not args.no_workloads:
            rows = other_workloads(args.workload, dev, prof_all)
            head = result.get("roofline")
            if head:
                rows[args.workload] = dict(workload_row(args.workload, prof_all.get(args.workload, {}), head["ms_extend"], head["ms_shadow"],
                                                        head["ms_replay"], head["_counted"]), ms=round(ms_per_step, 3), mrays=round(value, 1))
            # one flat dictionary per quantity, keyed by workload (every BASELINE configuration on this GPU, this build)
            for key in ("ms", "mrays", "dominant_kernel", "bound", "frac", "lane_util", "hbm_frac", "hbm_gbps", "nodes_per_ray", "tris_per_ray",
                        "hbm_over_tree_bytes"):
                result["workloads_" + key] = {w: r.get(key) for w, r in rows.items()}
        if result.get("roofline"):
            result["roofline"].pop("_counted", None)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene, opt, ms_per_step)
        print(json.dumps(result), flush=True)
    if world > 1 or rehearse:
        barrier()
        dist.destroy_process_group()
    return 0


def load_profile():
    """profiles/traffic.json: per-frame hardware-counter totals per workload and kernel (tools/make_traffic_json.py).
    Returns ({workload: {kernel: {...}}}, stale); the counters are only used when they were collected on exactly the
    kernel sources of this build (kernel_src_sha) - otherwise ({}, True)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return {}, None
    try:
        allp = json.load(open(tpath))
    except Exception:
        return {}, None
    if allp.get("kernel_src_sha") != kernel_source_sha():
        return {}, True
    return allp.get("workloads", {}), False


PEAK_GINST = SIMDS * CLOCK_HZ / ISSUE_CYCLES / 1e9            # G wave-instructions per second: 1228.8
NODE_BYTES = 48              # what a lane of the pruning traversal fetches of a 64-byte node (references packed into the planes)
TRI_BYTES = 48


def workload_row(name, prof, ms_ext, ms_sha, ms_replay, cs):
    """The dominant kernel of one workload's frame against the roofs.  ms_*: live per-frame kernel times (HIP events the
    library records on the call's stream, summed over the levels and bands of the frame); cs: statistics of a counted
    frame (nodes fetched, triangles tested); prof: the workload's entry of profiles/traffic.json ({} when stale)."""
    if ms_sha >= ms_ext:
        key, ms, rays, nn, nt = "k_shadow", ms_sha, cs.rays_any, cs.nodes_shadow, cs.tris_shadow
    else:
        key, ms, rays, nn, nt = "k_extend", ms_ext, cs.rays_closest, cs.nodes_extend, cs.tris_extend
    if ms_replay > ms:
        key, ms, rays, nn, nt = "replay", ms_replay, 0, 0, 0
    k = prof.get(key) or prof.get("k_replay") or prof.get("k_shade") or {} if key == "replay" else prof.get(key, {})
    row = {"dominant_kernel": key, "ms_dominant": round(ms, 4)}
    insts, hbm = k.get("valu_insts"), k.get("hbm_bytes")
    frac = insts / (ms * 1e-3) / 1e9 / PEAK_GINST if insts and ms > 0 else None
    hbm_gbps = hbm / (ms * 1e-3) / 1e9 if hbm and ms > 0 else None
    hbm_frac = hbm_gbps / HBM_PEAK_GBS if hbm_gbps else None
    row["frac"] = round(frac, 4) if frac else None
    row["hbm_frac"] = round(hbm_frac, 4) if hbm_frac else None
    row["hbm_gbps"] = round(hbm_gbps, 1) if hbm_gbps else None
    row["lane_util"] = k.get("lane_utilisation")
    # which roof is nearer: instruction issue of the vector ALUs, or HBM
    row["bound"] = None if frac is None or hbm_frac is None else ("hbm" if hbm_frac > frac else "valu_issue")
    row["nodes_per_ray"] = round(nn / rays, 2) if rays else None
    row["tris_per_ray"] = round(nt / rays, 2) if rays else None
    tree_bytes = NODE_BYTES * nn + TRI_BYTES * nt
    # measured HBM bytes over the bytes of tree the kernel asks for: far below 1 = the tree is served by the caches
    row["hbm_over_tree_bytes"] = round(hbm / tree_bytes, 4) if (hbm and tree_bytes) else None
    return row


def other_workloads(skip, dev, prof_all):
    """Frame time, ray rate and roofline figures of every other benchmark configuration on this GPU (outside the timed
    headline): a warm-up frame with statistics, the mean of a few frames (one for the multi-second 4K field), and one
    counted frame (nodes fetched / triangles tested)."""
    import torch
    from raytracer_amd import api
    from raytracer_amd._abi import Options, RT_JITTER_HASH
    rows = {}
    stream = torch.cuda.current_stream().cuda_stream
    for name in WORKLOAD_NAMES:
        if name == skip:
            continue
        make, kw = WORKLOADS[name]
        tracer = api.TracerScene(make())
        opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=0, **kw)
        out = torch.empty((opt.height, opt.width, 3), dtype=torch.float32, device=dev)
        _, st = tracer.render_rect(opt, 0, 0, opt.width, opt.height, out=out, stats=True, stream=stream)
        frames = 1 if st.ms_total > 500 else (3 if st.ms_total > 20 else 20)
        if st.ms_total <= 20:            # millisecond frames: bring the GPU back to speed first (see the spin-up in main)
            for _ in range(30):
                tracer.render_rect(opt, 0, 0, opt.width, opt.height, out=out, stats=False, stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            tracer.render_rect(opt, 0, 0, opt.width, opt.height, out=out, stats=False, stream=stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / frames
        _, st = tracer.render_rect(opt, 0, 0, opt.width, opt.height, out=out, stats=True, stream=stream)   # per-kernel times, GPU at speed
        copt = Options.from_buffer_copy(opt)
        copt.count_traversal = 1
        _, cs = tracer.render_rect(copt, 0, 0, opt.width, opt.height, out=out, stats=True, stream=stream)
        row = workload_row(name, prof_all.get(name, {}), st.ms_extend, st.ms_shadow, st.ms_shade, cs)
        row["ms"] = round(dt * 1e3, 3)
        row["mrays"] = round((st.rays_closest + st.rays_any) / dt / 1e6, 1)
        rows[name] = row
        tracer.close()
        del out
        torch.cuda.empty_cache()
    return rows


def roofline(tracer, opt, W, H, dev, stream, workload, prof, stale):
    """The dominant kernel of the timed workload's frame against the roof that binds it.

    The bunny scene's tree (about 1 MB) stays in the L1/L2 caches, so its traversal kernels do not run against HBM:
    hardware counters show a few per cent of the HBM peak.  What they run against is the rate at which a SIMD can issue
    vector instructions: `achieved` is wave-level vector instructions per second of the kernel (SQ_INSTS_VALU per frame
    from the committed counter passes over its time per frame), `peak` one instruction per SIMD per 2 cycles.  Counter
    values are only used when they were collected on exactly the kernel sources of this build (kernel_src_sha), else
    those fields are null.  The algorithmic-bytes figure (node and triangle fetches, which the caches serve) is kept for
    information.  All per-frame: a frame of the heavier workloads is many launches of each kernel."""
    import numpy as np
    import torch
    from raytracer_amd._abi import Options
    out = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    # work: one counted frame (kernels that also tally fetched nodes / tested triangles; not timed)
    copt = Options.from_buffer_copy(opt)
    copt.count_traversal = 1
    _, cs = tracer.render_rect(copt, 0, 0, W, H, out=out, stats=True, stream=stream)
    # time: a few plain frames with per-kernel HIP events recorded by the library on the stream the kernels run on
    # (after some frames that bring the GPU back to speed: the counted frame above is mostly host-side waiting)
    n_spin = 0 if cs.ms_total > 20 else 30
    for _ in range(n_spin):
        tracer.render_rect(opt, 0, 0, W, H, out=out, stats=False, stream=stream)
    ext, sha, shade = [], [], []
    for _ in range(9 if cs.ms_total < 200 else 3):
        _, st = tracer.render_rect(opt, 0, 0, W, H, out=out, stats=True, stream=stream)
        ext.append(st.ms_extend); sha.append(st.ms_shadow); shade.append(st.ms_shade)
    ms_ext, ms_sha, ms_shade = float(np.median(ext)), float(np.median(sha)), float(np.median(shade))
    # algorithmic bytes: NODE_BYTES per inner node fetched (a lane of the pruning traversal reads 48 of a node's 64
    # bytes) + 48 B per triangle tested, plus the queue records: a nearest-hit ray writes normal+t (16 B), triangle (4 B)
    # and, on a hit, the hit point (16 B) after reading 36 B of vertex normals; one (node, light) item of spr+1 light rays
    # reads triangle + hit point (20 B) and writes its count (4 B)
    b_ext = cs.rays_closest * (20 + 36 + 16) + NODE_BYTES * cs.nodes_extend + TRI_BYTES * cs.tris_extend
    b_sha = cs.rays_any // (opt.spr + 1) * 24 + NODE_BYTES * cs.nodes_shadow + TRI_BYTES * cs.tris_shadow
    row = workload_row(workload, prof, ms_ext, ms_sha, ms_shade, cs)
    key = row["dominant_kernel"]
    name = {"k_shadow": "k_shadow (any-hit light rays)", "k_extend": "k_extend (nearest hit)", "replay": "replay (colour walk)"}[key]
    if key == "k_shadow":
        b, ms, rays, nn, nt = b_sha, ms_sha, cs.rays_any, cs.nodes_shadow, cs.tris_shadow
    elif key == "k_extend":
        b, ms, rays, nn, nt = b_ext, ms_ext, cs.rays_closest, cs.nodes_extend, cs.tris_extend
    else:
        b, ms, rays, nn, nt = 0, ms_shade, 0, 0, 0
    k = prof.get(key, {}) if key != "replay" else (prof.get("k_replay") or prof.get("k_shade") or {})
    insts, traffic, util, salu = k.get("valu_insts"), k.get("hbm_bytes"), k.get("lane_utilisation"), k.get("salu_insts")
    bound = row["bound"] or "valu_issue"
    if bound == "hbm":
        achieved, peak, unit = row["hbm_gbps"], HBM_PEAK_GBS, "GB/s"
    else:
        achieved, peak, unit = (insts / (ms * 1e-3) / 1e9 if insts else None), PEAK_GINST, "Gwave-inst/s"
    res = {"bound": bound, "kernel": name,
           "achieved": round(achieved, 1) if achieved else None, "peak": round(peak, 1), "unit": unit,
           "frac": round(achieved / peak, 4) if achieved else None,
           "traffic": traffic,
           "launch_ms": round(ms / max(k.get("launches_per_frame") or 1, 1), 4), "kernel_ms_per_frame": round(ms, 4),
           "kernel_ms_per_frame_rocprof": k.get("rocprof_ms_per_frame"), "launches_per_frame": k.get("launches_per_frame"),
           "lane_utilisation": util,
           "useful_lane_frac": round(row["frac"] * util, 4) if (row["frac"] and util) else None,
           "valu_issue_frac": row["frac"], "hbm_frac_measured": row["hbm_frac"],
           "valu_insts_per_frame": insts, "salu_insts_per_frame": salu, "vmem_rd_insts_per_frame": k.get("vmem_rd_insts"),
           "l1_data_path_busy_frac": k.get("td_busy_frac"), "l1_address_path_busy_frac": k.get("ta_busy_frac"),
           "l1_accesses_per_frame": k.get("l1_accesses"), "l1_hit_rate": k.get("l1_hit_rate_est"), "l2_hit_rate": k.get("l2_hit_rate"),
           "hbm_traffic_over_algorithmic_queue_bytes": round(traffic / max(b - NODE_BYTES * nn - TRI_BYTES * nt, 1), 3) if (traffic and b) else None,
           "hbm_traffic_over_tree_bytes": row["hbm_over_tree_bytes"],
           "algorithmic_bytes_per_frame": int(b), "algorithmic_GBps_cache_served": round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
           "rays_per_frame": int(rays), "nodes_per_ray": row["nodes_per_ray"], "tris_per_ray": row["tris_per_ray"],
           "ms_extend": round(ms_ext, 4), "ms_shadow": round(ms_sha, 4), "ms_replay": round(ms_shade, 4),
           "kernel_src_sha": kernel_source_sha(), "counters_stale": stale, "_counted": cs,
           "note": "valu_issue: SQ_INSTS_VALU of the kernel per frame (profiles/traffic.json, collected on the sources kernel_src_sha names) "
                   "over its time per frame (HIP events on the call's stream), against 1 wave instruction per SIMD per 2 cycles on 1024 SIMDs "
                   "at 2.4 GHz; hbm: measured HBM bytes per frame of the kernel over the same time against 8 TB/s; bound = the nearer of the "
                   "two roofs; l1_*_busy_frac = TD_TD_BUSY / TA_TA_BUSY over SQ_BUSY_CU_CYCLES (the CUs' vector-memory return and address "
                   "paths); traffic = HBM bytes per frame from separate FETCH_SIZE / WRITE_SIZE passes (reads doubled, gfx950); where the "
                   "tree is cache resident the algorithmic node/triangle bytes are not HBM traffic (hbm_traffic_over_tree_bytes)"}
    return res


def cpu_baseline(scene, opt, gpu_ms):
    """The CPU oracle (a port of the reference's loop, with the reference's fork/join tile partitioning) timed on this
    box's host cores on the same frame."""
    import numpy as np
    from oracle.oracle import OracleScene
    ncpu = os.cpu_count() or 1
    # the GPU box shares its host between leases and exports the share it gives one GPU in OMP_NUM_THREADS; the
    # reference sizes its thread pool from hardware_concurrency (src/main.cpp:109), here from the cores actually ours
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = ncpu
    cores = max(1, min(x for x in (ncpu, affinity, share or ncpu)))
    o = OracleScene(scene)
    runs = []
    for _ in range(3):                       # the fastest of three counts (shared host cores are noisy)
        _, st, secs = o.render_frame_mt(opt, cores)
        runs.append(secs)
    secs = min(runs)
    rays = st.rays_closest + st.rays_any
    return {"value": round(rays / secs / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "the whole %dx%d frame, best of 3 runs (%s s): %d get_hit calls (the reference traces every render ray "
                      "twice), reference renderFrame partitioning (100-pixel blocks, %d threads per block); host reports %d CPUs, "
                      "%d usable by this process, share of this GPU lease %s"
                      % (opt.width, opt.height, ", ".join("%.2f" % r for r in runs), rays, int(np.ceil(np.sqrt(cores))) ** 2, ncpu,
                         affinity, share or "unknown"),
            "host_cpu_count": ncpu,
            "frame_seconds": round(secs, 3), "gpu_frame_ms": round(gpu_ms, 3),
            "frame_time_ratio": round(secs * 1e3 / gpu_ms, 1)}


if __name__ == "__main__":
    sys.exit(main())
