#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: rays/s on lego 800x800, 64 coarse + 128 fine samples per ray, fp32.

A "step" is one full frame through the hot path (ray gen -> stratified -> coarse MLP -> resample -> fine MLP ->
composite [-> RCCL all-gather of the row bands when N > 1]) with weights and all intermediates resident in HBM.
    python bench.py --gpus N --steps K --warmup W          (N > 1: this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Both launch forms run the same rank code: one process per GPU, torch.distributed over RCCL ("nccl"), row bands + ONE
all-gather of the framebuffer.  Rank 0 prints ONE JSON line.  `roofline` is the fine-network MLP kernel (the dominant
launch) measured with HIP events on the render stream inside the timed region; `cpu_baseline` is the CPU oracle (a
port of the reference's algorithm) timed on this host's cores in the REFERENCE'S OWN LOOP ORDER (forward_fallback,
src/network.rs:124-147) on one 8x8 block per thread at the reference CLI's 256x256 geometry; the oracle's cache-blocked
nest (bit-identical results) is reported beside it as `blocked_rays_per_s`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA ~2.5 PF dense"


PROFILE_ROUND = "r04"  # the round whose committed rocprofv3 summaries (profiles/r03*) describe the kernels this bench.py runs


def pmc_traffic_bytes(kernel_prefix="void nerf_mlp_kernel<true"):
    """HBM bytes per launch of the dominant kernel from THIS round's committed rocprofv3 PMC passes (profiles/r04*_pmc_summary.csv:
    separate FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).  PMC counters cannot be
    collected from inside the run, so this is the profile of the same command, not a per-run measurement; there is no fallback to an
    earlier round's profile: without a current summary for the selected kernel `traffic` is null and the line says why.
    Returns (bytes or None, source file or None, reason or None)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", PROFILE_ROUND + "*_pmc_summary.csv")))
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["kernel"].startswith(kernel_prefix) and float(row.get("FETCH_SIZE", 0) or 0) > 0:
                n = max(int(row["dispatches"]), 1)
                return (2.0 * float(row["FETCH_SIZE"]) + float(row.get("WRITE_SIZE", 0) or 0)) * 1024.0 / n, os.path.relpath(f, ROOT), None
    return None, None, f"no profiles/{PROFILE_ROUND}*_pmc_summary.csv row for kernel '{kernel_prefix}...' (PMC passes of this round not committed for it)"


def host_cores():
    """Threads to use on the host: the affinity mask, capped by the cgroup CPU quota (the GPU box gives a share of a
    large host; oversubscribing it would measure the scheduler, not the code) and by 32."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_baseline(width, height, n_coarse, n_fine, seed, reference_order=True):
    """The CPU oracle (kind "port": a C restatement of the reference's algorithm, oracle/nerf_oracle.c) on this host's
    cores.  `value` = the reference's own loop order (forward_fallback: k-outer / batch-middle / out-inner with separate
    multiply and add, src/network.rs:134-143) on ONE 8x8 block (the reference's rayon task, src/lib.rs:491,533-550) per
    thread, at the reference CLI's 256x256 geometry (src/lib.rs:657-658), same sample counts and seed: a fixed amount of
    work per thread, ~2 min on the round-1 GPU box.  `blocked_rays_per_s` = the oracle's cache-blocked nest (bit-identical
    results, ~400x faster) on a crop of the benchmark frame itself."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    scene = os.path.join(ROOT, "lego_rust")
    S = O.load_samples(os.path.join(scene, "tf_reference_samples.json"))
    co, fi = O.Net(os.path.join(scene, "coarse")), O.Net(os.path.join(scene, "fine"))
    cores = host_cores()
    # (1) cache-blocked nest on a crop of the benchmark frame
    cam = O.camera_from_samples(S, width, height)
    side = 8 * max(4, int(round((cores * 48) ** 0.5)))  # ~3000 rays (48 8x8 blocks) per thread: ~15 s on the GPU box
    side = min(side, (min(width, height) // 8) * 8)
    crop = ((width - side) // 2, (height - side) // 2, side, side)
    t0 = time.time()
    O.render_image(co, fi, cam, O.make_opts(n_coarse, n_fine, crop=crop, seed=seed, naive=False, threads=cores))
    dt_blocked = time.time() - t0
    blocked = side * side / dt_blocked
    blocked_note = (f"{side}x{side} crop at ({crop[0]},{crop[1]}) of the {width}x{height} frame in 8x8 blocks over {cores} threads, "
                    f"cache-blocked loop nest (bit-identical to the reference's), {dt_blocked:.1f} s")
    if not reference_order:
        return {"value": blocked, "unit": "rays/s", "cores": cores, "kind": "port, cache-blocked loop nest (NOT the reference loop order)",
                "sample": blocked_note + "; baseline, not target", "reference_loop_order_rays_per_s": None, "blocked_rays_per_s": blocked}
    # (2) the reference's loop order: one 8x8 block per thread, centred in the 256x256 frame of the reference's CLI
    cam256 = O.camera_from_samples(S, 256, 256)
    bh = max(d for d in range(1, int(cores ** 0.5) + 1) if cores % d == 0)
    bw = cores // bh
    c2 = (max(0, 128 - 4 * bw) // 8 * 8, max(0, 128 - 4 * bh) // 8 * 8, min(8 * bw, 256), min(8 * bh, 256))
    n_rays = c2[2] * c2[3]
    t0 = time.time()
    O.render_image(co, fi, cam256, O.make_opts(n_coarse, n_fine, crop=c2, seed=seed, naive=True, threads=cores))
    dt = time.time() - t0
    return {"value": n_rays / dt, "unit": "rays/s", "cores": cores, "kind": "port, reference loop order",
            "sample": f"{c2[2]}x{c2[3]} window at ({c2[0]},{c2[1]}) of the reference CLI's 256x256 frame (src/lib.rs:657-658) = {n_rays} rays = "
                      f"one 8x8 block (the reference's rayon task, src/lib.rs:491,533-550) per thread over {cores} threads, {n_coarse}+{n_fine} "
                      f"samples, forward_fallback's loop nest (src/network.rs:124-147: bias fill, k-outer / batch-middle / out-inner, "
                      f"separate mul and add), {dt:.1f} s; baseline, not target",
            "reference_loop_order_rays_per_s": n_rays / dt, "blocked_rays_per_s": blocked, "blocked_sample": blocked_note}


def launch_ranks(n, argv, script=None):
    """`python bench.py --gpus N` without an outer launcher: start the N ranks as child processes (one per GPU) BEFORE this
    process touches torch or HIP (it never does), relay rank 0's JSON line, exit non-zero if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    # HSA_ENABLE_IPC_MODE_LEGACY=0: RCCL's intra-node transport shares device buffers between the rank processes through HIP IPC
    # handles; the host driver of this pool supports only the dmabuf flavour, and with the legacy mode left on
    # `hipIpcGetMemHandle` fails with "invalid argument" as soon as two ranks on different GPUs connect.  The image exports the
    # variable already; it is passed on explicitly (an existing value wins) so that the ranks never depend on the caller's shell.
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
    for r, p in enumerate(procs):  # a failed rank leaves the others waiting in a collective: stop exactly those children
        if p.poll() is None:
            p.kill()
        p.wait()
        if p.returncode != 0 and failed is None:
            failed = (r, p.returncode)
    reader.join(timeout=10)
    for ln in (out0[0] if out0 and out0[0] else "").splitlines():  # library chatter on rank 0's stdout (gloo) goes to stderr
        print(ln, file=sys.stdout if ln.lstrip().startswith("{") else sys.stderr, flush=True)
    if failed:
        raise SystemExit(f"bench.py: rank {failed[0]} exited with code {failed[1]}")


def main_inproc(args):
    """--launch inproc: time nerf_render_image_multi -- N contexts in THIS process (one per GPU; fewer GPUs than contexts: they share
    device 0 and the line says so), bands on per-context host threads + streams, ONE gather (host / peer / rccl).  This is the path a
    Rust host takes; torch is not imported.  The entry point hands back a HOST frame, so `value` here includes
    the frame's device-to-host copy (7.68 MB; the per-rank launch keeps the frame in HBM) -- stated in `config.frame`."""
    import ctypes as C
    import nerf_rs_amd as N
    n = args.gpus
    n_dev = C.c_int(0)
    if C.CDLL("libamdhip64.so").hipGetDeviceCount(C.byref(n_dev)) != 0 or n_dev.value < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    distinct = n_dev.value >= n
    scene = os.path.join(ROOT, "lego_rust")
    rs = [N.Renderer(i if distinct else 0) for i in range(n)]
    for r in rs:
        r.load_scene(scene)
    cam = N.camera_from_samples(os.path.join(scene, "tf_reference_samples.json"), args.width, args.height, args.coarse)
    kw = dict(gather=args.gather, seed=args.seed, ssaa=args.ssaa, dtype=args.dtype, skip_empty=args.skip_empty, skip_dead=args.skip_dead,
              hybrid_sampling=args.hybrid_sampling, certify_zero=args.certify_zero)
    for _ in range(args.warmup):
        N.render_image_multi(rs, cam, args.fine, **kw)
    for r in rs:
        r.kernel_time_query(reset=True)
    walls, stats = [], None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        img, stats = N.render_image_multi(rs, cam, args.fine, return_stats=True, **kw)   # synchronous: returns with the frame on the host
        walls.append(1e3 * (time.perf_counter() - t1))
    dt = time.perf_counter() - t0
    dom = [r.kernel_time_query(reset=True) for r in rs]
    one = None
    if n > 1:  # the same frame from one context: the bits must be the same
        one = N.render_image(rs[0].coarse, rs[0].fine, cam, args.fine, **{k: v for k, v in kw.items() if k != "gather"})
    import numpy as np
    n_rays = args.width * args.height * args.ssaa * args.ssaa
    flop_ray = N.flop_per_ray(args.coarse, args.fine)
    bf16, split = args.dtype == "bf16", args.dtype in ("bf16x3", "f16x2")
    peak = PEAK_BF16_MFMA_TFLOPS if (bf16 or split) else PEAK_FP32_MFMA_TFLOPS
    mfma_per_flop = 6.0 if args.dtype == "bf16x3" else 3.0 if args.dtype == "f16x2" else 1.0
    ms_dom = sum(d[0] for d in dom); n_dom = sum(d[2] for d in dom)
    plain = not (args.skip_empty or args.skip_dead or args.certify_zero)
    # plain renders: every sample of every band goes through the dominant (fine-network) kernel once per step; the sum of the launches'
    # durations over all contexts prices the per-GPU rate (contexts that share a device run their launches one after another)
    flops_dom = n_rays * (args.coarse + args.fine) * args.steps * N.FLOP_PER_POINT_FULL
    ach = mfma_per_flop * flops_dom / (ms_dom * 1e-3) / 1e12 if (plain and ms_dom > 0) else None
    renders = [s.ms_total for s in stats]
    stripe = N.partition_for(args.skip_empty, args.skip_dead, args.certify_zero) if n > 1 else 0
    line = {
        "metric": "rays/sec, lego 800x800 (64 coarse + 128 fine samples per ray)", "value": n_rays * args.steps / dt, "unit": "rays/s",
        "n_gpus": n, "launch": "inproc: nerf_render_image_multi (C ABI, one process, one context + host thread + stream per GPU)",
        "gather": args.gather,
        "backend": ((("RCCL ncclAllGather (grouped, in place) over xGMI" if args.gather == "rccl" else
                      "hipMemcpyPeerAsync over xGMI into context 0's frame" if args.gather == "peer" else "per-band device-to-host copies")
                     if distinct or n == 1 else
                     f"REHEARSAL: {n} contexts share {n_dev.value} GPU(s) -- no speed-up to expect; " +
                     ("the RCCL path's slot layout with the collective step as device-to-device copies (RCCL refuses two ranks on one device)"
                      if args.gather == "rccl" else "same-device peer copies" if args.gather == "peer" else "per-band device-to-host copies"))),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "real lego weights + tf_reference_samples.json camera; sample positions from the seeded counter RNG",
        "config": {"workload": f"C3/C4-style: lego coarse+fine hierarchical, {args.width}x{args.height}, {args.coarse}+{args.fine} samples/ray, {args.dtype}, "
                               f"{n} context(s) in one process",
                   "partition": "rows round-robin (band_stripe_rows = 1)" if stripe else "contiguous row bands",
                   "rays_per_step": n_rays, "flop_per_ray": flop_ray, "seed": args.seed,
                   "frame": "this entry point returns the frame in a HOST buffer: the timed region includes its device-to-host copy (7.68 MB at 800x800), unlike "
                            "the per-rank launch whose `value` leaves the frame in HBM",
                   "skip_empty": bool(args.skip_empty), "skip_dead": bool(args.skip_dead), "hybrid_sampling": bool(args.hybrid_sampling),
                   "certify_zero": bool(args.certify_zero),
                   "whole_job_fraction_of_mfma_roofline": mfma_per_flop * (n_rays * args.steps / dt) * flop_ray / ((n if distinct else 1) * peak * 1e12)},
        "per_ctx": {"ms_render": {"max": max(renders), "min": min(renders), "by_ctx": renders},   # device time of each band, last step (nerf_stats.ms_total)
                    "ms_step_wall": {"mean": sum(walls) / len(walls), "by_step": walls},
                    # wall time of the last step beyond its slowest band: the gather, the frame's D2H and the host threads.  Not attributable
                    # when contexts share a device (their kernels interleave, a band's device time then includes its neighbours' work)
                    "ms_gather_and_host": (walls[-1] - max(renders)) if (distinct or n == 1) else None,
                    "rays_by_ctx": [s.n_rays for s in stats],
                    "ms_dominant_kernel_by_ctx": [d[0] / max(args.steps, 1) for d in dom]},
        "image_bit_identical_to_one_context": (bool(np.array_equal(img, one)) if one is not None else None),
    }
    if ach is not None:
        line["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s per GPU", "frac": ach / peak, "traffic": None,
                            "kernel": "fine-network MLP launch of every band (HIP events on each context's stream; sum over contexts)", "launches": n_dom,
                            "avg_launch_ms": ms_dom / max(n_dom, 1)}
    print(json.dumps(line), flush=True)
    for r in rs:
        r.close()
    N.load_library().nerf_multi_release()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--coarse", type=int, default=64)
    ap.add_argument("--fine", type=int, default=128)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16x3", "f16x2"], default="f32",
                    help="MLP arithmetic: f32 = BASELINE's headline config (C3, default, f32 MFMA); bf16 = the C5 study (not the headline); "
                         "bf16x3 / f16x2 = f32-accurate operand splitting on the 16-bit matrix cores (opt-in, meet the f32 tolerances)")
    ap.add_argument("--ssaa", type=int, default=1, help="s x s rays per pixel (C5: --dtype bf16 --ssaa 2)")
    ap.add_argument("--skip-dead", action="store_true",
                    help="SURVEY 8f.2 (reported separately, not the headline): exact dead-sample skipping as the timed path (profiling runs); "
                         "the roofline line is then the ray-sequential trunk kernel priced in EXECUTED flops")
    ap.add_argument("--hybrid-sampling", action="store_true",
                    help="with --skip-dead and --dtype bf16x3|f16x2 (profiling runs): sampling pass in the split arithmetic, ill-conditioned rays redone in f32")
    ap.add_argument("--certify-zero", action="store_true",
                    help="reported separately, not the headline: zero certification as the timed path (profiling runs; f32 only): a bf16 pass "
                         "certifies the samples whose density is certainly 0, the f32 kernel evaluates the rest; the roofline line then prices "
                         "the f32 list kernel in EXECUTED flops")
    ap.add_argument("--skip-empty", action="store_true",
                    help="SURVEY 8f.2 (reported separately, not the headline): skip the colour head of all-empty tiles; "
                         "the image is bit-identical, the roofline line then prices EXECUTED flops")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = ONE frame split in row bands + one RCCL all-gather (SURVEY 8e, default); weak = every rank "
                         "renders a whole frame of its own view (independent frames of a camera path, no data-path collective)")
    ap.add_argument("--launch", choices=["ranks", "inproc"], default="ranks",
                    help="ranks (default, the driver's contract): one process per GPU, torch.distributed over RCCL.  inproc: ONE process, one "
                         "context per GPU behind the C ABI -- nerf_render_image_multi, the call a Rust host makes (INTEGRATION.md section 2); "
                         "no torch involved.  On a box with fewer GPUs than --gpus the contexts share device 0 and the line says REHEARSAL")
    ap.add_argument("--gather", choices=["host", "peer", "rccl"], default="rccl",
                    help="--launch inproc: how the bands meet (NERF_GATHER_HOST / _PEER / _RCCL)")
    ap.add_argument("--no-extra", action="store_true", help="skip the separately reported skip_empty frames (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-reference-order", action="store_true",
                    help="quick runs: time only the oracle's cache-blocked nest (skips the ~2 min reference-loop-order sample)")
    args = ap.parse_args()
    if args.launch == "inproc":
        return main_inproc(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])

    import torch
    import torch.distributed as dist
    import nerf_rs_amd as N

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run")
    n_visible = torch.cuda.device_count()
    if n_visible < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    dev_index = local_rank % n_visible  # a launcher may expose one device per rank (then every rank uses index 0)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # NERF_BENCH_FORCE_DIST=1 drives the N > 1 code path (process group, band split, RCCL all-gather) at world size 1
    use_dist = world > 1 or os.environ.get("NERF_BENCH_FORCE_DIST") == "1"
    data_group = None  # the process group of the ONE data-path collective (None = the default group)
    backend = "none"
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        # nccl == RCCL on ROCm.  RCCL refuses two ranks on one device, so a rehearsal of N ranks on fewer GPUs (a 1-GPU
        # test box) falls back to gloo on host copies of the bands -- and says so in the JSON line.  If fewer devices are
        # visible than ranks, a launcher may still have given every rank its own GPU (per-rank *_VISIBLE_DEVICES): the ranks
        # exchange their device identities over gloo first and use RCCL when they are all different.
        backend = os.environ.get("NERF_BENCH_BACKEND", "nccl" if n_visible >= world else "")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            if backend == "":
                props = torch.cuda.get_device_properties(dev_index)
                ident = tuple(str(getattr(props, k)) for k in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(props, k)) or None
                ids = [None] * world
                dist.all_gather_object(ids, ident)
                if ident is not None and len(set(ids)) == world:
                    data_group = dist.new_group(backend="nccl")
                    backend = "nccl"
                else:
                    backend = "gloo"

    scene = os.path.join(ROOT, "lego_rust")
    r = N.Renderer(dev_index)
    r.load_scene(scene)
    cam = N.camera_from_samples(os.path.join(scene, "tf_reference_samples.json"), args.width, args.height, args.coarse)
    stream = torch.cuda.current_stream(dev).cuda_stream
    frame = torch.empty((args.height, args.width, 3), dtype=torch.float32, device=dev)

    weak = args.scaling == "weak" and world > 1
    view_seed = args.seed + (rank if weak else 0)  # weak scaling: rank r renders view r (its own sample-jitter stream)

    marks = []  # N > 1: per-step (render, gather) marks of this rank

    def step():
        if not use_dist or weak:
            N.render_image(r.coarse, r.fine, cam, args.fine, seed=view_seed, ssaa=args.ssaa, dtype=args.dtype,
                           skip_empty=args.skip_empty, skip_dead=args.skip_dead, hybrid_sampling=args.hybrid_sampling,
                           certify_zero=args.certify_zero, device_out=frame.data_ptr(), stream=stream)
            return frame
        return N.render_image_distributed(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, dtype=args.dtype,
                                          skip_empty=args.skip_empty, skip_dead=args.skip_dead, hybrid_sampling=args.hybrid_sampling,
                                          certify_zero=args.certify_zero, group=data_group, return_tensor=True, timings=marks)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    r.kernel_time_query(reset=True)
    marks.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    ms_dom, pts_dom, n_dom = r.kernel_time_query(reset=True)
    # MAX over ranks of the timed region, taken right here (nothing below may touch it)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if not use_dist or dist.get_backend() == "nccl" else "cpu")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # N > 1: attribute the step to render vs gather, per rank (events recorded inside the timed region, read here)
    per_rank = None
    if use_dist and marks:
        mine = [m.ms() for m in marks]
        v = torch.tensor([sum(a for a, _ in mine) / len(mine), sum(b for _, b in mine) / len(mine), ms_dom / max(n_dom, 1)],
                         dtype=torch.float64, device=tmax.device)
        allv = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(allv, v)
        cols = list(zip(*[t.tolist() for t in allv]))
        per_rank = {k: {"max": max(c), "min": min(c), "by_rank": list(c)} for k, c in zip(("ms_render", "ms_gather", "ms_dominant_kernel_per_launch"), cols)}
    skipped_per_launch = 0
    if args.skip_empty and world == 1:  # one extra untimed frame with stats: the skip count is deterministic per frame
        st = N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, dtype=args.dtype,
                            skip_empty=True, device_out=frame.data_ptr(), stream=stream, return_stats=True)
        skipped_per_launch = st.n_colour_skipped_points
        r.kernel_time_query(reset=True)
    dead_stats = None
    if args.skip_dead and world == 1:  # one extra untimed frame with stats: the executed-work counts are deterministic per frame
        dead_stats = N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, dtype=args.dtype, skip_dead=True,
                                    hybrid_sampling=args.hybrid_sampling, device_out=frame.data_ptr(), stream=stream, return_stats=True)
        r.kernel_time_query(reset=True)
    cert_stats = None
    if args.certify_zero and world == 1:  # one extra untimed frame with stats: the list lengths are deterministic per frame
        cert_stats = N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, dtype=args.dtype, certify_zero=True,
                                    device_out=frame.data_ptr(), stream=stream, return_stats=True)
        r.kernel_time_query(reset=True)
    # Reported separately, never part of `value`: the same frame with exact empty-tile skipping.
    extra_skip = None
    if world == 1 and not args.skip_empty and not args.skip_dead and not args.certify_zero and not args.no_extra:
        def skip_step():
            N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, dtype=args.dtype, skip_empty=True,
                           device_out=frame.data_ptr(), stream=stream)
        ref_frame = frame.clone()
        skip_step(); torch.cuda.synchronize(dev)
        identical = bool(torch.equal(frame, ref_frame))
        t1 = time.perf_counter()
        for _ in range(2):
            skip_step()
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t1) / 2
        extra_skip = {"rays_per_s": args.width * args.height * args.ssaa * args.ssaa / (ms * 1e-3), "ms_per_step": ms,
                      "image_bit_identical_to_headline_run": identical,
                      "note": "opt-in skip_empty: colour head skipped for workgroup tiles (128 samples f32, 256 bf16) whose densities are all 0 (exact)"}
        r.kernel_time_query(reset=True)
    # Reported separately (the rest of it), never part of `value`: the same frame with exact dead-sample skipping --
    # rays retired at the reference's T < 1e-4 cut, colour head only on samples with weight > 0 -- priced in EXECUTED flops.
    extra_dead = None
    if world == 1 and args.dtype == "f32" and not args.skip_empty and not args.skip_dead and not args.certify_zero and not args.no_extra:
        def dead_step(stats=False):
            return N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, skip_dead=True,
                                  device_out=frame.data_ptr(), stream=stream, return_stats=stats)
        step(); torch.cuda.synchronize(dev)
        ref_frame = frame.clone()
        st = dead_step(stats=True); torch.cuda.synchronize(dev)
        identical = bool(torch.equal(frame, ref_frame))
        t1 = time.perf_counter()
        for _ in range(3):
            dead_step()
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t1) / 3
        exec_flop = ((st.n_exec_coarse_trunk + st.n_exec_fine_trunk) * N.FLOP_PER_POINT_SIGMA +
                     st.n_exec_colour * (N.FLOP_PER_POINT_FULL - N.FLOP_PER_POINT_SIGMA))
        n_r = args.width * args.height * args.ssaa * args.ssaa
        extra_dead = {"rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms, "image_bit_identical_to_headline_run": identical,
                      "executed_fraction_coarse_trunk": st.n_exec_coarse_trunk / max(st.n_coarse_points, 1),
                      "executed_fraction_fine_trunk": st.n_exec_fine_trunk / max(st.n_fine_points, 1),
                      "executed_fraction_colour_head": st.n_exec_colour / max(st.n_fine_points, 1),
                      "executed_flop_per_frame": exec_flop, "algorithmic_flop_per_frame": n_r * N.flop_per_ray(args.coarse, args.fine),
                      "executed_tflops": exec_flop / (ms * 1e-3) / 1e12,
                      "executed_fraction_of_fp32_mfma_roofline": exec_flop / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                      "device_ms": {"total": st.ms_total, "coarse_trunk": st.ms_coarse_mlp, "fine_trunk_plus_colour": st.ms_fine_mlp,
                                    "other": st.ms_other, "passes": st.n_passes},
                      "note": "opt-in skip_dead (exact): ray-sequential kernel walks each ray front to back in 32-sample chunks off a "
                              "device-side ray queue and retires it at the reference's T < 1e-4 cut (src/lib.rs:276-279); the trunk outputs of the "
                              "samples with weight > 0 are compacted in LDS and bottleneck + viewdirs + rgb run on them in 64-column passes inside "
                              "the same launch (round 3: no HBM export, one pass per frame)"}
        # ... and with hybrid sampling on top: f16x2 sampling pass, ill-conditioned rays redone in f32, exact-f32 fine pass
        def dead_hyb_step(stats=False):
            return N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, skip_dead=True, hybrid_sampling=True,
                                  device_out=frame.data_ptr(), stream=stream, return_stats=stats)
        dead_hyb_step(); torch.cuda.synchronize(dev)
        hd = (frame - ref_frame).abs()
        t1 = time.perf_counter()
        for _ in range(3):
            dead_hyb_step()
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t1) / 3
        st = dead_hyb_step(stats=True)
        extra_dead["with_hybrid_sampling"] = {
            "rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms, "max_abs_diff_vs_headline_frame": float(hd.max().item()),
            "mean_abs_diff_vs_headline_frame": float(hd.mean().item()), "fraction_of_rays_redone_in_f32": st.n_hybrid_rays / max(st.n_rays, 1),
            "device_ms": {"total": st.ms_total, "coarse_f16x2_plus_f32_redo": st.ms_coarse_mlp, "fine_trunk_plus_colour_f32": st.ms_fine_mlp, "other": st.ms_other}}
        r.kernel_time_query(reset=True)
    # Reported separately, never part of `value`: the same frame in the opt-in f32-accurate operand-splitting arithmetics
    # (three bf16 parts, six products; DESIGN 4.7 f16x2: two f16 parts, three products), each also with
    # exact dead-sample skipping on top.  The coarse (sampling) pass stays on the f32 MFMA kernel in both.
    # Reported separately, never part of `value`: zero certification -- the same f32 frame, bit for bit, with the f32 kernel
    # evaluating only the samples a 16-bit pass (f16 where the network fits its range, else bf16) could not certify as zero-density.
    extra_cert = None
    if world == 1 and args.dtype == "f32" and not args.skip_empty and not args.skip_dead and not args.certify_zero and not args.no_extra:
        def cert_step(stats=False):
            return N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, certify_zero=True,
                                  device_out=frame.data_ptr(), stream=stream, return_stats=stats)
        step(); torch.cuda.synchronize(dev)
        ref_frame = frame.clone()
        st = cert_step(stats=True); torch.cuda.synchronize(dev)
        identical = bool(torch.equal(frame, ref_frame))
        t1 = time.perf_counter()
        for _ in range(3):
            cert_step()
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t1) / 3
        n_r = args.width * args.height * args.ssaa * args.ssaa
        extra_cert = {"rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms, "image_bit_identical_to_headline_run": identical,
                      "f32_evaluated_fraction_coarse": st.n_exec_coarse_trunk / max(st.n_coarse_points, 1),
                      "f32_evaluated_fraction_fine": st.n_exec_fine_trunk / max(st.n_fine_points, 1),
                      "colour_head_fraction_fine": st.n_exec_colour / max(st.n_fine_points, 1),
                      "device_ms": {"total": st.ms_total, "coarse_bf16_pass_plus_f32_list": st.ms_coarse_mlp, "fine_bf16_pass_plus_f32_list": st.ms_fine_mlp, "other": st.ms_other},
                      "audit": {"certified_samples_evaluated_all_the_same": st.n_certify_audited, "violations": st.n_certify_violations,
                                "margins_coarse_fine": list(st.certify_margin), "least_headroom_coarse_fine": list(st.certify_headroom),
                                "largest_prefilter_error_on_an_audited_certificate_coarse_fine": list(st.certify_max_error),
                                "frame_rendered_again": st.n_certify_retries},
                      "rays_whose_predicted_cut_was_not_confirmed": st.n_certify_fallback_rays,
                      "note": "opt-in certify_zero (DESIGN 4.9): a 16-bit pass (f16 operands where the network fits the f16 range -- lego does --, else bf16) over all samples certifies those whose density pre-activation is below minus the "
                              "network's margin as zeros of the f32 network too and predicts each ray's T < 1e-4 cut; the f32 MFMA kernel evaluates only the "
                              "other samples in front of the predicted cut (a device-side list), the exact transmittance confirms the cut; certified samples "
                              "and samples behind the cut have weight 0, so the frame is the headline frame bit for bit; 1 in 16 of the samples certified by less than twice the margin and 1 in 128 of the others are "
                              "evaluated all the same (the audit: a wrong or nearly wrong certificate widens the margin and the frame is rendered again); "
                              "fuzzed: tools/fuzz_certify.py"}
        r.kernel_time_query(reset=True)
    extra_split = {}
    if world == 1 and args.dtype == "f32" and not args.skip_empty and not args.skip_dead and not args.certify_zero and not args.no_extra:
        n_r = args.width * args.height * args.ssaa * args.ssaa
        step(); torch.cuda.synchronize(dev)
        f32_frame = frame.clone()
        notes = {"bf16x3": "opt-in mlp_dtype bf16x3: the fine (colour) pass computes every f32 product as the six significant bf16 x bf16 "
                           "products of three-way splits, f32 accumulate; the coarse (sampling) pass stays on the f32 MFMA kernel, so the "
                           "fine sample positions equal the f32 path's bit for bit; passes the UNRELAXED Gate 1 against the oracle "
                           "(tests/test_gpu_parity.py::test_bf16x3_render_matches_oracle_crop, tests/test_gpu_frame_fixture.py)",
                 "f16x2": "opt-in mlp_dtype f16x2: the fine (colour) pass computes every f32 product as the three significant f16 x f16 products "
                          "of two-way splits (operands exact to 2^-22), f32 accumulate; f32 sampling pass as for bf16x3; passes the UNRELAXED "
                          "Gate 1 against the oracle (tests/test_gpu_f16x2.py); f16 range: activations must stay below 65504"}
        for arith in ("bf16x3", "f16x2"):
            def split_step(dead=False, stats=False):
                return N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, dtype=arith, skip_dead=dead,
                                      device_out=frame.data_ptr(), stream=stream, return_stats=stats)

            def timed(dead):
                split_step(dead); split_step(dead); torch.cuda.synchronize(dev)  # warm frames: the clock settles at the power limit of the 16-bit MFMA stream
                t1 = time.perf_counter()
                for _ in range(3):
                    split_step(dead)
                torch.cuda.synchronize(dev)
                return 1e3 * (time.perf_counter() - t1) / 3
            ms = timed(False)
            diff = (frame - f32_frame).abs()
            split_frame = frame.clone()
            e = {"rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms,
                 "max_abs_diff_vs_f32_frame": float(diff.max().item()), "mean_abs_diff_vs_f32_frame": float(diff.mean().item()),
                 "fraction_of_values_differing_by_more_than_5e-5": float((diff > 5e-5).float().mean().item()),
                 "psnr_vs_f32_frame_db": float(-10.0 * torch.log10((diff.double() ** 2).mean().clamp_min(1e-30)).item()),
                 "note": notes[arith]}
            ms = timed(True)
            identical = bool(torch.equal(frame, split_frame))
            st = split_step(True, stats=True)
            e["with_skip_dead"] = {
                "rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms, f"image_bit_identical_to_the_{arith}_frame": identical,
                "executed_fraction_coarse_trunk": st.n_exec_coarse_trunk / max(st.n_coarse_points, 1),
                "executed_fraction_fine_trunk": st.n_exec_fine_trunk / max(st.n_fine_points, 1),
                "executed_fraction_colour_head": st.n_exec_colour / max(st.n_fine_points, 1),
                "device_ms": {"total": st.ms_total, "coarse_trunk_f32": st.ms_coarse_mlp, f"fine_trunk_plus_colour_{arith}": st.ms_fine_mlp, "other": st.ms_other}}
            # ... and with zero certification: exact-f32 sampling pass and the split fine pass, each on its certified list
            def cz_step(stats=False):
                return N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, dtype=arith, certify_zero=True,
                                      device_out=frame.data_ptr(), stream=stream, return_stats=stats)
            cz_step(); cz_step(); torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(3):
                cz_step()
            torch.cuda.synchronize(dev)
            ms = 1e3 * (time.perf_counter() - t1) / 3
            identical = bool(torch.equal(frame, split_frame))
            st = cz_step(stats=True)
            e["with_certify_zero"] = {
                "rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms, f"image_bit_identical_to_the_{arith}_frame": identical,
                "evaluated_fraction_coarse_f32": st.n_exec_coarse_trunk / max(st.n_coarse_points, 1),
                f"evaluated_fraction_fine_{arith}": st.n_exec_fine_trunk / max(st.n_fine_points, 1),
                "audit": {"certified_samples_evaluated_all_the_same": st.n_certify_audited, "violations": st.n_certify_violations,
                          "least_headroom_coarse_fine": list(st.certify_headroom), "margins_coarse_fine": list(st.certify_margin),
                          "frame_rendered_again": st.n_certify_retries},
                "device_ms": {"total": st.ms_total, "coarse_bf16_pass_plus_f32_list": st.ms_coarse_mlp, f"fine_bf16_pass_plus_{arith}_list": st.ms_fine_mlp, "other": st.ms_other}}
            # ... and with the sampling pass in the split arithmetic too, ill-conditioned rays redone in f32
            def hyb_step(stats=False):
                return N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=args.ssaa, dtype=arith, skip_dead=True,
                                      hybrid_sampling=True, device_out=frame.data_ptr(), stream=stream, return_stats=stats)
            hyb_step(); hyb_step(); torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(3):
                hyb_step()
            torch.cuda.synchronize(dev)
            ms = 1e3 * (time.perf_counter() - t1) / 3
            hd = (frame - split_frame).abs()
            st = hyb_step(stats=True)
            e["with_skip_dead_and_hybrid_sampling"] = {
                "rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms,
                f"max_abs_diff_vs_the_{arith}_frame": float(hd.max().item()), f"mean_abs_diff_vs_the_{arith}_frame": float(hd.mean().item()),
                "fraction_of_rays_redone_in_f32": st.n_hybrid_rays / max(st.n_rays, 1),
                "device_ms": {"total": st.ms_total, f"coarse_{arith}_plus_f32_redo": st.ms_coarse_mlp, f"fine_trunk_plus_colour_{arith}": st.ms_fine_mlp,
                              "other": st.ms_other},
                "note": "opt-in hybrid_sampling: coarse pass in the split arithmetic; rays with a hierarchical draw predicted to move by more than 1e-5 in t "
                        "(light CDF bins, nearly empty rays, transmittance near the cut) are redone in exact f32 and resampled (bit-identical "
                        "positions to the f32 sampling pass there, <= 1e-5 in t elsewhere); "
                        "Gate 1 against the oracle's whole frame: tests/test_gpu_f16x2.py::test_hybrid_sampling"}
            extra_split[arith] = e
        r.kernel_time_query(reset=True)
    # Reported separately, never part of `value`: BASELINE config C5's geometry on this one GPU -- 800x800 output, 2x2 SSAA
    # (1600x1600 = 2.56 M rays), bf16 operands / f32 accumulate (PSNR-level parity: tests/test_gpu_frame_fixture.py).
    extra_c5 = None
    if world == 1 and args.dtype == "f32" and args.ssaa == 1 and not args.skip_empty and not args.skip_dead and not args.no_extra:
        def c5_step(stats=False):
            return N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=2, dtype="bf16",
                                  device_out=frame.data_ptr(), stream=stream, return_stats=stats)
        c5_step(); c5_step(); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(3):
            c5_step()
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t1) / 3
        st = c5_step(stats=True)
        n_r = 4 * args.width * args.height
        extra_c5 = {"workload": f"C5 on one GPU: {args.width}x{args.height} output, 2x2 SSAA = {n_r} rays, {args.coarse}+{args.fine} samples/ray, bf16 MLP",
                    "rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms, "n_rays": st.n_rays,
                    "whole_job_fraction_of_bf16_mfma_roofline": n_r / (ms * 1e-3) * N.flop_per_ray(args.coarse, args.fine) / (PEAK_BF16_MFMA_TFLOPS * 1e12),
                    "device_ms": {"total": st.ms_total, "coarse": st.ms_coarse_mlp, "fine": st.ms_fine_mlp, "other": st.ms_other}}
        # ... and the same frame with exact dead-sample skipping in the bf16 arithmetic (two ray cursors per wave)
        c5_frame = frame.clone()

        def c5_dead_step(stats=False):
            return N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=2, dtype="bf16", skip_dead=True,
                                  device_out=frame.data_ptr(), stream=stream, return_stats=stats)
        c5_dead_step(); c5_dead_step(); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(3):
            c5_dead_step()
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t1) / 3
        st = c5_dead_step(stats=True)
        extra_c5["with_skip_dead"] = {
            "rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms, "image_bit_identical_to_the_bf16_frame": bool(torch.equal(frame, c5_frame)),
            "executed_fraction_coarse_trunk": st.n_exec_coarse_trunk / max(st.n_coarse_points, 1),
            "executed_fraction_fine_trunk": st.n_exec_fine_trunk / max(st.n_fine_points, 1),
            "executed_fraction_colour_head": st.n_exec_colour / max(st.n_fine_points, 1),
            "device_ms": {"total": st.ms_total, "coarse_trunk": st.ms_coarse_mlp, "fine_trunk_plus_colour": st.ms_fine_mlp, "other": st.ms_other,
                          "passes": st.n_passes}}
        # ... and, beside the bf16 study, what the same geometry costs at f32 ACCURACY (Gate 1): f16x2 fine pass + certify_zero
        def c5_x2_step(stats=False):
            return N.render_image(r.coarse, r.fine, cam, args.fine, seed=args.seed, ssaa=2, dtype="f16x2", certify_zero=True,
                                  device_out=frame.data_ptr(), stream=stream, return_stats=stats)
        try:
            c5_x2_step(); c5_x2_step(); torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(3):
                c5_x2_step()
            torch.cuda.synchronize(dev)
            ms = 1e3 * (time.perf_counter() - t1) / 3
            st = c5_x2_step(stats=True)
            extra_c5["same_geometry_at_f32_accuracy_f16x2_certify_zero"] = {
                "rays_per_s": n_r / (ms * 1e-3), "ms_per_step": ms,
                "exact_fraction_coarse": st.n_exec_coarse_trunk / max(st.n_coarse_points, 1), "exact_fraction_fine": st.n_exec_fine_trunk / max(st.n_fine_points, 1),
                "note": "not bf16: the f16x2 frame (Gate 1 against the CPU oracle at 800x800) at C5's geometry, for comparison with the bf16 rows above"}
        except N.NerfError as e:
            extra_c5["same_geometry_at_f32_accuracy_f16x2_certify_zero"] = {"error": e.msg}
        r.kernel_time_query(reset=True)
    # The timed region leaves the frame in HBM (`value` never includes PCIe); the host-pointer entry point additionally pays
    # one D2H copy of the frame (BASELINE.md section 4 counts it on the GPU side): measured here, reported beside `value`.
    d2h_ms = None
    if world == 1:
        host_frame = torch.empty(frame.shape, dtype=frame.dtype, pin_memory=True)
        host_frame.copy_(frame, non_blocking=True); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(5):
            host_frame.copy_(frame, non_blocking=True)
        torch.cuda.synchronize(dev)
        d2h_ms = 1e3 * (time.perf_counter() - t1) / 5

    if rank == 0:
        n_rays = args.width * args.height * args.ssaa * args.ssaa
        flop_ray = N.flop_per_ray(args.coarse, args.fine)
        bf16 = args.dtype == "bf16"
        x3 = args.dtype == "bf16x3"
        x2 = args.dtype == "f16x2"
        split = x3 or x2
        sfx = "x3" if x3 else "f16x2" if x2 else "bf16"  # kernel-name suffixes of the two-launch skip_dead kernels
        two_launch = split or bf16                         # skip_dead: trunk launch + colour launch on the HBM-compacted live samples
        peak = PEAK_BF16_MFMA_TFLOPS if (bf16 or split) else PEAK_FP32_MFMA_TFLOPS  # the f16 MFMA forms run at the bf16 rate
        mfma_per_flop = 6.0 if x3 else 3.0 if x2 else 1.0  # executed 16-bit MFMA flops per algorithmic f32 flop
        value = n_rays * args.steps * (world if weak else 1) / dt  # whole-job rays/s over all ranks
        # executed flops of the dominant launches: a skipped sample still runs dense0..7 + alpha (sigma-only cost)
        flops_dom = pts_dom * N.FLOP_PER_POINT_FULL - n_dom * skipped_per_launch * (N.FLOP_PER_POINT_FULL - N.FLOP_PER_POINT_SIGMA)
        if dead_stats is not None:
            # dominant launch = the ray-sequential fine kernel: dense0..7 + alpha on the samples in front of the cut and -- f32: in the same
            # launch (colour passes on the LDS-compacted live samples); bf16 / split arithmetics: in a second launch, not priced here -- the colour head
            flops_dom = (n_dom / max(dead_stats.n_passes, 1)) * (dead_stats.n_exec_fine_trunk * N.FLOP_PER_POINT_SIGMA +
                                                                  (0 if two_launch else dead_stats.n_exec_colour * (N.FLOP_PER_POINT_FULL - N.FLOP_PER_POINT_SIGMA)))
        if cert_stats is not None:  # dominant launch = the list kernel of the fine network: trunk of the listed samples + the colour heads not skipped tile-wise
            flops_dom = (n_dom / max(cert_stats.n_passes, 1)) * (cert_stats.n_exec_fine_trunk * N.FLOP_PER_POINT_SIGMA +
                                                                  cert_stats.n_exec_colour * (N.FLOP_PER_POINT_FULL - N.FLOP_PER_POINT_SIGMA))
        ach = mfma_per_flop * flops_dom / (ms_dom * 1e-3) / 1e12 if ms_dom > 0 else 0.0
        traffic, traffic_src, traffic_why = pmc_traffic_bytes(
            (f"void nerf_trunk_seq_kernel_{sfx}<true" if two_launch else "void nerf_trunk_seq_kernel<true") if args.skip_dead else
            "void nerf_mlp_kernel_bf16v2<true" if bf16 else
            ("void nerf_mlp_kernel_bf16x3<true, 2" if x3 else "void nerf_mlp_kernel_f16x2<true, 2" if x2 else "void nerf_mlp_kernel<true, 2") if args.certify_zero else
            "void nerf_mlp_kernel_bf16x3<true" if x3 else "void nerf_mlp_kernel_f16x2<true" if x2 else "void nerf_mlp_kernel<true")
        line = {  # noqa: E501
            "metric": "rays/sec, lego 800x800 (64 coarse + 128 fine samples per ray)", "value": value, "unit": "rays/s",
            "n_gpus": world, "ranks": dist.get_world_size() if use_dist else 1,
            "backend": ((backend + (" (RCCL over xGMI)" if backend == "nccl" else
                                    f" (REHEARSAL: {world} ranks share {n_visible} GPU(s); the collective runs on host copies)"))
                        if use_dist else "none (single process, no collective)"),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": ("bf16 operands / f32 accumulate (C5 study, PSNR-level parity)" if bf16 else
                      "f32 as three-way bf16 split: 6 bf16 MFMA products per f32 product, f32 accumulate (f32-level parity)" if x3 else
                      "f32 as two-way f16 split: 3 f16 MFMA products per f32 product, f32 accumulate (f32-level parity)" if x2 else "f32"),
            "data": "real lego weights (lego_rust/, 2 x 595,844 f32 parameters) + tf_reference_samples.json camera; "
                    "sample positions from the seeded counter RNG (no dataset involved)",
            "config": {"workload": (f"C5-style: {args.ssaa}x{args.ssaa} SSAA, bf16 MLP, " if bf16 or args.ssaa > 1 else f"C3 ({args.dtype} arithmetic): " if split else "C3: ") +
                                   f"lego coarse+fine hierarchical, {args.width}x{args.height}, {args.coarse}+{args.fine} "
                                   f"samples/ray, {args.dtype}, {world}xMI355X" +
                                   ("" if world == 1 else ", one frame per rank, no collective" if weak else ", row bands + RCCL all-gather"),
                       "rays_per_step": n_rays, "flop_per_ray": flop_ray, "seed": args.seed,
                       "frame": "left in HBM inside the timed region: `value` excludes the 7.68 MB device-to-host copy of the frame, as the bench "
                                "contract requires (inputs and outputs resident in HBM).  This supersedes BASELINE.md section 4's wording (\"includes "
                                "the device-to-host copy of the image\"); the PCIe-inclusive rate is reported beside it as rays_per_s_including_d2h",
                       "d2h_ms_per_frame": d2h_ms,
                       "rays_per_s_including_d2h": (n_rays / (dt / args.steps + 1e-3 * d2h_ms)) if d2h_ms is not None else None,
                       "skip_empty": bool(args.skip_empty), "skip_dead": bool(args.skip_dead), "hybrid_sampling": bool(args.hybrid_sampling),
                       "certify_zero": bool(args.certify_zero),
                       "colour_head_skipped_samples_per_frame": skipped_per_launch,
                       "whole_job_fraction_of_mfma_roofline": mfma_per_flop * value * flop_ray / (world * peak * 1e12)},  # per-GPU average
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak, "traffic": traffic,
                         "traffic_source": (f"HBM bytes per launch from this round's committed rocprofv3 PMC passes of the same command ({traffic_src}; "
                                            "2 x FETCH_SIZE + WRITE_SIZE in separate --pmc runs; counters cannot be read from inside the run); "
                                            "algorithmic: 20 B/point") if traffic_src else traffic_why,
                         "kernel": ((f"nerf_trunk_seq_kernel_{sfx}<EXPORT=true> (fine network, ray-sequential trunk; executed flops)" if two_launch else
                                     "nerf_trunk_seq_kernel<EXPORT=true> (fine network, ray-sequential trunk + in-kernel colour passes; executed flops)")
                                    if dead_stats is not None else
                                    ("nerf_mlp_kernel" + ("_bf16x3" if x3 else "_f16x2" if x2 else "") + "<FULL=true, MODE_LIST> (fine network, the samples a 16-bit pass could neither certify as zeros nor place behind the cut; executed flops)")
                                    if cert_stats is not None else
                                    ("nerf_mlp_kernel_bf16v2" if bf16 else "nerf_mlp_kernel_bf16x3" if x3 else "nerf_mlp_kernel_f16x2" if x2 else "nerf_mlp_kernel") +
                                    "<FULL=true, MODE_RAYS> (fine network)"),
                         "launches": n_dom, "avg_launch_ms": ms_dom / max(n_dom, 1),
                         "points_per_launch": (cert_stats.n_exec_fine_trunk if cert_stats is not None else pts_dom // max(n_dom, 1)),
                         "flop_per_point": N.FLOP_PER_POINT_SIGMA if dead_stats is not None else N.FLOP_PER_POINT_FULL,
                         "flop_per_live_point_colour_head": (N.FLOP_PER_POINT_FULL - N.FLOP_PER_POINT_SIGMA) if dead_stats is not None and not two_launch else None},
        }
        if per_rank is not None:
            line["per_rank"] = per_rank  # max/min over ranks: render (HIP events around the band render), gather (events around the collective)
        if split:
            line["roofline"]["note"] = (f"achieved/peak price the EXECUTED 16-bit MFMA flops ({mfma_per_flop:.0f} per algorithmic f32 flop) against the "
                                        "bf16/f16 peak; f32_equivalent_tflops = algorithmic f32 flops / time")
            line["roofline"]["f32_equivalent_tflops"] = ach / mfma_per_flop
        for arith, e in extra_split.items():
            line["extra_" + arith] = e
        if extra_skip:
            line["extra_skip_empty"] = extra_skip
        if extra_dead:
            line["extra_skip_dead"] = extra_dead
        if extra_cert:
            line["extra_certify_zero"] = extra_cert
        if extra_c5:
            line["extra_c5_bf16_ssaa2"] = extra_c5
        if world == 1 and not args.no_cpu_baseline and not bf16 and not split:
            line["cpu_baseline"] = cpu_baseline(args.width, args.height, args.coarse, args.fine, args.seed, not args.no_cpu_reference_order)
        print(json.dumps(line), flush=True)
    del out
    r.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
