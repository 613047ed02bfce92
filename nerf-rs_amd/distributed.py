"""Multi-GPU rendering: rays shard embarrassingly (every ray is independent -- the reference already exploits this
with rayon over 8x8 blocks, src/lib.rs:533-550).  One process per GPU; rank r renders band r of the rows on its own device and
ONE collective -- an all-gather of the bands over RCCL/xGMI -- assembles the framebuffer on every rank.  Weights are replicated
(4.8 MB).  No other exchange exists on this path.

The partition (nerf_render_opts.band_*, the same rule as nerf_render_image_multi): rayon balances by work stealing, a static
partition has to know where the cost is.  Plain renders cost the same for every ray: contiguous bands.  skip_dead / skip_empty /
certify_zero make the cost follow the scene (the lego background, 75 % of the rays, is nearly free and sits in the top rows): single
rows dealt out round-robin, so that every rank gets the same mix; the gathered bands are then put in place by one index_copy.
"""
import numpy as np


def partition_for(skip_empty=False, skip_dead=False, certify_zero=False):
    """band_stripe_rows of the partition: 0 = contiguous bands (uniform cost), 1 = rows round-robin (cost follows the scene)."""
    return 1 if (skip_empty or skip_dead or certify_zero) else 0


def band_row_indices(n_rows, rank, world_size, stripe_rows=0):
    """Rows of the window that rank `rank` renders, in the order its band holds them (= nerf-rs_amd/api.py band_row_indices)."""
    h, n = int(n_rows), int(world_size)
    if stripe_rows <= 0:
        y0, rows = band_of_rank(h, rank, n)
        return np.arange(y0, y0 + rows)
    rows = np.arange(h)
    return rows[(rows // stripe_rows) % n == rank]


def band_of_rank(n_rows, rank, world_size):
    """Contiguous, balanced row bands: cost per ray is uniform (the reference never skips work), so equal row counts
    balance.  Returns (first_row, n_rows_of_rank); the first (n_rows % world) ranks get one extra row."""
    base, rem = divmod(int(n_rows), int(world_size))
    y0 = rank * base + min(rank, rem)
    return y0, base + (1 if rank < rem else 0)


def render_image_distributed(coarse, fine, camera, fine_samples_per_ray=128, *, seed=0, coarse_only=False, crop=None,
                             ssaa=1, dtype="f32", skip_empty=False, skip_dead=False, hybrid_sampling=False, certify_zero=False, group=None,
                             band_renderer=None, device=None, return_tensor=False, timings=None, stripe_rows=None):
    """render_image over all ranks of `group` (torch.distributed; backend nccl == RCCL on ROCm, gloo in CPU tests).

    Every rank returns the full (h, w, 3) frame.  `band_renderer(crop) -> ndarray (rows, w, 3)` overrides the GPU
    renderer (used by the CPU gloo tests, where the band is produced by the oracle; with a striped partition it is called once per
    run of consecutive rows).  stripe_rows: None = partition_for(...) of the options; 0 = contiguous bands; S > 0 = stripes of S rows.

    `timings` (a list) receives one `StepMarks` per call so that a scaling curve can be attributed to render vs gather:
    read them with `.ms()` once the device is idle (bench.py does, after its timed region).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    x0, y0, w, h = crop if crop else (0, 0, camera.nx, camera.ny)
    stripe = partition_for(skip_empty, skip_dead, certify_zero) if stripe_rows is None else int(stripe_rows)
    mine = band_row_indices(h, rank, world, stripe)
    rows = len(mine)
    max_rows = len(band_row_indices(h, 0, world, stripe))  # band 0 is never shorter than another band
    marks = StepMarks() if timings is not None else None
    if band_renderer is not None:
        dev = torch.device("cpu") if device is None else device
        band = torch.zeros((max_rows, w, 3), dtype=torch.float32, device=dev)
        if marks: marks.stamp(dev)
        k = 0
        while k < rows:  # one call per run of consecutive rows (contiguous bands: one run)
            e = k + 1
            while e < rows and mine[e] == mine[e - 1] + 1:
                e += 1
            part = band_renderer((x0, y0 + int(mine[k]), w, e - k))
            band[k:e] = torch.from_numpy(np.ascontiguousarray(part, dtype=np.float32)).to(dev)
            k = e
    else:
        from .api import render_image
        dev = torch.device("cuda", coarse.renderer.device) if device is None else device
        band = torch.zeros((max_rows, w, 3), dtype=torch.float32, device=dev)
        if marks: marks.stamp(dev)
        if rows > 0:
            stream = torch.cuda.current_stream(dev).cuda_stream
            render_image(coarse, fine, camera, fine_samples_per_ray, seed=seed, coarse_only=coarse_only, crop=crop,
                         ssaa=ssaa, dtype=dtype, skip_empty=skip_empty, skip_dead=skip_dead, hybrid_sampling=hybrid_sampling,
                         certify_zero=certify_zero, band=(rank, world, stripe), device_out=band.data_ptr(), stream=stream)
    if marks: marks.stamp(band.device)  # band rendered (GPU: an event on the render stream)
    if band.device.type == "cuda" and dist.get_backend(group) != "nccl":
        # rehearsal only (e.g. gloo with several ranks on one GPU): the collective runs on host copies of the bands
        torch.cuda.current_stream(band.device).synchronize()
        if marks: marks.host_gather_begin()
        host = band.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        gathered = torch.stack(parts).to(band.device)
    else:
        gathered = torch.empty((world, max_rows, w, 3), dtype=torch.float32, device=band.device)
        dist.all_gather_into_tensor(gathered.view(-1), band.view(-1), group=group) if band.device.type == "cuda" else \
            dist.all_gather(list(gathered.unbind(0)), band, group=group)
    if marks:
        marks.stamp(band.device)  # bands gathered
        timings.append(marks)
    if stripe <= 0 and h % world == 0:
        frame = gathered.view(h, w, 3)
    elif stripe <= 0:
        frame = torch.cat([gathered[r, :band_of_rank(h, r, world)[1]] for r in range(world)], dim=0)
    else:  # packed rows -> their places: slot r, local row j is frame row band_row_indices(h, r, world, stripe)[j]
        src = np.concatenate([r * max_rows + np.arange(len(band_row_indices(h, r, world, stripe))) for r in range(world)])
        dst = np.concatenate([band_row_indices(h, r, world, stripe) for r in range(world)])
        frame = torch.empty((h, w, 3), dtype=torch.float32, device=gathered.device)
        frame.index_copy_(0, torch.from_numpy(dst).to(gathered.device), gathered.view(world * max_rows, w, 3).index_select(0, torch.from_numpy(src).to(gathered.device)))
    return frame if return_tensor else frame.cpu().numpy()


class StepMarks:
    """Three marks of one distributed step -- before the band render, after it, after the collective -- as HIP events on the
    current stream (GPU path) or host clock stamps (CPU path).  On the host-rehearsal path the gather runs on the host after a
    stream synchronisation, so it is bracketed by host stamps instead."""

    def __init__(self):
        self.m = []
        self.host_t0 = None

    def stamp(self, device):
        import time
        import torch
        if self.host_t0 is not None:  # host rehearsal: close the host-side gather bracket
            self.m.append(time.perf_counter())
        elif device.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(device))
            self.m.append(e)
        else:
            self.m.append(time.perf_counter())

    def host_gather_begin(self):
        import time
        self.host_t0 = time.perf_counter()

    def ms(self):
        """(ms_render, ms_gather); call once the device is idle."""
        def span(a, b):
            return a.elapsed_time(b) if hasattr(a, "elapsed_time") else 1e3 * (b - a)
        a, b, c = self.m
        return span(a, b), (1e3 * (c - self.host_t0) if self.host_t0 is not None else span(b, c))
