"""Multi-GPU rendering: rays shard embarrassingly (every ray is independent -- the reference already exploits this
with rayon over 8x8 blocks, src/lib.rs:533-550).  One process per GPU; rank r renders a contiguous band of rows on
its own device and ONE collective -- an all-gather of the bands over RCCL/xGMI -- assembles the framebuffer on every
rank.  Weights are replicated (4.8 MB).  No other exchange exists on this path.
"""
import numpy as np


def band_of_rank(n_rows, rank, world_size):
    """Contiguous, balanced row bands: cost per ray is uniform (the reference never skips work), so equal row counts
    balance.  Returns (first_row, n_rows_of_rank); the first (n_rows % world) ranks get one extra row."""
    base, rem = divmod(int(n_rows), int(world_size))
    y0 = rank * base + min(rank, rem)
    return y0, base + (1 if rank < rem else 0)


def render_image_distributed(coarse, fine, camera, fine_samples_per_ray=128, *, seed=0, coarse_only=False, crop=None,
                             ssaa=1, dtype="f32", skip_empty=False, skip_dead=False, hybrid_sampling=False, certify_zero=False, group=None,
                             band_renderer=None, device=None, return_tensor=False, timings=None):
    """render_image over all ranks of `group` (torch.distributed; backend nccl == RCCL on ROCm, gloo in CPU tests).

    Every rank returns the full (h, w, 3) frame.  `band_renderer(crop) -> ndarray (rows, w, 3)` overrides the GPU
    renderer (used by the CPU gloo tests, where the band is produced by the oracle).

    `timings` (a list) receives one `StepMarks` per call so that a scaling curve can be attributed to render vs gather:
    read them with `.ms()` once the device is idle (bench.py does, after its timed region).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    x0, y0, w, h = crop if crop else (0, 0, camera.nx, camera.ny)
    b0, rows = band_of_rank(h, rank, world)
    max_rows = band_of_rank(h, 0, world)[1]
    band_crop = (x0, y0 + b0, w, rows)
    marks = StepMarks() if timings is not None else None
    if band_renderer is not None:
        dev = torch.device("cpu") if device is None else device
        band = torch.zeros((max_rows, w, 3), dtype=torch.float32, device=dev)
        if marks: marks.stamp(dev)
        if rows > 0:
            band[:rows] = torch.from_numpy(np.ascontiguousarray(band_renderer(band_crop), dtype=np.float32)).to(dev)
    else:
        from .api import render_image
        dev = torch.device("cuda", coarse.renderer.device) if device is None else device
        band = torch.zeros((max_rows, w, 3), dtype=torch.float32, device=dev)
        if marks: marks.stamp(dev)
        if rows > 0:
            stream = torch.cuda.current_stream(dev).cuda_stream
            render_image(coarse, fine, camera, fine_samples_per_ray, seed=seed, coarse_only=coarse_only, crop=band_crop,
                         ssaa=ssaa, dtype=dtype, skip_empty=skip_empty, skip_dead=skip_dead, hybrid_sampling=hybrid_sampling,
                         certify_zero=certify_zero, device_out=band.data_ptr(), stream=stream)
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
    if h % world == 0:
        frame = gathered.view(h, w, 3)
    else:
        frame = torch.cat([gathered[r, :band_of_rank(h, r, world)[1]] for r in range(world)], dim=0)
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
